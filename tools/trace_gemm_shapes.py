#!/usr/bin/env python3
"""Group GEMM launches of the last replay in a rocprofv3 kernel trace by (kernel, grid) -> count, avg duration."""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
t_end = max(int(r['End_Timestamp']) for r in rows)
acc = defaultdict(lambda: [0, 0])
for r in rows:
    if int(r['Start_Timestamp']) < t_end - step_ms * 1e6 or 'gemm' not in r['Kernel_Name']:
        continue
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'\(.*', '', n).replace('void ', '')
    key = (n, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r.get('Grid_Size_Y', 1)))
    acc[key][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    acc[key][1] += 1
for (n, gx, gy), (d, c) in sorted(acc.items(), key=lambda x: -x[1][0])[:40]:
    print(f'{d / 1e6:6.2f} ms  {c:4d} x {d / c / 1e3:7.1f} us  blocks {gx:5d} x{gy}  {n}')
