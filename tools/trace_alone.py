#!/usr/bin/env python3
"""For the last graph replay in a rocprofv3 kernel_trace.csv: per kernel family, how much of its run time it was the ONLY kernel on
the GPU (a stream schedule wants that small for everything that does not fill the GPU by itself).
Usage: trace_alone.py kernel_trace.csv step_ms"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda x: x[0])
t_end = max(e[1] for e in ev)
win = [e for e in ev if e[0] >= t_end - step_ms * 1e6]
pts = sorted({t for s, e, _ in win for t in (s, e)})
# sweep: for each elementary interval count active kernels
import bisect
active = [0] * (len(pts) - 1)
for s, e, _ in win:
    for k in range(bisect.bisect_left(pts, s), bisect.bisect_left(pts, e)):
        active[k] += 1
fam_tot, fam_alone, fam_n = defaultdict(int), defaultdict(int), defaultdict(int)
for s, e, n in win:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)[:64]
    a = sum(pts[k + 1] - pts[k] for k in range(bisect.bisect_left(pts, s), bisect.bisect_left(pts, e)) if active[k] == 1)
    fam_tot[n] += e - s; fam_alone[n] += a; fam_n[n] += 1
print(f'{"family":64s} launches   total ms   alone ms')
for n in sorted(fam_tot, key=lambda k: -fam_alone[k])[:24]:
    print(f'{n:64s} {fam_n[n]:8d} {fam_tot[n] / 1e6:10.2f} {fam_alone[n] / 1e6:10.2f}')
print(f'alone total {sum(fam_alone.values()) / 1e6:.2f} ms')
