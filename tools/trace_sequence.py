#!/usr/bin/env python3
"""The kernel sequence of the LAST graph replay in a rocprofv3 kernel_trace.csv: start offset, duration, gap to the previous
kernel's end and name, one line per launch — to see which small launches sit on the critical path of a layer.
Usage: trace_sequence.py kernel_trace.csv step_ms [first [count]]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 10 ** 9
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda x: x[0])
t_end = max(e[1] for e in ev)
win = [e for e in ev if e[0] >= t_end - step_ms * 1e6]
t0 = win[0][0]
prev_end = t0
gaps = 0
for i, (s, e, n) in enumerate(win):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*', '', n)[:84]
    gap = s - prev_end
    gaps += max(gap, 0)
    if first <= i < first + count:
        print(f'{i:4d} t={(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:6.1f}  {n}')
    prev_end = max(prev_end, e)
print(f'{len(win)} launches, sum of gaps {gaps / 1e6:.3f} ms of {(win[-1][1] - t0) / 1e6:.2f} ms')
