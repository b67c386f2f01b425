#!/usr/bin/env python3
"""Timeline analysis of the last graph replay in a rocprofv3 kernel_trace.csv:
busy/idle time, concurrency, per-kernel-family time.  Usage: trace_timeline.py kernel_trace.csv step_ms"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda x: x[0])
t_end = max(e[1] for e in ev)
win = [e for e in ev if e[0] >= t_end - step_ms * 1e6]
t0, t1 = win[0][0], max(e[1] for e in win)
# union coverage + concurrency-weighted time
pts = []
for s, e, _ in win:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; conc_time = defaultdict(int); cur = 0; last = pts[0][0]
for t, d in pts:
    if cur > 0:
        busy += t - last
    conc_time[cur] += t - last
    cur += d; last = t
span = t1 - t0
print(f'window {span / 1e6:.2f} ms, kernels {len(win)}, GPU busy (>=1 kernel) {busy / 1e6:.2f} ms, idle {(span - busy) / 1e6:.2f} ms')
print('time by #concurrent kernels:', {k: round(v / 1e6, 2) for k, v in sorted(conc_time.items())})
fam = defaultdict(lambda: [0, 0])
for s, e, n in win:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:70]
    fam[n][0] += e - s; fam[n][1] += 1
tot = sum(v[0] for v in fam.values())
print(f'sum of kernel durations {tot / 1e6:.2f} ms')
for n, (d, c) in sorted(fam.items(), key=lambda x: -x[1][0])[:22]:
    print(f'  {d / 1e6:7.2f} ms  {c:5d} launches  avg {d / c / 1e3:7.1f} us  {n}')
