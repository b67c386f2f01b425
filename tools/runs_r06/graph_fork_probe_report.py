#!/usr/bin/env python3
"""Reads the kernel trace of graph_fork_probe.py: for the last replay, per queue first start / last end / busy, relative to the fork node."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
# replays are separated by idle gaps: take the last burst
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-8:]) > 200000:
        bursts.append(cur); cur = []
    cur.append(e)
bursts.append(cur)
last = bursts[-1]
t0 = last[0][0]
byq = defaultdict(list)
for e in last:
    byq[e[3]].append(e)
print(f'{sys.argv[2] if len(sys.argv) > 2 else ""}: last replay {len(last)} kernels, {(max(e[1] for e in last) - t0) / 1e3:.0f} us')
for q, l in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in l)
    print(f'   queue {q}: {len(l):3d} kernels, first start +{(l[0][0] - t0) / 1e3:7.1f} us, last end +{(l[-1][1] - t0) / 1e3:7.1f} us, busy {busy / 1e3:7.1f} us')
