#!/usr/bin/env python3
"""Per-queue view of the last replayed step in a rocprofv3 kernel_trace.csv of bench.py: for every hardware queue the graph replay uses, busy
time and kernel count per phase (encoder forward / decoders forward / decoders backward / encoder backward), and — for the encoder forward —
which queue a layer WAITS for: the step's streams re-join after every layer (the next layer's towers read the fusion tokens as context rows),
so a layer is as long as its longest chain.  Usage: trace_streams.py kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
qcol = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get(qcol, '0') if qcol else '0') for r in rows), key=lambda x: x[0])
adams = [e for e in ev if 'adamw_flat' in e[2]]
t_prev, t_last = (adams[-2][1] if len(adams) > 1 else ev[0][0]), adams[-1][1]          # (a file cut to one step: all of it)
win = [e for e in ev if t_prev <= e[0] <= t_last]
t0 = win[0][0]
ms = lambda t: (t - t0) / 1e6
short = lambda n: re.sub(r'\(.*', '', re.sub(r'\(anonymous namespace\)::', '', n))[:48]


def first(pred, after=0):
    for e in win:
        if e[0] >= after and pred(e[2]):
            return e
    return None


unsh = first(lambda n: 'unshuffle_fwd' in n)                  # first decoder kernel
loss_b = first(lambda n: 'patch_mse_bwd' in n)
gang1 = first(lambda n: 'gemm_tn_gang_kernel' in n)
gang2 = first(lambda n: 'gemm_tn_gang_kernel' in n, gang1[1])
phases = [('encoder forward', t0, unsh[0]), ('decoders forward', unsh[0], loss_b[0]), ('decoders backward', loss_b[0], gang1[0]),
          ('encoder backward', gang1[1], gang2[0])]
print(f'# queue column: {qcol}; phases of the last replayed step')
for name, a, b in phases:
    print(f'{name:18s} {ms(a):6.2f} .. {ms(b):6.2f} ms ({(b - a) / 1e6:.2f})')
    per = defaultdict(lambda: [0, 0])
    for s, e, n, q in win:
        if a <= s < b:
            per[q][0] += min(e, b) - s
            per[q][1] += 1
    for q, (d, c) in sorted(per.items(), key=lambda x: -x[1][0]):
        print(f'    queue {q:>4s}: busy {d / 1e6:6.2f} ms = {100 * d / (b - a):5.1f} %  {c:4d} kernels')
# per phase and queue: the chain of kernels — time in kernels vs time waiting between them (gap = idle before a kernel of that queue)
chains = {}
for name, a, b in phases:
    print(f'\n# {name}: per queue, running vs gaps')
    byq = defaultdict(list)
    for s, e, n, q in win:
        if a <= s < b:
            byq[q].append((s, e, n))
    chains[name] = byq
    for q, lst in sorted(byq.items(), key=lambda x: -len(x[1])):
        gaps = [lst[i][0] - lst[i - 1][1] for i in range(1, len(lst))]
        run = sum(e - s for s, e, _ in lst)
        hist = [sum(1 for g in gaps if lo <= g < hi) for lo, hi in ((-1 << 60, 2000), (2000, 10000), (10000, 30000), (30000, 1 << 60))]
        print(f'queue {q}: {len(lst)} kernels, running {run / 1e6:.2f} ms, gaps {sum(max(g, 0) for g in gaps) / 1e6:.2f} ms; gaps < 2 us / 2-10 / 10-30 / > 30 us: {hist}; '
              f'median kernel {sorted(e - s for s, e, _ in lst)[len(lst) // 2] / 1e3:.1f} us')
if len(sys.argv) > 3:          # dump one queue's chain in one phase: trace_streams.py trace.csv "decoders forward" 1 [count]
    lst = chains[sys.argv[2]][sys.argv[3]][:int(sys.argv[4]) if len(sys.argv) > 4 else 80]
    for i, (s, e, n) in enumerate(lst):
        print(f'  {ms(s):8.3f}  {(e - s) / 1e3:7.1f} us  gap {(s - lst[i - 1][1]) / 1e3 if i else 0:6.1f}  {short(n)}')
