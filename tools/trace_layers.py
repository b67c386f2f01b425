#!/usr/bin/env python3
"""Encoder-forward layers of every replayed step in a rocprofv3 kernel_trace.csv of bench.py: duration of each layer (from the image tower's
first LayerNorm of a layer to that of the next), time with no / one / several kernels resident — a layer whose three streams run one after
the other instead of side by side shows as a long layer with idle time.  Also: GPU-idle time of every whole step.
Usage: trace_layers.py kernel_trace.csv [verbose]"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
adam = [e for e in ev if 'adamw_flat' in e[2]]
steps = [(adam[i][1], adam[i + 1][1]) for i in range(len(adam) - 1)]
print(f'# {len(steps)} replayed steps')


def cover(ks, a, b):
    pts = sorted([(max(e[0], a), 1) for e in ks] + [(min(e[1], b), -1) for e in ks])
    cur, last, idle, one, multi = 0, a, 0, 0, 0
    for t, d in pts:
        if cur == 0:
            idle += t - last
        elif cur == 1:
            one += t - last
        else:
            multi += t - last
        cur += d
        last = t
    idle += b - last
    return idle / 1e3, one / 1e3, multi / 1e3


all_layers = []
for si, (a, b) in enumerate(steps):
    ks = [e for e in ev if a <= e[0] < b]
    unsh = next(e for e in ks if 'unshuffle_fwd' in e[2])[0]
    enc = [e for e in ks if e[0] < unsh]
    qs = sorted({e[3] for e in enc if 'attn_fwd_kernel<64, 64' in e[2]})
    q1 = qs[0]
    att = [e for e in enc if 'attn_fwd_kernel<64, 64' in e[2] and e[3] == q1]
    starts = [[e for e in enc if e[3] == q1 and 'ln_fwd_kernel' in e[2] and e[0] < x[0]][-1][0] for x in att] + [unsh]
    lay = []
    for i in range(len(starts) - 1):
        la, lb = starts[i], starts[i + 1]
        idle, one, multi = cover([e for e in enc if la <= e[0] < lb], la, lb)
        lay.append(((lb - la) / 1e3, idle, one, multi))
    all_layers += [l[0] for l in lay]
    idle_step = cover(ks, ks[0][0], b)[0]
    med = statistics.median(l[0] for l in lay)
    slow = [(i, round(l[0]), round(l[1])) for i, l in enumerate(lay) if l[0] > 1.4 * med]
    print(f'step {si}: {(b - a) / 1e6:6.2f} ms, GPU idle {idle_step / 1e3:5.2f} ms; encoder forward {sum(l[0] for l in lay) / 1e3:5.2f} ms, median layer {med:4.0f} us, '
          f'slow layers (index, us, idle us): {slow}')
    if len(sys.argv) > 2:
        for i, l in enumerate(lay):
            print(f'    layer {i:2d}: {l[0]:6.0f} us  idle {l[1]:5.0f}  one kernel {l[2]:5.0f}  overlapped {l[3]:5.0f}')
print(f'# all layers: median {statistics.median(all_layers):.0f} us, mean {statistics.mean(all_layers):.0f} us, max {max(all_layers):.0f} us')
