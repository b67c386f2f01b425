#!/usr/bin/env python3
"""Phase boundaries of the last replayed step in a rocprofv3 kernel_trace.csv of bench.py: forward (to the first loss-backward kernel),
decoders' backward (to the start of the first gang weight-gradient launch), that launch, encoder backward (to the second gang launch),
that launch, optimizer.  Usage: trace_phases.py kernel_trace.csv step_ms"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
step_ms = float(sys.argv[2])
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda x: x[0])
# the last complete step: from the end of the second-to-last AdamW launch to the end of the last one
adams = [e for e in ev if 'adamw_flat' in e[2]]
t_prev, t_last = adams[-2][1], adams[-1][1]
win = [e for e in ev if t_prev <= e[0] <= t_last]
t0 = win[0][0]
ms = lambda t: (t - t0) / 1e6


def first(pred, after=0):
    for e in win:
        if e[0] >= after and pred(e[2]):
            return e
    return None


loss_b = first(lambda n: 'patch_mse_bwd' in n)
gang1 = first(lambda n: 'gemm_tn_gang_kernel' in n)
gang2 = first(lambda n: 'gemm_tn_gang_kernel' in n, gang1[1]) if gang1 else None
adam = adams[-1]
print(f'forward            0.00 .. {ms(loss_b[0]):6.2f} ms')
if gang1:
    print(f'decoders backward  {ms(loss_b[0]):6.2f} .. {ms(gang1[0]):6.2f} ms   ({ms(gang1[0]) - ms(loss_b[0]):.2f})')
    print(f'gang launch 1      {ms(gang1[0]):6.2f} .. {ms(gang1[1]):6.2f} ms   ({(gang1[1] - gang1[0]) / 1e6:.2f})')
if gang2:
    print(f'encoder backward   (decoders done) .. {ms(gang2[0]):6.2f} ms')
    print(f'gang launch 2      {ms(gang2[0]):6.2f} .. {ms(gang2[1]):6.2f} ms   ({(gang2[1] - gang2[0]) / 1e6:.2f})')
if adam:
    print(f'AdamW              {ms(adam[0]):6.2f} .. {ms(adam[1]):6.2f} ms   ({(adam[1] - adam[0]) / 1e6:.2f})')
print(f'window             {ms(max(e[1] for e in win)):6.2f} ms')
