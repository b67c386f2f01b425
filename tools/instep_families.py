#!/usr/bin/env python3
"""Per-family kernel time INSIDE the step from a rocprofv3 --kernel-trace --stats kernel_stats.csv of bench.py (graph replays +
the eager / capture passes around them; normalised by the number of adamw_flat_kernel launches = optimizer steps in the trace).
Kernel time, not wall time: the default schedule runs up to three streams side by side.
Usage: instep_families.py kernel_stats.csv"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = sum(int(r['Calls']) for r in rows if 'adamw_flat_kernel' in r['Name']) or 1


def family(n):
    if 'gemm_tn' in n: return 'weight-gradient GEMMs (gemm_tn_grouped / gemm_tn2)'
    m = re.search(r'gemm_nt2(_grouped)?_kernel<(\d+), (\d+)', n)
    if m:
        bm, bn = int(m.group(2)), int(m.group(3))
        if bm * bn >= 128 * 128: return 'big NT GEMMs (128x128 / 128x256 tiles: towers, decoders)'
        if bm * bn >= 128 * 64: return 'NT GEMMs on 128x64 tiles (N = 768 outputs, mid-size dgrads)'
        return 'small NT GEMMs (64x64 tiles: fusion block)'
    if 'gemm_nt256' in n: return 'big NT GEMMs (128x128 / 128x256 tiles: towers, decoders)'
    if 'ft_tail' in n: return 'fused fusion-block tails'
    if 'attn' in n:
        if '<32, 32' in n: return 'attention, decoders (d = 32)'
        if '<16, 64' in n or 'attn_grouped' in n: return 'attention, fusion block (aggregation + pair)'
        return 'attention, towers (d = 64)'
    if 'ln_' in n: return 'LayerNorm (forward / backward / dgamma-dbeta reduce)'
    if 'adamw' in n or 'sumsq' in n or 'step_guard' in n: return 'AdamW + grad norm'
    if 'pair_' in n or 'add_cast' in n or 'cast_' in n or 'unshuffle' in n or 'patch_' in n or 'mask_build' in n or 'rows_' in n: return 'movers / loss / masking'
    return 'other (torch fills, copies, RNG)'


fam = {}
for r in rows:
    f = family(r['Name'])
    t, c = float(r['TotalDurationNs']), int(r['Calls'])
    a = fam.setdefault(f, [0.0, 0])
    a[0] += t; a[1] += c
tot = sum(v[0] for v in fam.values())
print(f'# {steps} optimizer steps in the trace; kernel time per step by family (streams overlap: the sum exceeds the step time)')
print(f'{"family":66s} {"ms/step":>8s} {"share":>6s} {"launches/step":>14s} {"avg us":>8s}')
for f, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f'{f:66s} {t / steps / 1e6:8.2f} {100 * t / tot:5.1f}% {c / steps:14.1f} {t / c / 1e3:8.1f}')
print(f'{"total":66s} {tot / steps / 1e6:8.2f}')
print()
print('# the individual kernels (per step)')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:40]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:7.2f} ms  {int(r['Calls']) / steps:7.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {re.sub(r'.anonymous namespace.::', '', r['Name'])[:110]}")
