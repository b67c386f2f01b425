#!/usr/bin/env python3
"""Where do the cycles of the dominant NT GEMM configuration (128x128 tile, 8 waves, 2-stage LDS ring) go?
Runs the instrumented instantiation (tile configuration 30: s_memtime stamps around the k-loop phases) on the step's big
shapes and prints mean shader cycles per wave: wait for DMA / barrier / DMA issue / fragment reads + MFMAs per k-step,
prologue and epilogue per tile.  Usage: gemm_phase_prof.py [M N K ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import ops  # noqa: E402

dev = 'cuda'
bf = torch.bfloat16
args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(5184, 2304, 768), (11264, 2304, 768), (7168, 3072, 768), (7168, 768, 3072), (22528, 2048, 512)]
NT3 = os.environ.get('PROF_NT3', '0') != '0'
NT3_CFG = 32 + int(os.environ.get('PROF_NT3', '0'))      # 1: full, 2..5: ablations (no DMA / no reads / no setprio / no MFMA)      # profile the staggered-halves kernel (configuration 33 / 32) instead
for M, N, K in shapes:
    A = torch.randn(M, K, device=dev).to(bf)
    W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    tiles = ((M + 255) // 256) * ((N + 127) // 128) if NT3 else ((M + 127) // 128) * ((N + 127) // 128)
    buf = torch.zeros(tiles * 8 * 10, dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.gemm_nt(A, W, M, N, K, C_out=C, c_bf16=True, res_rows=buf.view(torch.int32), variant=(NT3_CFG if NT3 else 30) << 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm_nt(A, W, M, N, K, C_out=C, c_bf16=True, res_rows=buf.view(torch.int32), variant=(NT3_CFG if NT3 else 30) << 4)
    e1.record()
    torch.cuda.synchronize()
    us_prof = e0.elapsed_time(e1) * 1e3
    e0.record()
    ops.gemm_nt(A, W, M, N, K, C_out=C, c_bf16=True, variant=(32 if NT3 else 3) << 4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    raw = buf.view(tiles, 8, 10).cpu()
    r = raw.double()
    nk = K // (32 if NT3 else 64)
    ph = r[:, :, :6].mean(dim=(0, 1))
    tot = float(ph.sum())
    print(f'{M}x{N}x{K}: {tiles} tiles, {nk} k-steps, plain {us:.1f} us ({2.0 * M * N * K / us / 1e6:.0f} TF), instrumented {us_prof:.1f} us')
    print(f'  per k-step cycles/wave: wait-DMA {ph[0] / nk:7.0f}  barrier {ph[1] / nk:7.0f}  DMA-issue {ph[2] / nk:6.0f}  reads+MFMA {ph[3] / nk:7.0f}'
          f'   (ideal MFMA-bound k-step at 2 WG/CU: 1024)')
    print(f'  per tile   cycles/wave: prologue {ph[4]:7.0f}  k-loop {float(ph[:4].sum()):8.0f}  epilogue {ph[5]:7.0f}   total {tot:8.0f} = {tot / 2.4e3:.1f} us at 2.4 GHz')
    t0, t1 = raw[:, 0, 6], raw[:, 0, 7]
    span = int(t1.max() - t0.min())
    print(f'  first start -> last end: {span} ticks over {us_prof:.1f} us -> counter rate {span / us_prof / 1e3:.2f} GHz; mean tile residency {float((t1 - t0).double().mean()):.0f} ticks')
    hw, xcc = raw[:, 0, 9], raw[:, 0, 8] & 15
    cu = ((xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)).tolist()      # (xcc, se, sh, cu)
    from collections import defaultdict
    per = defaultdict(list)
    for c, a, b in zip(cu, t0.tolist(), t1.tolist()):
        per[c].append((a, b))
    conc = []
    for c, iv in per.items():
        pts = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
        cur = mx = 0
        for _, d in pts:
            cur += d
            mx = max(mx, cur)
        conc.append(mx)
    print(f'  distinct CUs seen {len(per)}, tiles per CU min/max {min(len(v) for v in per.values())}/{max(len(v) for v in per.values())}, max co-resident workgroups per CU: {min(conc)}..{max(conc)}')
    w = r[:, :, :6].mean(dim=0)
    print('  reads+MFMA per k-step by wave:', [int(x / nk) for x in w[:, 3].tolist()], ' wait-DMA by wave:', [int(x / nk) for x in w[:, 0].tolist()])
