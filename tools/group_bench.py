#!/usr/bin/env python3
"""Grouped vs individual launches of the tower GEMMs of one ViT-B layer (B = 64: image 3136 / 5184 rows, audio 4032 / 6080
rows), timed as hipGraph replays so that host launch overhead is out of the picture.
Usage: group_bench.py [cfgs...]   cfg = explicit tile configuration number of dav_gemm_nt_bf16 (0 = auto)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import engine as E  # noqa: E402
from deepavfusion_amd import ops  # noqa: E402

dev = 'cuda'
bf = torch.bfloat16
REPS = 20


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        for _ in range(REPS):
            fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
    return best


NSETS = 5          # operand sets rotated per repetition so that inputs / outputs do not sit in the 256 MB infinity cache


def problem(M, N, K, kind):
    return [problem1(M, N, K, kind) for _ in range(NSETS)]


def problem1(M, N, K, kind):
    A = torch.randn(M, K, device=dev).to(bf)
    kw = {}
    if kind == 'dgrad':            # b_kn: B given as W[contraction=K][out=N]... dgrad reads W [N_fwd, K_fwd] itself
        W = (torch.randn(K, N, device=dev) * 0.05).to(bf)
        kw.update(ldb=N, variant=1 << 12)
    else:
        W = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    if kind == 'res':              # fp32 out + fp32 residual (proj / fc2 forward)
        C = torch.empty(M, N, device=dev)
        kw.update(res=torch.randn(M, N, device=dev), ldres=N, C_out=C)
    elif kind == 'gelu':           # fc1 forward: bf16 out + GELU' twin
        C = torch.empty(M, N, device=dev, dtype=bf)
        kw.update(act=1, C_out=C, c_bf16=True, C2=torch.empty(M, N, device=dev, dtype=bf), ldc2=N, c2_mode=4,
                  bias=torch.randn(N, device=dev))
    else:
        C = torch.empty(M, N, device=dev, dtype=bf)
        kw.update(C_out=C, c_bf16=True)
    return (A, W, M, N, K), kw


_ROT = [0]


def run(p, cfg):
    (A, W, M, N, K), kw = p[_ROT[0] % NSETS]
    kw = dict(kw)
    kw['variant'] = kw.get('variant', 0) | (cfg << 4)
    ops.gemm_nt(A, W, M, N, K, **kw)


ONLY = os.environ.get('GB_ONLY', '').split(',') if os.environ.get('GB_ONLY') else None
SHAPES = [('qkv', 5184, 6080, 2304, 768, 'plain'), ('proj', 3136, 4032, 768, 768, 'res'), ('fc1', 3136, 4032, 3072, 768, 'gelu'),
          ('fc2', 3136, 4032, 768, 3072, 'res'), ('d_fc2', 3136, 4032, 3072, 768, 'dgrad'), ('d_fc1', 3136, 4032, 768, 3072, 'dgrad'),
          ('d_proj', 3136, 4032, 768, 768, 'dgrad'), ('d_qkv', 5184, 6080, 768, 2304, 'dgrad'),
          ('dec_qkv', 14592, 22528, 1536, 512, 'plain'), ('dec_fc1', 14592, 22528, 2048, 512, 'gelu'), ('dec_fc2', 14592, 22528, 512, 2048, 'res'),
          ('dec_proj', 14592, 22528, 512, 512, 'res'), ('d_decprj', 14592, 22528, 512, 512, 'dgrad'), ('d_decfc2', 14592, 22528, 2048, 512, 'dgrad'),
          ('f_kv', 3136, 4032, 1536, 768, 'plain')]
cfgs = [int(a) for a in sys.argv[1:]] or [0, 3, 8, 5]
print(f'{"gemm":8s} {"GF":>6s} | ' + ' | '.join(f'cfg{c}: indiv us   TF  group us   TF' for c in cfgs))
for name, Mi, Ma, N, K, kind in SHAPES:
    if ONLY and name not in ONLY:
        continue
    pi, pa = problem(Mi, N, K, kind), problem(Ma, N, K, kind)
    gf = 2.0 * (Mi + Ma) * N * K / 1e9
    cells = []
    for c in cfgs:
        def indiv():
            _ROT[0] += 1
            run(pi, c)
            run(pa, c)

        def grouped():
            _ROT[0] += 1
            with E.batch() as bt:
                bt.lane()
                run(pi, c)
                bt.lane()
                run(pa, c)
        ti, tg = timed(indiv), timed(grouped)
        cells.append(f'      {ti:7.1f} {gf / ti * 1e3:5.0f}  {tg:7.1f} {gf / tg * 1e3:5.0f}')
    print(f'{name:8s} {gf:6.1f} | ' + ' | '.join(cells), flush=True)
