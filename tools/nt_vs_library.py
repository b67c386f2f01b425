#!/usr/bin/env python3
"""Every distinct big NT GEMM shape of a step, product kernel vs the vendor library, alone on the GPU.

The comparator (hipBLASLt behind torch.matmul) is NOT on the product path: it is the yardstick for what a library-class body
reaches on this chip for these shapes (bf16 in, bf16 out, no epilogue).  The product kernel runs with the tile configuration
the step would pick for a one-problem launch of that shape (tuned table / rules; variant 0) and, for comparison, as configuration
60 (the 256 x 256 ping-pong body that is faster alone and slower in the step).

Input: the launch list a bench run wrote (DAV_DUMP_MIX=<file> python bench.py --roofline-only [--config ...]).
Usage: python tools/nt_vs_library.py <mix.json> [<mix.json> ...]      (one table per file)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: E402,F401
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
MIN_LAUNCH_TILES = int(os.environ.get('NTL_MIN_TILES', '150'))      # (the bench's "dominant kernel" set is >= 400 tile equivalents per launch)
NSETS = 4                       # operand sets rotated per call: no launch finds its own operands in L2


def graph_us(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def one(M, N, K, b_kn):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NSETS)]
    W = [(torch.randn(K, N, device=dev) * 0.05).bfloat16() if b_kn else (torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(NSETS)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NSETS)]
    it = [0]

    def ours(cfg):
        def fn():
            i = it[0] % NSETS; it[0] += 1
            ops.gemm_nt(A[i], W[i], M, N, K, ldb=N if b_kn else K, C_out=C[i], c_bf16=True, variant=(cfg << 4) | (b_kn << 12))
        return fn

    def lib():
        i = it[0] % NSETS; it[0] += 1
        torch.matmul(A[i], W[i] if b_kn else W[i].t(), out=C[i])

    ref = (A[0].float() @ (W[0].float() if b_kn else W[0].float().t()))
    it[0] = 0; ours(0)(); torch.cuda.synchronize()
    err = float((C[0].float() - ref).norm() / ref.norm())
    assert err < 1e-2, (M, N, K, b_kn, err)
    t0 = graph_us(ours(0))
    try:
        t60 = graph_us(ours(60)) if (K % 64 == 0 and N % 8 == 0) else float('nan')
    except Exception:                       # (shapes configuration 60 refuses)
        t60 = float('nan')
    tl = graph_us(lib)
    return t0, t60, tl


for path in sys.argv[1:]:
    mix = json.load(open(path))['nt']
    shapes = {}
    for ent in mix:
        cfg, b_kn, probs = ent[0], ent[1], ent[2]
        tiles = sum(-(-m // 128) * -(-n // 128) for (m, n, k) in probs)
        if tiles < MIN_LAUNCH_TILES:
            continue
        for (m, n, k) in probs:
            if -(-m // 128) * -(-n // 128) < 100:
                continue                    # (small members of a grouped launch)
            key = (m, n, k, b_kn)
            c = shapes.setdefault(key, [0, set()])
            c[0] += 1; c[1].add(cfg)
    print(f'# {path}: {len(shapes)} distinct big NT shapes (launch count per step in the last column)')
    print(f'# {"M x N x K":>22} {"form":>5} {"product us":>10} {"TF":>6} {"cfg 60 us":>10} {"library us":>10} {"TF":>6} {"product / library":>8} {"n":>4}   tile configuration(s) in the step')
    tot = [0.0, 0.0]
    for (m, n, k, b_kn), (cnt, cfgs) in sorted(shapes.items(), key=lambda kv: -kv[1][0] * kv[0][0] * kv[0][1] * kv[0][2]):
        t0, t60, tl = one(m, n, k, b_kn)
        gf = 2.0 * m * n * k / 1e6
        tot[0] += t0 * cnt; tot[1] += tl * cnt
        print(f'  {f"{m} x {n} x {k}":>22} {"b_kn" if b_kn else "nt":>5} {t0:10.1f} {gf / t0:6.0f} {t60:10.1f} {tl:10.1f} {gf / tl:6.0f} {t0 / tl:8.2f} {cnt:4d}   cfg {sorted(cfgs)}')
    print(f'# weighted by launch count: product {tot[0] / 1e3:.2f} ms, library {tot[1] / 1e3:.2f} ms per step  (ratio {tot[0] / max(tot[1], 1e-9):.2f})')
