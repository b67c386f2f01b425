#!/usr/bin/env python3
"""One NT shape, both operand forms (nt: B [N, K]; b_kn: B [K, N]), across explicit tile configurations, alone on the GPU with
rotating operand sets.  Usage: python tools/nt_cfg_compare.py M N K cfg [cfg ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: E402,F401
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
M, N, K = (int(x) for x in sys.argv[1:4])
cfgs = [int(c) for c in sys.argv[4:]] or [0]
NS = 4
for b_kn in (0, 1):
    A = [torch.randn(M, K, device=dev).bfloat16() for _ in range(NS)]
    W = [(torch.randn(K, N, device=dev) * 0.05).bfloat16() if b_kn else (torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(NS)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(NS)]
    ref = A[0].float() @ (W[0].float() if b_kn else W[0].float().t())
    for c in cfgs:
        it = [0]

        def fn():
            i = it[0] % NS; it[0] += 1
            ops.gemm_nt(A[i], W[i], M, N, K, ldb=N if b_kn else K, C_out=C[i], c_bf16=True, variant=(c << 4) | (b_kn << 12))
        try:
            fn(); torch.cuda.synchronize()
        except RuntimeError as e:
            print(f'{M}x{N}x{K} {"b_kn" if b_kn else "nt  "} cfg{c:3d}: refused ({e})'); continue
        err = float((C[0].float() - ref).norm() / ref.norm())
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                for _ in range(20):
                    fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
        print(f'{M}x{N}x{K} {"b_kn" if b_kn else "nt  "} cfg{c:3d}: {best:8.1f} us  {2.0 * M * N * K / best / 1e6:7.0f} TF   rel err {err:.2e}')
