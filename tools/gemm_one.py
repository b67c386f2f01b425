#!/usr/bin/env python3
"""One NT GEMM shape across explicit tile configurations: correctness vs fp32 torch + graph-timed TFLOP/s.
Usage: python tools/gemm_one.py M N K cfg [cfg ...]"""
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
A = torch.randn(M, K, device=dev).bfloat16()
B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
ref = A.float() @ B.float().t() + bias
for c in cfgs:
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm_nt(A, B, M, N, K, bias=bias, C_out=C, c_bf16=True, variant=c << 4)
    fn(); torch.cuda.synchronize()
    err = float((C.float() - ref).norm() / ref.norm())
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'{M}x{N}x{K} cfg{c:3d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF   rel err {err:.2e}')
