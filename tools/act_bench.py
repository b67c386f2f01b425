#!/usr/bin/env python3
"""Cost of the fused activation epilogues of dav_gemm_nt_bf16 on the step's MLP shapes (graph-timed).
act 0 = none, 1 = GELU (fc1 forward, + pre-activation twin), 2 = * GELU'(aux) (fc2 input gradient)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (M, N, K, kn) in [(4032, 3072, 768, 0), (22528, 2048, 512, 0), (4032, 3072, 768, 1), (22528, 2048, 512, 1), (14592, 2048, 512, 1)]:
    A = torch.randn(M, K, device=dev).to(BF16)
    Bm = (torch.randn(K, N, device=dev) * 0.05).to(BF16) if kn else (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    C = torch.empty(M, N, device=dev, dtype=BF16)
    C2 = torch.empty(M, N, device=dev, dtype=BF16)
    aux = torch.randn(M, N, device=dev).to(BF16)
    bias = torch.randn(N, device=dev)
    kw = dict(ldb=N if kn else K, C_out=C, c_bf16=True, variant=kn << 12)
    t0 = timed(lambda: ops.gemm_nt(A, Bm, M, N, K, **kw))
    t1 = timed(lambda: ops.gemm_nt(A, Bm, M, N, K, bias=bias, act=1, C2=C2, ldc2=N, c2_mode=1, **kw))
    t2 = timed(lambda: ops.gemm_nt(A, Bm, M, N, K, act=2, aux=aux, ldaux=N, **kw))
    t4 = timed(lambda: ops.gemm_nt(A, Bm, M, N, K, bias=bias, act=1, C2=C2, ldc2=N, c2_mode=4, **kw))
    t3 = timed(lambda: ops.gemm_nt(A, Bm, M, N, K, act=3, aux=aux, ldaux=N, **kw))
    print(f'{M}x{N}x{K} kn={kn}: plain {t0:7.1f} us   gelu+z twin {t1:7.1f}   *gelu\'(aux) {t2:7.1f}   gelu+gelu\' twin {t4:7.1f}   *aux {t3:7.1f}')
