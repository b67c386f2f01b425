#!/usr/bin/env python3
"""Does the row stride of the operands matter (L2 channel interleave)?  Same GEMM with A / B / C rows padded by `pad` elements.
Usage: [PAD_WHICH=abc] gemm_pad.py M N K [pads...]     PAD_WHICH: which of a (activations), b (weights), c (output) carry the padding"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import ops  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4])
pads = [int(x) for x in sys.argv[4:]] or [0, 8, 32, 64, 128]
dev = 'cuda'
WHICH = os.environ.get('PAD_WHICH', 'abc')
NSETS = 4
for pad in pads:
    sets = []
    for _ in range(NSETS):
        pa, pb, pc = (pad if 'a' in WHICH else 0), (pad if 'b' in WHICH else 0), (pad if 'c' in WHICH else 0)
        A = torch.randn(M, K + pa, device=dev).bfloat16()
        W = (torch.randn(N, K + pb, device=dev) * 0.05).bfloat16()
        C = torch.empty(M, N + pc, device=dev, dtype=torch.bfloat16)
        sets.append((A, W, C))
    it = [0]

    def fn():
        A, W, C = sets[it[0] % NSETS]
        it[0] += 1
        ops.gemm_nt(A, W, M, N, K, lda=K + pa, ldb=K + pb, C_out=C, ldc=N + pc, c_bf16=True, variant=3 << 4)
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'{M}x{N}x{K} pad[{WHICH}] {pad:4d} elements: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF')
