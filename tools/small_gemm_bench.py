#!/usr/bin/env python3
"""Graph-timed (no host launch overhead) NT GEMMs on the small / medium shapes of the step — the fusion block's
projections and the N = 768 tower GEMMs — across tile configurations.  Usage: small_gemm_bench.py [cfg ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16
SHAPES = [(512, 768, 768), (512, 1536, 768), (512, 192, 768), (1024, 192, 768), (1024, 768, 768), (2048, 768, 768),
          (512, 768, 192), (4096, 768, 768), (3136, 768, 768), (4032, 768, 768), (3136, 768, 3072), (4032, 768, 3072)]


def timed(fn, reps=50):
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


cfgs = [int(c) for c in sys.argv[1:]] or [0, 5, 6, 7, 8]
print(f'{"shape":>18} ' + ' '.join(f'{"cfg%d" % c:>9}' for c in cfgs) + '   (us; cfg0 = library choice)')
for (M, N, K) in SHAPES:
    A = torch.randn(M, K, device=dev).to(BF16)
    Bm = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
    C = torch.empty(M, N, device=dev, dtype=BF16)
    row = []
    for c in cfgs:
        row.append(timed(lambda: ops.gemm_nt(A, Bm, M, N, K, C_out=C, c_bf16=True, variant=c << 4)))
    print(f'{M:6d}x{N:5d}x{K:5d} ' + ' '.join(f'{t:9.1f}' for t in row) + f'   best {2.0 * M * N * K / min(row) / 1e6:6.0f} TF')
