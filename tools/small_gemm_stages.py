#!/usr/bin/env python3
"""Latency of the fusion block's tiny GEMMs (a few dozen 64x64 tiles, nothing else on the GPU) per ring depth:
cfg 5 / 6 / 7 = 64x64 tiles with a 2- / 3- / 4-stage ring.  hipGraph replays, rotating operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import ops  # noqa: E402

dev, bf = 'cuda', torch.bfloat16


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin()
        for _ in range(reps):
            fn()
        g.capture_end()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for (M, N, K) in [(1024, 192, 768), (1024, 768, 192), (512, 768, 768), (512, 192, 768), (2048, 768, 768), (1024, 768, 768)]:
    sets = [(torch.randn(M, K, device=dev).to(bf), (torch.randn(N, K, device=dev) * 0.05).to(bf), torch.empty(M, N, device=dev, dtype=bf))
            for _ in range(4)]
    cells = []
    for cfg in [int(a) for a in sys.argv[1:]] or [5, 6, 7]:
        it = [0]

        def fn():
            A, W, C = sets[it[0] % 4]
            it[0] += 1
            ops.gemm_nt(A, W, M, N, K, C_out=C, c_bf16=True, variant=cfg << 4)
        cells.append(f'cfg{cfg}: {timed(fn):6.1f} us')
    print(f'{M}x{N}x{K}: ' + '  '.join(cells), flush=True)
