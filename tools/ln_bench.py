#!/usr/bin/env python3
"""Isolated timing of the LayerNorm kernels on the step's shapes, over launch-geometry knobs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda')
lib = _lib.load()


def timeit(fn, reps=20):
    """GPU time per call with the host out of the way: capture `reps` calls in a hipGraph and replay it."""
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


for (B, r0, r1, D) in [(64, 32, 49, 768), (64, 32, 63, 768), (64, 0, 49, 768), (64, 0, 352, 512), (64, 0, 228, 512)]:
    R = r0 + r1
    x0 = torch.randn(B, max(r0, 1), D, device=dev)[:, :r0].contiguous() if r0 else None
    x1 = torch.randn(B, r1, D, device=dev)
    g, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    a0, a1, n0, n1 = (x0, x1, r0, r1) if r0 else (x1, None, r1, 0)
    y = torch.empty(B * R, D, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(B * R, device=dev), torch.empty(B * R, device=dev)
    fwd = lambda: ops.layernorm_fwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, g, bt, 1e-6, y, None, mean, rstd)
    t_f = timeit(fwd)
    dy = torch.randn(B * R, D, device=dev).bfloat16()
    dx0 = torch.zeros(B, n0, D, device=dev); tw0 = torch.empty(B, n0, D, device=dev, dtype=torch.bfloat16)
    res0 = torch.randn(B, n0, D, device=dev)
    dx1 = torch.zeros(B, max(n1, 1), D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    bytes_b = B * R * D * (4 + 2 + 4 + 4 + 2)
    row = f'B{B} {r0}+{r1} D{D}: fwd {t_f:6.1f}us ({B * R * D * 6 / t_f / 1e6:5.2f} TB/s) | bwd'
    for waves in (2, 4, 8):
        for cap in (256, 512, 1024, 2048):
            lib.dav_tune(1, waves); lib.dav_tune(2, cap)
            bwd = lambda: ops.layernorm_bwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, dy, None, g, mean, rstd,
                                            dx0, n0 * D, 0, res0, n0 * D, tw0, n0 * D, dx1 if n1 else None, n1 * D, 0, None, 0, None, 0, dg, db,
                                            defer=[] if os.environ.get('LN_NO_REDUCE') else None)
            t = timeit(bwd)
            row += f' w{waves}c{cap}:{t:5.1f}'
    print(row + f'  (ideal {bytes_b / 5e6:.1f}us @5TB/s)', flush=True)
