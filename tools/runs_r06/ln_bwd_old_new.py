#!/usr/bin/env python3
"""fp32-source LayerNorm backward + forward, this tree's library against another build (DAV_BENCH_LIB), rotating buffers."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tools._libsel  # noqa: E402,F401
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
ROT = 6


def timeit(fn, reps):
    for _ in range(ROT):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


print('library:', os.environ.get('DAV_BENCH_LIB', 'tree'))
for (B, r0, r1, D) in [(64, 16, 49, 768), (64, 0, 64, 768), (64, 0, 352, 512), (64, 0, 228, 512), (64, 0, 16, 768)]:
    R = r0 + r1
    n0, n1 = (r0, r1) if r0 else (r1, 0)
    sets = []
    for _ in range(ROT):
        sets.append(dict(x0=torch.randn(B, n0, D, device=dev), x1=torch.randn(B, max(n1, 1), D, device=dev)[:, :n1].contiguous() if n1 else None,
                         dy=torch.randn(B * R, D, device=dev).bfloat16(), res0=torch.randn(B, n0, D, device=dev), dx0=torch.empty(B, n0, D, device=dev),
                         tw0=torch.empty(B, n0, D, device=dev, dtype=torch.bfloat16), dx1=torch.empty(B, max(n1, 1), D, device=dev),
                         y=torch.empty(B * R, D, device=dev, dtype=torch.bfloat16), mean=torch.zeros(B * R, device=dev), rstd=torch.ones(B * R, device=dev)))
    g, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    it = [0]

    def bwd():
        d = sets[it[0] % ROT]; it[0] += 1
        ops.layernorm_bwd(d['x0'], n0 * D, n0, d['x1'], n1 * D, n1, B, D, d['dy'], None, g, d['mean'], d['rstd'],
                          d['dx0'], n0 * D, 0, d['res0'], n0 * D, d['tw0'], n0 * D, d['dx1'] if n1 else None, n1 * D, 0, None, 0, None, 0, dg, db, defer=[])

    def fwd():
        d = sets[it[0] % ROT]; it[0] += 1
        ops.layernorm_fwd(d['x0'], n0 * D, n0, d['x1'], n1 * D, n1, B, D, g, bt, 1e-6, d['y'], None, d['mean'], d['rstd'])
    print(f'B{B} {r0:>3}+{r1:<3} D{D:<4}  bwd {timeit(bwd, 4 * ROT):7.1f} us   fwd {timeit(fwd, 4 * ROT):7.1f} us', flush=True)
