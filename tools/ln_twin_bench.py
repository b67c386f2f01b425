#!/usr/bin/env python3
"""LayerNorm backward: the fp32-source kernel (saved mean / rstd) against the twin-source form that also re-makes the LayerNorm output
(dav_layernorm_bwd_twin), alone on the GPU, rotating buffers, on the step's shapes.  Usage: python tools/ln_twin_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


print('shape                 fp32-source   twin-source(+h_out)   twin without h_out   [us]')
for (B, r0, r1, D) in [(64, 16, 49, 768), (64, 16, 64, 768), (64, 0, 49, 768), (64, 0, 64, 768), (64, 0, 352, 512), (64, 0, 228, 512), (64, 0, 16, 768)]:
    R = r0 + r1
    n0, n1 = (r0, r1) if r0 else (r1, 0)
    sets = []
    for _ in range(ROT):
        x0 = torch.randn(B, n0, D, device=dev)
        x1 = torch.randn(B, max(n1, 1), D, device=dev)[:, :n1].contiguous() if n1 else None
        d = dict(x0=x0, x1=x1, dy=torch.randn(B * R, D, device=dev).bfloat16(), res0=torch.randn(B, n0, D, device=dev),
                 dx0=torch.empty(B, n0, D, device=dev), tw0=torch.empty(B, n0, D, device=dev, dtype=torch.bfloat16),
                 dx1=torch.empty(B, max(n1, 1), D, device=dev), h=torch.empty(B * R, D, device=dev, dtype=torch.bfloat16),
                 mean=torch.zeros(B * R, device=dev), rstd=torch.ones(B * R, device=dev))
        for k, xs, n in (('t0', x0, n0), ('t1', x1, n1)):
            if xs is None:
                d[k] = (None, None)
                continue
            tw, st = torch.empty(B * n, D, device=dev, dtype=torch.bfloat16), torch.empty(B * n, D // 64, 2, device=dev)
            ops.rowstats_cast(xs, n * D, B, n, D, tw, st)
            d[k] = (tw, st)
        sets.append(d)
    g, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    it = [0]

    def old():
        d = sets[it[0] % ROT]; it[0] += 1
        ops.layernorm_bwd(d['x0'], n0 * D, n0, d['x1'], n1 * D, n1, B, D, d['dy'], None, g, d['mean'], d['rstd'],
                          d['dx0'], n0 * D, 0, d['res0'], n0 * D, d['tw0'], n0 * D, d['dx1'] if n1 else None, n1 * D, 0, None, 0, None, 0, dg, db, defer=[])

    def new(h=True):
        d = sets[it[0] % ROT]; it[0] += 1
        ops.layernorm_bwd_twin(d['t0'][0], n0 * D, d['t0'][1], n0, d['t1'][0], n1 * D, d['t1'][1], n1, B, D, 1e-6, d['dy'], None, g, bt,
                               d['dx0'], n0 * D, 0, d['res0'], n0 * D, d['tw0'], n0 * D, d['dx1'] if n1 else None, n1 * D, 0, None, 0, None, 0,
                               h_out=d['h'] if h else None, dgamma=dg, dbeta=db, defer=[])
    a, b, c = timeit(old, 4 * ROT), timeit(new, 4 * ROT), timeit(lambda: new(False), 4 * ROT)
    print(f'B{B} {r0:>3}+{r1:<3} D{D:<4}   {a:9.1f}   {b:9.1f} ({b / a:5.2f})   {c:9.1f} ({c / a:5.2f})', flush=True)
