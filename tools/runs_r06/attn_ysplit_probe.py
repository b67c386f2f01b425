#!/usr/bin/env python3
"""Resident attention kernels with y workgroups per (batch, head) (dav_tune knob 4): forward / dQ / dK-dV alone at the decoder shapes and a
tower shape, captured graphs, checked against y = 1 bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepavfusion_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16
lib = _lib.load()


def timed(fn, reps=40):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 4 * 1e3


for (B, H, N, d, nF) in [(64, 16, 352, 32, 0), (64, 16, 228, 32, 0), (64, 12, 65, 64, 16), (32, 16, 352, 32, 0)]:
    R = N
    torch.manual_seed(0)
    qkv = torch.randn(B, R, 3, H, d, device=dev).to(BF16)
    nq = R - nF
    st = (R * 3 * H * d, 3 * H * d) * 3
    p = lambda t, off: t.data_ptr() + 2 * off
    dO = torch.randn(B * nq, H * d, device=dev).to(BF16)
    ref = None
    line = f'B{B} H{H} {nq}x{R} d{d}:'
    for ys in (1, 2, 3, 4):
        _lib.check(lib.dav_tune(4, ys), 'dav_tune')
        O = torch.empty(B * nq, H * d, device=dev, dtype=BF16)
        LSE = torch.empty(B, H, nq, device=dev)
        dqkv = torch.zeros_like(qkv)
        Delta = torch.empty_like(LSE)
        fwd = lambda: ops.attn_fwd(p(qkv, nF * 3 * H * d), p(qkv, H * d), p(qkv, 2 * H * d), O, LSE, B, H, nq, R, d, d, *st, nq * H * d, H * d, d ** -0.5)
        bwd = lambda part: ops.attn_bwd(p(qkv, nF * 3 * H * d), p(qkv, H * d), p(qkv, 2 * H * d), O, dO, LSE, Delta, p(dqkv, nF * 3 * H * d), p(dqkv, H * d),
                                        p(dqkv, 2 * H * d), B, H, nq, R, d, d, *st, nq * H * d, H * d, nq * H * d, H * d, *st, d ** -0.5, part=part,
                                        dq_ctx_rows=nF)
        fwd(); bwd(3)
        torch.cuda.synchronize()
        got = (O.clone(), LSE.clone(), dqkv.clone())
        if ref is None:
            ref = got
        same = all(torch.equal(a, b) for a, b in zip(got, ref))
        tf, t1, t2 = timed(fwd), timed(lambda: bwd(1)), timed(lambda: bwd(2))
        line += f'  y={ys}: fwd {tf:5.1f} dQ {t1:5.1f} dK/dV {t2:5.1f} us{"" if same else " (DIFFERS)"}'
    print(line, flush=True)
_lib.check(lib.dav_tune(4, 0), 'dav_tune')
