#!/usr/bin/env python3
"""Would the two attention-backward kernels (dQ, dK/dV) of one attention run faster side by side than one after the other?  They depend on each
other only through Delta (dO . O per query row, written by the dQ kernel, read by the dK/dV kernel); neither is bound by a pipe.
Decoder shapes (B = 64, 16 heads of 32, 352 / 228 rows) and a tower shape; captured graphs, HIP events, Delta from a previous pass."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16


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


side = torch.cuda.Stream()
for (B, H, N, d, nF) in [(64, 16, 352, 32, 0), (64, 16, 228, 32, 0), (64, 12, 65, 64, 16)]:
    R = N
    qkv = torch.randn(B, R, 3, H, d, device=dev).to(BF16)
    nq = R - nF
    O = torch.empty(B * nq, H * d, device=dev, dtype=BF16)
    LSE = torch.empty(B, H, nq, device=dev)
    st = (R * 3 * H * d, 3 * H * d) * 3
    p = lambda t, off: t.data_ptr() + 2 * off
    ops.attn_fwd(p(qkv, nF * 3 * H * d), p(qkv, H * d), p(qkv, 2 * H * d), O, LSE, B, H, nq, R, d, d, *st, nq * H * d, H * d, d ** -0.5)
    dO = torch.randn(B * nq, H * d, device=dev).to(BF16)
    dqkv = torch.zeros_like(qkv)
    Delta, Delta2 = torch.empty_like(LSE), torch.empty_like(LSE)

    def bwd(part, delta):
        ops.attn_bwd(p(qkv, nF * 3 * H * d), p(qkv, H * d), p(qkv, 2 * H * d), O, dO, LSE, delta, p(dqkv, nF * 3 * H * d), p(dqkv, H * d),
                     p(dqkv, 2 * H * d), B, H, nq, R, d, d, *st, nq * H * d, H * d, nq * H * d, H * d, *st, d ** -0.5, part=part)
    bwd(1, Delta)          # Delta of this problem, for the dK/dV kernel

    def serial():
        bwd(1, Delta2)
        bwd(2, Delta)

    def parallel():
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            bwd(2, Delta)
        bwd(1, Delta2)
        main.wait_stream(side)
    t1, t2 = timed(lambda: bwd(1, Delta2)), timed(lambda: bwd(2, Delta))
    ts, tp = timed(serial), timed(parallel)
    print(f'B{B} H{H} {nq}x{R} d{d}: dQ {t1:6.1f} us, dK/dV {t2:6.1f} us; one after the other {ts:6.1f} us, side by side {tp:6.1f} us', flush=True)
