#!/usr/bin/env python3
"""Forward attention with one / two query tiles per wave (dav_tune knob 3) alone at the decoder shapes, captured graphs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepavfusion_amd import _lib, ops  # noqa: E402

dev = torch.device('cuda')
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


for (B, H, N, d) in [(64, 16, 352, 32), (64, 16, 228, 32), (64, 12, 65, 64)]:
    qkv = torch.randn(B, N, 3, H, d, device=dev).to(torch.bfloat16)
    st = (N * 3 * H * d, 3 * H * d) * 3
    O = torch.empty(B * N, H * d, device=dev, dtype=torch.bfloat16)
    LSE = torch.empty(B, H, N, device=dev)
    fwd = lambda: ops.attn_fwd(qkv.data_ptr(), qkv.data_ptr() + 2 * H * d, qkv.data_ptr() + 4 * H * d, O, LSE, B, H, N, N, d, d, *st, N * H * d, H * d, d ** -0.5)
    line = f'B{B} H{H} {N}x{N} d{d}:'
    for qt in (1, 2):
        _lib.check(lib.dav_tune(3, qt), 'dav_tune')
        line += f'  {qt} tile(s) per wave: {timed(fwd):5.1f} us'
    print(line, flush=True)
_lib.check(lib.dav_tune(3, 0), 'dav_tune')
