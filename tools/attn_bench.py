#!/usr/bin/env python3
"""Times dav_attn_fwd / dav_attn_bwd on the path's shapes (hipGraph replay of `reps` launches).
Usage: python tools/attn_bench.py [name ...]   names: enc_img enc_aud dec_img dec_aud pair video eval_aud"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: E402,F401
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
BF16 = torch.bfloat16
SHAPES = {   # B, H, Nq, Nk, dqk, dv, query row offset in the fused buffer
    'enc_img': (64, 12, 49, 81, 64, 64, 32),
    'enc_aud': (64, 12, 63, 95, 64, 64, 32),
    'dec_img': (64, 16, 228, 228, 32, 32, 0),
    'dec_aud': (64, 16, 352, 352, 32, 32, 0),
    'video': (16, 12, 784, 816, 64, 64, 32),
    'video_aud': (16, 12, 96, 128, 64, 64, 32),
    'eval_aud': (64, 12, 320, 352, 64, 64, 32),
    'long': (4, 12, 4096, 4096, 64, 64, 0),
}


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
    return e0.elapsed_time(e1) * 1e3 / reps      # us


def main():
    for kv in os.environ.get('DAV_TUNE', '').split(','):
        if ':' in kv:
            from deepavfusion_amd import _lib
            _lib.load().dav_tune(int(kv.split(':')[0]), int(kv.split(':')[1]))
    names = sys.argv[1:] or list(SHAPES)
    for n in names:
        B, H, Nq, Nk, dqk, dv, off = SHAPES[n]
        buf = torch.randn(B, Nk, 3, H, dqk, device=dev).to(BF16)
        dbuf = torch.zeros_like(buf)
        O = torch.empty(B * Nq, H * dv, device=dev, dtype=BF16)
        dO = torch.randn(B * Nq, H * dv, device=dev).to(BF16)
        LSE = torch.empty(B, H, Nq, device=dev); Delta = torch.empty_like(LSE)
        st = (Nk * 3 * H * dqk, 3 * H * dqk) * 3
        e = 2
        q, k, v = buf.data_ptr() + e * off * 3 * H * dqk, buf.data_ptr() + e * H * dqk, buf.data_ptr() + e * 2 * H * dqk
        dq, dk, dvp = dbuf.data_ptr() + e * off * 3 * H * dqk, dbuf.data_ptr() + e * H * dqk, dbuf.data_ptr() + e * 2 * H * dqk
        scale = dqk ** -0.5
        f = lambda: ops.attn_fwd(q, k, v, O, LSE, B, H, Nq, Nk, dqk, dv, *st, Nq * H * dv, H * dv, scale)
        bw = lambda: ops.attn_bwd(q, k, v, O, dO, LSE, Delta, dq, dk, dvp, B, H, Nq, Nk, dqk, dv, *st,
                                  Nq * H * dv, H * dv, Nq * H * dv, H * dv, *st, scale)
        tf, tb = timed(f), timed(bw)
        fl = 2.0 * B * H * Nq * Nk * (dqk + dv)
        print(f'{n:10s} B{B} H{H} {Nq}x{Nk} d{dqk}: fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF   bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:7.1f} TF')


if __name__ == '__main__':
    main()
