#!/usr/bin/env python3
"""Do the two MAE decoders overlap in a replayed hipGraph WITHOUT a profiler attached?  Forward of the bench workload's two decoders
(ViT-B encoder width 768 -> decoder 512 x 8 blocks, B = 64: 228 and 352 rows per sample) from fixed encoder outputs, captured
 (a) each alone, (b) both on two streams with a cross-join every k decoder blocks (k = 0: the two unjoined branches of rounds 1-5).
overlap = (alone_i + alone_a - both) / min(alone_i, alone_a): 0 = one after the other, 1 = the shorter one entirely hidden."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepavfusion_amd import autograd_bridge as bridge  # noqa: E402
from deepavfusion_amd import engine as E  # noqa: E402
from deepavfusion_amd.build_model import build_avmae  # noqa: E402
from deepavfusion_amd.configs import CONFIGS  # noqa: E402

cfg = CONFIGS[sys.argv[1] if len(sys.argv) > 1 else 'base']
B = 32 if cfg.embed_dim >= 1024 else 64
torch.manual_seed(0)
model = build_avmae(cfg).cuda()
dev = torch.device('cuda')
enc = model.encoder
nF = enc.fusion_tokens.shape[1]
D = enc.fusion_tokens.shape[2]
Li, La = model.image_gs[0] * model.image_gs[1], model.audio_gs[0] * model.audio_gs[1]
nki, nka = int(Li * (1 - model.image_mask_ratio)), int(La * (1 - model.audio_mask_ratio))
xi_b = torch.randn(B * nki, D, device=dev).to(E.BF16)
xa_b = torch.randn(B * nka, D, device=dev).to(E.BF16)
xf_b = torch.randn(B * nF, D, device=dev).to(E.BF16)
ir32 = torch.stack([torch.randperm(Li, device=dev) for _ in range(B)]).to(torch.int32)
ar32 = torch.stack([torch.randperm(La, device=dev) for _ in range(B)]).to(torch.int32)
dec_i, dec_a = model.decoder('image'), model.decoder('audio')


def timed(fn, reps=30):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
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
    return e0.elapsed_time(e1) / reps


def both(k):
    def fn():
        main, sa, _ = bridge._streams(dev)
        sa.wait_stream(main)
        bridge.paired_steps(E.decoder_fwd_steps(dec_a, xa_b, xf_b, ar32, B, nka, nF), sa,
                            E.decoder_fwd_steps(dec_i, xi_b, xf_b, ir32, B, nki, nF), main, k)
        main.wait_stream(sa)
    return fn


def lanes():
    with E.batch() as bt:
        bt.lane()
        E.decoder_fwd(dec_i, xi_b, xf_b, ir32, B, nki, nF)
        bt.lane()
        E.decoder_fwd(dec_a, xa_b, xf_b, ar32, B, nka, nF)


for rnd in range(2):
    ti = timed(lambda: E.decoder_fwd(dec_i, xi_b, xf_b, ir32, B, nki, nF))
    ta = timed(lambda: E.decoder_fwd(dec_a, xa_b, xf_b, ar32, B, nka, nF))
    line = f'round {rnd}: image decoder alone {ti:6.3f} ms, audio decoder alone {ta:6.3f} ms;  both, join every k blocks:'
    for k in (0, 1, 2, 4):
        tb = timed(both(k))
        line += f'  k={k}: {tb:6.3f} ms (overlap {(ti + ta - tb) / min(ti, ta):4.2f})'
    line += f'  as lanes of one launch batch (grouped grids): {timed(lanes):6.3f} ms'
    print(line, flush=True)
