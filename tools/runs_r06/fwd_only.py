#!/usr/bin/env python3
"""Forward alone (no backward follows: eval, kNN probe, encoder-only inference) with the LayerNorms folded into the GEMMs (what
DAV_LN_FUSE=auto picks there) against the LayerNorm kernels: AVMAE forward of the bench workload under no_grad, replayed hipGraph."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepavfusion_amd import autograd_bridge as bridge  # noqa: E402
from deepavfusion_amd import engine as E  # noqa: E402
from deepavfusion_amd.build_model import build_avmae  # noqa: E402
from deepavfusion_amd.configs import CONFIGS  # noqa: E402

cfg = CONFIGS['base']
B = 64
torch.manual_seed(0)
model = build_avmae(cfg).cuda()
image = torch.randn(B, 3, *cfg.image_size, device='cuda')
audio = torch.randn(B, 1, *cfg.audio_size, device='cuda')
Li, La = model.image_gs[0] * model.image_gs[1], model.audio_gs[0] * model.audio_gs[1]
ni, na = torch.rand(B, Li, device='cuda'), torch.rand(B, La, device='cuda')
res = {}
for rnd in range(2):
    for mode in ('off', 'on'):
        E.set_ln_fuse(mode)
        with torch.no_grad():
            for _ in range(3):
                bridge.avmae_fwd(model, image, audio, ni, na)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    out = bridge.avmae_fwd(model, image, audio, ni, na)[0]
            torch.cuda.synchronize()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
        print(f'round {rnd}: LayerNorms {"folded into the GEMMs" if mode == "on" else "as kernels":22s} forward {ms:7.3f} ms   loss {float(out[0]):.5f} {float(out[1]):.5f}', flush=True)
        del g
