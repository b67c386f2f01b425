#!/usr/bin/env python3
"""Stability probe: N captured steps of the bench workload; loss must stay finite and fall, device memory must not grow.
Usage: python tests/long_run_probe.py [config] [steps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from deepavfusion_amd.util import lr_sched                            # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW                      # noqa: E402
from deepavfusion_amd.util.misc import GraphedStep, Trainer           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'base'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
cfg = CONFIGS[name]
torch.manual_seed(0)
model = build_avmae(cfg).cuda()
nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
opt = FlatAdamW(groups, lr=1.5e-4, betas=(0.9, 0.95), model=model)
tr = Trainer(model, optimizer=opt, accum_iter=1)
g = torch.Generator(device='cuda'); g.manual_seed(1)
# a learnable signal: low-frequency images / spectrograms (pure noise has nothing to reconstruct)
yy, xx = torch.meshgrid(torch.linspace(0, 6.28, cfg.image_size[0], device='cuda'), torch.linspace(0, 6.28, cfg.image_size[1], device='cuda'), indexing='ij')
ph = torch.rand(B, 3, 1, 1, device='cuda', generator=g) * 6.28
image = torch.sin(yy * 2 + ph) + torch.cos(xx * 3 + ph) + 0.05 * torch.randn(B, 3, *cfg.image_size, device='cuda', generator=g)
ya, xa = torch.meshgrid(torch.linspace(0, 6.28, cfg.audio_size[0], device='cuda'), torch.linspace(0, 25.0, cfg.audio_size[1], device='cuda'), indexing='ij')
pa = torch.rand(B, 1, 1, 1, device='cuda', generator=g) * 6.28
audio = (2 * torch.sin(ya * 2 + xa + pa) - 3).clamp(-7, 4)
gs = GraphedStep(tr, image.shape, audio.shape)
torch.cuda.synchronize()
mem0 = torch.cuda.memory_allocated()
trace = torch.zeros(steps, 3, device='cuda')
for s in range(steps):
    li, la, gn = gs(image, audio)
    trace[s, 0].copy_(li.reshape(())); trace[s, 1].copy_(la.reshape(())); trace[s, 2].copy_(gn.reshape(()))
torch.cuda.synchronize()
t = trace.cpu()
print(f'{name} B={B}: {steps} steps; loss {float(t[0, 0] + t[0, 1]):.4f} -> {float(t[-1, 0] + t[-1, 1]):.4f}; '
      f'all finite {bool(torch.isfinite(t).all())}; grad norm first/last {float(t[0, 2]):.3f}/{float(t[-1, 2]):.3f}; '
      f'memory growth {(torch.cuda.memory_allocated() - mem0) / 2 ** 20:.1f} MiB, peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB')
for k in range(0, steps, max(1, steps // 10)):
    print(f'   step {k:4d}: {float(t[k, 0] + t[k, 1]):.4f}  gnorm {float(t[k, 2]):.3f}')
