#!/usr/bin/env python3
"""Where do the step's device-to-device copies and fills come from?  One eager step (engine path, as GraphedStep captures it) under
torch.profiler with Python stacks; prints the call sites of every aten::copy_ / clone / fill_ / zero_ with bytes and counts."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae           # noqa: E402
from deepavfusion_amd.configs import CONFIGS                   # noqa: E402
from deepavfusion_amd.util import lr_sched                     # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW               # noqa: E402
from deepavfusion_amd.util.misc import GraphedStep, Trainer    # noqa: E402

dev = torch.device('cuda', 0)
cfg = CONFIGS['base']
B = 64
torch.manual_seed(0)
model = build_avmae(cfg).to(dev)
nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
opt = FlatAdamW(groups, lr=1e-4, betas=(0.9, 0.95), model=model)
tr = Trainer(model, optimizer=opt, accum_iter=1, use_amp=True, distributed=False)
image = torch.randn(B, 3, *cfg.image_size, device=dev)
audio = torch.randn(B, 1, *cfg.audio_size, device=dev)
gs = GraphedStep(tr, image.shape, audio.shape)      # (warm-up + capture: allocator warm)
gs.image.copy_(image); gs.audio.copy_(audio)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    gs._fwd_bwd(None)
    torch.cuda.synchronize()
sites = collections.Counter()
nbytes = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::cat', 'aten::sum', 'aten::add_', 'aten::rand', 'aten::ones', 'aten::expand'):
        frames = [f for f in (ev.stack or []) if 'deepavfusion_amd' in f or 'tools/' in f]
        site = frames[0].split('deepavfusion_amd/')[-1] if frames else '?'
        shapes = str(ev.input_shapes)[:60]
        sites[(ev.name, site, shapes)] += 1
for (name, site, shapes), n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print(f'{n:5d}  {name:18s} {site[:90]:90s} {shapes}')
