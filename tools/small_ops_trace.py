#!/usr/bin/env python3
"""Which torch-side ops (copies, fills, adds ...) are left in the step: one eager forward + backward under torch.profiler,
aten ops grouped by the innermost deepavfusion_amd source line that issued them.  Usage: python tools/small_ops_trace.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae       # noqa: E402
from deepavfusion_amd.configs import CONFIGS               # noqa: E402
from deepavfusion_amd import autograd_bridge as bridge     # noqa: E402

cfg = CONFIGS['base']
model = build_avmae(cfg).cuda()
B = 64
image = torch.randn(B, 3, *cfg.image_size, device='cuda')
audio = torch.randn(B, 1, *cfg.audio_size, device='cuda')
Li, La = model.image_gs[0] * model.image_gs[1], model.audio_gs[0] * model.audio_gs[1]


def step():
    ni, na = torch.rand(B, Li, device='cuda'), torch.rand(B, La, device='cuda')
    outs, tape, aux = bridge.avmae_fwd(model, image, audio, ni, na)
    one = torch.ones((), device='cuda')
    bridge.avmae_bwd(model, tape, one, one)


step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if not e.name.startswith('aten::') or e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue
    if e.name in ('aten::empty', 'aten::view', 'aten::empty_like', 'aten::as_strided', 'aten::slice', 'aten::select', 'aten::reshape',
                  'aten::empty_strided', 'aten::_unsafe_view', 'aten::expand', 'aten::permute', 'aten::transpose', 'aten::unsqueeze'):
        continue
    where = '?'
    for fr in e.stack:
        if 'deepavfusion_amd' in fr and 'ops.py' not in fr:
            where = fr.split('deepavfusion_amd/')[-1]
            break
    cnt[(e.name, where)] += 1
for (name, where), n in sorted(cnt.items(), key=lambda x: -x[1])[:60]:
    print(f'{n:5d}  {name:28s} {where}')
