#!/usr/bin/env python3
"""How long does the HOST spend in one replay of the captured step (hipGraphLaunch of ~1000 nodes), and does it run ahead of the
GPU?  Usage: python tools/replay_host_time.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from deepavfusion_amd.util import lr_sched                            # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW                      # noqa: E402
from deepavfusion_amd.util.misc import GraphedStep, Trainer           # noqa: E402

cfg = CONFIGS['base']
model = build_avmae(cfg).cuda()
nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
opt = FlatAdamW(groups, lr=1e-4, betas=(0.9, 0.95), model=model)
tr = Trainer(model, optimizer=opt, accum_iter=1)
B = 64
image = torch.randn(B, 3, *cfg.image_size, device='cuda')
audio = torch.randn(B, 1, *cfg.audio_size, device='cuda')
gs = GraphedStep(tr, image.shape, audio.shape)
for _ in range(3):
    gs(image, audio)
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(20):
    t0 = time.perf_counter()
    gs(image, audio)
    host.append((time.perf_counter() - t0) * 1e3)
t_issue = (time.perf_counter() - t_all) * 1e3
torch.cuda.synchronize()
t_total = (time.perf_counter() - t_all) * 1e3
print('host ms per replay call:', ' '.join(f'{h:.2f}' for h in host))
print(f'20 replays: issued in {t_issue:.1f} ms, finished after {t_total:.1f} ms ({t_total / 20:.2f} ms per step)')
g = gs.graphs[0]
t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'bare graph replay: host {1e3 * (t1 - t0):.2f} ms, until done {1e3 * (t2 - t0):.2f} ms')
