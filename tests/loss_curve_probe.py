#!/usr/bin/env python3
"""The full 1k-step ViT-Tiny loss curve of the reference (tests/golden/curve_tiny.npz, fp32 CPU, same data / masking noise /
lr schedule) against the bf16 HIP path: per-step and smoothed deviations.  The GPU test runs a 40-step prefix of this."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from deepavfusion_amd.util import lr_sched                            # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW                      # noqa: E402
from deepavfusion_amd.util.misc import Trainer                        # noqa: E402
from oracle import avmae_oracle as O                                  # noqa: E402
from oracle.configs import CONFIGS as OC                              # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'curve_tiny.npz'))
cfg = OC['tiny']
model = build_avmae(CONFIGS['tiny']).cuda()
model.load_state_dict(O.closed_form_state(cfg, 0), strict=True)
nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
lr, B, spe = float(g['lr']), int(g['B']), int(g['steps_per_epoch'])
opt = FlatAdamW(groups, lr=lr, betas=(0.9, 0.95), model=model)
tr = Trainer(model, optimizer=opt, accum_iter=1)


class NS(dict):
    __getattr__ = dict.__getitem__


n_total = len(g['loss_image'])
steps = int(sys.argv[1]) if len(sys.argv) > 1 else n_total
args = NS(opt=NS(lr=lr, warmup_epochs=1, epochs=n_total // spe, pt_warmup_epochs=f'{n_total // spe}/2', pt_lr_mult_start=0, pt_lr_mult_end=1))
got = []
for s in range(steps):
    lr_sched.adjust_learning_rate(opt, s / spe, args)
    image, audio, ni, na = O.structured_batch(cfg, B, seed=10_000 + s)
    li, la = tr.model(image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda())[:2]
    tr.step(li + la)
    got.append(float(li) + float(la))
got = np.array(got)
ref = (g['loss_image'] + g['loss_audio'])[:steps]
dev = np.abs(got - ref) / ref
k = 25
sm = lambda x: np.convolve(x, np.ones(k) / k, mode='valid')
devs = np.abs(sm(got) - sm(ref)) / sm(ref)
print(f'{steps} steps: loss {ref[0]:.4f} -> ref {ref[-1]:.4f} / hip {got[-1]:.4f}; per-step deviation max {dev.max() * 100:.2f} % '
      f'(at step {int(dev.argmax())}), mean {dev.mean() * 100:.3f} %; {k}-step moving average: max {devs.max() * 100:.2f} %')
for a in range(0, steps, max(1, steps // 10)):
    print(f'   step {a:4d}: ref {ref[a]:.4f} hip {got[a]:.4f}  ({dev[a] * 100:.2f} %)')
