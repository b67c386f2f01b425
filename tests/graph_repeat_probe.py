#!/usr/bin/env python3
"""Repeatability probe for the captured step: same weights, data and seeds, several fresh builds of each mode;
prints per-step (loss_image, loss_audio, grad_norm).  Diagnostic companion of tests/test_hip_parity.py."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import engine                                   # noqa: E402
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from deepavfusion_amd.util import lr_sched                            # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW                      # noqa: E402
from deepavfusion_amd.util.misc import GraphedStep, Trainer           # noqa: E402
from oracle import avmae_oracle as O                                  # noqa: E402
from oracle.configs import CONFIGS as OC                              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'micro'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
for segments, distributed in ((1, False), (2, True)):
    runs = []
    for rep in range(reps):
        model = build_avmae(CONFIGS[name]).cuda()
        model.load_state_dict(O.closed_form_state(OC[name], 0), strict=True)
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1, distributed=distributed, bucket_mb=0.5, first_bucket_mb=0.25)
        image, audio, _, _ = O.structured_batch(OC[name], B, seed=3)
        gs = GraphedStep(tr, image.shape, audio.shape, segments=segments)
        run = []
        for s in range(5):
            torch.manual_seed(500 + s)
            li, la, gn = gs(image.cuda(), audio.cuda())
            run.append((float(li), float(la), float(gn)))
        runs.append(run)
        engine.set_grad_ready_hook(None)
    print(f'--- segments={segments} distributed={distributed}')
    for s in range(5):
        vals = [r[s] for r in runs]
        same = all(v == vals[0] for v in vals)
        print(f'  step {s}: ' + ('identical ' if same else 'DIFFER    ') + ' | '.join(f'{v[0]:.6f} {v[1]:.6f} {v[2]:.6f}' for v in vals))
