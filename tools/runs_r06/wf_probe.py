import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import test_hip_parity as T
from deepavfusion_amd.util import lr_sched
from deepavfusion_amd.util.flat import FlatAdamW
from deepavfusion_amd.util.misc import GraphedStep, Trainer
for rep in range(3):
    finals = []
    for mode in ('1', '0'):
        os.environ['DAV_WGRAD_OVERWRITE'] = mode
        model, sd, cfg, O = T._build('micro')
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
        tr = Trainer(model, optimizer=opt, accum_iter=1)
        image, audio, _, _ = O.structured_batch(cfg, 64, seed=3)
        image, audio = image.cuda(), audio.cuda()
        torch.manual_seed(77)
        gs = GraphedStep(tr, image.shape, audio.shape)
        for s in range(6):
            torch.manual_seed(500 + s)
            if s == 3 and os.environ.get('NO_EAGER') != '1':
                li, la = tr.model(image, audio)[:2]
                tr.step(li + la)
            else:
                li, la, gn = gs(image, audio)
        torch.cuda.synchronize()
        finals.append(opt.flat.flat_p.clone())
    d = (finals[0] - finals[1]).abs()
    names = [n for n, p in model.named_parameters()]
    print('rel', T.rel(finals[0], finals[1]), 'max abs diff', float(d.max()))
    # which parameters differ most
    off = 0
    worst = []
    for p, o in zip(opt.flat.params, opt.flat.offsets):
        n = p.numel()
        e = float((finals[0][o:o+n] - finals[1][o:o+n]).double().norm())
        worst.append((e, [k for k, v in model.named_parameters() if v is p][0]))
    print(sorted(worst, reverse=True)[:6])
