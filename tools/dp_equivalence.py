#!/usr/bin/env python3
"""N-GPU data-parallel equivalence (reference util/misc.py:32-34, util/distributed.py:66-100):
N ranks, each on its own GPU with its own slice of a global batch and the same injected masking noise, all-reduce-averaged
gradients  ==  one GPU on the concatenated batch.  Compared through what the optimizer sees: parameters after two AdamW steps
and the losses of both steps.  Needs >= 2 GPUs (skips with exit code 0 below that).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 tools/dp_equivalence.py
    (the bench itself: python bench.py --gpus 8   — starts its ranks itself —   or the same under torch.distributed.run)"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

world = int(os.environ.get('WORLD_SIZE', '1'))
rank = int(os.environ.get('RANK', '0'))
local = int(os.environ.get('LOCAL_RANK', '0'))
force1 = os.environ.get('DP_EQ_FORCE1') == '1'             # mechanics check on a 1-GPU box: a 1-rank RCCL group (DAV_FORCE_DIST=1)
if (world < 2 or torch.cuda.device_count() < 2) and not force1:
    if rank == 0:
        print('dp_equivalence: needs >= 2 GPUs under torch.distributed.run — skipped')
    sys.exit(0)
if force1:
    os.environ['DAV_FORCE_DIST'] = '1'
    os.environ.setdefault('MASTER_PORT', '29547')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
torch.cuda.set_device(local)
dist.init_process_group('nccl', init_method='env://', world_size=world, rank=rank)

from deepavfusion_amd.build_model import build_avmae       # noqa: E402
from deepavfusion_amd.configs import CONFIGS               # noqa: E402
from deepavfusion_amd.util import lr_sched                 # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW           # noqa: E402
from deepavfusion_amd.util.misc import Trainer             # noqa: E402

cfg = CONFIGS[os.environ.get('DP_EQ_CONFIG', 'tiny')]
B_PER, STEPS = 4, 2
dev = torch.device('cuda', local)


def make(distributed):
    torch.manual_seed(0)                                   # identical initial weights on every rank and in both arms
    model = build_avmae(cfg).to(dev)
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
    return model, opt, Trainer(model, optimizer=opt, accum_iter=1, distributed=distributed)


g = torch.Generator().manual_seed(99)                      # the GLOBAL batch and its masking noise, the same on every rank
Bg = B_PER * world
image = torch.randn(Bg, 3, *cfg.image_size, generator=g)
audio = (torch.randn(Bg, 1, *cfg.audio_size, generator=g) * 2.0 - 3.0).clamp(-7, 4)
Li = (cfg.image_size[0] // 16) * (cfg.image_size[1] // 16)
La = (cfg.audio_size[0] // 16) * (cfg.audio_size[1] // 16)
noise_i, noise_a = torch.rand(STEPS, Bg, Li, generator=g), torch.rand(STEPS, Bg, La, generator=g)
sl = slice(rank * B_PER, (rank + 1) * B_PER)

model, opt, tr = make(True)
dp_losses = []
for s in range(STEPS):
    li, la = tr.model(image[sl].to(dev), audio[sl].to(dev), noise_i[s, sl].to(dev), noise_a[s, sl].to(dev))[:2]
    tr.step(li + la)
    t = torch.stack([li.detach(), la.detach()])
    dist.all_reduce(t, op=dist.ReduceOp.AVG)               # equal mask counts per sample: the global masked mean = mean of the ranks'
    dp_losses.append(t.cpu())
torch.cuda.synchronize()
p_dp = opt.flat.flat_p.detach().clone()
ok = True
if rank == 0:
    model1, opt1, tr1 = make(False)
    for s in range(STEPS):
        li, la = tr1.model(image.to(dev), audio.to(dev), noise_i[s].to(dev), noise_a[s].to(dev))[:2]
        tr1.step(li + la)
        ref = torch.stack([li.detach(), la.detach()]).cpu()
        dl = float((dp_losses[s] - ref).abs().max() / ref.abs().max())
        print(f'step {s}: losses data-parallel {dp_losses[s].tolist()} vs one GPU {ref.tolist()}  rel {dl:.2e}')
        ok = ok and dl < 2e-3
    torch.cuda.synchronize()
    p1 = opt1.flat.flat_p.detach()
    rel = float((p_dp - p1).norm() / p1.norm())
    print(f'world {world}: parameters after {STEPS} steps, data-parallel vs one GPU on the concatenated batch: rel L2 {rel:.3e}')
    ok = ok and rel < 5e-4
    print('dp_equivalence:', 'OK' if ok else 'FAILED')
flag = torch.tensor([1 if ok else 0], device=dev)
dist.broadcast(flag, 0)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag) else 1)
