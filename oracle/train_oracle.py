"""CPU restatement of the host-side step logic around the hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, on plain Python /
torch-fp32-CPU:

* the per-iteration lr schedule           util/lr_sched.py:4-24
* the optimizer param-group split         train.py:89-93, util/lr_sched.py:77-93
                                          (+ timm optim_factory.param_groups_weight_decay, Appendix B)
* Trainer.backward/step semantics         util/misc.py:69-79, 96-136
* global grad norm                        util/misc.py:151-163
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch

from . import avmae_oracle as O


def lr_at(epoch: float, lr: float, warmup_epochs: float, epochs: float,
          pt_warmup_epochs: float = -1, pt_mult_start: float = 0., pt_mult_end: float = 1.):
    """util/lr_sched.py:4-24 -> (lr, lr for 'pretrained' groups)."""
    if epoch < warmup_epochs:
        cur = lr * epoch / warmup_epochs
    else:
        cur = lr * 0.5 * (1. + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))
    if epoch < pt_warmup_epochs:
        s = (0.5 - 0.5 * math.cos(math.pi * epoch / pt_warmup_epochs)) * (pt_mult_end - pt_mult_start) + pt_mult_start
    else:
        s = pt_mult_end
    return cur, cur * s


def param_groups(names_shapes: Dict[str, Sequence[int]], weight_decay: float,
                 image_pt, audio_pt, frozen=O.FROZEN) -> List[dict]:
    """Group assignment by NAME: returns 2 (+2 +2) groups in the order the
    reference builds them: [no_decay, decay] for the non-pretrained rest, then
    [no_decay, decay] of encoder.image (if image_pt is not None), then of
    encoder.audio (if audio_pt is not None).  The empty string still tags a
    tower as pretrained (util/lr_sched.py:83-86)."""
    def is_no_decay(n, shape):
        return len(shape) <= 1 or n.endswith('.bias') or ('bias' in n or 'norm' in n)
    names = [n for n in names_shapes if n not in frozen]
    pt_prefixes = []
    if image_pt is not None:
        pt_prefixes.append('encoder.image.')
    if audio_pt is not None:
        pt_prefixes.append('encoder.audio.')
    rest = [n for n in names if not any(n.startswith(p) for p in pt_prefixes)]
    groups = [dict(names=[n for n in rest if is_no_decay(n, names_shapes[n])], weight_decay=0., pretrained=False),
              dict(names=[n for n in rest if not is_no_decay(n, names_shapes[n])], weight_decay=weight_decay, pretrained=False)]
    for p in pt_prefixes:
        mine = [n for n in names if n.startswith(p)]
        groups.append(dict(names=[n for n in mine if is_no_decay(n, names_shapes[n])], weight_decay=0., pretrained=True))
        groups.append(dict(names=[n for n in mine if not is_no_decay(n, names_shapes[n])], weight_decay=weight_decay, pretrained=True))
    return groups


def global_grad_norm(grads: Sequence[torch.Tensor]) -> float:
    """util/misc.py:151-163 — L2 norm of the per-tensor L2 norms."""
    return float(torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0))


class OracleTrainer:
    """Pre-training loop state on the CPU oracle: fp32 weights as leaf tensors,
    torch.optim.AdamW(betas=(0.9, 0.95)) (train.py:93), accumulation and
    grad-norm reporting as util/misc.py:69-136 (no GradScaler: the build runs
    bf16 without loss scaling, the reference's scaler is disabled on CPU)."""

    def __init__(self, cfg: O.PathConfig, sd: Dict[str, torch.Tensor], lr: float, weight_decay: float = 0.05,
                 accum_iter: int = 1, image_pt='', audio_pt=''):
        self.cfg = cfg
        self.sd = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd.items()}
        shapes = {k: tuple(v.shape) for k, v in sd.items()}
        self.groups = param_groups(shapes, weight_decay, image_pt, audio_pt)
        self.opt = torch.optim.AdamW(
            [dict(params=[self.sd[n] for n in g['names']], weight_decay=g['weight_decay'], pretrained=g['pretrained'])
             for g in self.groups], lr=lr, betas=(0.9, 0.95))
        self.accum_iter, self.accums, self.n_steps = accum_iter, 0, 0

    def set_lr(self, epoch: float, lr: float, warmup_epochs: float, epochs: float, pt_warmup_epochs: float = -1):
        cur, cur_pt = lr_at(epoch, lr, warmup_epochs, epochs, pt_warmup_epochs)
        for g in self.opt.param_groups:
            g['lr'] = cur_pt if g.get('pretrained', False) else cur
        return cur

    def forward(self, image, audio, noise_i, noise_a):
        return O.avmae_forward(self.sd, self.cfg, image, audio, noise_i, noise_a)

    def step(self, loss: torch.Tensor) -> float:
        loss.backward()
        self.accums += 1
        grads = [p.grad for p in self.sd.values() if p.grad is not None]
        norm = global_grad_norm(grads) / self.accums
        if self.accums == self.accum_iter:
            if self.accum_iter > 1:
                for p in self.sd.values():
                    if p.grad is not None:
                        p.grad /= self.accum_iter
            self.opt.step()
            self.opt.zero_grad()
            self.accums = 0
            self.n_steps += 1
        return norm
