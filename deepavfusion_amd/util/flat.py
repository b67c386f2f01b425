"""Flat fp32 parameter / gradient / optimizer-state buffers and the fused AdamW on them.

One contiguous fp32 buffer holds every trainable parameter (``p.data`` become views), a second one
every gradient (``p.grad`` views: the wgrad GEMMs accumulate straight into it), so that
  * zero_grad is one memset, the global grad norm one reduction (K11 in SURVEY.md section 2c),
  * the data-parallel all-reduce runs on contiguous bucket slices with no flatten/unflatten copies,
  * AdamW (train.py:93: torch.optim.AdamW(betas=(0.9, 0.95))) is ONE kernel (dav_adamw_flat) that also
    could refresh the bf16 weight mirror.
Parameters are laid out in REVERSE registration order (decoders first, patch embeds last): that is
the order in which the hand-written backward finishes their gradients, so leading buckets are
complete early and their all-reduce overlaps the rest of the backward.
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional

import torch

from .. import engine, ops

ALIGN = 64   # elements (256 B): keeps every view 16-byte aligned for float4 / DMA access


class FlatParams:
    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('no trainable parameters')
        dev = self.params[0].device          # CPU is accepted for the host-logic / gloo tests; kernels need cuda
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            v = self.flat_p[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self.seg_end = torch.tensor([o + (p.numel() + ALIGN - 1) // ALIGN * ALIGN for p, o in zip(self.params, self.offsets)],
                                    dtype=torch.int64, device=dev)
        self._norm_ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self._norm_out = torch.zeros(1, dtype=torch.float32, device=dev)

    def reattach_grads(self):
        """Re-point p.grad at the flat buffer (after something set grads to None)."""
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.flat_g.zero_()
        self.stale = False

    stale = False      # True while gradients a captured step kept (not zero-filled) are lying in flat_g

    def grad_norm(self, scale: float = 1.0) -> torch.Tensor:
        """Global L2 norm of all gradients (util/misc.py:151-163) as a device scalar; no host sync."""
        ops.l2norm(self.flat_g, self._norm_out, self._norm_ws, scale)
        return self._norm_out


class FlatAdamW(torch.optim.Optimizer):
    """AdamW with decoupled weight decay on FlatParams; accepts the param groups of train.py:89-93
    (incl. the extra 'pretrained' / 'lr_scale' keys util/lr_sched.py reads)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.01, model: Optional[torch.nn.Module] = None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        in_groups = {id(p) for g in self.param_groups for p in g['params']}
        if model is not None:
            ordered = [p for p in reversed(list(model.parameters())) if id(p) in in_groups and p.requires_grad]
        else:
            ordered = [p for g in self.param_groups for p in g['params'] if p.requires_grad]
        self.flat = FlatParams(ordered)
        dev = self.flat.flat_p.device
        self.exp_avg = torch.zeros_like(self.flat.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_p)
        # bf16 mirror of every master weight, rewritten by the AdamW kernel itself: the GEMMs read views of it
        self.flat_bf16 = torch.empty(self.flat.total, dtype=torch.bfloat16, device=dev)
        for p, o in zip(self.flat.params, self.flat.offsets):
            if p.ndim >= 2 and p.shape[0] > 1:
                engine.adopt_weight_mirror(p, self.flat_bf16[o:o + p.numel()])
        self.sync_bf16()
        if model is not None:      # weights loaded later (also after a step was captured in a hipGraph) reach the mirror at once
            model.register_load_state_dict_post_hook(lambda module, incompatible_keys: self.sync_bf16())
        self._group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g['params']:
                self._group_of[id(p)] = gi
        n = len(self.flat.params)
        self._hyper = torch.zeros(n, 2, dtype=torch.float32, device=dev)
        self._bc = torch.ones(2, dtype=torch.float32, device=dev)
        self._gidx = torch.tensor([self._group_of[id(p)] for p in self.flat.params], dtype=torch.long)
        self.step_count = 0
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # one byte per parameter: 1 = the fused pass does not zero-fill this gradient (its next value will be written, not
        # accumulated: engine.wgrad_overwrite_begin / util.misc.GraphedStep).  All zero unless a captured step sets it.
        self.keep_grad = torch.zeros(n, dtype=torch.uint8, device=dev)
        for p, o in zip(self.flat.params, self.flat.offsets):    # torch-compatible per-parameter state (views)
            self.state[p] = dict(step=torch.tensor(0.), exp_avg=self.exp_avg[o:o + p.numel()].view(p.shape),
                                 exp_avg_sq=self.exp_avg_sq[o:o + p.numel()].view(p.shape))

    @classmethod
    def from_torch(cls, opt: torch.optim.Optimizer, model: Optional[torch.nn.Module] = None) -> 'FlatAdamW':
        """Adopt a stock ``torch.optim.AdamW`` (train.py:93 builds exactly that): same parameter groups — every key a group carries,
        incl. the 'pretrained' / 'lr_scale' tags util/lr_sched.py reads — same hyper-parameters, and whatever state it already holds
        (a resumed run) through its own state_dict, which is also this class's format.  The parameters' storage moves into the flat
        buffers; the torch optimizer object is not used afterwards."""
        if not isinstance(opt, torch.optim.AdamW):
            raise TypeError(f'only torch.optim.AdamW can be adopted into the flat optimizer, got {type(opt).__name__}')
        d = opt.defaults
        if d.get('amsgrad') or d.get('maximize'):
            raise NotImplementedError('FlatAdamW: amsgrad / maximize are not implemented (no reference config sets them)')
        for g in opt.param_groups:
            if tuple(g['betas']) != tuple(d['betas']) or g['eps'] != d['eps'] or g.get('amsgrad') or g.get('maximize'):
                raise NotImplementedError('FlatAdamW: betas / eps must be the same in every parameter group')
        skip = ('params', 'amsgrad', 'maximize', 'foreach', 'capturable', 'differentiable', 'fused', 'decoupled_weight_decay')
        groups = [dict({k: v for k, v in g.items() if k not in skip}, params=list(g['params'])) for g in opt.param_groups]
        had_state = len(opt.state) > 0
        sd = opt.state_dict() if had_state else None
        new = cls(groups, lr=d['lr'], betas=tuple(d['betas']), eps=d['eps'], weight_decay=d['weight_decay'], model=model)
        if had_state:
            new.load_state_dict(sd)
        return new

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def sync_bf16(self):
        """Re-derive the bf16 mirror from the fp32 masters (after load_state_dict / resume)."""
        ops.cast_bf16(self.flat.flat_p, self.flat_bf16)
        engine.invalidate_weight_cache(self.flat.params)

    def group_hyper(self):
        """[(lr, weight_decay)] per parameter group as they stand now (what ``prepare_step`` would upload)."""
        return [(g['lr'], g['weight_decay']) for g in self.param_groups]

    def prepare_step(self, group_hyper=None):
        """Host side of a step (kept outside hipGraph capture): bump t, upload per-tensor {lr, wd} and bias corrections.
        ``group_hyper``: a ``group_hyper()`` snapshot to upload instead of the groups' current values (a deferred update uses the
        learning rate of the step its gradients belong to, not the one the scheduler has set since).
        The staging tensors are FRESH pinned allocations every step: the host may be many replayed steps ahead of the GPU,
        and a reused staging buffer would be overwritten with a later step's values before this step's asynchronous copy
        has run (torch's pinned-memory allocator recycles a block only after the copy that read it has completed)."""
        self.step_count += 1
        gh = group_hyper if group_hyper is not None else self.group_hyper()
        lrs = torch.tensor([h[0] for h in gh], dtype=torch.float32)
        wds = torch.tensor([h[1] for h in gh], dtype=torch.float32)
        pin = self._hyper.is_cuda
        hyper = torch.empty(self._hyper.shape, dtype=torch.float32, pin_memory=pin)
        hyper[:, 0] = lrs[self._gidx]
        hyper[:, 1] = wds[self._gidx]
        b1, b2 = self.defaults['betas']
        bc = torch.empty(2, dtype=torch.float32, pin_memory=pin)
        bc[0] = 1 - b1 ** self.step_count
        bc[1] = math.sqrt(1 - b2 ** self.step_count)
        self._hyper.copy_(hyper, non_blocking=True)
        self._bc.copy_(bc, non_blocking=True)
        for st in self.state.values():
            st['step'] += 1

    def launch_step(self, grad_scale: float = 1.0, fused_norm_and_zero: bool = False, keep_grad=None, gscale_dev=None):
        """Device side of a step: one kernel over all parameters (capturable).  With ``fused_norm_and_zero`` the same
        pass also leaves sum(g^2) in ``self.sumsq`` (grad norm = sqrt) and zeroes the gradients — except those of the
        parameters flagged in ``keep_grad`` (uint8 per parameter; every captured step owns its table, default: this
        optimizer's all-zero one).  ``gscale_dev``: device scalar of ``ops.step_guard`` (clip factor, 0 = skip)."""
        b1, b2 = self.defaults['betas']
        f = self.flat
        if keep_grad is None:
            keep_grad = self.keep_grad
        ops.adamw_flat(f.flat_p, f.flat_g, self.exp_avg, self.exp_avg_sq, self.flat_bf16, f.seg_end, self._hyper, len(f.params),
                       b1, b2, self.defaults['eps'], self._bc, grad_scale,
                       sumsq_out=self.sumsq if fused_norm_and_zero else None, zero_grad=fused_norm_and_zero,
                       keep_grad=keep_grad if fused_norm_and_zero else None, gscale_dev=gscale_dev)
        engine.invalidate_weight_cache(f.params)      # fp32 masters changed behind torch's back

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Accepts what ``torch.optim.AdamW.state_dict()`` produces for the same parameter groups (the 'optimizer' entry
        of a reference checkpoint, util/misc.py:251-262) as well as this class's own ``state_dict()`` — they are the same
        format.  The loaded moments are copied into the flat buffers and the per-parameter state becomes views again."""
        super().load_state_dict(state_dict)
        steps = []
        for p, o in zip(self.flat.params, self.flat.offsets):
            st, n = self.state.get(p, {}), p.numel()
            for key, flat in (('exp_avg', self.exp_avg), ('exp_avg_sq', self.exp_avg_sq)):
                if key in st:
                    flat[o:o + n].copy_(st[key].reshape(-1).to(flat.device, torch.float32))
                else:
                    flat[o:o + n].zero_()
            step = float(st['step']) if 'step' in st else 0.0
            steps.append(int(step))
            self.state[p] = dict(step=torch.tensor(step), exp_avg=self.exp_avg[o:o + n].view(p.shape),
                                 exp_avg_sq=self.exp_avg_sq[o:o + n].view(p.shape))
        self.step_count = max(steps) if steps else 0

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        self.prepare_step()
        self.launch_step(grad_scale)
