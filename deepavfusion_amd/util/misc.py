"""Training-step runtime (reference util/misc.py: Trainer :27-148, get_grad_norm_ :151-163,
CheckpointManager :222-309) on the MI355X path.

Differences that are deliberate (and documented in DESIGN.md):
  * bf16 compute with fp32 master weights instead of fp16 autocast + GradScaler: ``autocast()`` is a
    null context, ``get_scale()`` is 1.0 so ``train_one_epoch`` keeps logging ``amp_scale``;
  * DistributedDataParallel is replaced by ``util.distributed.DataParallel`` (bucketed RCCL all-reduce
    on the flat gradient buffer, launched by the hand-written backward);
  * the per-step global grad norm is ONE reduction over the flat gradient buffer; ``Trainer.step``
    still returns it as a Python float (host sync) like the reference, ``GraphedStep`` keeps it on device.
"""
from __future__ import annotations

import contextlib
import copy
import math
import os

import torch

from .. import engine, ops
from . import distributed as dist_utils
from .flat import FlatAdamW


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """util/misc.py:151-163 (generic per-tensor form; the Trainer uses the flat single-kernel form)."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    grads = [p.grad.detach() for p in parameters if p.grad is not None]
    if not grads:
        return torch.tensor(0.)
    if norm_type == math.inf:
        return max(g.abs().max() for g in grads)
    return torch.norm(torch.stack([torch.norm(g, norm_type) for g in grads]), norm_type)


class Trainer:
    def __init__(self, model, criterion=None, optimizer=None, accum_iter=1, use_amp=True, distributed=False, bucket_mb=None,
                 first_bucket_mb=None):
        self.distributed = distributed
        self.model_without_ddp = model
        self.n_steps = torch.tensor([0])
        # train.py:93 hands over a plain torch.optim.AdamW (util/misc.py:27-40 accepts any optimizer): on the GPU it is adopted into the
        # flat optimizer — same groups, hyper-parameters and state; ``trainer.optimizer`` (what train.py:155 adjusts and the checkpoints
        # store) is the adopted one.  Other optimizer classes keep torch's own step (single process only).
        if isinstance(optimizer, torch.optim.AdamW) and not isinstance(optimizer, FlatAdamW) and any(p.is_cuda for g in optimizer.param_groups for p in g['params']):
            optimizer = FlatAdamW.from_torch(optimizer, model)
        self.flat = optimizer.flat if isinstance(optimizer, FlatAdamW) else None
        if self.distributed:
            if self.flat is None:
                raise RuntimeError(f'data-parallel training reduces the flat gradient buffer of the AdamW path; {type(optimizer).__name__} has none')
            model = dist_utils.DataParallel(model, self.flat, bucket_mb=bucket_mb, first_bucket_mb=first_bucket_mb)
            optimizer.sync_bf16()        # rank 0's fp32 weights were just broadcast: re-derive the bf16 mirror the GEMMs read
        self.model = model
        self.criterion = criterion
        self.optimizer = optimizer
        self.scaler = None               # bf16: no loss scaling
        self.use_amp = use_amp
        self.accum_iter = accum_iter
        self.accums = 0
        self.eval_model = self.model_without_ddp
        self.zero_grad()

    def module_dict(self):
        d = {'state_dict': self.model_without_ddp, 'n_steps': self.n_steps}
        if self.criterion is not None:
            d['criterion'] = self.criterion
        if self.optimizer is not None:
            d['optimizer'] = self.optimizer
        return d

    def zero_grad(self):
        if self.flat is not None:
            self.flat.zero_grad()
        elif self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=False)
        self.accums = 0

    def get_scale(self):
        return 1.

    def grad_norm_tensor(self) -> torch.Tensor:
        if self.flat is not None:
            return self.flat.grad_norm(1.0 / max(self.accums, 1))
        return get_grad_norm_(self.model_without_ddp.parameters()) / max(self.accums, 1)

    def backward(self, loss, create_graph=False):
        if self.flat is not None and self.flat.stale and self.accums == 0:      # eager step after replays of a captured one
            self.flat.zero_grad()
        loss.backward(create_graph=create_graph)
        self.accums += 1
        if self.distributed:
            self.model.reducer.finish()          # no-op when this backward's forward ran under no_sync()
        return self.grad_norm_tensor().item(), self.get_scale()

    def step(self, loss, create_graph=False, clip_grad=None, skip_grad=None):
        """util/misc.py:96-136.  ``skip_grad`` (util/misc.py:81-104, never enabled by train.py): the gradients accumulated so far
        are set aside, this micro-step's gradients are taken alone, and if their norm exceeds ``skip_grad`` they are dropped and
        the micro-step is not counted; then the set-aside gradients are added back.  (One device-to-device copy of the flat
        gradient buffer per call: this is a safety option, not a fast path.)"""
        if skip_grad is not None:
            if self.flat is None:
                raise NotImplementedError('skip_grad needs the flat gradient buffer (FlatAdamW)')
            if self.distributed and self.accum_iter > 1:
                # the set-aside gradients of earlier no_sync micro-steps were never reduced and the per-rank "norm > skip_grad"
                # decision can differ between ranks (different accums -> collectives out of step): the reference has the same
                # flaw (util/misc.py:81-104 under DDP); refuse instead of training ranks on different gradients
                raise NotImplementedError('skip_grad with data-parallel gradient accumulation (accum_iter > 1) is not supported: '
                                          'ranks would step with different gradients')
            if self.flat.stale and self.accums == 0:         # kept gradients of a replayed captured step: nothing of value in the buffer
                self.flat.zero_grad()
            backup = self.flat.flat_g.clone()
            self.flat.flat_g.zero_()
            norm, scale = self.backward(loss, create_graph=create_graph)
            if norm > skip_grad:
                self.flat.flat_g.zero_()
                self.accums -= 1
            self.flat.flat_g.add_(backup)
            del backup
        else:
            norm, scale = self.backward(loss, create_graph=create_graph)
        if self.accums == self.accum_iter:
            gscale = 1.0 / self.accum_iter if self.accum_iter > 1 else 1.0
            if clip_grad is not None:
                total = norm * self.accums * gscale
                gscale *= min(1.0, clip_grad / (total + 1e-6))
            if isinstance(self.optimizer, FlatAdamW):
                self.optimizer.step(grad_scale=gscale)
            else:
                if gscale != 1.0:
                    for g in self.optimizer.param_groups:
                        for p in g['params']:
                            if p.grad is not None:
                                p.grad.mul_(gscale)
                self.optimizer.step()
                engine.invalidate_weight_cache(self.model_without_ddp.parameters())
            self.zero_grad()
            self.n_steps += 1
        return norm, scale

    def autocast(self):
        return contextlib.nullcontext()

    def autosync(self):
        if self.distributed and self.accums < self.accum_iter - 1:
            return self.model.no_sync()
        return contextlib.nullcontext()


# hipStreamCaptureModeThreadLocal: with the default (global) mode ANY thread's event query is an error while a capture is
# open — and the RCCL process group's watchdog thread polls the events of earlier collectives (the rank-0 broadcast, an eager
# warm-up step) at its own pace: "operation not permitted when stream is capturing" kills the process (seen with a 1-rank
# RCCL group under torch.distributed.run).  Only the capturing thread's calls need policing.
CAPTURE_MODE = 'thread_local'


def segment_cuts(depth: int, segments: int):
    """Where the captured backward is cut into ``segments`` graphs (pure function: tested on CPU).  The backward runs both
    decoders, then encoder layers depth-1 .. 0; a segment ends after "layer" l for every l in the returned list, l == depth
    standing for "after the decoders".  That is always the first cut — the decoders' gradients (a sixth of the bytes) start
    their all-reduce while the whole encoder backward is still ahead — and the remaining cuts leave quadratically fewer
    layers behind them (depth 12, 5 segments: 11..7 | 6..3 | 2..1 | 0), so the last reduction, the only one nothing is left
    to overlap with, is one layer's worth of bytes."""
    segments = max(1, min(segments, depth + 1))
    if segments < 2:
        return []
    enc_parts = segments - 1
    return sorted({depth} | ({round(depth * ((enc_parts - s) / enc_parts) ** 2) for s in range(1, enc_parts)} - {0, depth}), reverse=True)


class GraphedStep:
    """One pre-training step (forward -> backward -> grad norm -> AdamW) captured in hipGraphs and replayed per
    iteration: ~1900 kernel launches become a handful of graph launches.

    world_size == 1: one graph holds everything.  world_size > 1 (or ``segments`` > 1): the step is captured as
    ``segments`` consecutive graphs that share one memory pool — [forward + decoders' backward], then equal groups of
    encoder layers, last to first — and after each replayed segment the gradient buckets that segment
    completed (known from capture time) are all-reduced on the comm stream, i.e. overlapped with the next segment;
    grad norm + AdamW form a last graph behind the final reduction.  Collectives themselves are never captured.

    Non-finite guard: a replay whose loss (or clipped norm) is not finite leaves parameters / moments untouched ON THE DEVICE, but
    the host-side counters still advance for it (``n_steps``, Adam's step count and bias corrections in ``prepare_step``): a run
    is NOT meant to continue past such a step — ``check()`` raises (train.py calls it every print_freq steps, at the end of every
    epoch and before every checkpoint), exactly where the reference raises on the step itself (train.py:166-167)."""

    def __init__(self, trainer: Trainer, image_shape, audio_shape, warmup: int = 2, segments: int = 0, clip_grad=None):
        """``clip_grad``: max global gradient norm (``opt.clip_grad``; util/misc.py:118-120) — the norm is then taken in a
        pass of its own in front of AdamW and the factor min(1, clip / (norm + 1e-6)) reaches the update as a device scalar.
        Always on: the non-finite guard of train.py:166-167 — a step whose loss (or, with clipping, gradient norm) is not
        finite leaves parameters and moments untouched; ``check()`` raises on the host when it is convenient to look."""
        assert trainer.accum_iter == 1 and isinstance(trainer.optimizer, FlatAdamW)
        self.clip_grad = float(clip_grad) if clip_grad else None
        from .. import autograd_bridge as bridge
        self.bridge = bridge
        self.tr = trainer
        self.model = trainer.model_without_ddp
        self.opt = trainer.optimizer
        dev = self.opt.flat.flat_p.device
        self.image = torch.zeros(image_shape, device=dev)
        self.audio = torch.zeros(audio_shape, device=dev)
        self.world = dist_utils.get_world_size()
        self.reducer = trainer.model.reducer if trainer.distributed else None
        self.dist_active = self.world > 1 or (self.reducer is not None and self.reducer.force)
        if segments <= 0:
            segments = (int(os.environ.get('DAV_DP_SEGMENTS', '0')) if self.dist_active else 0) or int(os.environ.get('DAV_SEGMENTS', '0')) or (5 if self.dist_active else 1)
        enc = self.model.encoder
        vis = enc.visual if hasattr(enc, 'visual') else (enc.video if hasattr(enc, 'video') else enc.image)      # image or video tower
        depth = len(vis.blocks)
        segments = max(1, min(segments, depth + 1))
        # The backward runs: both decoders, then encoder layers depth-1 .. 0.  A segment ends after "layer" l when l is in
        # `cuts`; l == depth stands for "after the decoders".  That is always the first cut: the decoders' gradients
        # (a sixth of the bytes) start their all-reduce while the whole encoder backward is still ahead; the remaining cuts
        # leave quadratically fewer layers behind them (depth 12, 5 segments: layers 11..7 | 6..3 | 2..1 | 0), so the last
        # reduction — the only one nothing is left to overlap with — is one layer's worth of bytes.
        self.cuts = segment_cuts(depth, segments)
        self.n_seg = len(self.cuts) + 1
        saved_hook = engine._GRAD_READY
        engine.set_grad_ready_hook(None)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._fwd_bwd(None)
                self.opt.flat.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        # ---- capture ------------------------------------------------------------------------------------------
        self.graphs = [torch.cuda.CUDAGraph() for _ in range(self.n_seg)]
        self.bucket_sched = [[] for _ in range(self.n_seg)]
        seg = [0]
        pending = [len(b[2]) for b in self.reducer.buckets] if self.reducer is not None else []

        def on_ready(p):
            if self.reducer is None:
                return
            bi = self.reducer._bucket_of.get(id(p))
            if bi is not None:
                pending[bi] -= 1
                if pending[bi] == 0:
                    self.bucket_sched[seg[0]].append(bi)
        engine.set_grad_ready_hook(on_ready)

        def layer_cb(l):
            if l in self.cuts:
                self.graphs[seg[0]].capture_end()
                seg[0] += 1
                self.graphs[seg[0]].capture_begin(pool=self.graphs[0].pool(), capture_error_mode=CAPTURE_MODE)
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        self.opt.flat.zero_grad()            # the AdamW pass leaves the gradients zeroed for the next replay
        # per captured step (several GraphedSteps may share one optimizer): which gradients its backward WRITES (not zero-filled
        # by its AdamW pass), the device scalar its update is scaled by, and how many of its replays were skipped
        self.keep_grad = torch.zeros_like(self.opt.keep_grad)
        self.step_scale = torch.ones(1, dtype=torch.float32, device=dev)
        self.bad_steps = torch.zeros(1, dtype=torch.int32, device=dev)
        self._presq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._presq_ws = torch.zeros(1024, dtype=torch.float32, device=dev)

        def optimizer_pass():
            gnorm = None
            if self.clip_grad is not None:       # the clip factor needs the norm BEFORE the update: one extra read of the gradients
                ops.l2norm(self.opt.flat.flat_g, self._presq, self._presq_ws, 1.0)
                gnorm = self._presq
            ops.step_guard(self.loss_image, self.loss_audio, gnorm, self.clip_grad, 1.0, self.step_scale, self.bad_steps)
            self.opt.launch_step(fused_norm_and_zero=True, keep_grad=self.keep_grad, gscale_dev=self.step_scale)      # AdamW + sum(g^2) + zero_grad in one pass
            self.grad_norm = self.opt.sumsq.sqrt()
        kept = set()
        with torch.cuda.stream(cap):
            self.graphs[0].capture_begin(capture_error_mode=CAPTURE_MODE)
            try:
                if os.environ.get('DAV_WGRAD_OVERWRITE', '1') != '0':
                    engine.wgrad_overwrite_begin()       # first weight-gradient GEMM into a Linear weight writes its tile (see engine)
                # the encoder layers' weight gradients are merged into one launch per captured segment: they must be out where a
                # segment ends
                engine.WGRAD_FLUSH_LAYERS = set(self.cuts)
                # every derived bf16 copy (casts of un-mirrored weights, TRANSPOSED copies) must be re-derived INSIDE the graph:
                # copies left over from the warm-up passes would otherwise be read, stale, by every replay
                engine.invalidate_weight_cache(self.model.parameters())
                engine.refresh_weight_cache(self.model)
                self.loss_image, self.loss_audio = self._fwd_bwd(layer_cb)
            finally:
                # (also on an exception: a write-first mode left on would make later EAGER backwards overwrite instead of
                # accumulate the first contribution to every Linear weight's gradient)
                kept = {id(p) for p in engine.wgrad_overwrite_end()}
                engine.WGRAD_FLUSH_LAYERS = None
            if not self.dist_active:
                optimizer_pass()
            self.graphs[seg[0]].capture_end()
            self.opt_graph = None
            if self.dist_active:
                self.opt_graph = torch.cuda.CUDAGraph()
                self.opt_graph.capture_begin(pool=self.graphs[0].pool(), capture_error_mode=CAPTURE_MODE)
                optimizer_pass()
                self.opt_graph.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()
        if self.reducer is not None:          # anything not reported (should not happen) goes with the last segment
            done = {bi for s in self.bucket_sched for bi in s}
            self.bucket_sched[-1] += [bi for bi in range(len(self.reducer.buckets)) if bi not in done]
        engine.set_grad_ready_hook(saved_hook)
        self.opt.flat.zero_grad()
        # the captured AdamW pass reads this table at replay time: gradients the captured backward WRITES are not zero-filled
        self.keep_grad.copy_(torch.tensor([1 if id(p) in kept else 0 for p in self.opt.flat.params], dtype=torch.uint8))
        self.kept_params = len(kept)

    def check(self):
        """Host side of the non-finite guard (one device read: call it where the loss is read anyway, e.g. every
        ``print_freq`` steps).  Raises like train.py:166-167 if any replay since construction was skipped."""
        n = int(self.bad_steps.item())
        if n:
            raise RuntimeError(f'Loss is {float(self.loss_image + self.loss_audio)}, stopping training '
                               f'({n} captured step(s) had a non-finite loss or gradient norm; their updates were skipped)')

    def _fwd_bwd(self, layer_cb):
        """Forward + hand-written backward straight on the engine (no autograd), unit upstream gradients."""
        B, dev = self.image.shape[0], self.image.device
        Li, La = self.model.image_gs[0] * self.model.image_gs[1], self.model.audio_gs[0] * self.model.audio_gs[1]
        noise_i, noise_a = torch.rand(B, Li, device=dev), torch.rand(B, La, device=dev)
        with engine.ln_fuse_for(True):
            outs, tape, aux = self.bridge.avmae_fwd(self.model, self.image, self.audio, noise_i, noise_a)
        one = torch.ones((), device=dev)
        self.bridge.avmae_bwd(self.model, tape, one, one, layer_cb=layer_cb)
        return outs[0], outs[1]

    def __call__(self, image, audio):
        self.image.copy_(image, non_blocking=True)
        self.audio.copy_(audio, non_blocking=True)
        self.opt.prepare_step()
        if self.reducer is not None:
            self.reducer.begin_backward()
        for s, g in enumerate(self.graphs):
            g.replay()
            if self.reducer is not None and self.dist_active:
                self.reducer.launch_buckets(self.bucket_sched[s])      # overlaps the next segment's replay
        if self.opt_graph is not None:
            self.reducer.finish()
            self.opt_graph.replay()
        self.opt.flat.stale = self.kept_params > 0
        engine.bump_fold_generation()          # the replayed optimizer pass moved the masters: gamma-folded weight copies are re-made on next eager use
        self.tr.n_steps += 1
        return self.loss_image, self.loss_audio, self.grad_norm


class CheckpointManager:
    """util/misc.py:222-309, same file format: ``checkpoint_latest.pth`` (+ ``checkpoint_best.pth``, + numbered every
    save_freq) = {'state_dict': model.state_dict(), 'optimizer': <torch AdamW format>, 'n_steps', 'scaler', 'epoch', ...};
    ``resume()`` -> (start_epoch, metrics).  A checkpoint written by the reference resumes here and vice versa: the
    optimizer entry is the torch format (FlatAdamW.state_dict / load_state_dict), and because this path trains in bf16
    without loss scaling a neutral GradScaler state is written under 'scaler' for the reference's resume loop."""

    NEUTRAL_SCALER = {'scale': 1.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 0}

    def __init__(self, modules, ckpt_dir, epochs, save_freq=None):
        self.modules, self.ckpt_dir, self.epochs, self.save_freq = modules, ckpt_dir, epochs, save_freq
        self.world_size, self.rank = dist_utils.get_world_size(), dist_utils.get_rank()
        if self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)

    def map_location(self, state, device):
        if isinstance(state, dict):
            return {k: self.map_location(v, device) for k, v in state.items()}
        if isinstance(state, (list, tuple)):
            return type(state)(self.map_location(v, device) for v in state)
        if isinstance(state, torch.Tensor):
            return state.detach().to(device, copy=True)
        return state

    def create_state_dict(self, save_dict=None):
        state = {}
        for k, m in self.modules.items():
            if m is None:
                state[k] = None
            elif isinstance(m, torch.Tensor):
                state[k] = m.detach().clone().cpu()
            else:
                state[k] = self.map_location(m.state_dict(), 'cpu')
        state.setdefault('scaler', dict(self.NEUTRAL_SCALER))
        if save_dict is not None:
            state.update(save_dict)
        return state

    def checkpoint(self, epoch, save_dict=None, is_best=False):
        if self.rank != 0:
            return
        state = self.create_state_dict(save_dict)
        torch.save(state, os.path.join(self.ckpt_dir, 'checkpoint_latest.pth'))
        if is_best:
            torch.save(state, os.path.join(self.ckpt_dir, 'checkpoint_best.pth'))
        if self.save_freq is not None and (epoch % self.save_freq == 0 or epoch == self.epochs):
            torch.save(state, os.path.join(self.ckpt_dir, f'checkpoint_{epoch:04d}.pth'))

    def resume(self):
        path = os.path.join(self.ckpt_dir, 'checkpoint_latest.pth')
        start_epoch, metrics = 0, {}
        if not os.path.isfile(path):
            return start_epoch, metrics
        ckpt = torch.load(path, map_location='cpu')
        for k, m in self.modules.items():
            if m is None:
                continue
            if isinstance(m, torch.Tensor):
                m.data[:] = ckpt[k].data
            else:
                m.load_state_dict(ckpt[k])
        start_epoch = ckpt['epoch']
        metrics = {k: v for k, v in ckpt.items() if k not in set(self.modules.keys()) and k not in ('epoch', 'scaler')}
        engine.invalidate_weight_cache(self.modules['state_dict'].parameters())
        for m in self.modules.values():
            if isinstance(m, FlatAdamW):
                m.sync_bf16()                # fp32 masters were overwritten in place: refresh the bf16 mirror the GEMMs read
        return start_epoch, metrics
