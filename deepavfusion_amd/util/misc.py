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
        self.flat = optimizer.flat if isinstance(optimizer, FlatAdamW) else None
        if self.distributed:
            if self.flat is None:
                raise RuntimeError('data-parallel training needs FlatAdamW (flat gradient buffer for the RCCL reducer)')
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

    def _flush_deferred(self):
        """A captured step with a deferred optimizer pass (GraphedStep(defer=True)) may hold one update back: anything that is about to
        touch the gradients from the eager side applies it first."""
        g = getattr(self, 'deferred_step', None)
        if g is not None:
            g.flush()

    def zero_grad(self):
        self._flush_deferred()
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
        self._flush_deferred()
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

    def __init__(self, trainer: Trainer, image_shape, audio_shape, warmup: int = 2, segments: int = 0, clip_grad=None, defer=None, fuse=None):
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
        # Early AdamW (one GPU, no clipping — the clip factor needs every gradient first, a data-parallel step its all-reduce):
        # the update of a parameter needs nothing but that parameter's final gradient, so the ranges of the flat buffer whose
        # gradients are complete at a few points of the backward (after the decoders, after encoder layers 9 / 6 / 3 / 1 by
        # default: DAV_EARLY_ADAMW_CUTS) get their AdamW pass THERE, on a side stream, under the rest of the backward — instead
        # of one 1.8 ms memory-bound kernel running alone at the end of the step.  Which parameters are final where is learnt
        # from one of the warm-up passes (the engine reports every parameter once its gradient is complete).
        single = not self.dist_active and self.n_seg == 1 and self.clip_grad is None
        # Deferred AdamW (same conditions): the update with step i's gradients is the FIRST thing of replay i + 1, issued layer by
        # layer on a side stream while the forward of replay i + 1 is already running — each forward stage waits only for the update
        # of its own parameters (autograd_bridge ``fwd_gate``).  The loss sequence is that of the plain schedule (forward i + 1 sees
        # parameters updated i times either way); what lags is the state BETWEEN calls: after call i the parameters carry i - 1
        # updates and the gradients of step i, until the next call or ``flush()`` (before anything reads the parameters:
        # checkpoint, evaluation, end of training).  The returned grad norm is the previous step's.
        self.defer = single and (defer if defer is not None else os.environ.get('DAV_DEFER_ADAMW', '0') == '1')
        self.early = single and not self.defer and os.environ.get('DAV_EARLY_ADAMW', '0') == '1'
        # Fused AdamW (same conditions, and the written-first weight gradients on): a Linear weight whose one weight-gradient problem of the
        # step is a written tile set is updated by the workgroups that own those tiles (dav_gemm_tn_grouped_adamw_bf16) — its gradient is
        # never stored, and the optimizer kernel behind the backward covers only what is left (biases, norms, embeddings, column blocks).
        self.fuse = (single and not self.defer and not self.early and os.environ.get('DAV_WGRAD_OVERWRITE', '1') != '0'
                     and (fuse if fuse is not None else os.environ.get('DAV_FUSED_ADAMW', '0') == '1'))
        cuts_env = os.environ.get('DAV_EARLY_ADAMW_CUTS', '')
        self.early_cuts = ({int(c) for c in cuts_env.split(',') if c.strip()} if cuts_env
                           else {depth} | {l for l in (9, 6, 3, 1) if l < depth}) if self.early else set()
        if self.defer:
            self.early_cuts = set(range(1, depth + 1))
        learn = self.early or self.defer
        index_of = {id(p): i for i, p in enumerate(self.opt.flat.params)}
        learnt, fresh = {}, []                 # cut -> parameter indices whose gradients became final since the previous cut

        def learn_ready(p):
            i = index_of.get(id(p))
            if i is not None:
                fresh.append(i)

        def learn_cb(l):
            if l in self.early_cuts:
                learnt[l] = sorted(set(fresh))
                fresh.clear()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            contrib = {}
            for it in range(max(warmup, 1 if (learn or self.fuse) else 0)):
                if self.fuse and it == 0:
                    engine.wgrad_contrib_begin()
                    self._fwd_bwd(None)
                    contrib = engine.wgrad_contrib_end()
                elif learn and it == 0:
                    engine.set_grad_ready_hook(learn_ready)
                    self._fwd_bwd(learn_cb)
                    engine.set_grad_ready_hook(None)
                else:
                    self._fwd_bwd(None)
                self.opt.flat.zero_grad()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        def as_ranges(idx):                    # consecutive parameter indices -> [first, last] ranges of the flat buffer
            out = []
            for i in idx:
                if out and i == out[-1][1] + 1:
                    out[-1][1] = i
                else:
                    out.append([i, i])
            return [self.opt.make_range(a, b) for a, b in out]
        taken = set()
        self.early_ranges = {}
        for l in sorted(learnt, reverse=True):     # (the backward meets the cuts in descending order)
            idx = [i for i in learnt[l] if i not in taken]
            taken.update(idx)
            self.early_ranges[l] = as_ranges(idx)
        self.late_ranges = as_ranges([i for i in range(len(self.opt.flat.params)) if i not in taken]) if learn else []
        self.opt_stream = torch.cuda.Stream() if learn else None
        if self.defer:
            # update order = the order the forward needs the parameters: [embeddings + layer 0] (what the backward finishes last),
            # layers 1 .. depth-1 (the final norms ride with the last layer), the decoders
            self.chunks = [self.late_ranges] + [self.early_ranges.get(l, []) for l in range(1, depth + 1)]
            chunk_of = {}
            for k, rs in enumerate(self.chunks):
                for r in rs:
                    for i in range(r['first'], r['first'] + r['n']):
                        chunk_of[i] = k
            blocks = [vis.blocks, enc.audio.blocks, enc.fusion_blocks]
            norms = [m for m in (vis.norm, enc.audio.norm, getattr(enc, 'fusion_norm', None)) if m is not None]
            in_layers = {id(p) for bl in blocks for b in bl if b is not None for p in b.parameters()} | {id(p) for m in norms for p in m.parameters()}
            enc_ids = {id(p) for p in enc.parameters()}
            stage_params = {-1: [p for p in enc.parameters() if id(p) not in in_layers],
                            depth: [p for m in norms for p in m.parameters()],
                            depth + 1: [p for p in self.model.parameters() if id(p) not in enc_ids]}
            for l in range(depth):
                stage_params[l] = [p for bl in blocks if bl[l] is not None for p in bl[l].parameters()]
            # (a stage waits for the LAST chunk any of its parameters is in; chunks run in order on one stream)
            self.stage_chunk = {st: max([chunk_of[index_of[id(p)]] for p in ps if id(p) in index_of], default=-1)
                                for st, ps in stage_params.items()}

        # ---- capture ------------------------------------------------------------------------------------------
        self.graphs = [torch.cuda.CUDAGraph() for _ in range(self.n_seg)]
        self.bucket_sched = [[] for _ in range(self.n_seg)]
        seg = [0]
        pending = [len(b[2]) for b in self.reducer.buckets] if self.reducer is not None else []

        final_now = set()

        def on_ready(p):
            final_now.add(index_of.get(id(p)))
            if self.reducer is None:
                return
            bi = self.reducer._bucket_of.get(id(p))
            if bi is not None:
                pending[bi] -= 1
                if pending[bi] == 0:
                    self.bucket_sched[seg[0]].append(bi)
        engine.set_grad_ready_hook(on_ready)

        guard_done = [False]

        def guard():                          # the device scalar the update is scaled by (0 = skip): needs the losses only
            if not guard_done[0]:
                ops.step_guard(self.loss_image_dev, self.loss_audio_dev, None, None, 1.0, self.step_scale, self.bad_steps)
                guard_done[0] = True

        def layer_cb(l):
            if l in self.cuts or (self.early and self.early_ranges.get(l)):
                engine.join_wgrad_stream(dev)      # (DAV_WGRAD_SIDE: the weight gradients launched so far are part of what ends here)
            if l in self.cuts:
                self.graphs[seg[0]].capture_end()
                seg[0] += 1
                self.graphs[seg[0]].capture_begin(pool=self.graphs[0].pool(), capture_error_mode=CAPTURE_MODE)
            if self.early and self.early_ranges.get(l):
                # every stream of the backward is joined at a layer boundary and the layer's weight gradients are launched on
                # this stream (autograd_bridge.encoder_bwd): whatever was reported final is final in stream order here
                for r in self.early_ranges[l]:
                    missing = [i for i in range(r['first'], r['first'] + r['n']) if i not in final_now]
                    assert not missing, ('gradients not final at cut', l, missing[:4])
                cur = torch.cuda.current_stream()
                self.opt_stream.wait_stream(cur)
                with torch.cuda.stream(self.opt_stream):
                    guard()
                    for r in self.early_ranges[l]:
                        self.opt.launch_range(r, keep_grad=self.keep_grad, gscale_dev=self.step_scale)
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
        self.prev_loss = torch.zeros(2, dtype=torch.float32, device=dev)      # deferred AdamW: the losses its guard judges
        self.pending = False                                                  # deferred AdamW: gradients waiting for their update
        if self.defer:
            # nothing on the eager side may see the lagging state: Trainer.backward / zero_grad flush before touching the gradients,
            # an eager forward (training or evaluation) and state_dict() flush before reading the parameters
            trainer.deferred_step = self
            self.model.register_forward_pre_hook(lambda m, args: self.flush())
            self.model.register_state_dict_pre_hook(lambda m, prefix, keep_vars: self.flush())
        if self.defer:
            self.grad_norm = torch.zeros((), dtype=torch.float32, device=dev)

        def deferred_update():                # AdamW of the PREVIOUS replay's gradients, chunk by chunk on the side stream
            self.opt_stream.wait_stream(torch.cuda.current_stream())
            self.chunk_events = []
            with torch.cuda.stream(self.opt_stream):
                ops.step_guard(self.prev_loss[0:1], self.prev_loss[1:2], None, None, 1.0, self.step_scale, self.bad_steps)
                for rs in self.chunks:
                    for r in rs:
                        self.opt.launch_range(r, keep_grad=self.keep_grad, gscale_dev=self.step_scale)
                    ev = torch.cuda.Event()
                    ev.record(self.opt_stream)
                    self.chunk_events.append(ev)
                # (a buffer of its own: this pass is captured twice — in the step and in the flush graph — over one memory pool)
                torch.sqrt(torch.cat([r['sumsq'] for rs in self.chunks for r in rs]).sum(), out=self.grad_norm)
            engine.invalidate_weight_cache(self.opt.flat.params)
        waited = [-1]

        def fwd_gate(stage):
            k = self.stage_chunk.get(stage, -1)
            if k > waited[0]:
                torch.cuda.current_stream().wait_event(self.chunk_events[k])
                waited[0] = k

        self.fused_sumsq = torch.zeros(1, dtype=torch.float32, device=dev)      # sum(g^2) of the tiles whose update the weight-gradient launches carry

        def optimizer_pass():
            if self.fuse:
                guard()                              # (already issued behind the forward: a no-op here)
                self.opt.launch_step(fused_norm_and_zero=True, keep_grad=self.keep_grad, gscale_dev=self.step_scale)
                self.grad_norm = (self.opt.sumsq + self.fused_sumsq).sqrt()
                return
            if self.early:                       # what the backward did not take along: the first layers, the embeddings
                torch.cuda.current_stream().wait_stream(self.opt_stream)
                guard()
                for r in self.late_ranges:
                    self.opt.launch_range(r, keep_grad=self.keep_grad, gscale_dev=self.step_scale)
                parts = [r['sumsq'] for rs in self.early_ranges.values() for r in rs] + [r['sumsq'] for r in self.late_ranges]
                self.grad_norm = torch.cat(parts).sum().sqrt()
                engine.invalidate_weight_cache(self.opt.flat.params)
                return
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
                # every derived bf16 copy (casts of un-mirrored weights, TRANSPOSED copies) must be re-derived INSIDE the graph:
                # copies left over from the warm-up passes would otherwise be read, stale, by every replay
                engine.invalidate_weight_cache(self.model.parameters())
                engine.refresh_weight_cache(self.model)
                if self.defer:
                    deferred_update()
                if self.fuse:
                    flat = self.opt.flat
                    b1, b2 = self.opt.defaults['betas']
                    grad_at = {int(flat.flat_g[o:o + 1].data_ptr()): id(p) for p, o in zip(flat.params, flat.offsets)}
                    allowed = {grad_at[a] for a, n in contrib.items() if n == 1 and a in grad_at}
                    self.fused_sumsq.zero_()
                    engine.fused_adamw_begin(dict(g=flat.flat_g, p=flat.flat_p, m=self.opt.exp_avg, v=self.opt.exp_avg_sq, bf16=self.opt.flat_bf16,
                                                  hyper=self.opt._hyper, bias_corr=self.opt._bc, gscale_dev=self.step_scale, sumsq=self.fused_sumsq,
                                                  beta1=b1, beta2=b2, eps=self.opt.defaults['eps']), index_of, allowed)
                self.loss_image, self.loss_audio = self._fwd_bwd(layer_cb, fwd_gate if self.defer else None, after_fwd=guard if self.fuse else None)
                if self.defer:
                    torch.cuda.current_stream().wait_stream(self.opt_stream)
                    self.prev_loss[0:1].copy_(self.loss_image.reshape(1))
                    self.prev_loss[1:2].copy_(self.loss_audio.reshape(1))
            finally:
                # (also on an exception: a write-first mode left on would make later EAGER backwards overwrite instead of
                # accumulate the first contribution to every Linear weight's gradient)
                kept = {id(p) for p in engine.wgrad_overwrite_end()}
                fused = {id(p) for p in engine.fused_adamw_end()}
            if not self.dist_active and not self.defer:
                optimizer_pass()
            self.graphs[seg[0]].capture_end()
            self.opt_graph = None
            self.flush_graph = None
            if self.defer:                       # the update alone: what ``flush()`` replays
                self.flush_graph = torch.cuda.CUDAGraph()
                self.flush_graph.capture_begin(pool=self.graphs[0].pool(), capture_error_mode=CAPTURE_MODE)
                deferred_update()
                torch.cuda.current_stream().wait_stream(self.opt_stream)
                self.flush_graph.capture_end()
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
        # (byte bit 1: the optimizer kernel skips the parameter altogether — its update came with its gradient, dav_adamw_flat)
        self.keep_grad.copy_(torch.tensor([(3 if id(p) in fused else 1) if id(p) in kept else 0 for p in self.opt.flat.params], dtype=torch.uint8))
        self.kept_params = len(kept)
        self.fused_params = len(fused)

    def check(self):
        """Host side of the non-finite guard (one device read: call it where the loss is read anyway, e.g. every
        ``print_freq`` steps).  Raises like train.py:166-167 if any replay since construction was skipped."""
        n = int(self.bad_steps.item())
        if n:
            raise RuntimeError(f'Loss is {float(self.loss_image + self.loss_audio)}, stopping training '
                               f'({n} captured step(s) had a non-finite loss or gradient norm; their updates were skipped)')

    def flush(self):
        """Deferred AdamW only: apply the update the last call left pending (a no-op otherwise).  Call before anything reads the
        parameters or the optimizer state — checkpoint, evaluation, the end of training."""
        if self.defer and self.pending:
            self.opt.prepare_step(self.pending_hyper)
            self.flush_graph.replay()
            self.pending = False

    def _fwd_bwd(self, layer_cb, fwd_gate=None, after_fwd=None):
        """Forward + hand-written backward straight on the engine (no autograd), unit upstream gradients."""
        B, dev = self.image.shape[0], self.image.device
        Li, La = self.model.image_gs[0] * self.model.image_gs[1], self.model.audio_gs[0] * self.model.audio_gs[1]
        noise_i, noise_a = torch.rand(B, Li, device=dev), torch.rand(B, La, device=dev)
        outs, tape, aux = self.bridge.avmae_fwd(self.model, self.image, self.audio, noise_i, noise_a, fwd_gate=fwd_gate)
        self.loss_image_dev, self.loss_audio_dev = outs[0], outs[1]      # (the early AdamW passes' guard reads them inside the backward)
        if after_fwd is not None:
            after_fwd()                                                  # (fused AdamW: the step guard's scalar before the first weight-gradient launch)
        one = torch.ones((), device=dev)
        self.bridge.avmae_bwd(self.model, tape, one, one, layer_cb=layer_cb)
        return outs[0], outs[1]

    def __call__(self, image, audio):
        self.image.copy_(image, non_blocking=True)
        self.audio.copy_(audio, non_blocking=True)
        if self.defer:
            if self.pending:
                self.opt.prepare_step(self.pending_hyper)
                self.graphs[0].replay()
            else:
                # nothing to apply yet (first call, or right after flush()): a non-finite "previous loss" makes the guard skip the
                # update on the device — the one replay the skip counter must not see
                self.prev_loss.fill_(float('nan'))
                self.graphs[0].replay()
                self.bad_steps.sub_(1)
            self.pending, self.pending_hyper = True, self.opt.group_hyper()      # (the learning rate of THIS step's gradients)
            self.opt.flat.stale = self.kept_params > 0
            self.tr.n_steps += 1
            return self.loss_image, self.loss_audio, self.grad_norm
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
