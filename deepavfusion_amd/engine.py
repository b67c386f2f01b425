"""Explicit forward/backward of the AVMAE pre-training step on the HIP kernels.

Every function here launches kernels of libdavfusion_hip.so through ``ops`` on torch's
current stream and keeps its own "tape" (a dict of saved activations) for the
hand-written backward.  Nothing goes through autograd inside; ``autograd_bridge``
exposes the whole step (or the encoder alone) as ONE autograd node so that
``loss.backward()`` / the reference's ``Trainer.step`` flow keeps working.

Precision: fp32 residual stream and master weights, bf16 GEMM/attention operands,
fp32 accumulation, fp32 LayerNorm / softmax / loss.

Weight gradients are accumulated IN PLACE into ``param.grad`` (fp32) by the wgrad GEMM
epilogues (the bridge returns ``None`` for parameters): no per-parameter autograd
accumulation pass, and a data-parallel reducer is told when a parameter's gradient is
final through ``set_grad_ready_hook``.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional

import torch

from . import ops
from .ops import F32

BF16 = torch.bfloat16        # dtype of GEMM / attention OPERAND buffers (set_precision('fp32') rebinds it to torch.float32)
PRECISION = 'bf16'


def set_precision(p: str):
    """'bf16' (product path: bf16 operands, fp32 accumulate, MFMA kernels) or 'fp32' (every operand and intermediate fp32 on
    the kernels of csrc/f32_path.hip: the parity cross-check of the engine itself, see include/dav_kernels.h).  The fp32
    path issues its launches one by one (no launch batching)."""
    global BF16, PRECISION
    if p not in ('bf16', 'fp32'):
        raise ValueError(p)
    PRECISION = p
    BF16 = torch.float32 if p == 'fp32' else torch.bfloat16

_GRAD_READY: Optional[Callable[[torch.nn.Parameter], None]] = None


def set_grad_ready_hook(fn: Optional[Callable[[torch.nn.Parameter], None]]):
    """Called once per parameter per backward, right after its last gradient kernel was enqueued."""
    global _GRAD_READY
    _GRAD_READY = fn


def _ready(*params):
    if _GRAD_READY is not None:
        if _BATCH is not None:          # the kernels that finish these gradients are only recorded so far: report at batch end
            _BATCH.pending_ready.extend(params)
            return
        for p in params:
            if p is not None and p.requires_grad:
                _GRAD_READY(p)


# ------------------------------------------------------------------------------------------------
# launch batching (csrc/batch.h): ``with batch() as bt: bt.lane(); <chain A>; bt.lane(); <chain B>`` records the kernel
# launches of independent chains and issues them in lockstep, equal kernels of equal rank as ONE grouped grid — the image
# and the audio tower block of a layer (models/deepavfusion.py:104-105), the two decoders (models/avmae.py:147-180), the
# two aggregation cross-attentions of a fusion block.  ``batch(auto_lanes=True)``: every launch inside is independent of
# the others.  Only library launches may touch live buffers inside a batch (torch ops would run BEFORE the recorded
# kernels).  Nested use joins the outer batch (lane() then has no effect).  DAV_BATCH=0 turns batching off (BATCH_POLICY below).
# ------------------------------------------------------------------------------------------------
_BATCH = None
BATCH_STATS = [0, 0]          # launches recorded / issued since the last reset (diagnostics, tests)

# Schedules of the independent chains of a step (image block | audio block | fusion block of a layer; the two decoders):
#   streams   one HIP stream per chain (parallel hipGraph branches), launch batching only INSIDE a chain: regions of independent
#             launches (the fusion block's projections / cross-attentions / LayerNorms) and the layer's grouped weight gradients.
#   lanes     the two tower blocks (the two decoders) as LANES of one launch batch on the main stream — equal-rank kernels merged
#             into grouped grids — with the fusion block on its own stream beside them (DAV_FUSION_STREAM=0: as a third lane
#             of the batch, the towers idling through its extra steps: everything on ONE queue; (round 3 also ran the decoders
#             on two streams all the same).
# BATCH_POLICY 'auto' (default): streams, unless DAV_LANE_MIN_ROWS=n asks for lanes from n rows (B x tokens per tower block)
# upwards; 'on' (DAV_BATCH=1): lanes always; 'off' (DAV_BATCH=0): streams and no launch batching at all.
# Same-box A/B (tools/ab_env.sh; ms per step at ViT-B B = 64 / ViT-B B = 32 / ViT-L B = 32):
#   streams                                  28.8-28.9 / 22.2 / 46.8-47.5
#   lanes + fusion stream, decoders streams  29.1-29.2
#   lanes + fusion stream                    29.5-29.8 / 24.3 / 48.0
#   three lanes on one queue                 30.0-30.2 / 26.2 / 52.0      (the default of the first half of round 2)
#   'off'                                    31.0 / 24.2 / 49.0-49.5      (round 1's schedule)
# The merged tower grids are ~10 % faster than their parts in isolation (0.256 vs 0.233 of the MFMA peak for the big GEMM
# launches), but on one queue the fusion block's ~11 small dependent steps run alone on the GPU (4-8 per layer, DESIGN.md
# section 4), and concurrent streams also fill each other's launch tails — once the regions are batched that is worth more.
BATCH_POLICY = {'1': 'on', '0': 'off'}.get(os.environ.get('DAV_BATCH', ''), 'auto')
LANE_MIN_ROWS = int(os.environ.get('DAV_LANE_MIN_ROWS', str(1 << 30)))
FUSION_ON_STREAM = os.environ.get('DAV_FUSION_STREAM', '1') != '0'


def set_batch_policy(policy: str):
    """'on' | 'off' | 'auto' (see BATCH_POLICY); tests use it to run one model under both schedules."""
    global BATCH_POLICY
    assert policy in ('on', 'off', 'auto')
    BATCH_POLICY = policy


def batching_allowed() -> bool:
    return BATCH_POLICY != 'off' and PRECISION == 'bf16'


def lanes_for(rows: int) -> bool:
    """Should the independent chains of a step with ``rows`` = B x (tokens per tower) go out as lanes of one batch?"""
    if not batching_allowed():
        return False
    return BATCH_POLICY == 'on' or rows >= LANE_MIN_ROWS


class batch:
    def __init__(self, auto_lanes: bool = False):
        self.auto_lanes = auto_lanes
        self.active = False
        self.pending_ready = []

    def __enter__(self):
        global _BATCH
        if _BATCH is None and batching_allowed():
            ops.batch_begin(self.auto_lanes)
            _BATCH = self
            self.active = True
        return self

    def lane(self):
        if self.active:
            ops.batch_lane()

    @staticmethod
    def current():
        return _BATCH

    def __exit__(self, et, ev, tb):
        global _BATCH
        if not self.active:
            return False
        _BATCH = None
        rec, iss = ops.batch_end(abort=et is not None)
        BATCH_STATS[0] += rec
        BATCH_STATS[1] += iss
        if et is None and self.pending_ready:
            _ready(*self.pending_ready)
        return False


def lane_skip(steps: int):
    """Inside a batch lane: idle for ``steps`` steps, so that this lane's next launches line up with (and are grouped
    with) equal launches of a longer lane.  No effect outside a batch."""
    if _BATCH is not None and steps > 0:
        ops.batch_skip(steps)


def gbuf(p: torch.nn.Parameter) -> torch.Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# ------------------------------------------------------------------------------------------------
# bf16 weight cache (W and W^T) — refreshed when the fp32 master changes
# ------------------------------------------------------------------------------------------------
class _WCache:
    """bf16 copy of a weight (and, lazily, of its transpose).  ``managed`` copies are views into the
    optimizer's flat bf16 mirror that the fused AdamW kernel rewrites every step."""
    __slots__ = ('ver', 'ptr', 'wb', 'wtb', 'wt_ver', 'managed')

    def __init__(self):
        self.ver = self.ptr = self.wt_ver = None
        self.wb = self.wtb = None
        self.managed = False


def wcache(p: torch.nn.Parameter, force: bool = False) -> torch.Tensor:
    """bf16 [out, in] copy of ``p`` (re-cast when the fp32 master changed); the fp32 master itself on the fp32 path."""
    if PRECISION == 'fp32':
        return p.detach().reshape(p.shape[0], -1)
    c = p.__dict__.get('_dav_cache')
    if c is None:
        c = p.__dict__['_dav_cache'] = _WCache()
    if c.managed and not force:
        if c.ver != p._version:        # written from the torch side (load_state_dict, p.copy_()) since the mirror was made:
            with ops.unbatched():
                ops.cast_bf16(p.detach().reshape(p.shape[0], -1), c.wb)  # the optimizer kernel's own updates keep it in sync
            c.ver = p._version
            p.__dict__['_dav_epoch'] = p.__dict__.get('_dav_epoch', 0) + 1     # transposed copies follow
        return c.wb
    ver, ptr = p._version, p.data_ptr()
    if force or c.wb is None or c.ver != ver or c.ptr != ptr:
        w2 = p.detach().reshape(p.shape[0], -1)
        if c.wb is None or c.wb.shape != w2.shape or c.wb.device != w2.device:
            c.wb = torch.empty(w2.shape, dtype=BF16, device=p.device)
        with ops.unbatched():
            ops.cast_bf16(w2, c.wb)
        c.ver, c.ptr = ver, ptr
    return c.wb


def wcache_t(p: torch.nn.Parameter) -> torch.Tensor:
    """bf16 [in, out] transposed copy — only for the rare dgrad whose contraction is not a multiple of 64."""
    wcache(p)
    c = p.__dict__['_dav_cache']
    stamp = (c.ver, c.ptr) if not c.managed else ('managed', p.__dict__.get('_dav_epoch', 0))
    if c.wtb is None or c.wt_ver != stamp:
        w2 = p.detach().reshape(p.shape[0], -1)
        if c.wtb is None:
            c.wtb = torch.empty((w2.shape[1], w2.shape[0]), dtype=BF16, device=p.device)
        with ops.unbatched():
            ops.cast_transpose_bf16(w2, c.wtb)
        c.wt_ver = stamp
    return c.wtb


def refresh_transposes(params):
    """Bring the transposed bf16 copies of ``params`` (the weights wcache_t will be asked for) up to date with ONE grouped
    launch instead of one cast_transpose launch per weight at first use (bf16 path only)."""
    if PRECISION == 'fp32':
        return
    todo = []
    for p in params:
        wcache(p)
        c = p.__dict__['_dav_cache']
        stamp = (c.ver, c.ptr) if not c.managed else ('managed', p.__dict__.get('_dav_epoch', 0))
        if c.wtb is None or c.wt_ver != stamp:
            w2 = p.detach().reshape(p.shape[0], -1)
            if c.wtb is None:
                c.wtb = torch.empty((w2.shape[1], w2.shape[0]), dtype=BF16, device=p.device)
            todo.append((w2, c.wtb))
            c.wt_ver = stamp
    if todo:
        with ops.unbatched():
            ops.cast_transpose_grouped(todo)


def invalidate_weight_cache(params):
    """The fp32 masters changed behind torch's version counters (optimizer kernel, flat-buffer cast): un-managed bf16 copies
    become stale; optimizer-managed mirrors were rewritten by the same kernel and are stamped in sync; transposed copies of
    either kind are re-derived on next use."""
    bump_fold_generation()
    for p in params:
        c = p.__dict__.get('_dav_cache')
        if c is not None:
            c.ver = p._version if c.managed else None
            p.__dict__['_dav_epoch'] = p.__dict__.get('_dav_epoch', 0) + 1


def adopt_weight_mirror(p: torch.nn.Parameter, view_bf16: torch.Tensor):
    """Make ``view_bf16`` (kept in sync by the optimizer) the bf16 copy of ``p``."""
    c = p.__dict__.get('_dav_cache')
    if c is None:
        c = p.__dict__['_dav_cache'] = _WCache()
    c.wb, c.managed = view_bf16.view(p.shape[0], -1), True
    c.ver = None                        # first use casts the current fp32 value into the mirror


def refresh_weight_cache(module: torch.nn.Module):
    """Re-cast every un-managed cached weight (top of a captured step when no optimizer mirror exists)."""
    for p in module.parameters():
        c = p.__dict__.get('_dav_cache')
        if c is not None and not c.managed and c.wb is not None:
            wcache(p, force=True)


def _e(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


# ------------------------------------------------------------------------------------------------
# LayerNorm folded into the GEMMs either side of it (dav_gemm_nt_ln_bf16; DAV_LN_FUSE=0 keeps the LayerNorm kernels).
# Every fp32 residual-stream tensor carries a bf16 TWIN and its row-statistics partials ({sum, sum of squares} per 64-column slot),
# written by the epilogue of the GEMM that produced it (``lin_fwd_tw``) or by ``ensure_tw``; the Linear behind a LayerNorm contracts
# the RAW twin with gamma-folded weights and normalises in its epilogue (``lin_fwd_ln``).  No LayerNorm-forward launch and no
# normalised tensor in memory on the forward path; the LayerNorm backward re-makes the weight-gradient operand from the twin
# (``ln_bwd_tw``: h_out).  The reference lines: timm Block norms (models/vits.py:32-34, models/avmae.py:53-57), the fusion blocks'
# norm1_img / norm1_aud / norm2 (models/fusion_blocks.py:281-288), decoder_norm -> decoder_pred (models/avmae.py:177-178).
# norm1_mm of the fusion blocks stays a kernel: its OUTPUT is the residual base (norm-then-residual, :281-283).
# ------------------------------------------------------------------------------------------------
# Where it stands (profiles/r06_ln_fuse.txt, ViT-B B = 64, same box, alternating): the forward loses its 123 LayerNorm launches, but the
# producers' epilogues pay 6-20 % for twin + statistics, the consumers 0-10 %, the weight fold moves 0.8 GB per step, and the backward
# must WRITE the LayerNorm outputs it no longer finds saved (+2 bytes per element on the serial LayerNorm-backward kernels): forward alone
# 8.60 ms against 8.48 ms, the whole step 25.5 ms against 24.8 ms.  The LayerNorm kernels the folding deletes are memory-bound fillers
# that run beside MFMA-bound GEMMs of the other streams; what replaces them sits in the GEMMs' own epilogues.  So DAV_LN_FUSE defaults to
# OFF (the LayerNorm kernels of rounds 1-5); '1' folds always, 'auto' folds when no backward follows.  The GEMM kernels that carry the
# folding code are separate instantiations (gemm.hip LNK): launches without a folded LayerNorm run the kernels of round 5 unchanged.
LN_FUSE_MODE = {'1': 'on', 'auto': 'auto'}.get(os.environ.get('DAV_LN_FUSE', ''), 'off')
LN_FUSE = LN_FUSE_MODE == 'on'


def set_ln_fuse(mode):
    """'on' | 'off' | 'auto' (True / False = 'on' / 'off'): tests run one model with the LayerNorms folded into their neighbour GEMMs
    and with the LayerNorm kernels."""
    global LN_FUSE, LN_FUSE_MODE
    LN_FUSE_MODE = {True: 'on', False: 'off'}.get(mode, mode)
    assert LN_FUSE_MODE in ('on', 'off', 'auto')
    LN_FUSE = LN_FUSE_MODE == 'on'


class ln_fuse_for:
    """``with ln_fuse_for(need_backward):`` around a forward: resolves the 'auto' mode for this pass."""

    def __init__(self, need_backward: bool):
        self.need = need_backward

    def __enter__(self):
        global LN_FUSE
        self.prev = LN_FUSE
        if LN_FUSE_MODE == 'auto':
            LN_FUSE = not self.need
        return self

    def __exit__(self, *exc):
        global LN_FUSE
        LN_FUSE = self.prev
        return False


def ln_fuse_ok(D: int) -> bool:
    return LN_FUSE and PRECISION == 'bf16' and D % 64 == 0 and D <= 1024


def tw_of(x):
    """(bf16 twin [rows, D], statistics partials [rows, D / 64, 2]) attached to the fp32 activation ``x``, or None."""
    return getattr(x, '_dav_tw', None)


def tw_set(x, xb, st):
    x._dav_tw = (xb, st)
    return x


def ensure_tw(x):
    """The twin + statistics of a contiguous fp32 activation [B, rows, D]: as attached by its producer, else made here
    (dav_rowstats_cast: the patch tokens of another API, the expanded fusion tokens, the un-shuffled decoder input)."""
    t = tw_of(x)
    if t is None:
        B, r, D = x.shape
        xb, st = _e((B * r, D), BF16, x.device), _e((B * r, D // 64, 2), F32, x.device)
        ops.rowstats_cast(x, r * D, B, r, D, xb, st)
        t = x._dav_tw = (xb, st)
    return t


class _Fold:
    """gamma-folded bf16 copy of a Linear's weight behind a LayerNorm + the two epilogue vectors (dav_ln_fold_grouped)."""
    __slots__ = ('wl', 'c', 'd', 'stamp', 'lin', 'norm')


_FOLD_SCOPE = None          # dict id(weight) -> _Fold of the model whose forward is running (filled lazily; refreshed as ONE launch)


_FOLD_GEN = [0]             # bumped whenever fp32 masters change behind torch's version counters (optimizer kernel, graph replay)


def bump_fold_generation():
    _FOLD_GEN[0] += 1


def _fold_stamp(lin, norm):
    ps = (lin.weight, norm.weight, norm.bias) + ((lin.bias,) if lin.bias is not None else ())
    return (_FOLD_GEN[0],) + tuple((p._version, p.data_ptr()) for p in ps)


def _fold_items(folds):
    return [(f.lin.weight.detach().view(f.lin.weight.shape[0], -1), f.norm.weight.detach(), f.norm.bias.detach(),
             f.lin.bias.detach() if f.lin.bias is not None else None, f.wl, f.c, f.d) for f in folds]


def ln_fold(lin, norm):
    """-> _Fold of (lin, norm), brought up to date when a master changed (optimizer step, load_state_dict)."""
    f = lin.weight.__dict__.get('_dav_fold')
    if f is None or f.norm is not norm or f.wl.device != lin.weight.device:
        f = _Fold()
        N, K = lin.weight.shape[0], lin.weight.numel() // lin.weight.shape[0]
        dev = lin.weight.device
        f.wl, f.c, f.d, f.stamp, f.lin, f.norm = _e((N, K), BF16, dev), _e((N,), F32, dev), _e((N,), F32, dev), None, lin, norm
        lin.weight.__dict__['_dav_fold'] = f
    if _FOLD_SCOPE is not None:
        _FOLD_SCOPE[id(lin.weight)] = f
        if id(f) in _FOLD_PENDING:          # refreshed on the side stream by this step's fold_scope: join it once per consuming stream
            cur = torch.cuda.current_stream()
            if cur.cuda_stream not in _FOLD_JOINED:
                cur.wait_stream(_FOLD_SIDE[0])
                _FOLD_JOINED.add(cur.cuda_stream)
    stamp = _fold_stamp(lin, norm)
    if f.stamp != stamp:
        with ops.unbatched():
            ops.ln_fold_grouped(_fold_items([f]))
        f.stamp = stamp
    return f


FOLD_EARLY = 8              # folds (in order of first use: the first encoder layer's) refreshed on the main stream; the rest beside it
_FOLD_PENDING = set()       # id(_Fold) of the folds this step's refresh put on the side stream
_FOLD_JOINED = set()        # streams that have waited for it
_FOLD_SIDE = [None]


class fold_scope:
    """``with fold_scope(model):`` around a forward: the folds the model has used so far are refreshed by ONE grouped launch when any
    master changed since (every training step) — always inside a stream capture, so that a replayed graph re-folds from the weights the
    captured optimizer pass wrote; folds met for the first time are made on the spot and join the scope."""

    def __init__(self, root):
        self.root = root
        self.main = None

    def __enter__(self):
        global _FOLD_SCOPE
        self.prev = _FOLD_SCOPE
        if self.prev is not None or not (LN_FUSE and PRECISION == 'bf16'):
            return self
        scope = self.root.__dict__.setdefault('_dav_fold_scope', {})
        _FOLD_SCOPE = scope
        folds = list(scope.values())
        if folds:
            capturing = folds[0].wl.is_cuda and torch.cuda.is_current_stream_capturing()
            stamps = [_fold_stamp(f.lin, f.norm) for f in folds]
            todo = folds if capturing else [f for f, s_ in zip(folds, stamps) if f.stamp != s_]
            if todo:
                # the first layer's folds at once; the bulk (0.8 GB of traffic) on a side stream under the first layer's kernels —
                # nothing else runs at the top of a step, on the main stream the refresh was 0.3 ms of step time
                early, late = todo[:FOLD_EARLY], todo[FOLD_EARLY:]
                with ops.unbatched():
                    ops.ln_fold_grouped(_fold_items(early))
                    if late and folds[0].wl.is_cuda and os.environ.get('DAV_STREAMS', '1') != '0':
                        if _FOLD_SIDE[0] is None:
                            _FOLD_SIDE[0] = torch.cuda.Stream(folds[0].wl.device)
                        self.main = torch.cuda.current_stream()
                        _FOLD_SIDE[0].wait_stream(self.main)
                        with torch.cuda.stream(_FOLD_SIDE[0]):
                            ops.ln_fold_grouped(_fold_items(late))
                        _FOLD_PENDING.update(id(f) for f in late)
                    elif late:
                        ops.ln_fold_grouped(_fold_items(late))
                for f in todo:
                    f.stamp = _fold_stamp(f.lin, f.norm)
        return self

    def __exit__(self, *exc):
        global _FOLD_SCOPE
        if self.prev is None and _FOLD_PENDING:
            if self.main.cuda_stream not in _FOLD_JOINED:          # (a capture must see every forked stream re-joined)
                self.main.wait_stream(_FOLD_SIDE[0])
            _FOLD_PENDING.clear()
            _FOLD_JOINED.clear()
        _FOLD_SCOPE = self.prev
        return False


class region:
    """``with region(): <independent launches>``: inside a batch the launches form ONE step of the current lane (so they
    are grouped with each other and with the other lanes' launches of that step); outside a batch the region is a batch of
    its own.  Every call inside must be a single-kernel library launch (or calls of identical kernel sequence)."""

    def __enter__(self):
        self.own = None
        if _BATCH is not None:
            ops.batch_region(True)
        elif batching_allowed():
            self.own = batch(auto_lanes=True)
            self.own.__enter__()
        return self

    def __exit__(self, et, ev, tb):
        if self.own is not None:
            return self.own.__exit__(et, ev, tb)
        if _BATCH is not None and et is None:
            ops.batch_region(False)
        return False


# ------------------------------------------------------------------------------------------------
# deferred weight gradients: inside `deferred_wgrads()` lin_bwd only records its wgrad problem;
# `flush_wgrads()` launches everything recorded so far as one grouped GEMM (dav_gemm_tn_grouped_bf16)
# and then reports the parameters ready.
# ------------------------------------------------------------------------------------------------
# DAV_WGRAD_GANG (default 1): a flush with at least WGRAD_GANG_MIN_TILES 256 x 256 tiles goes out as ONE gang-scheduled launch
# (dav_gemm_tn_gang_bf16); 0 = the 128 x 128 grouped kernel, <= 40 problems per launch.  DAV_WGRAD_MERGE (default 0 = all): encoder
# layers whose weight gradients share a launch — the flush after a layer's backward is skipped unless the layer closes a group (or a
# captured segment: ``WGRAD_FLUSH_LAYERS``, set by util.misc.GraphedStep to its cuts).  A whole tile per workgroup only balances
# when thousands of tiles share a launch (profiles/r05_tn_gang_*.txt: 12 launches 2.83 ms, one launch 1.95 ms).
WGRAD_GANG = os.environ.get('DAV_WGRAD_GANG', '1') != '0'
WGRAD_GANG_MIN_TILES = 512      # (one ViT-B layer = 346 tiles: 193 us on the 128 x 128 kernel, 202 us as a gang launch; two layers 386 vs 347 us)
WGRAD_MERGE = int(os.environ.get('DAV_WGRAD_MERGE', '0'))
WGRAD_FLUSH_LAYERS = None      # None: no captured step in progress; a set (possibly empty): the segment cuts of the step being captured


WGRAD_EAGER_DP_MERGE = 3      # eager data-parallel backward (a grad-ready hook, no captured segments): layers per weight-gradient launch


def wgrad_flush_due(layer: int, depth: int) -> bool:
    """Whether the queued weight gradients go out after encoder layer ``layer``'s backward (layers run depth-1 .. 0)."""
    if not WGRAD_GANG or layer == 0 or layer in (WGRAD_FLUSH_LAYERS or ()):
        return True
    if WGRAD_MERGE > 0:
        return (depth - layer) % WGRAD_MERGE == 0
    # merge-all, and no captured step in progress (a capture declares its segment cuts, an empty set for one graph): with a
    # gradient-ready hook installed — the data-parallel reducer of an EAGER step — one launch at layer 0 would report every encoder
    # gradient at the very end (no bucket could start its all-reduce under the backward) and keep every layer's operands alive until
    # then: a bounded merge instead
    return _GRAD_READY is not None and WGRAD_FLUSH_LAYERS is None and (depth - layer) % WGRAD_EAGER_DP_MERGE == 0


_DEFERRED = None
_DEFERRED_LN = None          # deferred LayerNorm dgamma/dbeta reductions (partial-row workspaces)
_DEFERRED_LN_READY = []


class deferred_wgrads:
    def __enter__(self):
        global _DEFERRED, _DEFERRED_LN
        self.prev = (_DEFERRED, _DEFERRED_LN)
        _DEFERRED, _DEFERRED_LN = [], []
        return self

    def __exit__(self, *exc):
        global _DEFERRED, _DEFERRED_LN
        if exc[0] is None:
            flush_wgrads()
        _DEFERRED, _DEFERRED_LN = self.prev


# ------------------------------------------------------------------------------------------------
# Written (not accumulated) first contributions.  A captured step (util.misc.GraphedStep) knows that every gradient is consumed
# and re-initialised by the fused AdamW pass at its end, so the FIRST weight-gradient GEMM into a Linear weight may write its
# tile instead of read-modify-writing it, and AdamW need not zero-fill that weight's gradient: the gradient tiles' fp32 reads
# (a burst at the end of every tile, 25 % of the weight-gradient kernel) and 4 of AdamW's 34 bytes per parameter go away.
# Only full-weight problems of the grouped launches qualify; a weight that saw an immediate (un-deferred) contribution earlier
# in the step, or whose problem covers a column block only, keeps the accumulate / zero-fill pair.
# ------------------------------------------------------------------------------------------------
_OVERWRITE = None          # None: off;  dict(written=set of C addresses, touched=set of C addresses, params=set of id(weight))


def wgrad_overwrite_begin():
    global _OVERWRITE
    _OVERWRITE = dict(written=set(), touched=set(), params={})


def wgrad_overwrite_end():
    """-> the weights (parameters) whose gradient the step wrote instead of accumulating."""
    global _OVERWRITE
    st, _OVERWRITE = _OVERWRITE, None
    return list(st['params'].values()) if st else []


def deferred_operands_to(stream):
    """The queued weight-gradient problems will be launched on ``stream`` although their operands were allocated while another
    stream was current: tell the caching allocator, so that a block is not handed out again on its home stream while the
    weight-gradient kernel may still be reading it."""
    for pr in (_DEFERRED or []):
        for k in ('A', 'B'):
            t = pr.get(k)
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)
    for item in (_DEFERRED_LN or []):          # (workspace of partial rows, dgamma, dbeta, rows, D)
        if torch.is_tensor(item[0]) and item[0].is_cuda:
            item[0].record_stream(stream)


def flush_wgrads():
    """Launch the queued weight-gradient problems on the current stream (their operands must be complete on it)."""
    if _DEFERRED_LN:
        items = list(_DEFERRED_LN)
        del _DEFERRED_LN[:]
        ops.layernorm_bwd_reduce_grouped(items)
        for pair in _DEFERRED_LN_READY:
            _ready(*pair)
        del _DEFERRED_LN_READY[:]
    if not _DEFERRED:
        return
    probs = list(_DEFERRED)
    del _DEFERRED[:]
    # a grouped launch lets ONE workgroup own each output tile (plain read-modify-write), so two problems that
    # accumulate into the same weight (decoder_embed: patch tokens and fusion tokens) must not share a launch
    while probs:
        seen, now, later = set(), [], []
        for pr in probs:
            key = pr['C'].data_ptr()
            (later if key in seen else now).append(pr)
            seen.add(key)
        # longest contractions first: tiles are dispatched in list order as workgroup slots free up, and a tile's run time is
        # proportional to its contraction length (49 .. 95 k-steps in one launch) — the short ones fill the tail
        now.sort(key=lambda pr: -pr['Mc'])
        if _OVERWRITE is not None:
            for pr in now:
                key = pr['C'].data_ptr()
                pr['overwrite'] = bool(pr.get('weight') is not None and key not in _OVERWRITE['written'] and key not in _OVERWRITE['touched'])
                _OVERWRITE['written'].add(key)
                if pr.get('weight') is None:                       # a column block: the weight as a whole stays on accumulate / zero-fill
                    _OVERWRITE['touched'].add(pr.get('gbase', key))
                if pr['overwrite']:
                    _OVERWRITE['params'][id(pr['weight'])] = pr['weight']
        if WGRAD_GANG and PRECISION == 'bf16' and (sum(-(-pr['N'] // 256) * -(-pr['K'] // 256) for pr in now) >= WGRAD_GANG_MIN_TILES
                                                   or any(pr['Mc'] % 64 for pr in now)):      # (a ragged contraction has no other fast kernel)
            ops.gemm_tn_gang(now)          # one persistent launch of 256 x 256 tiles, gangs of panel-sharing tiles per XCD
        else:
            grouped = [pr for pr in now if pr['Mc'] % 64 == 0 or pr['A'].dtype == F32]
            if grouped:
                ops.gemm_tn_grouped(grouped)
            for pr in now:                 # (a ragged contraction in a flush too small for the gang launch: the one-problem kernel)
                if pr['Mc'] % 64 and pr['A'].dtype != F32:
                    ops.gemm_tn(pr['A'], pr['B'], pr['Mc'], pr['N'], pr['K'], pr['C'], lda=pr['lda'], ldb=pr['ldb'], ldc=pr['ldc'],
                                a_rowmap=pr.get('a_rowmap'), b_rowmap=pr.get('b_rowmap'), beta=0 if pr.get('overwrite') else 1,
                                bias_grad=pr.get('bias_grad'))
        for pr in now:
            _ready(*pr['ready'])
        probs = later


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def ln_fwd(norm, x0, x1, B, eps=None, want_f32=False, want_bf16=True):
    """LayerNorm over rows [x0 rows | x1 rows] per batch element. x*: fp32 [B, r, D] (contiguous) or None."""
    D = norm.weight.shape[0]
    r0 = x0.shape[1] if x0 is not None else 0
    r1 = x1.shape[1] if x1 is not None else 0
    if x0 is None:
        x0, r0, x1, r1 = x1, r1, None, 0
    dev = x0.device
    M = B * (r0 + r1)
    y = _e((M, D), BF16, dev) if want_bf16 else None
    y32 = _e((M, D), F32, dev) if want_f32 else None
    if PRECISION == 'fp32' and y is not None and y32 is not None:
        y32 = y                           # one fp32 output serves as operand copy and as fp32 result
    mean, rstd = _e((M,), F32, dev), _e((M,), F32, dev)
    ops.layernorm_fwd(x0, r0 * D, r0, x1, r1 * D, r1, B, D, norm.weight, norm.bias, norm.eps if eps is None else eps,
                      y, y32, mean, rstd)
    return y, y32, (mean, rstd)


def ln_bwd(norm, x0, x1, B, stats, dy_bf16=None, dy_f32=None, *, dx0=None, acc0=0, res0=None, dx0_bf16=None,
           dx1=None, acc1=0, res1=None, dx1_bf16=None):
    D = norm.weight.shape[0]
    r0 = x0.shape[1] if x0 is not None else 0
    r1 = x1.shape[1] if x1 is not None else 0
    if x0 is None:
        x0, r0, x1, r1 = x1, r1, None, 0
        dx0, acc0, res0, dx0_bf16, dx1, acc1, res1, dx1_bf16 = dx1, acc1, res1, dx1_bf16, None, 0, None, None
    ops.layernorm_bwd(x0, r0 * D, r0, x1, r1 * D, r1, B, D, dy_bf16, dy_f32, norm.weight, stats[0], stats[1],
                      dx0, r0 * D, acc0, res0, r0 * D, dx0_bf16, r0 * D,
                      dx1, r1 * D, acc1, res1, r1 * D, dx1_bf16, r1 * D,
                      gbuf(norm.weight), gbuf(norm.bias), defer=_DEFERRED_LN)
    if _DEFERRED_LN is None:
        _ready(norm.weight, norm.bias)
    else:
        _DEFERRED_LN_READY.append((norm.weight, norm.bias))


def lin_fwd(lin, a, M, *, a_rowmap=None, lda=None, act=0, res=None, res_rowmap=None, out=None, out_bf16=False,
            c_rowmap=None, ldc=None, C2=None, c2_mode=0, w_col_off=0, k=None, use_bias=True):
    """y = a @ W[:, w_col_off : w_col_off+k]^T (+ bias) with the fused epilogue of dav_gemm_nt_bf16."""
    W = wcache(lin.weight)
    N, Kfull = W.shape
    K = Kfull if k is None else k
    dev = a.device
    if out is None:
        out = _e((M, N), BF16 if out_bf16 else F32, dev)
    Bw = W if w_col_off == 0 else W.view(-1)[w_col_off:]
    ops.gemm_nt(a, Bw, M, N, K, lda=lda if lda is not None else K, ldb=Kfull, a_rowmap=a_rowmap,
                bias=lin.bias if use_bias else None, act=act, res=res, ldres=N, res_rowmap=res_rowmap,
                C_out=out, ldc=ldc if ldc is not None else N, c_bf16=out.dtype == BF16, c_rowmap=c_rowmap,
                C2=C2, ldc2=N, c2_mode=c2_mode)
    return out


def lin_bwd(lin, dy, a, M, *, dy_rowmap=None, a_rowmap=None, lda=None, need_dx=True, gelu_aux=None, dx=None, dx_bf16=True,
            dx_rowmap=None, dx_beta=0, dx_C2=None, dx_c2_mode=0, w_col_off=0, k=None, use_bias=True, final=True,
            dx_res=None, dx_res_rowmap=None, wgrad=True):
    """Backward of lin_fwd: dx = dy @ W (optionally * GELU'(aux)), dW += dy^T a, db += colsum(dy).

    dy: bf16 [*, N]; a: bf16 [*, K] (the forward input).  Returns dx (bf16 [M, K] unless given)."""
    W = wcache(lin.weight)
    N, Kfull = W.shape
    K = Kfull if k is None else k
    dev = dy.device
    if need_dx:
        if dx is None:
            dx = _e((M, K), BF16 if dx_bf16 else F32, dev)
        if N % 64 == 0 or PRECISION == 'fp32':      # read W [N, K] itself as the [contraction, out] operand (transposing LDS reads)
            Bt, ldb, variant = (W if w_col_off == 0 else W.view(-1)[w_col_off:]), Kfull, 1 << 12
        else:
            WT = wcache_t(lin.weight)
            Bt, ldb, variant = (WT if w_col_off == 0 else WT[w_col_off:]), N, 0
        ops.gemm_nt(dy, Bt, M, K, N, lda=N, ldb=ldb, a_rowmap=dy_rowmap, act=3 if gelu_aux is not None else 0,
                    aux=gelu_aux, ldaux=K, res=dx_res, ldres=K, res_rowmap=dx_res_rowmap,
                    C_out=dx, ldc=K, c_bf16=dx.dtype == BF16, c_rowmap=dx_rowmap, beta=dx_beta,
                    C2=dx_C2, ldc2=K, c2_mode=dx_c2_mode, variant=variant)
    if wgrad:
        lin_wgrad(lin, dy, a, M, dy_rowmap=dy_rowmap, a_rowmap=a_rowmap, lda=lda, w_col_off=w_col_off, k=k, use_bias=use_bias, final=final)
    return dx


def lin_wgrad(lin, dy, a, M, *, dy_rowmap=None, a_rowmap=None, lda=None, w_col_off=0, k=None, use_bias=True, final=True):
    """The weight-gradient half of lin_bwd: dW += dy^T a, db += colsum(dy) (queued inside ``deferred_wgrads()``).  On its own for the
    Linears behind a folded LayerNorm, whose operand ``a`` only exists once the LayerNorm backward has re-made it (ln_bwd_tw)."""
    N, Kfull = lin.weight.shape[0], lin.weight.numel() // lin.weight.shape[0]
    K = Kfull if k is None else k
    gw = gbuf(lin.weight)
    gwv = gw.view(N, -1)
    Cw = gwv if w_col_off == 0 else gwv.view(-1)[w_col_off:]
    gb = gbuf(lin.bias) if (use_bias and lin.bias is not None) else None
    if _DEFERRED is not None and N % 8 == 0 and K % 8 == 0 and (M % 64 == 0 or (WGRAD_GANG and PRECISION == 'bf16')):
        # weight gradients are off the dependency chain: queue them and launch ONE grouped GEMM per flush
        _DEFERRED.append(dict(A=dy, B=a, Mc=M, N=N, K=K, C=Cw, lda=N, ldb=lda if lda is not None else K, ldc=Kfull,
                              a_rowmap=dy_rowmap, b_rowmap=a_rowmap, bias_grad=gb,
                              ready=(lin.weight, lin.bias) if final else (),
                              gbase=gwv.data_ptr(),
                              weight=lin.weight if (w_col_off == 0 and K == Kfull) else None))      # full weight: may be written
        return
    if _OVERWRITE is not None:
        _OVERWRITE['touched'].add(gwv.data_ptr())
    ops.gemm_tn(dy, a, M, N, K, Cw, lda=N, ldb=lda if lda is not None else K, ldc=Kfull, a_rowmap=dy_rowmap,
                b_rowmap=a_rowmap, beta=1, bias_grad=gb)
    if final:
        _ready(lin.weight, lin.bias)


def lin_fwd_ln(lin, norm, eps, t0, M, *, t1=None, r0=0, r1=0, a_rowmap=None, act=0, out_bf16=True, out=None, C2=None, c2_mode=0):
    """lin(norm(x)) with the LayerNorm folded into the GEMM (dav_gemm_nt_ln_bf16): t0 = (twin, statistics) of x.  With t1 the rows of
    a batch element are r0 rows of t0 followed by r1 rows of t1 — norm(cat((x_fusion, x_mod))) of models/deepavfusion.py:104-105."""
    f = ln_fold(lin, norm)
    N, K = f.wl.shape
    if out is None:
        out = _e((M, N), BF16 if out_bf16 else F32, f.wl.device)
    ln = dict(stats=t0[1], ln_c=f.c, eps=eps)
    if t1 is not None:
        ln.update(A2=t1[0], stats2=t1[1], a_r0=r0, a_r1=r1)
    ops.gemm_nt_ln(t0[0], f.wl, M, N, K, ln=ln, a_rowmap=a_rowmap, bias=f.d, act=act, C_out=out, c_bf16=out.dtype == BF16,
                   C2=C2, ldc2=N, c2_mode=c2_mode)
    return out


def lin_fwd_tw(lin, a, M, res, *, res_rowmap=None, res_rows=None, out=None, tw=None, c_rowmap=None, C2=None, c2_mode=0, rows_out=None):
    """res + lin(a) in fp32 together with its bf16 twin and row-statistics partials, all from the one epilogue -> (out, (twin, stats)).
    ``out`` / ``tw`` given: several GEMMs fill row groups of one activation through ``c_rowmap`` (the fusion block's three proj)."""
    W = wcache(lin.weight)
    N, K = W.shape
    dev = a.device
    R = M if rows_out is None else rows_out
    if out is None:
        out = _e((R, N), F32, dev)
    if tw is None:
        tw = (_e((R, N), BF16, dev), _e((R, N // 64, 2), F32, dev))
    ops.gemm_nt_ln(a, W, M, N, K, prod=dict(stats_out=tw[1], twin_out=tw[0], ld_twin=N), bias=lin.bias, res=res, ldres=N,
                   res_rowmap=res_rowmap, res_rows=res_rows, C_out=out, c_rowmap=c_rowmap, C2=C2, ldc2=N, c2_mode=c2_mode)
    return out, tw


def ln_bwd_tw(norm, eps, t0, r0, t1, r1, B, dy_bf16=None, dy_f32=None, *, h_out=None, dx0=None, acc0=0, res0=None, dx0_bf16=None,
              dx1=None, acc1=0, res1=None, dx1_bf16=None):
    """Backward of a LayerNorm that was folded away on the forward path (dav_layernorm_bwd_twin): x comes as twin + statistics
    (t0: r0 rows | t1: r1 rows per batch element, both dense), ``h_out`` receives the LayerNorm OUTPUT — the operand of the consuming
    Linear's weight gradient."""
    D = norm.weight.shape[0]
    if t0 is None:
        t0, r0, t1, r1 = t1, r1, None, 0
        dx0, acc0, res0, dx0_bf16, dx1, acc1, res1, dx1_bf16 = dx1, acc1, res1, dx1_bf16, None, 0, None, None
    ops.layernorm_bwd_twin(t0[0], r0 * D, t0[1], r0, t1[0] if t1 is not None else None, r1 * D, t1[1] if t1 is not None else None, r1,
                           B, D, eps, dy_bf16, dy_f32, norm.weight, norm.bias,
                           dx0, r0 * D, acc0, res0, r0 * D, dx0_bf16, r0 * D,
                           dx1, r1 * D, acc1, res1, r1 * D, dx1_bf16, r1 * D,
                           h_out=h_out, dgamma=gbuf(norm.weight), dbeta=gbuf(norm.bias), defer=_DEFERRED_LN)
    if _DEFERRED_LN is None:
        _ready(norm.weight, norm.bias)
    else:
        _DEFERRED_LN_READY.append((norm.weight, norm.bias))


def attention_fwd(q, k, v, B, H, Nq, Nk, dqk, dv, scale, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, dev, keep=None):
    """q/k/v are (tensor, element_offset) pairs into bf16 buffers. Returns O [B*Nq, H*dv] bf16 and LSE.
    ``keep`` = (bytes 0 / 1 [B, H, Nq, ld], ld, 1 / (1 - p)): attention dropout (see draw_attn_keep)."""
    O = _e((B * Nq, H * dv), BF16, dev)
    LSE = _e((B, H, Nq), F32, dev)
    ops.hold(q[0], k[0], v[0])
    es = q[0].element_size()
    if keep is not None:
        ops.attn_drop_fwd(q[0].data_ptr() + es * q[1], k[0].data_ptr() + es * k[1], v[0].data_ptr() + es * v[1], O, LSE, B, H, Nq, Nk,
                          dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, Nq * H * dv, H * dv, scale, keep[0], keep[1], keep[2])
        return O, LSE
    ops.attn_fwd(q[0].data_ptr() + es * q[1], k[0].data_ptr() + es * k[1], v[0].data_ptr() + es * v[1], O, LSE, B, H, Nq, Nk,
                 dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, Nq * H * dv, H * dv, scale)
    return O, LSE


_ATTN_CTX = True      # (False: a torch fill pass zeroes the context rows' dq slots, rounds 1-2)


def attention_bwd(q, k, v, O, dO, LSE, dq, dk, dvv, B, H, Nq, Nk, dqk, dv, scale, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                  dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, part=3, Delta=None, dq_ctx_rows=0, keep=None):
    """``part`` 1 / 2: only the dQ (+ Delta) / only the dK-dV kernel, with the same ``Delta`` buffer passed to both calls
    (lets a caller put the two kernels of several attentions into two regions of a launch batch)."""
    if Delta is None:
        Delta = torch.empty_like(LSE)
    ops.hold(q[0], k[0], v[0], dq[0], dk[0], dvv[0])
    p = lambda t: t[0].data_ptr() + t[0].element_size() * t[1]
    if keep is not None:
        ops.attn_drop_bwd(p(q), p(k), p(v), O, dO, LSE, Delta, p(dq), p(dk), p(dvv), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                          v_bs, v_rs, Nq * H * dv, H * dv, Nq * H * dv, H * dv, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale,
                          keep[0], keep[1], keep[2], part=part, dq_ctx_rows=dq_ctx_rows)
        return
    ops.attn_bwd(p(q), p(k), p(v), O, dO, LSE, Delta, p(dq), p(dk), p(dvv), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                 v_bs, v_rs, Nq * H * dv, H * dv, Nq * H * dv, H * dv, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, part=part,
                 dq_ctx_rows=dq_ctx_rows)


def to_bf16(x):
    if PRECISION == 'fp32':
        return x.contiguous()
    y = torch.empty(x.shape, dtype=BF16, device=x.device)
    ops.cast_bf16(x.contiguous(), y)
    return y


# ------------------------------------------------------------------------------------------------
# timm Block (pre-LN) with optional fusion-token context rows
# ------------------------------------------------------------------------------------------------
# ---- dropout (nn.Dropout with p > 0 in training mode: the fine-tuning constructors' attn_drop / drop; every pre-training config
# has 0 and none of this runs).  The DRAW is torch's (draw_attn_keep / draw_keep, made by the caller BEFORE a launch batch opens: a
# torch RNG kernel must not overtake recorded launches); the kernels are deterministic in the byte masks.
def draw_attn_keep(p, B, H, Nq, Nk, dev, sample=None):
    """Keep mask of an attention-probability dropout: (bytes 0 / 1 [B, H, Nq, ld], ld, 1 / (1 - p)), ld = Nk rounded up to 32.
    ``sample((B, H, Nq, Nk))`` -> 0 / 1 tensor replaces the Bernoulli draw (parity tests inject the reference's masks)."""
    ld = (Nk + 31) // 32 * 32
    if sample is None:
        m = (torch.rand((B, H, Nq, ld), device=dev) >= p).to(torch.uint8)
    else:
        m = torch.zeros((B, H, Nq, ld), device=dev, dtype=torch.uint8)
        m[..., :Nk] = sample((B, H, Nq, Nk)).to(device=dev, dtype=torch.uint8)
    return m, ld, 1.0 / (1.0 - p)


def draw_keep(p, rows, D, dev, sample=None):
    """Keep mask of a dropout on a [rows, D] activation: (bytes 0 / 1 [rows, D], 1 / (1 - p))."""
    if sample is None:
        m = (torch.rand((rows, D), device=dev) >= p).to(torch.uint8)
    else:
        m = sample((rows, D)).to(device=dev, dtype=torch.uint8).contiguous()
    return m, 1.0 / (1.0 - p)


def _drop(x, m, B, rows, D):
    """x <- keep * x / (1 - p), in place (x: operand-dtype or fp32 [B * rows, D]; forward and backward of an nn.Dropout alike)."""
    ops.dropout_rows(x, m[0], m[1], B, rows, D, x)


def _branch_grad(g, scale, drop, B, rows, D):
    """Gradient entering a residual BRANCH that ended in [Dropout ->] DropPath: operand-dtype(scale[b] * keep / (1 - p) * g); the
    residual path passes g through unchanged."""
    out = _e((B * rows, D), BF16, g.device)
    if drop is None and PRECISION != 'fp32':
        ops.rows_scale_cast(g, scale, B, rows, D, out)
    else:
        ops.dropout_rows(g, None if drop is None else drop[0], 1.0 if drop is None else drop[1], B, rows, D, out, rowscale=scale)
    return out


def _res_add(lin, a, M, res, B, rows, D, scale, drop=None, **kw):
    """res + [scale[b] *] [dropout] lin(a): fused into the GEMM epilogue, or — with a DropPath scale and / or a dropout mask — GEMM
    to a temporary followed by one masked, per-sample scaled add (timm: x + drop_path(proj_drop(proj(...))))."""
    if scale is None and drop is None:
        return lin_fwd(lin, a, M, res=res, **kw)
    y = lin_fwd(lin, a, M, **kw)
    out = _e((M, D), F32, a.device)
    if drop is None:
        ops.rows_axpy(res, y, scale, B, rows, D, out)
    else:
        ops.dropout_rows(y, drop[0], drop[1], B, rows, D, out, res=res, rowscale=scale)
    return out


def block_fwd(blk, x_mod, x_fus, heads, eps, dp=None, idle_before_mlp=0, dr=None):
    """x_mod fp32 [B,n,D]; x_fus fp32 [B,nF,D] or None (context rows: keys/values only —
    models/deepavfusion.py:104-105).  ``dp`` = (s_attn, s_mlp): per-sample DropPath scales (fp32 [B], 0 or 1/keep) of the
    two residual branches, None when inactive.  ``dr``: dropout masks of this call — {'attn': draw_attn_keep(...), 'proj' / 'fc1' /
    'fc2': draw_keep(...)}, a missing key = p 0 (timm Attention.attn_drop / proj_drop, Mlp.drop1 / drop2); None when inactive.
    Returns (x_out fp32 [B,n,D], tape)."""
    B, n, D = x_mod.shape
    nF = x_fus.shape[1] if x_fus is not None else 0
    R, hd, dev = nF + n, D // heads, x_mod.device
    M, Mq = B * R, B * n
    if dp is None and not dr and ln_fuse_ok(D):
        return _block_fwd_ln(blk, x_mod, x_fus, heads, eps, idle_before_mlp)
    dr = dr or {}
    h1, _, st1 = ln_fwd(blk.norm1, x_fus, x_mod, B, eps)
    qkv = lin_fwd(blk.attn.qkv, h1, M, out_bf16=True)                                       # [B*R, 3D]
    o, lse = attention_fwd((qkv, nF * 3 * D), (qkv, D), (qkv, 2 * D), B, heads, n, R, hd, hd, hd ** -0.5,
                           R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, dev, keep=dr.get('attn'))
    x1 = _res_add(blk.attn.proj, o, Mq, x_mod, B, n, D, None if dp is None else dp[0], drop=dr.get('proj')).view(B, n, D)
    lane_skip(idle_before_mlp)          # batched beside a fusion block: line norm2 / fc1 / fc2 up with its norm2 / fc1 / fc2
    h2, _, st2 = ln_fwd(blk.norm2, None, x1, B, eps)
    Hd = blk.mlp.fc1.weight.shape[0]
    z = _e((Mq, Hd), BF16, dev)
    u = lin_fwd(blk.mlp.fc1, h2, Mq, act=1, out_bf16=True, C2=z, c2_mode=4)
    if 'fc1' in dr:
        _drop(u, dr['fc1'], B, n, Hd)       # fc2 (and its weight gradient) contract the DROPPED activation; z keeps GELU's input
    x2 = _res_add(blk.mlp.fc2, u, Mq, x1, B, n, D, None if dp is None else dp[1], drop=dr.get('fc2')).view(B, n, D)
    tape = dict(x_mod=x_mod, x_fus=x_fus, h1=h1, st1=st1, qkv=qkv, o=o, lse=lse, x1=x1, h2=h2, st2=st2, z=z, u=u,
                heads=heads, nF=nF, dp=dp, dr=dr or None)
    return x2, tape


def _block_fwd_ln(blk, x_mod, x_fus, heads, eps, idle_before_mlp=0):
    """block_fwd with both LayerNorms folded into the GEMMs around them: qkv / fc1 contract the raw twins of their inputs, proj / fc2
    write the twin + row statistics of the residual stream they produce.  Saved for the backward: twins and statistics instead of the
    two normalised tensors and the two fp32 residual tensors."""
    B, n, D = x_mod.shape
    nF = x_fus.shape[1] if x_fus is not None else 0
    R, hd, dev = nF + n, D // heads, x_mod.device
    M, Mq = B * R, B * n
    tm = ensure_tw(x_mod)
    tf = ensure_tw(x_fus) if nF else None
    if nF:
        qkv = lin_fwd_ln(blk.attn.qkv, blk.norm1, eps, tf, M, t1=tm, r0=nF, r1=n)            # [B*R, 3D]
    else:
        qkv = lin_fwd_ln(blk.attn.qkv, blk.norm1, eps, tm, M)
    o, lse = attention_fwd((qkv, nF * 3 * D), (qkv, D), (qkv, 2 * D), B, heads, n, R, hd, hd, hd ** -0.5,
                           R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, dev)
    x1, tw1 = lin_fwd_tw(blk.attn.proj, o, Mq, x_mod)
    lane_skip(idle_before_mlp)
    Hd = blk.mlp.fc1.weight.shape[0]
    z = _e((Mq, Hd), BF16, dev)
    u = lin_fwd_ln(blk.mlp.fc1, blk.norm2, eps, tw1, Mq, act=1, C2=z, c2_mode=4)
    x2, tw2 = lin_fwd_tw(blk.mlp.fc2, u, Mq, x1)
    x2 = tw_set(x2.view(B, n, D), *tw2)
    tape = dict(ln_fused=True, eps=eps, x_mod=x_mod, x_fus=x_fus, tm=tm, tf=tf, tw1=tw1, qkv=qkv, o=o, lse=lse, z=z, u=u,
                heads=heads, nF=nF, dp=None)
    return x2, tape


def block_bwd(blk, t, g2, g2b, *, dx_fus=None, dx_fus_acc=0, dx_mod=None, dx_mod_acc=0, need_dx=True, before_ln1=None):
    """g2 fp32 [B,n,D] (+ bf16 twin g2b or None).  Writes/accumulates the fusion-row gradient into dx_fus
    and the modality-row gradient (incl. the residual path) into dx_mod; returns (dx_mod, dx_mod_bf16, dx_fus).
    = block_bwd_head (everything down to the qkv input gradient) + block_bwd_tail (the norm1 backward, the only kernel
    that touches dx_fus / dx_mod): callers that batch two blocks in lockstep wait for the producers of those buffers
    between the two halves."""
    st = block_bwd_head(blk, t, g2, g2b)
    if before_ln1 is not None and need_dx:
        before_ln1()          # e.g. wait for the stream that produced the buffers this LayerNorm accumulates into
    return block_bwd_tail(blk, t, st, dx_fus=dx_fus, dx_fus_acc=dx_fus_acc, dx_mod=dx_mod, dx_mod_acc=dx_mod_acc, need_dx=need_dx)


def block_bwd_head(blk, t, g2, g2b, idle_before_attn=0):
    x_mod, x_fus, heads, nF = t['x_mod'], t['x_fus'], t['heads'], t['nF']
    B, n, D = x_mod.shape
    R, hd, dev = nF + n, D // heads, x_mod.device
    M, Mq = B * R, B * n
    dp = t.get('dp')
    dr = t.get('dr') or {}
    if dp is not None or 'fc2' in dr:       # the branch sees s[b] * keep / (1 - p) * g, the residual path g itself
        g2b = _branch_grad(g2, None if dp is None else dp[1], dr.get('fc2'), B, n, D)
    elif g2b is None:
        g2b = to_bf16(g2)
    fused = t.get('ln_fused', False)
    dz = lin_bwd(blk.mlp.fc2, g2b, t['u'], Mq, gelu_aux=t['z'])                              # [Mq, Hd] bf16 (already * GELU')
    if 'fc1' in dr:
        _drop(dz, dr['fc1'], B, n, dz.shape[1])
    g1 = _e((B, n, D), F32, dev)
    g1b = _e((Mq, D), BF16, dev)
    if fused:       # the norm2 backward re-makes fc1's weight-gradient operand (the LayerNorm output) from the twin of x1
        dh2 = lin_bwd(blk.mlp.fc1, dz, None, Mq, wgrad=False)
        h2 = _e((Mq, D), BF16, dev)
        ln_bwd_tw(blk.norm2, t['eps'], None, 0, t['tw1'], n, B, dy_bf16=dh2, h_out=h2, dx1=g1, res1=g2, dx1_bf16=g1b)
        lin_wgrad(blk.mlp.fc1, dz, h2, Mq)
    else:
        dh2 = lin_bwd(blk.mlp.fc1, dz, t['h2'], Mq)
        ln_bwd(blk.norm2, None, t['x1'], B, t['st2'], dy_bf16=dh2, dx1=g1, res1=g2, dx1_bf16=g1b)
    if dp is not None or 'proj' in dr:
        g1b = _branch_grad(g1, None if dp is None else dp[0], dr.get('proj'), B, n, D)
    do = lin_bwd(blk.attn.proj, g1b, t['o'], Mq)
    lane_skip(idle_before_attn)         # batched beside a fusion block: line the attention backward up with its cross-attentions
    dqkv = _e((M, 3 * D), BF16, dev)
    # attention writes dq for the modality rows and dk / dv for all rows: only the (dropped) queries of the fusion context rows
    # are never written and must read as zero — the dQ kernel fills them (dav_attn_bwd_ctx; the fp32 kernels have no such entry)
    ctx_in_kernel = PRECISION != 'fp32' and _ATTN_CTX
    if nF > 0 and not ctx_in_kernel:
        dqkv.view(B, R, 3 * D)[:, :nF, :D].zero_()
    qkv = t['qkv']
    attention_bwd((qkv, nF * 3 * D), (qkv, D), (qkv, 2 * D), t['o'], do, t['lse'],
                  (dqkv, nF * 3 * D), (dqkv, D), (dqkv, 2 * D), B, heads, n, R, hd, hd, hd ** -0.5,
                  R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D,
                  R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, dq_ctx_rows=nF if ctx_in_kernel else 0, keep=dr.get('attn'))
    if fused:
        dh1 = lin_bwd(blk.attn.qkv, dqkv, None, M, wgrad=False)
        return dict(dh1=dh1, g1=g1, dqkv=dqkv)
    dh1 = lin_bwd(blk.attn.qkv, dqkv, t['h1'], M)
    return dict(dh1=dh1, g1=g1)


def block_bwd_tail(blk, t, st, *, dx_fus=None, dx_fus_acc=0, dx_mod=None, dx_mod_acc=0, need_dx=True):
    x_mod, x_fus, nF = t['x_mod'], t['x_fus'], t['nF']
    B, n, D = x_mod.shape
    dev = x_mod.device
    dh1, g1 = st['dh1'], st['g1']
    if t.get('ln_fused', False):
        h1 = _e((B * (nF + n), D), BF16, dev)
        if not need_dx:
            ln_bwd_tw(blk.norm1, t['eps'], t['tf'], nF, t['tm'], n, B, dy_bf16=dh1, h_out=h1)
            lin_wgrad(blk.attn.qkv, st['dqkv'], h1, B * (nF + n))
            return None, None, None
        if dx_mod is None:
            dx_mod = _e((B, n, D), F32, dev)
            dx_mod_acc = 0
        dx_mod_b = _e((B * n, D), BF16, dev)
        if nF > 0 and dx_fus is None:
            dx_fus = _e((B, nF, D), F32, dev)
            dx_fus_acc = 0
        ln_bwd_tw(blk.norm1, t['eps'], t['tf'], nF, t['tm'], n, B, dy_bf16=dh1, h_out=h1, dx0=dx_fus, acc0=dx_fus_acc,
                  dx1=dx_mod, acc1=dx_mod_acc, res1=g1, dx1_bf16=dx_mod_b)
        lin_wgrad(blk.attn.qkv, st['dqkv'], h1, B * (nF + n))
        return dx_mod, dx_mod_b, dx_fus
    if not need_dx:
        ln_bwd(blk.norm1, x_fus, x_mod, B, t['st1'], dy_bf16=dh1)
        return None, None, None
    if dx_mod is None:
        dx_mod = _e((B, n, D), F32, dev)
        dx_mod_acc = 0
    dx_mod_b = _e((B * n, D), BF16, dev)
    if nF > 0 and dx_fus is None:
        dx_fus = _e((B, nF, D), F32, dev)
        dx_fus_acc = 0
    ln_bwd(blk.norm1, x_fus, x_mod, B, t['st1'], dy_bf16=dh1, dx0=dx_fus, acc0=dx_fus_acc,
           dx1=dx_mod, acc1=dx_mod_acc, res1=g1, dx1_bf16=dx_mod_b)
    return dx_mod, dx_mod_b, dx_fus


# ------------------------------------------------------------------------------------------------
# FusionBlock_FactorizedAVInteractions (models/fusion_blocks.py:216-289)
# ------------------------------------------------------------------------------------------------
def _factorized_fwd(fb, x_f, x_i, x_a, heads, tkns, dp=None, dr=None):
    """FusionBlock_FactorizedAVInteractions forward (models/fusion_blocks.py:266-289) as 11 dependent steps; the
    independent launches of a step sit in a ``region()`` so that they go out as one grouped grid (and, when the block is a
    lane of the layer's launch batch, together with the tower blocks' launches of that step).  With DropPath scales or dropout
    masks the sequential form below is used (it needs torch ops between kernels)."""
    if dp is not None or dr:
        return _factorized_fwd_seq(fb, x_f, x_i, x_a, heads, tkns, dp, dr)
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    nmm, nv, na = tkns
    dev, at = x_f.device, fb.attn
    Da, hd = at.q.weight.shape[0], D // heads
    rm2, rmv, rma = (nmm, nF, 0), (nv, nF, nmm), (na, nF, nmm + nv)
    fused = ln_fuse_ok(D)      # norm1_img / norm1_aud / norm2 folded into the kv / fc1 GEMMs; norm1_mm's OUTPUT is the residual base: a kernel
    # norm-then-residual: the residual base is the NORMED xmm (models/fusion_blocks.py:281-283)
    if fused:
        ti, ta = ensure_tw(x_i), ensure_tw(x_a)        # (the towers' twins of the same layer inputs)
        xmm_b, xmm32, st_mm = ln_fwd(fb.norm1_mm, None, x_f, B, want_f32=True)
        xv_b = xa_b = st_v = st_a = None
    else:
        with region():
            xmm_b, xmm32, st_mm = ln_fwd(fb.norm1_mm, None, x_f, B, want_f32=True)
            xv_b, _, st_v = ln_fwd(fb.norm1_img, None, x_i, B)
            xa_b, _, st_a = ln_fwd(fb.norm1_aud, None, x_a, B)
    with region():      # CrossAttention q / kv (models/fusion_blocks.py:46-59) of both aggregations + the pair query
        q_v = lin_fwd(at.attn_v.q, xmm_b, B * nv, a_rowmap=rmv, out_bf16=True)               # [B*nv, D]
        q_a = lin_fwd(at.attn_a.q, xmm_b, B * na, a_rowmap=rma, out_bf16=True)
        if fused:
            kv_v = lin_fwd_ln(at.attn_v.kv, fb.norm1_img, fb.norm1_img.eps, ti, B * nI)          # [B*nI, 2D]
            kv_a = lin_fwd_ln(at.attn_a.kv, fb.norm1_aud, fb.norm1_aud.eps, ta, B * nA)
        else:
            kv_v = lin_fwd(at.attn_v.kv, xv_b, B * nI, out_bf16=True)
            kv_a = lin_fwd(at.attn_a.kv, xa_b, B * nA, out_bf16=True)
        q2 = lin_fwd(at.q, xmm_b, B * nmm, a_rowmap=rm2, out_bf16=True)                      # [B*nmm, Da]
    with region():
        o_v, lse_v = attention_fwd((q_v, 0), (kv_v, 0), (kv_v, D), B, heads, nv, nI, hd, hd, hd ** -0.5,
                                   nv * D, D, nI * 2 * D, 2 * D, nI * 2 * D, 2 * D, dev)
        o_a, lse_a = attention_fwd((q_a, 0), (kv_a, 0), (kv_a, D), B, heads, na, nA, hd, hd, hd ** -0.5,
                                   na * D, D, nA * 2 * D, 2 * D, nA * 2 * D, 2 * D, dev)
    cv, ca = dict(q=q_v, kv=kv_v, o=o_v, lse=lse_v), dict(q=q_a, kv=kv_a, o=o_a, lse=lse_a)
    xmm1 = _e((B, nF, D), F32, dev)
    # proj of the two aggregations: fp32 result lands in its rows of xmm1 (+ normed-xmm residual),
    # bf16 twin of the pre-residual value feeds the pair projections
    xvo_b, xao_b = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    xmm1_2d = xmm1.view(B * nF, D)
    tw1 = (_e((B * nF, D), BF16, dev), _e((B * nF, D // 64, 2), F32, dev)) if fused else None      # twin + statistics of xmm1, by row group
    with region():
        if fused:
            lin_fwd_tw(at.attn_v.proj, o_v, B * nv, xmm32, res_rowmap=rmv, out=xmm1_2d, tw=tw1, c_rowmap=rmv, C2=xvo_b, c2_mode=2)
            lin_fwd_tw(at.attn_a.proj, o_a, B * na, xmm32, res_rowmap=rma, out=xmm1_2d, tw=tw1, c_rowmap=rma, C2=xao_b, c2_mode=2)
        else:
            lin_fwd(at.attn_v.proj, o_v, B * nv, res=xmm32, res_rowmap=rmv, out=xmm1, c_rowmap=rmv, C2=xvo_b, c2_mode=2)
            lin_fwd(at.attn_a.proj, o_a, B * na, res=xmm32, res_rowmap=rma, out=xmm1, c_rowmap=rma, C2=xao_b, c2_mode=2)
    # all (v, a) pairs: Linear(cat(xv_i, xa_j)) = W[:, :D] xv_i + W[:, D:] xa_j + b  (never materialised)
    with region():
        kv_p = lin_fwd(at.k, xvo_b, B * nv, k=D)
        ka_p = lin_fwd(at.k, xao_b, B * na, k=D, w_col_off=D, use_bias=False)
        vv_p = lin_fwd(at.v, xvo_b, B * nv, k=D)
        va_p = lin_fwd(at.v, xao_b, B * na, k=D, w_col_off=D, use_bias=False)
    P = nv * na
    Kp, Vp = _e((B * P, Da), BF16, dev), _e((B * P, D), BF16, dev)
    with region():
        ops.pair_expand(kv_p, ka_p, B, nv, na, Da, Kp)
        ops.pair_expand(vv_p, va_p, B, nv, na, D, Vp)
    scale = hd ** -0.5                                                                       # NOT (Da/heads)^-0.5 (:220-222)
    o2, lse2 = attention_fwd((q2, 0), (Kp, 0), (Vp, 0), B, heads, nmm, P, Da // heads, hd, scale,
                             nmm * Da, Da, P * Da, Da, P * D, D, dev)
    Hd = fb.mlp.fc1.weight.shape[0]
    z = _e((B * nF, Hd), BF16, dev)
    if fused:
        lin_fwd_tw(at.proj, o2, B * nmm, xmm32, res_rowmap=rm2, out=xmm1_2d, tw=tw1, c_rowmap=rm2)
        u = lin_fwd_ln(fb.mlp.fc1, fb.norm2, fb.norm2.eps, tw1, B * nF, act=1, C2=z, c2_mode=4)
        out, two = lin_fwd_tw(fb.mlp.fc2, u, B * nF, xmm1_2d)
        out = tw_set(out.view(B, nF, D), *two)
        h2 = st2 = None
    else:
        lin_fwd(at.proj, o2, B * nmm, res=xmm32, res_rowmap=rm2, out=xmm1, c_rowmap=rm2)
        h2, _, st2 = ln_fwd(fb.norm2, None, xmm1, B)
        u = lin_fwd(fb.mlp.fc1, h2, B * nF, act=1, out_bf16=True, C2=z, c2_mode=4)
        out = lin_fwd(fb.mlp.fc2, u, B * nF, res=xmm1).view(B, nF, D)
    tape = dict(dp=None, x_f=x_f, x_i=x_i, x_a=x_a, xmm_b=xmm_b, st_mm=st_mm, xv_b=xv_b, st_v=st_v, xa_b=xa_b, st_a=st_a, cv=cv, ca=ca,
                xvo_b=xvo_b, xao_b=xao_b, Kp=Kp, Vp=Vp, q2=q2, o2=o2, lse2=lse2, xmm1=xmm1, h2=h2, st2=st2, z=z, u=u,
                heads=heads, tkns=tkns, ln_fused=fused, ti=ti if fused else None, ta=ta if fused else None, tw1=tw1)
    return out, tape


def _factorized_bwd(fb, t, g, gb, *, dx_i=None, dx_a=None):
    """g fp32 [B,nF,D] grad of the block output.  Returns (dx_f fp32, dx_i, dx_a);
    dx_i / dx_a are freshly STORED (fp32 [B,n,D]) unless buffers are passed.  Same step / region structure as the forward;
    library launches only (no torch op touches a buffer a recorded kernel produces)."""
    if t.get('dp') is not None or t.get('dr'):
        return _factorized_bwd_seq(fb, t, g, gb, dx_i=dx_i, dx_a=dx_a)
    x_f, x_i, x_a, heads = t['x_f'], t['x_i'], t['x_a'], t['heads']
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    nmm, nv, na = t['tkns']
    dev, at = x_f.device, fb.attn
    Da, hd = at.q.weight.shape[0], D // heads
    P = nv * na
    rm2, rmv, rma = (nmm, nF, 0), (nv, nF, nmm), (na, nF, nmm + nv)
    cv, ca = t['cv'], t['ca']
    if gb is None:
        gb = to_bf16(g)
    dz = lin_bwd(fb.mlp.fc2, gb, t['u'], B * nF, gelu_aux=t['z'])
    g1 = _e((B, nF, D), F32, dev)                 # gradient at xmm1 = residual-path gradient of the normed xmm
    g1b = _e((B * nF, D), BF16, dev)
    if t.get('ln_fused', False):
        dh2 = lin_bwd(fb.mlp.fc1, dz, None, B * nF, wgrad=False)
        h2 = _e((B * nF, D), BF16, dev)
        ln_bwd_tw(fb.norm2, fb.norm2.eps, None, 0, t['tw1'], nF, B, dy_bf16=dh2, h_out=h2, dx1=g1, res1=g, dx1_bf16=g1b)
        lin_wgrad(fb.mlp.fc1, dz, h2, B * nF)
    else:
        dh2 = lin_bwd(fb.mlp.fc1, dz, t['h2'], B * nF)
        ln_bwd(fb.norm2, None, t['xmm1'], B, t['st2'], dy_bf16=dh2, dx1=g1, res1=g, dx1_bf16=g1b)
    # d(normed xmm) from the three projections of its row groups lands in one bf16 buffer
    dxmm_b = _e((B * nF, D), BF16, dev)
    # --- pair attention branch (rows [0, nmm)) ---
    do2 = lin_bwd(at.proj, g1b, t['o2'], B * nmm, dy_rowmap=rm2)                                  # [B*nmm, D]
    dq2, dKp, dVp = _e((B * nmm, Da), BF16, dev), _e((B * P, Da), BF16, dev), _e((B * P, D), BF16, dev)
    attention_bwd((t['q2'], 0), (t['Kp'], 0), (t['Vp'], 0), t['o2'], do2, t['lse2'], (dq2, 0), (dKp, 0), (dVp, 0),
                  B, heads, nmm, P, Da // heads, hd, hd ** -0.5, nmm * Da, Da, P * Da, Da, P * D, D,
                  nmm * Da, Da, P * Da, Da, P * D, D)
    dkv_p, dka_p = _e((B * nv, Da), BF16, dev), _e((B * na, Da), BF16, dev)
    dvv_p, dva_p = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    with region():
        lin_bwd(at.q, dq2, t['xmm_b'], B * nmm, a_rowmap=rm2, dx=dxmm_b, dx_rowmap=rm2)
        ops.pair_reduce(dKp, B, nv, na, Da, dkv_p, dka_p)
        ops.pair_reduce(dVp, B, nv, na, D, dvv_p, dva_p)
    # d(xv_out) = g1[rows v] + dkv_p Wk[:, :D] + dvv_p Wv[:, :D]   (fp32 accumulate, bf16 twin on the last GEMM):
    # the first GEMM of each pair brings g1's rows in as its fp32 residual, the second accumulates
    dxvo, dxao = _e((B * nv, D), F32, dev), _e((B * na, D), F32, dev)
    dxvo_b, dxao_b = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    g1r = g1.view(B * nF, D)
    with region():
        lin_bwd(at.k, dkv_p, t['xvo_b'], B * nv, k=D, dx=dxvo, dx_res=g1r, dx_res_rowmap=rmv, final=False)
        lin_bwd(at.k, dka_p, t['xao_b'], B * na, k=D, w_col_off=D, use_bias=False, dx=dxao, dx_res=g1r, dx_res_rowmap=rma)
    with region():
        lin_bwd(at.v, dvv_p, t['xvo_b'], B * nv, k=D, dx=dxvo, dx_beta=1, dx_C2=dxvo_b, dx_c2_mode=3, final=False)
        lin_bwd(at.v, dva_p, t['xao_b'], B * na, k=D, w_col_off=D, use_bias=False, dx=dxao, dx_beta=1, dx_C2=dxao_b, dx_c2_mode=3)
    # --- the two aggregation cross-attentions ---
    with region():
        dov = lin_bwd(at.attn_v.proj, dxvo_b, cv['o'], B * nv)
        doa = lin_bwd(at.attn_a.proj, dxao_b, ca['o'], B * na)
    return _factorized_bwd_cross(fb, t, g1, dov, doa, dxmm_b, dx_i, dx_a)


def _factorized_bwd_cross(fb, t, g1, dov, doa, dxmm_b, dx_i, dx_a):
    """The backward of the two aggregation cross-attentions, their q / kv projections and the three input LayerNorms ."""
    x_f, x_i, x_a, heads = t['x_f'], t['x_i'], t['x_a'], t['heads']
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    nmm, nv, na = t['tkns']
    dev, at = x_f.device, fb.attn
    hd = D // heads
    rmv, rma = (nv, nF, nmm), (na, nF, nmm + nv)
    cv, ca = t['cv'], t['ca']
    dq_v, dkv_v = _e((B * nv, D), BF16, dev), _e((B * nI, 2 * D), BF16, dev)
    dq_a, dkv_a = _e((B * na, D), BF16, dev), _e((B * nA, 2 * D), BF16, dev)
    del_v, del_a = torch.empty_like(cv['lse']), torch.empty_like(ca['lse'])
    for part in (1, 2):      # dQ kernels of both aggregations, then their dK/dV kernels: two steps, like a tower block's attention
        with region():
            attention_bwd((cv['q'], 0), (cv['kv'], 0), (cv['kv'], D), cv['o'], dov, cv['lse'], (dq_v, 0), (dkv_v, 0), (dkv_v, D),
                          B, heads, nv, nI, hd, hd, hd ** -0.5, nv * D, D, nI * 2 * D, 2 * D, nI * 2 * D, 2 * D,
                          nv * D, D, nI * 2 * D, 2 * D, nI * 2 * D, 2 * D, part=part, Delta=del_v)
            attention_bwd((ca['q'], 0), (ca['kv'], 0), (ca['kv'], D), ca['o'], doa, ca['lse'], (dq_a, 0), (dkv_a, 0), (dkv_a, D),
                          B, heads, na, nA, hd, hd, hd ** -0.5, na * D, D, nA * 2 * D, 2 * D, nA * 2 * D, 2 * D,
                          na * D, D, nA * 2 * D, 2 * D, nA * 2 * D, 2 * D, part=part, Delta=del_a)
    fused = t.get('ln_fused', False)
    with region():
        lin_bwd(at.attn_v.q, dq_v, t['xmm_b'], B * nv, a_rowmap=rmv, dx=dxmm_b, dx_rowmap=rmv)
        lin_bwd(at.attn_a.q, dq_a, t['xmm_b'], B * na, a_rowmap=rma, dx=dxmm_b, dx_rowmap=rma)
        dxv_b = lin_bwd(at.attn_v.kv, dkv_v, t['xv_b'], B * nI, wgrad=not fused)
        dxa_b = lin_bwd(at.attn_a.kv, dkv_a, t['xa_b'], B * nA, wgrad=not fused)
    # --- the three input LayerNorms ---
    acc_i, acc_a = (1 if dx_i is not None else 0), (1 if dx_a is not None else 0)
    if dx_i is None:
        dx_i = _e((B, nI, D), F32, dev)
    if dx_a is None:
        dx_a = _e((B, nA, D), F32, dev)
    dx_f = _e((B, nF, D), F32, dev)
    with region():
        if fused:      # (folded on the forward path: the backward re-makes the kv projections' weight-gradient operands)
            hv, ha = _e((B * nI, D), BF16, dev), _e((B * nA, D), BF16, dev)
            ln_bwd_tw(fb.norm1_img, fb.norm1_img.eps, None, 0, t['ti'], nI, B, dy_bf16=dxv_b, h_out=hv, dx1=dx_i, acc1=acc_i)
            ln_bwd_tw(fb.norm1_aud, fb.norm1_aud.eps, None, 0, t['ta'], nA, B, dy_bf16=dxa_b, h_out=ha, dx1=dx_a, acc1=acc_a)
        else:
            ln_bwd(fb.norm1_img, None, x_i, B, t['st_v'], dy_bf16=dxv_b, dx1=dx_i, acc1=acc_i)
            ln_bwd(fb.norm1_aud, None, x_a, B, t['st_a'], dy_bf16=dxa_b, dx1=dx_a, acc1=acc_a)
        ln_bwd(fb.norm1_mm, None, x_f, B, t['st_mm'], dy_bf16=dxmm_b, dy_f32=g1, dx1=dx_f)
    if fused:
        lin_wgrad(at.attn_v.kv, dkv_v, hv, B * nI)
        lin_wgrad(at.attn_a.kv, dkv_a, ha, B * nA)
    return dx_f, dx_i, dx_a


# ---- sequential form (DropPath active: per-sample scaled branches need torch ops between the kernels) --------------------
def _cross_fwd_seq(ca, xq_b, q_rowmap, nq, xkv_b, nk, B, D, heads, dev, keep=None):
    """CrossAttention (models/fusion_blocks.py:46-59) up to (not incl.) proj. xq_b rows come from the
    normed fusion tokens through q_rowmap; xkv_b is the normed modality [B*nk, D]."""
    hd = D // heads
    q = lin_fwd(ca.q, xq_b, B * nq, a_rowmap=q_rowmap, out_bf16=True)                        # [B*nq, D]
    kv = lin_fwd(ca.kv, xkv_b, B * nk, out_bf16=True)                                        # [B*nk, 2D]
    o, lse = attention_fwd((q, 0), (kv, 0), (kv, D), B, heads, nq, nk, hd, hd, hd ** -0.5,
                           nq * D, D, nk * 2 * D, 2 * D, nk * 2 * D, 2 * D, dev, keep=keep)
    return dict(q=q, kv=kv, o=o, lse=lse, keep=keep)


def _cross_bwd_seq(ca, c, do, xq_b, q_rowmap, nq, xkv_b, nk, B, D, heads, dxq_out, dxq_rowmap):
    """do: bf16 [B*nq, D] grad wrt the attention output (pre-proj). Returns d(xkv normed) bf16 [B*nk, D];
    writes d(xq normed) into rows of dxq_out through dxq_rowmap."""
    hd, dev = D // heads, do.device
    dq = _e((B * nq, D), BF16, dev)
    dkv = _e((B * nk, 2 * D), BF16, dev)
    attention_bwd((c['q'], 0), (c['kv'], 0), (c['kv'], D), c['o'], do, c['lse'], (dq, 0), (dkv, 0), (dkv, D),
                  B, heads, nq, nk, hd, hd, hd ** -0.5, nq * D, D, nk * 2 * D, 2 * D, nk * 2 * D, 2 * D,
                  nq * D, D, nk * 2 * D, 2 * D, nk * 2 * D, 2 * D, keep=c.get('keep'))
    lin_bwd(ca.q, dq, xq_b, B * nq, a_rowmap=q_rowmap, dx=dxq_out, dx_rowmap=dxq_rowmap)
    return lin_bwd(ca.kv, dkv, xkv_b, B * nk)


def _factorized_fwd_seq(fb, x_f, x_i, x_a, heads, tkns, dp=None, dr=None):
    """``dr``: dropout masks {'attn_v.attn' / 'attn_a.attn' / 'attn': draw_attn_keep, 'attn_v.proj' [B*nv, D] / 'attn_a.proj' [B*na, D] /
    'proj' [B*nmm, D] / 'fc1' / 'fc2': draw_keep} (models/fusion_blocks.py:54,58 in both aggregations, :256,260, Mlp.drop1 / drop2)."""
    dr = dr or {}
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    nmm, nv, na = tkns
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    # norm-then-residual: the residual base is the NORMED xmm (models/fusion_blocks.py:281-283)
    xmm_b, xmm32, st_mm = ln_fwd(fb.norm1_mm, None, x_f, B, want_f32=True)
    xv_b, _, st_v = ln_fwd(fb.norm1_img, None, x_i, B)
    xa_b, _, st_a = ln_fwd(fb.norm1_aud, None, x_a, B)
    rm2, rmv, rma = (nmm, nF, 0), (nv, nF, nmm), (na, nF, nmm + nv)
    cv = _cross_fwd_seq(at.attn_v, xmm_b, rmv, nv, xv_b, nI, B, D, heads, dev, keep=dr.get('attn_v.attn'))
    ca = _cross_fwd_seq(at.attn_a, xmm_b, rma, na, xa_b, nA, B, D, heads, dev, keep=dr.get('attn_a.attn'))
    xmm1 = _e((B, nF, D), F32, dev)
    # proj of the two aggregations: fp32 result lands in its rows of xmm1 (+ normed-xmm residual),
    # bf16 twin of the pre-residual value feeds the pair projections
    xvo_b, xao_b = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    # with DropPath / projection dropout the three projections land WITHOUT the residual; it is added below, masked and scaled
    pdrop = [dr.get('proj'), dr.get('attn_v.proj'), dr.get('attn_a.proj')]
    seq_res = dp is not None or any(m is not None for m in pdrop)
    r32 = None if seq_res else xmm32
    lin_fwd(at.attn_v.proj, cv['o'], B * nv, res=r32, res_rowmap=rmv, out=xmm1, c_rowmap=rmv, C2=xvo_b, c2_mode=2)
    lin_fwd(at.attn_a.proj, ca['o'], B * na, res=r32, res_rowmap=rma, out=xmm1, c_rowmap=rma, C2=xao_b, c2_mode=2)
    # the pairs are built from what the two CrossAttention modules RETURN, i.e. after their proj_drop (:58, :240-248)
    if pdrop[1] is not None:
        _drop(xvo_b, pdrop[1], B, nv, D)
    if pdrop[2] is not None:
        _drop(xao_b, pdrop[2], B, na, D)
    # all (v, a) pairs: Linear(cat(xv_i, xa_j)) = W[:, :D] xv_i + W[:, D:] xa_j + b  (never materialised)
    kv_p = lin_fwd(at.k, xvo_b, B * nv, k=D)
    ka_p = lin_fwd(at.k, xao_b, B * na, k=D, w_col_off=D, use_bias=False)
    vv_p = lin_fwd(at.v, xvo_b, B * nv, k=D)
    va_p = lin_fwd(at.v, xao_b, B * na, k=D, w_col_off=D, use_bias=False)
    P = nv * na
    Kp, Vp = _e((B * P, Da), BF16, dev), _e((B * P, D), BF16, dev)
    ops.pair_expand(kv_p, ka_p, B, nv, na, Da, Kp)
    ops.pair_expand(vv_p, va_p, B, nv, na, D, Vp)
    q2 = lin_fwd(at.q, xmm_b, B * nmm, a_rowmap=rm2, out_bf16=True)                          # [B*nmm, Da]
    scale = (D // heads) ** -0.5                                                             # NOT (Da/heads)^-0.5 (:220-222)
    o2, lse2 = attention_fwd((q2, 0), (Kp, 0), (Vp, 0), B, heads, nmm, P, Da // heads, D // heads, scale,
                             nmm * Da, Da, P * Da, Da, P * D, D, dev, keep=dr.get('attn'))
    lin_fwd(at.proj, o2, B * nmm, res=r32, res_rowmap=rm2, out=xmm1, c_rowmap=rm2)
    keep_rows = None
    if any(m is not None for m in pdrop):
        # one [B * nF, D] mask over the rows of cat((xmm2, xmm_v, xmm_a)) (:262), a site without dropout all ones
        ks = next(m[1] for m in pdrop if m is not None)
        parts = [(m[0].view(B, r, D) if m is not None else torch.ones((B, r, D), device=dev, dtype=torch.uint8))
                 for m, r in zip(pdrop, (nmm, nv, na))]
        keep_rows = (torch.cat(parts, dim=1).view(B * nF, D).contiguous(), ks)
        ops.dropout_rows(xmm1, keep_rows[0], ks, B, nF, D, xmm1, res=xmm32, rowscale=None if dp is None else dp[0])
    elif dp is not None:
        ops.rows_axpy(xmm32, xmm1, dp[0], B, nF, D, xmm1)               # xmm + s[b] * attn(xmm, xv, xa)
    h2, _, st2 = ln_fwd(fb.norm2, None, xmm1, B)
    Hd = fb.mlp.fc1.weight.shape[0]
    z = _e((B * nF, Hd), BF16, dev)
    u = lin_fwd(fb.mlp.fc1, h2, B * nF, act=1, out_bf16=True, C2=z, c2_mode=4)
    if 'fc1' in dr:
        _drop(u, dr['fc1'], B, nF, Hd)
    out = _res_add(fb.mlp.fc2, u, B * nF, xmm1, B, nF, D, None if dp is None else dp[1], drop=dr.get('fc2')).view(B, nF, D)
    tape = dict(dp=dp, dr=dr or None, keep_rows=keep_rows, x_f=x_f, x_i=x_i, x_a=x_a, xmm_b=xmm_b, st_mm=st_mm, xv_b=xv_b, st_v=st_v, xa_b=xa_b, st_a=st_a, cv=cv, ca=ca,
                xvo_b=xvo_b, xao_b=xao_b, Kp=Kp, Vp=Vp, q2=q2, o2=o2, lse2=lse2, xmm1=xmm1, h2=h2, st2=st2, z=z, u=u,
                heads=heads, tkns=tkns)
    return out, tape


def _factorized_bwd_seq(fb, t, g, gb, *, dx_i=None, dx_a=None):
    """g fp32 [B,nF,D] grad of the block output.  Returns (dx_f fp32 + bf16 twin, dx_i, dx_a);
    dx_i / dx_a are freshly STORED (fp32 [B,n,D]) unless buffers are passed."""
    x_f, x_i, x_a, heads = t['x_f'], t['x_i'], t['x_a'], t['heads']
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    nmm, nv, na = t['tkns']
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    P = nv * na
    rm2, rmv, rma = (nmm, nF, 0), (nv, nF, nmm), (na, nF, nmm + nv)
    dp = t.get('dp')
    dr = t.get('dr') or {}
    if dp is not None or 'fc2' in dr:
        gb = _branch_grad(g, None if dp is None else dp[1], dr.get('fc2'), B, nF, D)
    elif gb is None:
        gb = to_bf16(g)
    dz = lin_bwd(fb.mlp.fc2, gb, t['u'], B * nF, gelu_aux=t['z'])
    if 'fc1' in dr:
        _drop(dz, dr['fc1'], B, nF, dz.shape[1])
    dh2 = lin_bwd(fb.mlp.fc1, dz, t['h2'], B * nF)
    g1 = _e((B, nF, D), F32, dev)
    g1b = _e((B * nF, D), BF16, dev)
    ln_bwd(fb.norm2, None, t['xmm1'], B, t['st2'], dy_bf16=dh2, dx1=g1, res1=g, dx1_bf16=g1b)
    # g1 = gradient at xmm1: unchanged it is the residual-path gradient of the normed xmm (norm1_mm backward below);
    # the attention branch sees it scaled per sample when DropPath is on, and masked where its projections were dropped:
    #   gy   gradient of the branch output cat((xmm2, xmm_v, xmm_a)) AFTER the dropouts (what the pairs' gradients add to),
    #   g1b  the same through the masks = gradient of the three projections' outputs (only proj's rows are read from it)
    gy = g1
    keep_rows = t.get('keep_rows')
    if dp is not None:
        gy = _e((B, nF, D), F32, dev)
        ops.dropout_rows(g1, None, 1.0, B, nF, D, gy, rowscale=dp[0])
    if dp is not None or keep_rows is not None:
        g1b = _branch_grad(g1, None if dp is None else dp[0], keep_rows, B, nF, D)
    # d(normed xmm) from the three projections of its row groups lands in one bf16 buffer
    dxmm_b = _e((B * nF, D), BF16, dev)
    # --- pair attention branch (rows [0, nmm)) ---
    do2 = lin_bwd(at.proj, g1b, t['o2'], B * nmm, dy_rowmap=rm2)                              # [B*nmm, D]
    dq2, dKp, dVp = _e((B * nmm, Da), BF16, dev), _e((B * P, Da), BF16, dev), _e((B * P, D), BF16, dev)
    attention_bwd((t['q2'], 0), (t['Kp'], 0), (t['Vp'], 0), t['o2'], do2, t['lse2'], (dq2, 0), (dKp, 0), (dVp, 0),
                  B, heads, nmm, P, Da // heads, D // heads, (D // heads) ** -0.5, nmm * Da, Da, P * Da, Da, P * D, D,
                  nmm * Da, Da, P * Da, Da, P * D, D, keep=dr.get('attn'))
    lin_bwd(at.q, dq2, t['xmm_b'], B * nmm, a_rowmap=rm2, dx=dxmm_b, dx_rowmap=rm2)
    dkv_p, dka_p = _e((B * nv, Da), BF16, dev), _e((B * na, Da), BF16, dev)
    dvv_p, dva_p = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    ops.pair_reduce(dKp, B, nv, na, Da, dkv_p, dka_p)
    ops.pair_reduce(dVp, B, nv, na, D, dvv_p, dva_p)
    # d(xv_out) = g1[rows v] + dkv_p Wk[:, :D] + dvv_p Wv[:, :D]   (fp32 accumulate, bf16 twin on the last GEMM)
    # copies (clone, not .contiguous(): at B == 1 the slices are already contiguous VIEWS of g1, and the GEMMs below
    # accumulate into them while g1 is still needed by the norm1_mm backward)
    cf = torch.contiguous_format
    dxvo = gy.view(B, nF, D)[:, nmm:nmm + nv].clone(memory_format=cf).view(B * nv, D)
    dxao = gy.view(B, nF, D)[:, nmm + nv:].clone(memory_format=cf).view(B * na, D)
    dxvo_b, dxao_b = _e((B * nv, D), BF16, dev), _e((B * na, D), BF16, dev)
    lin_bwd(at.k, dkv_p, t['xvo_b'], B * nv, k=D, dx=dxvo, dx_beta=1, final=False)
    lin_bwd(at.v, dvv_p, t['xvo_b'], B * nv, k=D, dx=dxvo, dx_beta=1, dx_C2=dxvo_b, dx_c2_mode=3, final=False)
    lin_bwd(at.k, dka_p, t['xao_b'], B * na, k=D, w_col_off=D, use_bias=False, dx=dxao, dx_beta=1)
    lin_bwd(at.v, dva_p, t['xao_b'], B * na, k=D, w_col_off=D, use_bias=False, dx=dxao, dx_beta=1, dx_C2=dxao_b, dx_c2_mode=3)
    # --- the two aggregation cross-attentions ---
    if 'attn_v.proj' in dr:          # through CrossAttention.proj_drop (:58): everything that consumed its output is summed by now
        _drop(dxvo_b, dr['attn_v.proj'], B, nv, D)
    if 'attn_a.proj' in dr:
        _drop(dxao_b, dr['attn_a.proj'], B, na, D)
    dov = lin_bwd(at.attn_v.proj, dxvo_b, t['cv']['o'], B * nv)
    doa = lin_bwd(at.attn_a.proj, dxao_b, t['ca']['o'], B * na)
    dxv_b = _cross_bwd_seq(at.attn_v, t['cv'], dov, t['xmm_b'], rmv, nv, t['xv_b'], nI, B, D, heads, dxmm_b, rmv)
    dxa_b = _cross_bwd_seq(at.attn_a, t['ca'], doa, t['xmm_b'], rma, na, t['xa_b'], nA, B, D, heads, dxmm_b, rma)
    # --- the three input LayerNorms ---
    acc_i, acc_a = (1 if dx_i is not None else 0), (1 if dx_a is not None else 0)
    if dx_i is None:
        dx_i = _e((B, nI, D), F32, dev)
    if dx_a is None:
        dx_a = _e((B, nA, D), F32, dev)
    ln_bwd(fb.norm1_img, None, x_i, B, t['st_v'], dy_bf16=dxv_b, dx1=dx_i, acc1=acc_i)
    ln_bwd(fb.norm1_aud, None, x_a, B, t['st_a'], dy_bf16=dxa_b, dx1=dx_a, acc1=acc_a)
    dx_f = _e((B, nF, D), F32, dev)
    ln_bwd(fb.norm1_mm, None, x_f, B, t['st_mm'], dy_bf16=dxmm_b, dy_f32=g1, dx1=dx_f)
    return dx_f, dx_i, dx_a


# ------------------------------------------------------------------------------------------------
# the two alternative fusion blocks behind fusion_arch (models/fusion_blocks.py:89-213).  Arguments are POSITIONAL as
# the encoder passes them — (x_fusion, x_image, x_audio), models/deepavfusion.py:106 — and the reference's swapped
# parameter names are reproduced:
#   token     : norm1_img normalises the 3rd argument (audio), norm1_aud the 2nd (image); keys = [audio rows | image rows]
#   dense_mmi : norms in order; inside the attention pairs are (audio_i, image_j), p = i*nI + j, features [audio || image]
# Both share the norm-then-residual form and the norm2 + MLP tail of the factorised block.
# ------------------------------------------------------------------------------------------------
def _alt_tail_fwd(fb, xmm1, B, nF, D, dev, dp=None, dr=None):
    dr = dr or {}
    h2, _, st2 = ln_fwd(fb.norm2, None, xmm1, B)
    Hd = fb.mlp.fc1.weight.shape[0]
    z = _e((B * nF, Hd), BF16, dev)
    u = lin_fwd(fb.mlp.fc1, h2, B * nF, act=1, out_bf16=True, C2=z, c2_mode=4)
    if 'fc1' in dr:
        _drop(u, dr['fc1'], B, nF, Hd)
    out = _res_add(fb.mlp.fc2, u, B * nF, xmm1, B, nF, D, None if dp is None else dp[1], drop=dr.get('fc2')).view(B, nF, D)
    return out, dict(h2=h2, st2=st2, z=z, u=u, xmm1=xmm1, dp=dp, dr=dr or None)


def _alt_tail_bwd(fb, tt, g, gb, B, nF, D, dev):
    """-> (g1 fp32 [B,nF,D], g1b bf16): gradient at xmm1 (the attention residual output); with DropPath g1b is the
    gradient of the attention BRANCH (scaled per sample), g1 stays the residual-path gradient."""
    dp = tt.get('dp')
    dr = tt.get('dr') or {}
    if dp is not None or 'fc2' in dr:
        gb = _branch_grad(g, None if dp is None else dp[1], dr.get('fc2'), B, nF, D)
    elif gb is None:
        gb = to_bf16(g)
    dz = lin_bwd(fb.mlp.fc2, gb, tt['u'], B * nF, gelu_aux=tt['z'])
    if 'fc1' in dr:
        _drop(dz, dr['fc1'], B, nF, dz.shape[1])
    dh2 = lin_bwd(fb.mlp.fc1, dz, tt['h2'], B * nF)
    g1 = _e((B, nF, D), F32, dev)
    g1b = _e((B * nF, D), BF16, dev)
    ln_bwd(fb.norm2, None, tt['xmm1'], B, tt['st2'], dy_bf16=dh2, dx1=g1, res1=g, dx1_bf16=g1b)
    if dp is not None or 'proj' in dr:
        g1b = _branch_grad(g1, None if dp is None else dp[0], dr.get('proj'), B, nF, D)
    return g1, g1b


def _token_fwd(fb, x_f, x_2, x_3, heads, dp=None, dr=None):
    dr = dr or {}
    B, nF, D = x_f.shape
    n2, n3 = x_2.shape[1], x_3.shape[1]
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    hd = Da // heads
    xmm_b, xmm32, st_mm = ln_fwd(fb.norm1_mm, None, x_f, B, want_f32=True)
    x3_b, _, st_3 = ln_fwd(fb.norm1_img, None, x_3, B)          # "xv" = 3rd argument (:135-136)
    x2_b, _, st_2 = ln_fwd(fb.norm1_aud, None, x_2, B)          # "xa" = 2nd argument
    nS = n3 + n2                                                 # cat(xv, xa) (:106): 3rd-argument rows first
    kv = _e((B * nS, 2 * Da), BF16, dev)
    lin_fwd(at.kv, x3_b, B * n3, out=kv, c_rowmap=(n3, nS, 0))
    lin_fwd(at.kv, x2_b, B * n2, out=kv, c_rowmap=(n2, nS, n3))
    q = lin_fwd(at.q, xmm_b, B * nF, out_bf16=True)
    o, lse = attention_fwd((q, 0), (kv, 0), (kv, Da), B, heads, nF, nS, hd, hd, hd ** -0.5,
                           nF * Da, Da, nS * 2 * Da, 2 * Da, nS * 2 * Da, 2 * Da, dev, keep=dr.get('attn'))
    xmm1 = _res_add(at.proj, o, B * nF, xmm32, B, nF, D, None if dp is None else dp[0], drop=dr.get('proj')).view(B, nF, D)
    out, tt = _alt_tail_fwd(fb, xmm1, B, nF, D, dev, dp, dr)
    tt.update(arch='token', x_f=x_f, x_2=x_2, x_3=x_3, xmm_b=xmm_b, st_mm=st_mm, x2_b=x2_b, st_2=st_2, x3_b=x3_b, st_3=st_3,
              kv=kv, q=q, o=o, lse=lse, heads=heads)
    return out, tt


def _token_bwd(fb, t, g, gb, *, dx_i=None, dx_a=None):
    x_f, x_2, x_3, heads = t['x_f'], t['x_2'], t['x_3'], t['heads']
    B, nF, D = x_f.shape
    n2, n3 = x_2.shape[1], x_3.shape[1]
    nS = n3 + n2
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    hd = Da // heads
    g1, g1b = _alt_tail_bwd(fb, t, g, gb, B, nF, D, dev)
    do = lin_bwd(at.proj, g1b, t['o'], B * nF)
    dq, dkv = _e((B * nF, Da), BF16, dev), _e((B * nS, 2 * Da), BF16, dev)
    attention_bwd((t['q'], 0), (t['kv'], 0), (t['kv'], Da), t['o'], do, t['lse'], (dq, 0), (dkv, 0), (dkv, Da),
                  B, heads, nF, nS, hd, hd, hd ** -0.5, nF * Da, Da, nS * 2 * Da, 2 * Da, nS * 2 * Da, 2 * Da,
                  nF * Da, Da, nS * 2 * Da, 2 * Da, nS * 2 * Da, 2 * Da, keep=(t.get('dr') or {}).get('attn'))
    dxmm_b = lin_bwd(at.q, dq, t['xmm_b'], B * nF)
    dx3_b = lin_bwd(at.kv, dkv, t['x3_b'], B * n3, dy_rowmap=(n3, nS, 0), final=False)
    dx2_b = lin_bwd(at.kv, dkv, t['x2_b'], B * n2, dy_rowmap=(n2, nS, n3))
    acc_2, acc_3 = (1 if dx_i is not None else 0), (1 if dx_a is not None else 0)
    if dx_i is None:
        dx_i = _e((B, n2, D), F32, dev)
    if dx_a is None:
        dx_a = _e((B, n3, D), F32, dev)
    ln_bwd(fb.norm1_aud, None, x_2, B, t['st_2'], dy_bf16=dx2_b, dx1=dx_i, acc1=acc_2)
    ln_bwd(fb.norm1_img, None, x_3, B, t['st_3'], dy_bf16=dx3_b, dx1=dx_a, acc1=acc_3)
    dx_f = _e((B, nF, D), F32, dev)
    ln_bwd(fb.norm1_mm, None, x_f, B, t['st_mm'], dy_bf16=dxmm_b, dy_f32=g1, dx1=dx_f)
    return dx_f, dx_i, dx_a


def _dense_fwd(fb, x_f, x_i, x_a, heads, dp=None, dr=None):
    dr = dr or {}
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    hd = Da // heads
    xmm_b, xmm32, st_mm = ln_fwd(fb.norm1_mm, None, x_f, B, want_f32=True)
    xi_b, _, st_i = ln_fwd(fb.norm1_img, None, x_i, B)
    xa_b, _, st_a = ln_fwd(fb.norm1_aud, None, x_a, B)
    # Linear(cat(audio_i, image_j)) = W[:, :D] audio_i + W[:, D:] image_j + b: two small GEMMs + a broadcast add
    # instead of the [B, nA*nI, 2D] pair tensor of models/fusion_blocks.py:171-174
    pa = lin_fwd(at.kv, xa_b, B * nA, k=D)                                   # fp32 [B*nA, 2Da]
    pi = lin_fwd(at.kv, xi_b, B * nI, k=D, w_col_off=D, use_bias=False)
    P = nA * nI
    KV = _e((B * P, 2 * Da), BF16, dev)
    ops.pair_expand(pa, pi, B, nA, nI, 2 * Da, KV)
    q = lin_fwd(at.q, xmm_b, B * nF, out_bf16=True)
    scale = (D // heads) ** -0.5                                  # from the FULL dim (:157-158)
    o, lse = attention_fwd((q, 0), (KV, 0), (KV, Da), B, heads, nF, P, hd, hd, scale,
                           nF * Da, Da, P * 2 * Da, 2 * Da, P * 2 * Da, 2 * Da, dev, keep=dr.get('attn'))
    xmm1 = _res_add(at.proj, o, B * nF, xmm32, B, nF, D, None if dp is None else dp[0], drop=dr.get('proj')).view(B, nF, D)
    out, tt = _alt_tail_fwd(fb, xmm1, B, nF, D, dev, dp, dr)
    tt.update(arch='dense_mmi', x_f=x_f, x_i=x_i, x_a=x_a, xmm_b=xmm_b, st_mm=st_mm, xi_b=xi_b, st_i=st_i, xa_b=xa_b, st_a=st_a,
              KV=KV, q=q, o=o, lse=lse, heads=heads)
    return out, tt


def _dense_bwd(fb, t, g, gb, *, dx_i=None, dx_a=None):
    x_f, x_i, x_a, heads = t['x_f'], t['x_i'], t['x_a'], t['heads']
    B, nF, D = x_f.shape
    nI, nA = x_i.shape[1], x_a.shape[1]
    P = nA * nI
    dev, at = x_f.device, fb.attn
    Da = at.q.weight.shape[0]
    hd = Da // heads
    g1, g1b = _alt_tail_bwd(fb, t, g, gb, B, nF, D, dev)
    do = lin_bwd(at.proj, g1b, t['o'], B * nF)
    dq, dKV = _e((B * nF, Da), BF16, dev), _e((B * P, 2 * Da), BF16, dev)
    scale = (D // heads) ** -0.5
    attention_bwd((t['q'], 0), (t['KV'], 0), (t['KV'], Da), t['o'], do, t['lse'], (dq, 0), (dKV, 0), (dKV, Da),
                  B, heads, nF, P, hd, hd, scale, nF * Da, Da, P * 2 * Da, 2 * Da, P * 2 * Da, 2 * Da,
                  nF * Da, Da, P * 2 * Da, 2 * Da, P * 2 * Da, 2 * Da, keep=(t.get('dr') or {}).get('attn'))
    dxmm_b = lin_bwd(at.q, dq, t['xmm_b'], B * nF)
    dpa, dpi = _e((B * nA, 2 * Da), BF16, dev), _e((B * nI, 2 * Da), BF16, dev)
    ops.pair_reduce(dKV, B, nA, nI, 2 * Da, dpa, dpi)
    dxa_b = lin_bwd(at.kv, dpa, t['xa_b'], B * nA, k=D, final=False)
    dxi_b = lin_bwd(at.kv, dpi, t['xi_b'], B * nI, k=D, w_col_off=D, use_bias=False)
    acc_i, acc_a = (1 if dx_i is not None else 0), (1 if dx_a is not None else 0)
    if dx_i is None:
        dx_i = _e((B, nI, D), F32, dev)
    if dx_a is None:
        dx_a = _e((B, nA, D), F32, dev)
    ln_bwd(fb.norm1_img, None, x_i, B, t['st_i'], dy_bf16=dxi_b, dx1=dx_i, acc1=acc_i)
    ln_bwd(fb.norm1_aud, None, x_a, B, t['st_a'], dy_bf16=dxa_b, dx1=dx_a, acc1=acc_a)
    dx_f = _e((B, nF, D), F32, dev)
    ln_bwd(fb.norm1_mm, None, x_f, B, t['st_mm'], dy_bf16=dxmm_b, dy_f32=g1, dx1=dx_f)
    return dx_f, dx_i, dx_a


# steps a tower block idles when it runs as a lane beside the factorised fusion block (see _factorized_fwd / _bwd):
# forward: k/v pair projections, pair_expand, pair attention, its proj (steps 5-8) sit between the towers' proj and norm2;
# backward: pair attention dQ, dK/dV, [q2 dgrad + pair_reduce], k dgrads, v dgrads, aggregation-proj dgrads (steps 5-10)
FUSION_IDLE_FWD, FUSION_IDLE_BWD = 4, 6


def fusion_block_batchable(fb, dp=None, dr=None):
    """True when the block's forward / backward consist of library launches only, i.e. may run as a lane of a launch batch
    (the factorised block without DropPath / dropout; the token / dense blocks, DropPath and dropout use torch ops between kernels)."""
    return dp is None and not dr and getattr(fb, 'arch', 'factorized_mmi') == 'factorized_mmi'


def fusion_block_fwd(fb, x_f, x_i, x_a, heads, tkns, dp=None, dr=None):
    """Dispatch on the block's architecture (models/deepavfusion.py:28-35); x_i / x_a = the 2nd / 3rd positional
    argument of the reference call ``blk_fusion(x_fusion, x_image, x_audio)``."""
    arch = getattr(fb, 'arch', 'factorized_mmi')
    if arch == 'token':
        return _token_fwd(fb, x_f, x_i, x_a, heads, dp, dr)
    if arch == 'dense_mmi':
        return _dense_fwd(fb, x_f, x_i, x_a, heads, dp, dr)
    return _factorized_fwd(fb, x_f, x_i, x_a, heads, tkns, dp, dr)


def fusion_block_bwd(fb, t, g, gb, *, dx_i=None, dx_a=None):
    """g fp32 [B,nF,D] grad of the block output -> (dx_f, dx_i, dx_a), gradients of the three positional inputs."""
    arch = t.get('arch', 'factorized_mmi')
    if arch == 'token':
        return _token_bwd(fb, t, g, gb, dx_i=dx_i, dx_a=dx_a)
    if arch == 'dense_mmi':
        return _dense_bwd(fb, t, g, gb, dx_i=dx_i, dx_a=dx_a)
    return _factorized_bwd(fb, t, g, gb, dx_i=dx_i, dx_a=dx_a)


# ------------------------------------------------------------------------------------------------
# patch embedding of the kept patches (timm PatchEmbed + pos_embed + gather; models/vits.py:91-100)
# ------------------------------------------------------------------------------------------------
def patch_embed_fwd(vit, img, ids_keep32):
    """Images [B,C,H,W] (timm PatchEmbed + models/vits.py:93-100: pos-embed added BEFORE the gather) and clips
    [B,C,T,H,W] (PatchEmbed3D + models/video_vits.py:229-232: gathered tokens + pos_embed in natural order, which
    only broadcasts when every patch is kept)."""
    B, C = img.shape[:2]
    pe = vit.patch_embed
    L = pe.num_patches
    nk = ids_keep32.shape[1] if ids_keep32 is not None else L
    D = vit.embed_dim
    clip = img.dim() == 5
    pt = pe.patch_size[0] if clip else 1
    if not clip:      # timm PatchEmbed.forward asserts the exact input size (the kernels would index a different grid)
        H, W = img.shape[2], img.shape[3]
        assert H == pe.img_size[0], f"Input image height ({H}) doesn't match model ({pe.img_size[0]})."
        assert W == pe.img_size[1], f"Input image width ({W}) doesn't match model ({pe.img_size[1]})."
    else:             # PatchEmbed3D has no check of its own: a clip of another size fails at `+ pos_embed` (models/video_vits.py:229-232)
        got = (img.shape[2] // pt) * (img.shape[3] // 16) * (img.shape[4] // 16)
        if tuple(img.shape[2:]) != tuple(pe.input_size):
            raise RuntimeError(f'The size of tensor a ({got}) must match the size of tensor b ({L}) at non-singleton dimension 1')
    if C != pe.proj.weight.shape[1]:
        raise RuntimeError(f'Given groups=1, weight of size {list(pe.proj.weight.shape)}, expected input{list(img.shape)} to have '
                           f'{pe.proj.weight.shape[1]} channels, but got {C} channels instead')
    if clip and nk != L:
        raise RuntimeError(f'The size of tensor a ({nk}) must match the size of tensor b ({L}) at non-singleton dimension 1')
    K = C * pt * 256
    A = _e((B * nk, K), BF16, img.device)
    ops.patch_gather(img, ids_keep32, nk, A, pt)
    Wb = wcache(pe.proj.weight)
    tok = _e((B, nk, D), F32, img.device)
    pos_by_id = ids_keep32 is not None and not clip
    if ln_fuse_ok(D) and K % 64 == 0:      # the tokens' twin + row statistics for the first block's folded norm1 (and the fusion block's norm1_*)
        tw = (_e((B * nk, D), BF16, img.device), _e((B * nk, D // 64, 2), F32, img.device))
        ops.gemm_nt_ln(A, Wb, B * nk, D, K, prod=dict(stats_out=tw[1], twin_out=tw[0], ld_twin=D), bias=pe.proj.bias, res=vit.pos_embed,
                       ldres=D, res_rows=ids_keep32 if pos_by_id else None, res_rowmap=None if pos_by_id else (L, 0, 0), C_out=tok)
        tw_set(tok, *tw)
    else:
        ops.gemm_nt(A, Wb, B * nk, D, K, bias=pe.proj.bias, res=vit.pos_embed, ldres=D,
                    res_rows=ids_keep32 if pos_by_id else None, res_rowmap=None if pos_by_id else (L, 0, 0), C_out=tok)
    return tok, dict(A=A, nk=nk)


def patch_embed_bwd(vit, t, g, gb):
    pe = vit.patch_embed
    D = vit.embed_dim
    A = t['A']
    M, K = A.shape
    if gb is None:
        gb = to_bf16(g)
    ops.gemm_tn(gb, A, M, D, K, gbuf(pe.proj.weight).view(D, K), beta=1, bias_grad=gbuf(pe.proj.bias))
    _ready(pe.proj.weight, pe.proj.bias)


# ------------------------------------------------------------------------------------------------
# MAE decoder (models/avmae.py:147-180, decoder_arch='plain')
# ------------------------------------------------------------------------------------------------
def drain(steps):
    """Run a step generator (decoder_fwd_steps / decoder_bwd_steps) to its end; returns its value."""
    try:
        while True:
            next(steps)
    except StopIteration as e:
        return e.value


def decoder_fwd(dec, x_b, xf_b, ids_restore32, B, nk, nF):
    """dec: namespace with embed, mask_token, pos_embed, blocks, norm, pred, heads.  x_b bf16 [B*nk, D]
    (normed encoder tokens), xf_b bf16 [B*nF, D].  Returns pred fp32 [B, L, P]."""
    return drain(decoder_fwd_steps(dec, x_b, xf_b, ids_restore32, B, nk, nF))


def decoder_fwd_steps(dec, x_b, xf_b, ids_restore32, B, nk, nF):
    """decoder_fwd as a generator that yields after every decoder block: the caller advances the two decoders alternately on their two
    streams and may cross-join the streams between blocks (autograd_bridge.paired_steps: why)."""
    L = ids_restore32.shape[1]
    Dd = dec.embed.weight.shape[0]
    dev = x_b.device
    emb = lin_fwd(dec.embed, x_b, B * nk)                                                    # fp32 [B*nk, Dd]
    x = _e((B, nF + L, Dd), F32, dev)
    lin_fwd(dec.embed, xf_b, B * nF, out=x, c_rowmap=(nF, nF + L, 0))                        # embed is shared (:158)
    ops.unshuffle_fwd(emb, dec.mask_token, dec.pos_embed, ids_restore32, B, L, nk, Dd, x, (nF + L) * Dd, nF)
    tapes = []
    swin = getattr(dec, 'arch', 'plain') == 'swin'
    for blk in dec.blocks:
        if swin:
            x, bt = swin_block_fwd(blk, x, nF)
        else:
            x, bt = block_fwd(blk, x, None, dec.heads, blk.norm1.eps)
        tapes.append(bt)
        yield
    # decoder_norm + pred on the patch rows only (x[:, nF:]) — the LN reads them through the batch stride
    D_ = Dd
    if not swin and ln_fuse_ok(Dd):       # decoder_norm folded into decoder_pred: the head contracts the twin's patch rows through a row map
        tw_last = ensure_tw(x)
        pred = lin_fwd_ln(dec.pred, dec.norm, dec.norm.eps, tw_last, B * L, a_rowmap=(L, nF + L, nF), out_bf16=False)
        P = pred.shape[1]
        tape = dict(x_b=x_b, xf_b=xf_b, tapes=tapes, tw_last=tw_last, ln_fused=True, ids_restore32=ids_restore32, nk=nk, nF=nF, L=L)
        return pred.view(B, L, P), tape
    hN = _e((B * L, D_), BF16, dev)
    mean, rstd = _e((B * L,), F32, dev), _e((B * L,), F32, dev)
    xs = x.view(-1)[nF * Dd:]
    ops.layernorm_fwd(xs, (nF + L) * Dd, L, None, 0, 0, B, Dd, dec.norm.weight, dec.norm.bias, dec.norm.eps, hN, None, mean, rstd)
    pred = lin_fwd(dec.pred, hN, B * L)
    P = pred.shape[1]
    tape = dict(x_b=x_b, xf_b=xf_b, x_last=x, tapes=tapes, hN=hN, stN=(mean, rstd), ids_restore32=ids_restore32, nk=nk, nF=nF, L=L)
    return pred.view(B, L, P), tape


def decoder_bwd(dec, t, dpred_b, ids_keep32, B):
    return drain(decoder_bwd_steps(dec, t, dpred_b, ids_keep32, B))


def decoder_bwd_steps(dec, t, dpred_b, ids_keep32, B):
    """decoder_bwd as a generator that yields after the head's backward and after every decoder block (see decoder_fwd_steps)."""
    nk, nF, L = t['nk'], t['nF'], t['L']
    Dd = dec.embed.weight.shape[0]
    dev = dpred_b.device
    fused = t.get('ln_fused', False)
    dhN = lin_bwd(dec.pred, dpred_b, None if fused else t['hN'], B * L, wgrad=not fused)
    g = _e((B, nF + L, Dd), F32, dev)                  # the LayerNorm backward writes the L patch rows; the fusion rows carry no
    gb = _e((B * (nF + L), Dd), BF16, dev)             # gradient from the head (models/avmae.py:173 drops them)
    g[:, :nF].zero_()
    gb.view(B, nF + L, Dd)[:, :nF].zero_()
    if fused:
        xb, st = t['tw_last']
        hN = _e((B * L, Dd), BF16, dev)
        ops.layernorm_bwd_twin((xb, nF * Dd), (nF + L) * Dd, (st, nF * (Dd // 64) * 2), L, None, 0, None, 0, B, Dd, dec.norm.eps,
                               dhN, None, dec.norm.weight, dec.norm.bias,
                               (g, nF * Dd), (nF + L) * Dd, 0, None, 0, (gb, nF * Dd), (nF + L) * Dd,
                               h_out=hN, dgamma=gbuf(dec.norm.weight), dbeta=gbuf(dec.norm.bias))
        lin_wgrad(dec.pred, dpred_b, hN, B * L)
    else:
        xs = t['x_last'].view(-1)[nF * Dd:]
        ops.layernorm_bwd(xs, (nF + L) * Dd, L, None, 0, 0, B, Dd, dhN, None, dec.norm.weight, t['stN'][0], t['stN'][1],
                          g.view(-1)[nF * Dd:], (nF + L) * Dd, 0, None, 0, gb.view(-1)[nF * Dd:], (nF + L) * Dd,
                          None, 0, 0, None, 0, None, 0, gbuf(dec.norm.weight), gbuf(dec.norm.bias))
    _ready(dec.norm.weight, dec.norm.bias)
    swin = getattr(dec, 'arch', 'plain') == 'swin'
    yield
    for blk, bt in zip(reversed(list(dec.blocks)), reversed(t['tapes'])):
        if swin:
            g, gb = swin_block_bwd(blk, bt, g, gb)
        else:
            g, gb, _ = block_bwd(blk, bt, g, gb)
        yield
    d_emb = _e((B * nk, Dd), BF16, dev)
    d_embf = _e((B * nF, Dd), BF16, dev)
    ops.rows_gather_cast(g, (nF + L) * Dd, nF, ids_keep32, B, nk, Dd, d_emb)
    ops.rows_gather_cast(g, (nF + L) * Dd, 0, None, B, nF, Dd, d_embf)
    ops.unshuffle_bwd_reduce(g, (nF + L) * Dd, nF, t['ids_restore32'], B, L, nk, Dd, gbuf(dec.pos_embed), gbuf(dec.mask_token))
    _ready(dec.pos_embed, dec.mask_token)
    dx_b = lin_bwd(dec.embed, d_emb, t['x_b'], B * nk, final=False)
    dxf_b = lin_bwd(dec.embed, d_embf, t['xf_b'], B * nF)
    return dx_b, dxf_b


# ------------------------------------------------------------------------------------------------
# Swin decoder block (models/swin.py:160-209 with x_fusion given, models/avmae.py:174-176)
# ------------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634


def swin_block_fwd(blk, x, nF):
    """x fp32 [B, nF + L, D] = [fusion rows | token rows] (the decoder's layout).  Every window attends over its A tokens
    and ALL nF fusion tokens with the relative-position bias (+ shift mask) on the A x A corner; the fusion rows of the
    result are the mean over the windows.  Returns (x_out fp32 [B, nF + L, D], tape)."""
    B, R, D = x.shape
    L, dev = R - nF, x.device
    heads, nW = blk.num_heads, blk.num_windows
    A = blk.attn.window_area
    N, hd, M, Mw = A + nF, D // heads, B * R, B * nW * (A + nF)
    h1, _, st1 = ln_fwd(blk.norm1, None, x, B)
    seq = _e((Mw, D), BF16, dev)
    ops.window_unfold(h1, blk.rows32, B, nW, A, nF, L, D, 1.0, seq)                        # roll + window_partition + cat (:172-185)
    qkv = lin_fwd(blk.attn.qkv, seq, Mw, out_bf16=True)                                    # [Mw, 3D]
    nb = nW if blk.attn_mask is not None else 1
    ld = (N + 31) // 32 * 32
    bias = _e((nb, heads, N, ld), F32, dev)                                                # rebuilt per step: the table is a parameter
    ops.relpos_bias_build(blk.attn.relative_position_bias_table, blk.index32, blk.attn_mask, nb, heads, A, N, ld,
                          1.0 if PRECISION == 'fp32' else LOG2E, bias)
    o = _e((Mw, D), BF16, dev)
    lse = _e((B * nW, heads, N), F32, dev)
    ops.hold(qkv)
    es = qkv.element_size()
    ops.attn_bias_fwd(qkv.data_ptr(), qkv.data_ptr() + es * D, qkv.data_ptr() + es * 2 * D, o, lse, B * nW, heads, N, N, hd, hd,
                      N * 3 * D, 3 * D, N * 3 * D, 3 * D, N * 3 * D, 3 * D, N * D, D, blk.attn.scale, bias, nb, ld)
    t = lin_fwd(blk.attn.proj, o, Mw)                                                      # fp32 [Mw, D]
    x1 = _e((B, R, D), F32, dev)
    ops.window_fold(t, blk.inv32, x, B, nW, A, nF, L, D, 1.0 / nW, x1)                     # window_reverse + roll back + mean (:191-201)
    h2, _, st2 = ln_fwd(blk.norm2, None, x1, B)
    Hd = blk.mlp.fc1.weight.shape[0]
    z = _e((M, Hd), BF16, dev)
    u = lin_fwd(blk.mlp.fc1, h2, M, act=1, out_bf16=True, C2=z, c2_mode=4)
    x2 = _res_add(blk.mlp.fc2, u, M, x1, B, R, D, None).view(B, R, D)
    tape = dict(x=x, nF=nF, h1=h1, st1=st1, seq=seq, qkv=qkv, bias=bias, nb=nb, ld=ld, o=o, lse=lse, x1=x1, h2=h2, st2=st2, z=z, u=u)
    return x2, tape


def swin_block_bwd(blk, t, g2, g2b=None):
    """g2 fp32 [B, nF + L, D] -> (dx fp32, dx bf16 twin)."""
    x, nF = t['x'], t['nF']
    B, R, D = x.shape
    L, dev = R - nF, x.device
    heads, nW = blk.num_heads, blk.num_windows
    A = blk.attn.window_area
    N, hd, M, Mw = A + nF, D // heads, B * R, B * nW * (A + nF)
    if g2b is None:
        g2b = to_bf16(g2.view(M, D))
    dz = lin_bwd(blk.mlp.fc2, g2b, t['u'], M, gelu_aux=t['z'])
    dh2 = lin_bwd(blk.mlp.fc1, dz, t['h2'], M)
    g1 = _e((B, R, D), F32, dev)
    ln_bwd(blk.norm2, None, t['x1'], B, t['st2'], dy_bf16=dh2, dx1=g1, res1=g2)
    dt = _e((Mw, D), BF16, dev)
    ops.window_unfold(g1, blk.rows32, B, nW, A, nF, L, D, 1.0 / nW, dt)                    # backward of the fold (mean -> 1 / nW)
    do = lin_bwd(blk.attn.proj, dt, t['o'], Mw)
    dqkv = _e((Mw, 3 * D), BF16, dev)
    dS = _e((B * nW, heads, N, t['ld']), F32, dev)
    qkv = t['qkv']
    ops.hold(qkv, dqkv)
    es = qkv.element_size()
    ops.attn_bias_bwd(qkv.data_ptr(), qkv.data_ptr() + es * D, qkv.data_ptr() + es * 2 * D, t['o'], do, t['lse'],
                      torch.empty_like(t['lse']), dqkv.data_ptr(), dqkv.data_ptr() + es * D, dqkv.data_ptr() + es * 2 * D,
                      B * nW, heads, N, N, hd, hd, N * 3 * D, 3 * D, N * 3 * D, 3 * D, N * 3 * D, 3 * D, N * D, D, N * D, D,
                      N * 3 * D, 3 * D, N * 3 * D, 3 * D, N * 3 * D, 3 * D, blk.attn.scale, t['bias'], t['nb'], t['ld'], dS)
    table = blk.attn.relative_position_bias_table
    ops.relpos_bias_bwd(dS, blk.index32, B * nW, heads, A, N, t['ld'], table.shape[0], gbuf(table))
    _ready(table)
    dseq = lin_bwd(blk.attn.qkv, dqkv, t['seq'], Mw, dx_bf16=False)                        # fp32 [Mw, D]
    dh1 = _e((M, D), F32, dev)
    ops.window_fold(dseq, blk.inv32, None, B, nW, A, nF, L, D, 1.0, dh1)                   # backward of the unfold (repeat -> sum)
    dx = _e((B, R, D), F32, dev)
    dxb = _e((M, D), BF16, dev)
    ln_bwd(blk.norm1, None, x, B, t['st1'], dy_f32=dh1, dx1=dx, res1=g1, dx1_bf16=dxb)
    return dx, dxb


# ------------------------------------------------------------------------------------------------
# loss (models/avmae.py:182-214)
# ------------------------------------------------------------------------------------------------
def loss_fwd(img, pred, mask, norm_pix):
    B, L = mask.shape
    dev = img.device
    lp, tm, tr = _e((B * L,), F32, dev), _e((B * L,), F32, dev), _e((B * L,), F32, dev)
    loss, msum = _e((1,), F32, dev), _e((1,), F32, dev)
    ops.patch_mse_fwd(img, pred, mask, norm_pix, lp, tm, tr, loss, msum)
    return loss.view(()), dict(tm=tm, tr=tr, msum=msum)


def loss_bwd(img, pred, mask, t, gout):
    B, L, P = pred.shape
    dpred = _e((B * L, P), BF16, img.device)
    ops.patch_mse_bwd(img, pred, mask, t['tm'], t['tr'], t['msum'], gout, dpred)
    return dpred
