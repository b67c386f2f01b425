"""Whole-path drivers + the single autograd node that exposes them.

``avmae_apply`` runs the complete AVMAE step (masking -> patch embed -> 12 fused layers ->
final norms -> two MAE decoders -> masked per-patch MSE) on the HIP engine and returns the
reference's 4-tuple; the backward of that one node runs the hand-written backward and
accumulates parameter gradients in place.  ``encoder_apply`` does the same for
``DeepAVFusion.forward`` alone (kNN probe / downstream heads).
"""
from __future__ import annotations

import os

import torch

from . import engine as E
from . import ops
from .ops import F32


_SIDE_STREAMS = {}


def _streams(dev):
    """(main, side_a, side_f): the three independent blocks of a layer (image tower, audio tower, fusion block)
    run on three HIP streams — inside a captured step they become parallel branches of the hipGraph.
    DAV_STREAMS=0 serialises everything on the current stream."""
    main = torch.cuda.current_stream(dev)
    if os.environ.get('DAV_STREAMS', '1') == '0':
        return main, main, main
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        # (a higher dispatch priority for the fusion block's stream costs 9 ms per step: profiles/r04_stream_priority_ab.txt)
        _SIDE_STREAMS[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    sa, sf = _SIDE_STREAMS[key]
    return main, sa, sf


# Cross-joins between the two decoders' streams, every DEC_JOIN decoder blocks (0 = never, the default).  The decoders are two independent
# chains of ~60 launches.  Under rocprofv3 the replayed graph starts the second of two such unjoined branches only when the first is most of
# the way through (profiles/r06_streams_base.txt: 2.2 ms late in the forward, 2.9 ms in the backward; tools/runs_r06/graph_fork_probe.py shows
# the same on a bare two-chain graph), and joins every 1-4 blocks give the overlap back (decoder phases 10.0 -> 9.2 ms in the trace).  WITHOUT
# the profiler the branches do start together: tools/runs_r06/dec_overlap_probe.py times the two decoders' forward as graphs — alone 1.84 and
# 2.80 ms, both 4.08-4.10 ms with or without joins — and the step A/B is flat (24.27-24.58 vs 24.28-24.45 ms).  So the late start is an
# artefact of the tracing, the kernel-trace timelines under profiles/ understate the overlap of the decoders, and the switch stays off;
# what limits the pair is that each decoder's kernels already fill most of the GPU alone (both = 0.88 x the sum).
DEC_JOIN = int(os.environ.get('DAV_DEC_JOIN', '0'))
_PENDING = object()


def paired_steps(gen_a, stream_a, gen_b, stream_b, every):
    """Advance two step generators alternately, each under its stream; while both still have steps, cross-join the two streams after
    every ``every`` steps (0: never).  Returns the generators' values."""
    ra = rb = _PENDING
    i = 0
    while ra is _PENDING or rb is _PENDING:
        if ra is _PENDING:
            with torch.cuda.stream(stream_a):
                try:
                    next(gen_a)
                except StopIteration as e:
                    ra = e.value
        if rb is _PENDING:
            with torch.cuda.stream(stream_b):
                try:
                    next(gen_b)
                except StopIteration as e:
                    rb = e.value
        i += 1
        if every and i % every == 0 and ra is _PENDING and rb is _PENDING and stream_a is not stream_b:
            stream_a.wait_stream(stream_b)
            stream_b.wait_stream(stream_a)
    return ra, rb


def _batched(rows):
    """Launch batching (engine.batch, csrc/batch.h) with the paired tower blocks / decoders as LANES of one batch, or one HIP
    stream per tower: engine.lanes_for (DAV_BATCH = 1 / 0 force either; default: by the step's size)."""
    return E.lanes_for(rows)


def _f32c(x):
    return x.detach().to(dtype=F32).contiguous()


def _ids32(ids):
    return None if ids is None else ids.to(torch.int32).contiguous()


def drop_path_scales(owner, blk, B, dev, tag):
    """Per-sample DropPath scales of the two residual branches of ``blk`` — (s_attn, s_mlp), each fp32 [B] with values
    0 or 1/keep (timm DropPath, scale_by_keep) — or None when the block has no DropPath or the model is in eval mode.
    ``owner._drop_path_sampler(tag, branch, B)`` (a 0/1 keep mask) replaces the Bernoulli draw when set: parity tests
    inject the masks, the two RNG streams can never match."""
    p = getattr(blk, 'drop_path_prob', 0.0)
    if p <= 0.0 or not owner.training:
        return None
    keep = 1.0 - p
    sampler = getattr(owner, '_drop_path_sampler', None)

    def one(branch):
        if sampler is not None:
            m = sampler(tag, branch, B).to(device=dev, dtype=F32)
        else:
            m = torch.empty(B, device=dev, dtype=F32).bernoulli_(keep)
        return m / keep if keep > 0.0 else m
    return one(0), one(1)


def _sampler(owner, tag):
    """``owner._dropout_sampler(name, shape)`` -> 0 / 1 keep tensor of the REFERENCE's shape for the nn.Dropout called ``name``
    (e.g. 'visual.0.attn', 'fusion.1.attn_v.proj'); set by parity tests (the two RNG streams can never match)."""
    smp = getattr(owner, '_dropout_sampler', None)
    if smp is None:
        return lambda site, shape, cut=None: None
    def make(site, shape, cut=None):
        def sample(_shape):
            m = smp(f'{tag}.{site}', shape)
            return m if cut is None else cut(m)
        return sample
    return make


def block_drops(owner, blk, B, n, nF, dev, tag):
    """Dropout masks of ONE tower-block call (engine.block_fwd's ``dr``), or None when the block has no dropout or the model is in
    eval mode.  The reference runs the block over cat((x_fusion, x_modality)) and drops the fusion rows' outputs
    (models/deepavfusion.py:104-105): its masks span nF + n rows, of which the engine computes — and draws — the n modality rows
    (a sampler is asked for the reference's shape and cut)."""
    pa, pp = getattr(blk, 'attn_drop_prob', 0.0), getattr(blk, 'proj_drop_prob', 0.0)
    if (pa <= 0.0 and pp <= 0.0) or not owner.training:
        return None
    S = _sampler(owner, tag)
    R, H, D, Hd = nF + n, blk.num_heads, blk.norm1.weight.shape[0], blk.mlp.fc1.weight.shape[0]
    rows = lambda m: m[:, nF:].reshape(B * n, -1)
    dr = {}
    if pa > 0.0:        # call order of the reference: attn_drop, proj_drop, Mlp.drop1, Mlp.drop2
        dr['attn'] = E.draw_attn_keep(pa, B, H, n, R, dev, S('attn', (B, H, R, R), lambda m: m[:, :, nF:]))
    if pp > 0.0:
        dr['proj'] = E.draw_keep(pp, B * n, D, dev, S('proj', (B, R, D), rows))
        dr['fc1'] = E.draw_keep(pp, B * n, Hd, dev, S('fc1', (B, R, Hd), rows))
        dr['fc2'] = E.draw_keep(pp, B * n, D, dev, S('fc2', (B, R, D), rows))
    return dr


def fusion_drops(owner, fb, B, nF, nI, nA, dev, tag):
    """Dropout masks of ONE fusion-block call (engine.fusion_block_fwd's ``dr``) in the reference's call order; nI / nA = rows of
    the 2nd / 3rd positional argument."""
    pa, pp = getattr(fb, 'attn_drop_prob', 0.0), getattr(fb, 'proj_drop_prob', 0.0)
    if (pa <= 0.0 and pp <= 0.0) or not owner.training:
        return None
    S = _sampler(owner, tag)
    H, D, Hd = fb.num_heads, fb.norm1_mm.weight.shape[0], fb.mlp.fc1.weight.shape[0]
    flat = lambda m: m.reshape(-1, m.shape[-1])
    arch = getattr(fb, 'arch', 'factorized_mmi')
    dr = {}
    if arch == 'factorized_mmi':
        nmm, nv, na = fb.fusion_tkns
        for name, nq, nk in (('attn_v', nv, nI), ('attn_a', na, nA)):       # models/fusion_blocks.py:241-242, each :54, :58
            if pa > 0.0:
                dr[name + '.attn'] = E.draw_attn_keep(pa, B, H, nq, nk, dev, S(name + '.attn', (B, H, nq, nk)))
            if pp > 0.0:
                dr[name + '.proj'] = E.draw_keep(pp, B * nq, D, dev, S(name + '.proj', (B, nq, D), flat))
        nq, nk = nmm, nv * na
    else:
        nq, nk = nF, (nI + nA if arch == 'token' else nI * nA)
    if pa > 0.0:
        dr['attn'] = E.draw_attn_keep(pa, B, H, nq, nk, dev, S('attn', (B, H, nq, nk)))
    if pp > 0.0:
        dr['proj'] = E.draw_keep(pp, B * nq, D, dev, S('proj', (B, nq, D), flat))
        dr['fc1'] = E.draw_keep(pp, B * nF, Hd, dev, S('fc1', (B, nF, Hd), flat))
        dr['fc2'] = E.draw_keep(pp, B * nF, D, dev, S('fc2', (B, nF, D), flat))
    return dr


def _vis(enc):
    """The visual tower: ``.image`` of DeepAVFusion (models/deepavfusion.py:20) or ``.video`` of VideoEarlyFusion
    (models/video_earlyfusion.py:32) — the layer loop is the same."""
    return enc.visual if hasattr(enc, 'visual') else (enc.video if hasattr(enc, 'video') else enc.image)


# ------------------------------------------------------------------------------------------------
# encoder (models/deepavfusion.py:88-118)
# ------------------------------------------------------------------------------------------------
def encoder_fwd(enc, image, audio, ik32, ak32, want_f32=False, collect_embs=False):
    with E.fold_scope(enc):          # (a no-op inside avmae_fwd's scope)
        return _encoder_fwd(enc, image, audio, ik32, ak32, want_f32, collect_embs)


def _encoder_fwd(enc, image, audio, ik32, ak32, want_f32=False, collect_embs=False):
    B = image.shape[0]
    vis = _vis(enc)
    x_i, t_pi = E.patch_embed_fwd(vis, image, ik32)
    x_a, t_pa = E.patch_embed_fwd(enc.audio, audio, ak32)
    x_f = enc.fusion_tokens.detach().expand(B, -1, -1).clone(memory_format=torch.contiguous_format)   # never an alias of the parameter (B == 1)
    Hi, Ha, Hf = vis.num_heads, enc.audio.num_heads, enc.fusion_num_heads
    if E.ln_fuse_ok(x_f.shape[2]):      # twin + row statistics of the layer inputs, made once before the three chains of a layer fork
        for x in (x_i, x_a, x_f):
            E.ensure_tw(x)
    layers, embs = [], []
    main, sa, sf = _streams(image.device)
    batched = _batched(B * (x_i.shape[1] + x_f.shape[1]))          # rows of the (smaller) visual tower's blocks
    for l, (bi, ba, fb) in enumerate(zip(vis.blocks, enc.audio.blocks, enc.fusion_blocks)):
        # the reference draws in call order: visual block (attn, mlp), audio block, fusion block (:104-106)
        dpi, dpa = drop_path_scales(enc, bi, B, image.device, f'visual.{l}'), drop_path_scales(enc, ba, B, image.device, f'audio.{l}')
        dpf = drop_path_scales(enc, fb, B, image.device, f'fusion.{l}') if fb is not None else None
        xf_ctx = x_f if fb is not None else None
        tf = None
        # dropout masks (attn_drop / drop > 0, training mode): drawn here, before the layer's launch batch opens
        nFc = x_f.shape[1] if fb is not None else 0
        dri = block_drops(enc, bi, B, x_i.shape[1], nFc, image.device, f'visual.{l}')
        dra = block_drops(enc, ba, B, x_a.shape[1], nFc, image.device, f'audio.{l}')
        drf = fusion_drops(enc, fb, B, x_f.shape[1], x_i.shape[1], x_a.shape[1], image.device, f'fusion.{l}') if fb is not None else None
        if batched:
            # ONE launch batch per layer, three lanes — image block, audio block, fusion block (all read the layer inputs
            # only, models/deepavfusion.py:104-107): LayerNorms / GEMMs / attentions of equal rank go out as grouped grids
            # on one stream.  (hipGraph branches of this weight start ~150 us late on MI355X)
            lane_f = fb is not None and E.fusion_block_batchable(fb, dpf, drf) and not E.FUSION_ON_STREAM
            if fb is not None and E.FUSION_ON_STREAM:
                # mixed schedule: the fusion block's ~11 small dependent steps run on their own stream (its independent launches
                # still grouped per region) BESIDE the two tower lanes, which then need no idle steps
                sf.wait_stream(main)
                with torch.cuda.stream(sf):
                    n_f, tf = E.fusion_block_fwd(fb, x_f, x_i, x_a, Hf, enc.num_fusion, dpf, drf)
            elif fb is not None and not lane_f:
                n_f, tf = E.fusion_block_fwd(fb, x_f, x_i, x_a, Hf, enc.num_fusion, dpf, drf)      # reads the layer INPUT x_i / x_a (:106-107)
            idle = E.FUSION_IDLE_FWD if lane_f else 0
            with E.batch() as bt:
                bt.lane()
                n_i, ti = E.block_fwd(bi, x_i, xf_ctx, Hi, bi.norm1.eps, dpi, idle_before_mlp=idle, dr=dri)
                bt.lane()
                n_a, ta = E.block_fwd(ba, x_a, xf_ctx, Ha, ba.norm1.eps, dpa, idle_before_mlp=idle, dr=dra)
                if lane_f:
                    bt.lane()
                    n_f, tf = E.fusion_block_fwd(fb, x_f, x_i, x_a, Hf, enc.num_fusion, dpf, drf)
            if fb is not None and E.FUSION_ON_STREAM:
                main.wait_stream(sf)
            if fb is not None:
                x_f = n_f
            x_i, x_a = n_i, n_a
        else:
            sa.wait_stream(main)
            sf.wait_stream(main)
            with torch.cuda.stream(sa):
                n_a, ta = E.block_fwd(ba, x_a, xf_ctx, Ha, ba.norm1.eps, dpa, dr=dra)
            if fb is not None:
                with torch.cuda.stream(sf):
                    n_f, tf = E.fusion_block_fwd(fb, x_f, x_i, x_a, Hf, enc.num_fusion, dpf, drf)
            n_i, ti = E.block_fwd(bi, x_i, xf_ctx, Hi, bi.norm1.eps, dpi, dr=dri)
            x_i, x_a = n_i, n_a
            if fb is not None:
                x_f = n_f
            main.wait_stream(sa)
            main.wait_stream(sf)
        layers.append((ti, ta, tf))
        if collect_embs:
            embs.append((x_i, x_a, x_f))
    xi_b, xi32, st_i = E.ln_fwd(vis.norm, None, x_i, B, want_f32=want_f32)
    xa_b, xa32, st_a = E.ln_fwd(enc.audio.norm, None, x_a, B, want_f32=want_f32)
    xf_b, xf32, st_f = E.ln_fwd(enc.fusion_norm, None, x_f, B, want_f32=want_f32)
    tape = dict(t_pi=t_pi, t_pa=t_pa, layers=layers, x_i=x_i, x_a=x_a, x_f=x_f, st_i=st_i, st_a=st_a, st_f=st_f, B=B, lanes=batched)
    return (xi_b, xa_b, xf_b), (xi32, xa32, xf32), embs, tape


def encoder_bwd(enc, t, dxi_b=None, dxa_b=None, dxf_b=None, dxi32=None, dxa32=None, dxf32=None, layer_cb=None):
    """Gradients w.r.t. the three NORMED outputs (bf16 and/or fp32 parts) -> parameter gradients.
    ``layer_cb(l)`` is called after layer l's backward (all streams joined, its weight gradients launched)."""
    with E.deferred_wgrads():
        _encoder_bwd(enc, t, dxi_b, dxa_b, dxf_b, dxi32, dxa32, dxf32, layer_cb)


def _encoder_bwd(enc, t, dxi_b, dxa_b, dxf_b, dxi32, dxa32, dxf32, layer_cb=None):
    B = t['B']
    dev = t['x_i'].device
    vis = _vis(enc)

    def final_norm(norm, x, st, dy_b, dy32):
        g = torch.empty_like(x)
        gb = torch.empty(x.shape, dtype=E.BF16, device=dev)
        if dy_b is None and dy32 is None:
            return g.zero_(), gb.zero_()
        E.ln_bwd(norm, None, x, B, st, dy_bf16=dy_b, dy_f32=dy32, dx1=g, dx1_bf16=gb)
        return g, gb
    g_i, g_ib = final_norm(vis.norm, t['x_i'], t['st_i'], dxi_b, dxi32)
    g_a, g_ab = final_norm(enc.audio.norm, t['x_a'], t['st_a'], dxa_b, dxa32)
    g_f, g_fb = final_norm(enc.fusion_norm, t['x_f'], t['st_f'], dxf_b, dxf32)
    blocks = list(zip(vis.blocks, enc.audio.blocks, enc.fusion_blocks))
    main, sa, sf = _streams(dev)
    batched = t['lanes']                  # the schedule the forward of this step chose
    for l, ((bi, ba, fb), (ti, ta, tf)) in reversed(list(enumerate(zip(blocks, t['layers'])))):
        # Memory lifetime across streams: the gradients entering this layer were allocated on one stream (main for the
        # final norms, the fusion / audio stream further down) and are READ by kernels of another.  Dropping the last
        # reference hands the block back to the allocating stream's pool at once, and a later allocation on THAT stream
        # could overwrite it while the reader (concurrent in the captured graph) has not run yet.  So everything consumed
        # here stays referenced until all streams have re-joined.
        hold = (g_i, g_ib, g_a, g_ab, g_f, g_fb)
        if batched:
            # one launch batch, three lanes: the two tower blocks down to their qkv input gradient, and the fusion block's
            # whole backward.  The towers' last kernel — the norm1 backward — accumulates into the buffers the fusion
            # block's backward produces, so it forms a second (two-lane) batch behind the first.
            dx_f = dx_i = dx_a = None
            lane_f = fb is not None and E.fusion_block_batchable(fb, tf.get('dp'), tf.get('dr')) and not E.FUSION_ON_STREAM
            if fb is not None and E.FUSION_ON_STREAM:
                sf.wait_stream(main)
                with torch.cuda.stream(sf):
                    dx_f, dx_i, dx_a = E.fusion_block_bwd(fb, tf, g_f, g_fb)
            elif fb is not None and not lane_f:
                dx_f, dx_i, dx_a = E.fusion_block_bwd(fb, tf, g_f, g_fb)
            idle = E.FUSION_IDLE_BWD if lane_f else 0
            with E.batch() as bt:
                bt.lane()
                st_i = E.block_bwd_head(bi, ti, g_i, g_ib, idle_before_attn=idle)
                bt.lane()
                st_a = E.block_bwd_head(ba, ta, g_a, g_ab, idle_before_attn=idle)
                if lane_f:
                    bt.lane()
                    dx_f, dx_i, dx_a = E.fusion_block_bwd(fb, tf, g_f, g_fb)
            acc = 1 if fb is not None else 0
            if fb is not None and E.FUSION_ON_STREAM:
                main.wait_stream(sf)          # the norm1 backwards accumulate into the buffers the fusion block's backward wrote
            with E.batch() as bt:
                bt.lane()
                g_i, g_ib, dxf_i = E.block_bwd_tail(bi, ti, st_i, dx_fus=dx_f, dx_fus_acc=acc, dx_mod=dx_i, dx_mod_acc=acc)
                bt.lane()
                g_a, g_ab, dxf_a = E.block_bwd_tail(ba, ta, st_a, dx_mod=dx_a, dx_mod_acc=acc)
            if fb is not None:
                g_f = dxf_i.add_(dxf_a)
                g_fb = E.to_bf16(g_f)          # outside the next layer's batch: its fusion lane starts with the fc2 dgrad, like the towers'
            del st_i, st_a
        elif fb is None:
            sa.wait_stream(main)
            with torch.cuda.stream(sa):
                g_a, g_ab, _ = E.block_bwd(ba, ta, g_a, g_ab)
            g_i, g_ib, _ = E.block_bwd(bi, ti, g_i, g_ib)
            main.wait_stream(sa)
        else:
            sa.wait_stream(main)
            sf.wait_stream(main)
            # the fusion block's backward (many tiny kernels) runs beside the two tower blocks; their last
            # LayerNorm backward accumulates into the buffers it produces, so that kernel waits for it
            with torch.cuda.stream(sf):
                dx_f, dx_i, dx_a = E.fusion_block_bwd(fb, tf, g_f, g_fb)
            with torch.cuda.stream(sa):
                g_a, g_ab, dxf_a = E.block_bwd(ba, ta, g_a, g_ab, dx_mod=dx_a, dx_mod_acc=1,
                                               before_ln1=lambda: sa.wait_stream(sf))
            g_i, g_ib, dx_f = E.block_bwd(bi, ti, g_i, g_ib, dx_fus=dx_f, dx_fus_acc=1, dx_mod=dx_i, dx_mod_acc=1,
                                          before_ln1=lambda: main.wait_stream(sf))
            main.wait_stream(sa)
            main.wait_stream(sf)
            if sa is main:
                g_f, g_fb = dx_f.add_(dxf_a), None
            elif _ADD_CAST and E.PRECISION != 'fp32' and dx_f.is_contiguous() and dxf_a.is_contiguous() and dx_f.numel() % 4 == 0:
                g_f, g_fb = ops.add_cast(dx_f, dxf_a)          # the sum and the next block's bf16 operand in one pass
            else:
                g_f, g_fb = dx_f + dxf_a, None
        del hold
        if E.wgrad_flush_due(l, len(blocks)):
            E.flush_wgrads()      # the queued weight gradients (this layer's and, merged, the layers' before it) as one launch
        if layer_cb is not None and l > 0:
            layer_cb(l)
    E.patch_embed_bwd(vis, t['t_pi'], g_i, g_ib)
    E.patch_embed_bwd(enc.audio, t['t_pa'], g_a, g_ab)
    E.gbuf(enc.fusion_tokens).add_(g_f.sum(dim=0, keepdim=True))          # backward of .expand(B, -1, -1)
    E._ready(enc.fusion_tokens)


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, image, audio, ik32, ak32, return_embs, need_tape, *params):
        ctx.set_materialize_grads(False)
        with E.ln_fuse_for(need_tape):
            _, f32s, embs, tape = encoder_fwd(enc, image, audio, ik32, ak32, want_f32=True, collect_embs=return_embs)
        ctx.enc, ctx.tape = enc, (tape if need_tape else None)
        B = image.shape[0]
        outs = [x.view(B, -1, x.shape[-1]) for x in f32s]
        flat_embs = [e for tri in embs for e in tri]
        ctx.mark_non_differentiable(*flat_embs)
        return (*outs, *flat_embs)

    @staticmethod
    def backward(ctx, gi, ga, gf, *_):
        if ctx.tape is None:
            raise RuntimeError('backward through DeepAVFusion.forward called without a saved tape')
        c = lambda g: None if g is None else g.contiguous().view(-1, g.shape[-1])
        encoder_bwd(ctx.enc, ctx.tape, dxi32=c(gi), dxa32=c(ga), dxf32=c(gf))
        ctx.tape = None
        return (None,) * (7 + len(ctx.enc._param_list))


def encoder_apply(enc, image, audio, image_ids_keep, audio_ids_keep, return_embs):
    enc._param_list = [p for p in enc.parameters()]
    need_tape = torch.is_grad_enabled() and any(p.requires_grad for p in enc._param_list)
    out = _EncoderFn.apply(enc, _f32c(image), _f32c(audio), _ids32(image_ids_keep), _ids32(audio_ids_keep),
                           bool(return_embs), need_tape, *enc._param_list)
    if not return_embs:
        return out[0], out[1], out[2]
    rest = out[3:]
    embs = [tuple(rest[3 * i:3 * i + 3]) for i in range(len(rest) // 3)]     # per-layer (x_image, x_audio, x_fusion), detached
    return out[0], out[1], out[2], embs


# ------------------------------------------------------------------------------------------------
# small stand-alone pieces of the reference API
# ------------------------------------------------------------------------------------------------
class _PatchTokensFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vit, x, ids32, *params):
        ctx.set_materialize_grads(False)
        tok, tape = E.patch_embed_fwd(vit, x, ids32)
        ctx.vit, ctx.tape = vit, tape
        return tok

    @staticmethod
    def backward(ctx, g):
        E.patch_embed_bwd(ctx.vit, ctx.tape, g.contiguous(), None)
        return (None,) * (3 + 2)


def patch_tokens(vit, x, ids_keep):
    return _PatchTokensFn.apply(vit, _f32c(x), _ids32(ids_keep), vit.patch_embed.proj.weight, vit.patch_embed.proj.bias)


class _CrossAttentionFn(torch.autograd.Function):
    """Standalone ``CrossAttention.forward(x1, x2)`` (models/fusion_blocks.py:46-59): q / kv projections, softmax attention,
    proj — on the HIP kernels, with the hand-written backward."""

    @staticmethod
    def forward(ctx, ca, x1, x2, *params):
        ctx.set_materialize_grads(False)
        B, N1, D = x1.shape
        N2 = x2.shape[1]
        x1b, x2b = E.to_bf16(x1.reshape(B * N1, D)), E.to_bf16(x2.reshape(B * N2, D))
        H, hd = ca.num_heads, D // ca.num_heads
        S = _sampler(ca, 'cross')
        pa, pp = (ca.attn_drop_prob, ca.proj_drop_prob) if ca.training else (0.0, 0.0)
        keep = E.draw_attn_keep(pa, B, H, N1, N2, x1.device, S('attn', (B, H, N1, N2))) if pa > 0.0 else None
        ctx.pm = E.draw_keep(pp, B * N1, D, x1.device, S('proj', (B, N1, D), lambda m: m.reshape(B * N1, D))) if pp > 0.0 else None
        c = E._cross_fwd_seq(ca, x1b, None, N1, x2b, N2, B, D, ca.num_heads, x1.device, keep=keep)
        out = E.lin_fwd(ca.proj, c['o'], B * N1)
        if ctx.pm is not None:
            E._drop(out, ctx.pm, B, N1, D)
        ctx.ca, ctx.c, ctx.x1b, ctx.x2b, ctx.dims, ctx.np = ca, c, x1b, x2b, (B, N1, N2, D), len(params)
        # the attention matrix the reference also returns: softmax rebuilt from the kernels' own q / k and log-sum-exp
        q = c['q'].view(B, N1, H, hd).permute(0, 2, 1, 3).float()
        k = c['kv'].view(B, N2, 2, H, hd)[:, :, 0].permute(0, 2, 1, 3).float()
        attn = torch.exp((q @ k.transpose(-2, -1)) * ca.scale - c['lse'].unsqueeze(-1))
        if keep is not None:          # returned AFTER attn_drop (models/fusion_blocks.py:54-59)
            attn = attn * keep[0][..., :N2].float() * keep[2]
        ctx.mark_non_differentiable(attn)
        return out.view(B, N1, D), attn

    @staticmethod
    def backward(ctx, g, _g_attn):
        B, N1, N2, D = ctx.dims
        ca = ctx.ca
        with E.deferred_wgrads():
            gb = E.to_bf16(g.contiguous().view(B * N1, D))
            if ctx.pm is not None:
                gb = gb.clone() if gb.data_ptr() == g.data_ptr() else gb      # (fp32 engine: to_bf16 is the identity; autograd's g stays)
                E._drop(gb, ctx.pm, B, N1, D)
            do = E.lin_bwd(ca.proj, gb, ctx.c['o'], B * N1)
            dx1 = torch.empty(B * N1, D, dtype=E.BF16, device=g.device)
            dx2 = E._cross_bwd_seq(ca, ctx.c, do, ctx.x1b, None, N1, ctx.x2b, N2, B, D, ca.num_heads, dx1, None)
        return (None, dx1.float().view(B, N1, D), dx2.float().view(B, N2, D)) + (None,) * ctx.np


def cross_attention(ca, x1, x2):
    params = [p for p in ca.parameters()]
    c = lambda x: x.to(dtype=F32).contiguous()          # no detach: the inputs receive gradients
    return _CrossAttentionFn.apply(ca, c(x1), c(x2), *params)


class _SwinBlockFn(torch.autograd.Function):
    """One SwinTransformerBlock with fusion tokens (models/swin.py:160-209) as a differentiable op: the module-level API
    of models/swin.py; AVMAE's decoder drives the same engine functions directly."""
    @staticmethod
    def forward(ctx, blk, xcat, nF, *params):
        ctx.set_materialize_grads(False)
        out, tape = E.swin_block_fwd(blk, xcat, nF)
        ctx.blk, ctx.tape, ctx.np = blk, tape, len(params)
        return out

    @staticmethod
    def backward(ctx, g):
        with E.deferred_wgrads():
            dx, _ = E.swin_block_bwd(ctx.blk, ctx.tape, g.contiguous(), None)
        return (None, dx, None) + (None,) * ctx.np


def swin_block_apply(blk, x, x_fusion):
    """(x [B, L, C], x_fusion [B, Lf, C]) -> (x, x_fusion) as models/swin.py:160-209 returns them."""
    nF = x_fusion.shape[1]
    xcat = torch.cat([x_fusion.to(dtype=F32), x.to(dtype=F32)], dim=1).contiguous()      # the engine's [fusion | tokens] layout
    out = _SwinBlockFn.apply(blk, xcat, nF, *list(blk.parameters()))
    return out[:, nF:], out[:, :nF]


class _FusionBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fb, xmm, xv, xa, *params):
        ctx.set_materialize_grads(False)
        out, tape = E.fusion_block_fwd(fb, xmm, xv, xa, fb.num_heads, getattr(fb, 'fusion_tkns', None),
                                       drop_path_scales(fb, fb, xmm.shape[0], xmm.device, 'fusion'),
                                       fusion_drops(fb, fb, xmm.shape[0], xmm.shape[1], xv.shape[1], xa.shape[1], xmm.device, 'fusion'))
        ctx.fb, ctx.tape, ctx.np = fb, tape, len(params)
        return out

    @staticmethod
    def backward(ctx, g):
        with E.deferred_wgrads():
            dx_f, dx_i, dx_a = E.fusion_block_bwd(ctx.fb, ctx.tape, g.contiguous(), None)
        return (None, dx_f, dx_i, dx_a) + (None,) * ctx.np


def fusion_block(fb, xmm, xv, xa):
    c = lambda x: x.to(dtype=F32).contiguous()
    return _FusionBlockFn.apply(fb, c(xmm), c(xv), c(xa), *list(fb.parameters()))


# ------------------------------------------------------------------------------------------------
# the whole AVMAE step (models/avmae.py:216-236)
# ------------------------------------------------------------------------------------------------
def avmae_fwd(model, image, audio, noise_i, noise_a):
    with E.fold_scope(model):      # the gamma-folded weights behind the LayerNorms: one grouped refresh per step
        return _avmae_fwd(model, image, audio, noise_i, noise_a)


def _avmae_fwd(model, image, audio, noise_i, noise_a):
    enc = model.encoder
    B = image.shape[0]
    Li, La = model.image_gs[0] * model.image_gs[1], model.audio_gs[0] * model.audio_gs[1]
    nki, nka = int(Li * (1 - model.image_mask_ratio)), int(La * (1 - model.audio_mask_ratio))     # models/avmae.py:132
    ik, im, ir, ik32, ir32 = ops.mask_build(noise_i, nki)
    ak, am, ar, ak32, ar32 = ops.mask_build(noise_a, nka)
    (xi_b, xa_b, xf_b), _, _, t_enc = encoder_fwd(enc, image, audio, ik32, ak32)
    nF = enc.fusion_tokens.shape[1]
    dec_i, dec_a = model.decoder('image'), model.decoder('audio')
    main, sa, _ = _streams(image.device)
    lanes = t_enc['lanes']                                                        # the two decoders follow the encoder's schedule (as lanes beside a STREAMED encoder: +1.6 ms, profiles/r06_dec_overlap.txt)
    if lanes:
        with E.batch() as bt:                 # the two MAE decoders (models/avmae.py:147-180) in lockstep
            bt.lane()
            pred_i, t_di = E.decoder_fwd(dec_i, xi_b, xf_b, ir32, B, nki, nF)
            loss_i, t_li = E.loss_fwd(image, pred_i, im, model.image_norm_loss)
            bt.lane()
            pred_a, t_da = E.decoder_fwd(dec_a, xa_b, xf_b, ar32, B, nka, nF)
            loss_a, t_la = E.loss_fwd(audio, pred_a, am, model.audio_norm_loss)
    else:
        si = main
        sa.wait_stream(main)
        if si is not main:
            si.wait_stream(main)

        def chain(dec, x_b, r32, nk, target, mask, norm_loss):
            pred, t_d = yield from E.decoder_fwd_steps(dec, x_b, xf_b, r32, B, nk, nF)
            loss, t_l = E.loss_fwd(target, pred, mask, norm_loss)
            return pred, t_d, loss, t_l
        (pred_a, t_da, loss_a, t_la), (pred_i, t_di, loss_i, t_li) = paired_steps(
            chain(dec_a, xa_b, ar32, nka, audio, am, model.audio_norm_loss), sa,
            chain(dec_i, xi_b, ir32, nki, image, im, model.image_norm_loss), si, DEC_JOIN)
        main.wait_stream(sa)
        if si is not main:
            main.wait_stream(si)
    tape = dict(image=image, audio=audio, im=im, am=am, ik32=ik32, ak32=ak32, dec_lanes=lanes, t_enc=t_enc, t_di=t_di, t_da=t_da, t_li=t_li,
                t_la=t_la, pred_i=pred_i, pred_a=pred_a, B=B)
    aux = dict(image_ids_keep=ik, image_mask=im, image_ids_restore=ir, audio_ids_keep=ak, audio_mask=am, audio_ids_restore=ar)
    return (loss_i, loss_a, pred_i, pred_a), tape, aux


_DEC_WGRAD_JOINT = True
_ADD_CAST = True


def avmae_bwd(model, t, g_li, g_la, g_pi=None, g_pa=None, layer_cb=None):
    B = t['B']
    dec_i, dec_a = model.decoder('image'), model.decoder('audio')
    main, sa, _ = _streams(t['image'].device)
    if t['dec_lanes']:
        dpi = E.loss_bwd(t['image'], t['pred_i'], t['im'], t['t_li'], g_li)
        dpa = E.loss_bwd(t['audio'], t['pred_a'], t['am'], t['t_la'], g_la)
        if g_pi is not None:
            dpi = (dpi.float() + g_pi.reshape(dpi.shape)).to(E.BF16)
        if g_pa is not None:
            dpa = (dpa.float() + g_pa.reshape(dpa.shape)).to(E.BF16)
        with E.deferred_wgrads():             # both decoders' weight gradients: ONE grouped GEMM after the batch
            with E.batch() as bt:
                bt.lane()
                dxi_b, dxf_i = E.decoder_bwd(dec_i, t['t_di'], dpi, t['ik32'], B)
                bt.lane()
                dxa_b, dxf_a = E.decoder_bwd(dec_a, t['t_da'], dpa, t['ak32'], B)
    elif _DEC_WGRAD_JOINT and sa is not main:
        # the decoders run on their two streams, but their weight gradients go out TOGETHER, after both: each decoder's tiles
        # alone fill the 512 workgroup slots an integer number of times plus a nearly empty last round (1536 tiles of one
        # contraction length, then two small heads: 1944 + 504 us and 914 + 227 us), the union of both lists fills the tail
        # of one with the tiles of the other (2.7 ms instead of 3.7 ms of weight-gradient time per step)
        with E.deferred_wgrads():
            si = main
            sa.wait_stream(main)
            if si is not main:
                si.wait_stream(main)

            def chain(dec, target, pred, mask, t_l, g_l, g_p, t_d, k32):
                dp = E.loss_bwd(target, pred, mask, t_l, g_l)
                if g_p is not None:
                    dp = (dp.float() + g_p.reshape(dp.shape)).to(E.BF16)
                return (yield from E.decoder_bwd_steps(dec, t_d, dp, k32, B))
            (dxa_b, dxf_a), (dxi_b, dxf_i) = paired_steps(
                chain(dec_a, t['audio'], t['pred_a'], t['am'], t['t_la'], g_la, g_pa, t['t_da'], t['ak32']), sa,
                chain(dec_i, t['image'], t['pred_i'], t['im'], t['t_li'], g_li, g_pi, t['t_di'], t['ik32']), si, DEC_JOIN)
            main.wait_stream(sa)
            if si is not main:
                main.wait_stream(si)
            E.deferred_operands_to(main)          # (the decoders' operands were allocated on their streams)
    else:
        sa.wait_stream(main)
        # each decoder's weight gradients are queued and launched as ONE grouped GEMM on that decoder's stream
        with torch.cuda.stream(sa), E.deferred_wgrads():
            dpa = E.loss_bwd(t['audio'], t['pred_a'], t['am'], t['t_la'], g_la)
            if g_pa is not None:
                dpa = (dpa.float() + g_pa.reshape(dpa.shape)).to(E.BF16)
            dxa_b, dxf_a = E.decoder_bwd(dec_a, t['t_da'], dpa, t['ak32'], B)
        with E.deferred_wgrads():
            dpi = E.loss_bwd(t['image'], t['pred_i'], t['im'], t['t_li'], g_li)
            if g_pi is not None:
                dpi = (dpi.float() + g_pi.reshape(dpi.shape)).to(E.BF16)
            dxi_b, dxf_i = E.decoder_bwd(dec_i, t['t_di'], dpi, t['ik32'], B)
        main.wait_stream(sa)
    dxf32 = dxf_i.float() + dxf_a.float()            # both decoders read the same normed fusion tokens
    if layer_cb is not None:
        layer_cb(len(model.encoder.fusion_blocks))    # both decoders' backward done (streams joined): a legal graph cut
    encoder_bwd(model.encoder, t['t_enc'], dxi_b=dxi_b, dxa_b=dxa_b, dxf32=dxf32, layer_cb=layer_cb)


class _AVMAEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, image, audio, noise_i, noise_a, need_tape, *params):
        ctx.set_materialize_grads(False)
        with E.ln_fuse_for(need_tape):
            outs, tape, aux = avmae_fwd(model, image, audio, noise_i, noise_a)
        ctx.model, ctx.tape, ctx.np = model, (tape if need_tape else None), len(params)
        model._last_masks = aux
        return outs

    @staticmethod
    def backward(ctx, g_li, g_la, g_pi, g_pa):
        z = lambda g: g.contiguous() if g is not None else torch.zeros((), dtype=F32, device=ctx.tape['image'].device)
        avmae_bwd(ctx.model, ctx.tape, z(g_li), z(g_la), g_pi, g_pa)
        ctx.tape = None
        return (None,) * (6 + ctx.np)


def avmae_apply(model, image, audio, noise_image=None, noise_audio=None):
    B, dev = image.shape[0], image.device
    Li, La = model.image_gs[0] * model.image_gs[1], model.audio_gs[0] * model.audio_gs[1]
    if noise_image is None:
        noise_image = torch.rand(B, Li, device=dev)           # models/avmae.py:127
    if noise_audio is None:
        noise_audio = torch.rand(B, La, device=dev)
    params = getattr(model, '_param_list', None)
    if params is None:
        params = model._param_list = [p for p in model.parameters()]
    need_tape = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _AVMAEFn.apply(model, _f32c(image), _f32c(audio), _f32c(noise_image), _f32c(noise_audio), need_tape, *params)
