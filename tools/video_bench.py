#!/usr/bin/env python3
"""BASELINE.json configs[4]: ViT-Base video early fusion (8-frame 224x224 clip + (128,192) log-mel), bf16, batch 16,
one MI355X — forward + backward (sum-of-outputs loss, models/video_earlyfusion.py:185-186) + AdamW, replayed from a
hipGraph.  A parity-test configuration, not the bench line (bench.py measures configs[1]); this tool reports its step
time, the necessary-FLOP MFMA fraction and the share of the long-sequence attention kernels.

    python tools/video_bench.py [--batch 16] [--steps 20] [--no-graph]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MFMA_BF16_PEAK_TFLOPS = 2500.0


def lin(t, i, o):
    return 2.0 * t * i * o


def attn(nq, nk, dqk, dv):
    return 2.0 * nq * nk * (dqk + dv)


def necessary_fwd_flops_per_clip(cfg):
    """SURVEY.md section 8(d) work model for the video encoder: modality blocks with c fusion context rows whose own
    outputs are dropped, factorised fusion blocks; no masking, no decoder."""
    D, c = cfg.embed_dim, sum(cfg.fusion_tkns)
    nmm, nv, na = cfg.fusion_tkns
    gv, ga = cfg.video_grid, cfg.audio_grid
    nV, nA = gv[0] * gv[1] * gv[2], ga[0] * ga[1]
    Hm = int(D * cfg.mlp_ratio)
    f = lin(nV, 3 * cfg.video_patch[0] * 256, D) + lin(nA, 256, D)
    for n in (nV, nA):
        f += cfg.depth * (lin(n, D, D) + lin(n + c, D, 2 * D) + attn(n, n + c, D, D) + lin(n, D, D) + 2 * lin(n, D, Hm))
    Da, Hf = int(D * cfg.fusion_attn_ratio), int(D * cfg.fusion_mlp_ratio)
    fus = 0.0
    for nq, nk in ((nv, nV), (na, nA)):
        fus += lin(nq, D, D) + lin(nk, D, 2 * D) + attn(nq, nk, D, D) + lin(nq, D, D)
    fus += lin(nv + na, D, Da) + lin(nv + na, D, D) + lin(nmm, D, Da) + attn(nmm, nv * na, Da, D) + lin(nmm, D, D) + 2 * lin(c, D, Hf)
    return f + len(cfg.fusion_layers) * fus


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='video_base')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--race', type=int, default=0, help='race screen: capture forward+backward only, replay N times, gradients must repeat')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    from deepavfusion_amd import autograd_bridge as bridge
    from deepavfusion_amd import engine
    from deepavfusion_amd.build_model import build_video_earlyfusion
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util.flat import FlatAdamW
    cfg = CONFIGS[a.config]
    B = a.batch
    torch.manual_seed(0)
    model = build_video_earlyfusion(cfg).to(dev)
    nd = [p for n, p in model.named_parameters() if p.requires_grad and ('bias' in n or 'norm' in n)]
    wd = [p for n, p in model.named_parameters() if p.requires_grad and not ('bias' in n or 'norm' in n)]
    opt = FlatAdamW([{'params': wd, 'weight_decay': 0.05}, {'params': nd, 'weight_decay': 0.0}], lr=1e-4, betas=(0.9, 0.95), model=model)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    video = torch.randn(B, 3, *cfg.video_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)

    def fwd_bwd():
        (_, _, _), f32s, _, tape = bridge.encoder_fwd(model, video, audio, None, None, want_f32=True)
        loss = f32s[0].sum() + f32s[1].sum() + f32s[2].sum()
        ones = [torch.ones_like(t) for t in f32s]                       # d(sum)/d(out)
        bridge.encoder_bwd(model, tape, dxi32=ones[0], dxa32=ones[1], dxf32=ones[2])
        return loss

    def eager_step():
        opt.prepare_step()
        engine.refresh_weight_cache(model)
        loss = fwd_bwd()
        opt.launch_step(fused_norm_and_zero=True)
        return loss

    if a.race:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                opt.flat.zero_grad()
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            graph.capture_begin()
            opt.flat.zero_grad()
            engine.invalidate_weight_cache(model.parameters())
            engine.refresh_weight_cache(model)
            fwd_bwd()
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        ref, bad = None, 0
        for it in range(a.race):
            graph.replay()
            torch.cuda.synchronize()
            g_ = opt.flat.flat_g.clone()
            if ref is None:
                ref = g_
            elif float((g_ - ref).abs().max()) > 1e-5 * float(ref.abs().max()):
                bad += 1
        print(f'race screen {a.config} B={B}: {a.race} replays, {bad} deviating')
        return
    opt.flat.zero_grad()
    if a.no_graph:
        step = eager_step
        step()
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            graph.capture_begin()
            engine.invalidate_weight_cache(model.parameters())
            engine.refresh_weight_cache(model)
            loss_t = fwd_bwd()
            opt.launch_step(fused_norm_and_zero=True)
            graph.capture_end()
        torch.cuda.current_stream().wait_stream(cap)
        torch.cuda.synchronize()

        def step():
            opt.prepare_step()
            graph.replay()
            return loss_t
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / a.steps * 1e3
    fl = 3.0 * necessary_fwd_flops_per_clip(cfg)
    gv = cfg.video_grid
    nV, c = gv[0] * gv[1] * gv[2], sum(cfg.fusion_tkns)
    attn_fl = 3.5 * cfg.depth * attn(nV, nV + c, cfg.embed_dim, cfg.embed_dim)      # fwd + 2.5x bwd of the video blocks' attention
    print(json.dumps({
        'metric': 'AV-clips/sec, video_efav_base step (8x224x224 clip + 3s audio), fwd+bwd+AdamW', 'value': round(B * a.steps / dt, 2),
        'unit': 'clips/s', 'n_gpus': 1, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms, 3), 'dtype': 'bf16',
        'data': 'synthetic', 'config': {'workload': f'VideoEarlyFusion({a.config}) B={B}, rows {nV}+{c} / {cfg.audio_grid[0] * cfg.audio_grid[1]}+{c}',
                                        'graph': not a.no_graph},
        'loss': round(float(loss), 3), 'step_necessary_gflop_per_clip': round(fl / 1e9, 1),
        'step_mfma_frac': round(fl * B / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
        'video_attention_gflop_per_clip': round(attn_fl / 1e9, 1)}), flush=True)


if __name__ == '__main__':
    main()
