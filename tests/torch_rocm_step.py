#!/usr/bin/env python3
"""Same-box STOCK-STACK comparator (round-3 review item 8): the oracle's pure-torch restatement of the reference step run on
``cuda`` under bf16 autocast — rocBLAS / hipBLASLt GEMMs, fused SDPA (what timm's Attention calls when fused attention is on),
ATen LayerNorm / GELU / softmax, torch.optim.AdamW — i.e. what the reference's own code would do on this MI355X.

TEST INFRASTRUCTURE / MEASUREMENT ONLY.  It lives under tests/ because it imports ``oracle`` (only tests/, smoke() and the
cpu_baseline leg of bench.py may); nothing in the package or in bench.py's timed region imports it, and it is not a target:
it calibrates "matching or beating" — where 2400 AV-pairs/s stands against the reference's stack on the same GPU.

    python tests/torch_rocm_step.py [--config base] [--batch 64] [--steps 20] [--warmup 5] [--no-sdpa] [--fp32]
Prints one JSON line (tools/torch_rocm_step.py forwards it into profiles/).
Follows /root/reference/train.py:151-180 (forward under autocast -> loss sum -> backward -> AdamW step -> zero_grad); the
reference's fp16 + GradScaler (util/misc.py:38,140) is replaced by bf16 without scaling, which can only favour this stack.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='base')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-sdpa', action='store_true', help='explicit softmax attention instead of F.scaled_dot_product_attention')
    ap.add_argument('--fp32', action='store_true', help='no autocast')
    ap.add_argument('--foreach', action='store_true', help='torch.optim.AdamW(foreach=True) instead of fused=True')
    a = ap.parse_args()

    from oracle import avmae_oracle as O
    from oracle.configs import CONFIGS as OC
    cfg = OC[a.config]
    dev = torch.device('cuda', 0)
    if not a.no_sdpa:        # timm 0.9.2 Attention: F.scaled_dot_product_attention when fused attention is available
        O.softmax_attention = lambda q, k, v, scale: F.scaled_dot_product_attention(q, k, v, scale=scale)
    sd0 = O.closed_form_state(cfg, 0)
    sd = {k: v.to(dev).clone().requires_grad_(k not in O.FROZEN and v.is_floating_point()) for k, v in sd0.items()}
    params = [p for p in sd.values() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1.5e-4 * a.batch / 256, betas=(0.9, 0.95), weight_decay=0.05,
                            **({'foreach': True} if a.foreach else {'fused': True}))
    B = a.batch
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    image = torch.randn(B, 3, *cfg.image_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)
    Li, La = cfg.image_grid[0] * cfg.image_grid[1], cfg.audio_grid[0] * cfg.audio_grid[1]

    def masking(L, ratio):          # models/avmae.py:120-142 on the device (argsort of uniform noise)
        noise = torch.rand(B, L, device=dev)
        ids_shuffle = torch.argsort(noise, dim=1)
        ids_restore = torch.argsort(ids_shuffle, dim=1)
        lk = O.len_keep_of(L, ratio)
        mask = torch.ones(B, L, device=dev)
        mask[:, :lk] = 0
        return ids_shuffle[:, :lk], torch.gather(mask, 1, ids_restore), ids_restore

    def step():
        ik, im, ir = masking(Li, cfg.image_mask_ratio)
        ak, am, ar = masking(La, cfg.audio_mask_ratio)
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=not a.fp32):
            x_i, x_a, x_f = O.deepavfusion_forward(sd, cfg, image, audio, ik, ak, prefix='encoder.')
            pred_i = O.forward_decoder(x_i, x_f, ir, sd, cfg, 'image')
            pred_a = O.forward_decoder(x_a, x_f, ar, sd, cfg, 'audio')
        li = O.forward_loss(O.patchify(image, (cfg.patch, cfg.patch)), pred_i.float(), im, cfg.image_norm_loss)
        la = O.forward_loss(O.patchify(audio, (cfg.patch, cfg.patch)), pred_a.float(), am, cfg.audio_norm_loss)
        (li + la).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return li, la

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        li, la = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({'what': 'stock PyTorch-ROCm stack (oracle restatement of the reference step on cuda: hipBLASLt/rocBLAS + '
                              + ('explicit softmax' if a.no_sdpa else 'SDPA') + ' + ATen, ' + ('fp32' if a.fp32 else 'bf16 autocast')
                              + ', torch.optim.AdamW ' + ('foreach' if a.foreach else 'fused') + '), eager, same synthetic workload; comparator only',
                      'config': a.config, 'B': B, 'steps': a.steps, 'warmup': a.warmup, 'value': round(B * a.steps / dt, 2), 'unit': 'AV-pairs/s',
                      'ms_per_step': round(dt / a.steps * 1e3, 3), 'loss': round(float(li) + float(la), 5),
                      'torch': torch.__version__, 'max_mem_GB': round(torch.cuda.max_memory_allocated() / 2**30, 2)}))


if __name__ == '__main__':
    main()
