#!/usr/bin/env python3
"""What does a whole kernel family cost INSIDE the captured step?  (measurement only; patches the engine in its own process)

Captured steps compared in one process, replays interleaved round-robin:
  base         the product step
  no_fusion    every factorised fusion block replaced by the identity (x_f passes through; backward: dx_f = g, dx_i = dx_a = 0):
               the upper bound of what ANY speed-up of the fusion block's ~11 forward / ~14 backward launches per layer can buy
  no_adamw     the optimizer pass dropped from the captured graph
Results are wrong by construction in every mode but base — the numbers are step times only.

    python tools/instep_knockout.py [--config base] [--batch 64] [--rounds 6] [--reps 10] [--modes base,no_fusion]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='base')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=6)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--modes', default='base,no_fusion')
    a = ap.parse_args()

    from deepavfusion_amd import engine as E
    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer

    dev = torch.device('cuda', 0)
    cfg = CONFIGS[a.config]
    B = a.batch
    torch.manual_seed(0)
    model = build_avmae(cfg).to(dev)
    nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
    groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
    opt = FlatAdamW(groups, lr=1.5e-4 * B / 256, betas=(0.9, 0.95), model=model)
    trainer = Trainer(model, optimizer=opt, accum_iter=1, use_amp=True, distributed=False)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    image = torch.randn(B, 3, *cfg.image_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)
    torch.manual_seed(0)

    orig_f, orig_b = E.fusion_block_fwd, E.fusion_block_bwd

    def id_fwd(fb, x_f, x_i, x_a, heads, tkns, dp=None):
        return x_f, dict(shape_i=x_i.shape, shape_a=x_a.shape)

    def id_bwd(fb, t, g_, gb, *, dx_i=None, dx_a=None):
        z = lambda s: torch.zeros(s, dtype=torch.float32, device=g_.device)
        return g_, (dx_i if dx_i is not None else z(t['shape_i'])), (dx_a if dx_a is not None else z(t['shape_a']))

    modes = [m for m in a.modes.split(',') if m]
    steps = {}
    for m in modes:
        E.fusion_block_fwd, E.fusion_block_bwd = (id_fwd, id_bwd) if m == 'no_fusion' else (orig_f, orig_b)
        steps[m] = GraphedStep(trainer, image.shape, audio.shape)
        E.fusion_block_fwd, E.fusion_block_bwd = orig_f, orig_b
    for m in modes:
        for _ in range(3):
            steps[m](image, audio)
    torch.cuda.synchronize()
    times = {m: [] for m in modes}
    for r in range(a.rounds):
        for m in modes:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                steps[m](image, audio)
            torch.cuda.synchronize()
            times[m].append((time.perf_counter() - t0) / a.reps * 1e3)
    med = {m: sorted(v)[len(v) // 2] for m, v in times.items()}
    sched = f"DAV_STREAMS={os.environ.get('DAV_STREAMS', '1')} DAV_BATCH={os.environ.get('DAV_BATCH', 'auto')}"
    print(f'# in-step knockouts, config {a.config} B={B}, schedule [{sched}], {a.rounds} interleaved rounds x {a.reps} replays')
    for m in modes:
        print(f'{m:10s} median {med[m]:7.3f} ms  min {min(times[m]):7.3f} ms   rounds ' + ' '.join(f'{t:.2f}' for t in times[m]))
    for m in modes:
        if m != 'base' and 'base' in med:
            print(f'{m}: the family costs {med["base"] - med[m]:.3f} ms of the {med["base"]:.3f} ms step')
    print('JSON ' + json.dumps({'schedule': sched, 'median_ms': med}))


if __name__ == '__main__':
    main()
