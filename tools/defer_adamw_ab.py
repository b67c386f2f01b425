#!/usr/bin/env python3
"""Deferred AdamW (GraphedStep(defer=True)) against the plain captured step, in ONE process, replays interleaved round-robin.

Two models / optimizers of the same configuration, one captured step each; every round times ``reps`` replays of each; the deferred
variant's ``flush()`` (the one update left pending) runs outside the timed region — inside it every replay already carries one
whole AdamW pass (the previous step's), so both variants do forward + backward + AdamW per replay.

    python tools/defer_adamw_ab.py [--config base] [--batch 64] [--rounds 8] [--reps 10]
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
    ap.add_argument('--rounds', type=int, default=8)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--variants', default='plain,defer,plainB,deferB')
    a = ap.parse_args()

    from deepavfusion_amd.build_model import build_avmae
    from deepavfusion_amd.configs import CONFIGS
    from deepavfusion_amd.util import lr_sched
    from deepavfusion_amd.util.flat import FlatAdamW
    from deepavfusion_amd.util.misc import GraphedStep, Trainer

    dev = torch.device('cuda', 0)
    cfg = CONFIGS[a.config]
    B = a.batch
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    image = torch.randn(B, 3, *cfg.image_size, device=dev, generator=g)
    audio = (torch.randn(B, 1, *cfg.audio_size, device=dev, generator=g) * 2.0 - 3.0).clamp(-7, 4)

    steps = {}
    for v in [v for v in a.variants.split(',') if v]:
        torch.manual_seed(0)
        model = build_avmae(cfg).to(dev)
        nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
        groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
        opt = FlatAdamW(groups, lr=1.5e-4 * B / 256, betas=(0.9, 0.95), model=model)
        trainer = Trainer(model, optimizer=opt, accum_iter=1, use_amp=True, distributed=False)
        torch.manual_seed(0)
        steps[v] = GraphedStep(trainer, image.shape, audio.shape, defer=v.startswith('defer'))
    for gs in steps.values():
        for _ in range(3):
            gs(image, audio)
        gs.flush()
    torch.cuda.synchronize()
    times = {v: [] for v in steps}
    losses = {}
    for r in range(a.rounds):
        for v, gs in steps.items():
            gs(image, audio)                    # (deferred: puts an update in flight, so every timed replay carries one)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                out = gs(image, audio)
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / a.reps * 1e3)
            gs.flush()
            torch.cuda.synchronize()
            losses[v] = float(out[0]) + float(out[1])
            gs.check()
    med = {v: sorted(t)[len(t) // 2] for v, t in times.items()}
    print(f'# deferred AdamW A/B, config {a.config} B={B}, {a.rounds} interleaved rounds x {a.reps} replays')
    for v in steps:
        print(f'{v:8s} median {med[v]:7.3f} ms  min {min(times[v]):7.3f} ms  loss {losses[v]:.5f}   rounds ' + ' '.join(f'{t:.2f}' for t in times[v]))
    print('JSON ' + json.dumps({'config': a.config, 'B': B, 'median_ms': med, 'min_ms': {v: min(t) for v, t in times.items()}}))


if __name__ == '__main__':
    main()
