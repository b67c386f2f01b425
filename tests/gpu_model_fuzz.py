#!/usr/bin/env python3
"""End-to-end fuzz: random path configurations on the micro towers (input sizes, fusion token counts, mask ratios, fusion
widths and architectures, batch sizes incl. 1) — HIP path vs the CPU oracle, losses + every parameter gradient.
Usage: python tests/gpu_model_fuzz.py [seed] [cases]"""
import dataclasses
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from oracle import avmae_oracle as O                                  # noqa: E402
from oracle.configs import CONFIGS as OC                              # noqa: E402

ZERO_GRADS = ('attn.k.bias',)


def one_case(rng, idx):
    over = dict(
        image_size=(16 * rng.randint(1, 6), 16 * rng.randint(1, 6)),
        audio_size=(16 * rng.randint(1, 4), 16 * rng.randint(1, 10)),
        fusion_tkns=(rng.randint(1, 6), rng.randint(1, 5), rng.randint(1, 5)),
        image_mask_ratio=rng.choice([0.5, 0.75, 0.8, 0.9]), audio_mask_ratio=rng.choice([0.5, 0.75, 0.8]),
        fusion_attn_ratio=rng.choice([0.25, 1.0]), fusion_mlp_ratio=rng.choice([1.0, 4.0]),
        fusion_arch=rng.choice(['factorized_mmi', 'factorized_mmi', 'token', 'dense_mmi']),
        image_norm_loss=rng.random() < 0.7, audio_norm_loss=rng.random() < 0.7)
    B = rng.choice([1, 2, 3, 5, 8])
    ocfg = dataclasses.replace(OC['micro'], **over)
    if O.len_keep_of(ocfg.image_grid[0] * ocfg.image_grid[1], ocfg.image_mask_ratio) < 1 or \
            O.len_keep_of(ocfg.audio_grid[0] * ocfg.audio_grid[1], ocfg.audio_mask_ratio) < 1:
        return None                                   # the reference itself cannot run with zero kept patches
    model = build_avmae(dataclasses.replace(CONFIGS['micro'], **over)).cuda()
    sd = O.closed_form_state(ocfg, idx)
    model.load_state_dict(sd, strict=True)
    image, audio, ni, na = O.synthetic_batch(ocfg, B, seed=100 + idx)
    out = model(image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda())
    (out[0] + out[1]).backward()
    sdo = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd.items()}
    li, la, pi, pa, aux = O.avmae_forward(sdo, ocfg, image, audio, ni, na)
    (li + la).backward()
    tag = f'case {idx}: B={B} ' + ' '.join(f'{k}={v}' for k, v in over.items())
    bad = []
    for k in ('image_ids_keep', 'image_mask', 'image_ids_restore', 'audio_ids_keep', 'audio_mask', 'audio_ids_restore'):
        if not np.array_equal(model._last_masks[k].cpu().numpy(), aux[k]):
            bad.append(f'{k} not bit-exact')
    for got, ref, nm in ((out[0], li, 'loss_image'), (out[1], la, 'loss_audio')):
        if abs(float(got) - float(ref)) > 2e-3 * abs(float(ref)):
            bad.append(f'{nm} {float(got):.6f} vs {float(ref):.6f}')
    g_all = sum(float(v.grad.double().norm()) ** 2 for v in sdo.values() if v.grad is not None) ** 0.5
    for n, p in model.named_parameters():
        if not p.requires_grad or n.endswith(ZERO_GRADS):
            continue
        if p.grad is None:
            bad.append(f'{n}: no gradient')
            continue
        ref = sdo[n].grad.double()
        d = float((p.grad.detach().double().cpu() - ref).norm())
        if d > 5e-2 * float(ref.norm()) + 2e-4 * g_all:
            bad.append(f'{n}: |d|={d:.3e} |g|={float(ref.norm()):.3e}')
    print(('FAIL ' if bad else 'ok   ') + tag, flush=True)
    for b in bad[:6]:
        print('      ', b)
    return bad


if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rng = random.Random(seed)
    nbad = 0
    for i in range(cases):
        r = one_case(rng, 1000 * seed + i)
        nbad += bool(r)
    print(f'model fuzz seed {seed}: {nbad} failing cases of {cases}')
    sys.exit(1 if nbad else 0)
