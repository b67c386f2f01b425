#!/usr/bin/env python3
"""End-to-end parity of the HIP path against the CPU oracle (and the committed golden fixtures),
printing per-tensor errors.  Diagnostic companion of tests/test_hip_parity.py.
Usage: python tests/gpu_parity.py [micro|tiny|base_b2] ..."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae, build_video_earlyfusion            # noqa: E402
from deepavfusion_amd.configs import CONFIGS as PCONFIGS         # noqa: E402
from oracle import avmae_oracle as O                            # noqa: E402
from oracle.configs import CONFIGS                              # noqa: E402

dev = torch.device('cuda')


def rel(a, b):
    a, b = torch.as_tensor(a).double().flatten().cpu(), torch.as_tensor(b).double().flatten().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def run(name, B, seed):
    cfg = CONFIGS[name]
    sd = O.closed_form_state(cfg, 0)
    model = build_avmae(cfg).to(dev)
    model.load_state_dict(sd, strict=True)
    image, audio, ni, na = O.synthetic_batch(cfg, B, seed=seed)
    # oracle (CPU fp32)
    t0 = time.time()
    sdo = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd.items()}
    li, la, pi, pa, aux = O.avmae_forward(sdo, cfg, image, audio, ni, na)
    (li + la).backward()
    t_or = time.time() - t0
    # HIP path
    model.zero_grad()
    out = model(image.to(dev), audio.to(dev), torch.from_numpy(ni).to(dev), torch.from_numpy(na).to(dev))
    (out[0] + out[1]).backward()
    torch.cuda.synchronize()
    print(f'[{name} B={B}] oracle {t_or:.1f}s  loss_image hip={float(out[0]):.6f} ref={float(li):.6f}  '
          f'loss_audio hip={float(out[1]):.6f} ref={float(la):.6f}')
    m = model._last_masks
    for k in ('image_ids_keep', 'image_mask', 'image_ids_restore', 'audio_ids_keep', 'audio_mask', 'audio_ids_restore'):
        same = np.array_equal(m[k].cpu().numpy(), aux[k])
        print(f'   {k}: {"bit-exact" if same else "MISMATCH"}')
    print(f'   pred_image rel {rel(out[2], pi):.3e}   pred_audio rel {rel(out[3], pa):.3e}')
    worst = []
    tot_h, tot_r = 0.0, 0.0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g_ref = sdo[n].grad
        if p.grad is None:
            print('   MISSING grad', n)
            continue
        e = rel(p.grad, g_ref)
        worst.append((e, n, float(g_ref.norm())))
        tot_h += float(p.grad.double().norm()) ** 2
        tot_r += float(g_ref.double().norm()) ** 2
    worst.sort(reverse=True)
    print(f'   grad-norm hip={tot_h ** 0.5:.6f} ref={tot_r ** 0.5:.6f}; worst rel-L2 grads:')
    for e, n, gn in worst[:22]:
        print(f'      {e:.3e}  |g|={gn:.3e}  {n}')
    med = sorted(w[0] for w in worst)[len(worst) // 2]
    print(f'   median grad rel-L2 {med:.3e} over {len(worst)} tensors')
    gpath = os.path.join(ROOT, 'tests', 'golden', f'e2e_{name}.npz')
    if os.path.exists(gpath):
        g = np.load(gpath)
        if int(g['B']) == B and int(g['seed']) == seed:
            print(f'   vs golden: loss_image ref={float(g["loss_image"]):.6f} loss_audio ref={float(g["loss_audio"]):.6f}')


def run_video(name, B, seed, **over):
    """VideoEarlyFusion (BASELINE configs[4] family) forward + backward under the fixtures' probe loss."""
    import dataclasses
    cfg = dataclasses.replace(CONFIGS[name], **over)
    sd = O.closed_form_state(cfg, 0)
    model = build_video_earlyfusion(dataclasses.replace(PCONFIGS[name], **over)).to(dev)
    model.load_state_dict(sd, strict=True)
    video, audio = O.synthetic_video_batch(cfg, B, seed=seed)
    t0 = time.time()
    sdo = {k: v.clone().requires_grad_(k not in O.FROZEN) for k, v in sd.items()}
    xv, xa, xf = O.video_earlyfusion_forward(sdo, cfg, video, audio)
    rs = np.random.RandomState(seed + 1)
    w = [torch.from_numpy(rs.standard_normal(tuple(t.shape)).astype(np.float32)) for t in (xv, xa, xf)]
    loss = (xv * w[0]).sum() + (xa * w[1]).sum() + (xf * w[2]).sum()
    loss.backward()
    t_or = time.time() - t0
    model.zero_grad()
    ov, oa, of = model(video.to(dev), audio.to(dev))
    lh = (ov * w[0].to(dev)).sum() + (oa * w[1].to(dev)).sum() + (of * w[2].to(dev)).sum()
    lh.backward()
    torch.cuda.synchronize()
    print(f'[{name} B={B}] oracle {t_or:.1f}s  probe loss hip={float(lh):.5f} ref={float(loss):.5f}')
    print(f'   x_video rel {rel(ov, xv):.3e}  x_audio rel {rel(oa, xa):.3e}  x_fusion rel {rel(of, xf):.3e}')
    worst = []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            print('   MISSING grad', n)
            continue
        worst.append((rel(p.grad, sdo[n].grad), n, float(sdo[n].grad.norm())))
    worst.sort(reverse=True)
    for e, n, gn in worst[:12]:
        print(f'      {e:.3e}  |g|={gn:.3e}  {n}')
    print(f'   median grad rel-L2 {sorted(w_[0] for w_ in worst)[len(worst) // 2]:.3e} over {len(worst)} tensors')
    if os.environ.get('DAV_PARITY_ALL'):
        import re
        groups = {}
        for e, n, gn in worst:
            key = re.sub(r'\.\d+\.', '.N.', n)
            groups.setdefault(key, []).append(e)
        for k, v in sorted(groups.items()):
            print(f'      {k:50s} min {min(v):.2e} max {max(v):.2e}')


if __name__ == '__main__':
    which = sys.argv[1:] or ['micro', 'tiny']
    for w in which:
        if w == 'micro':
            run('micro', 3, 21)
        elif w == 'tiny':
            run('tiny', 2, 22)
        elif w == 'base_b2':
            run('base', 2, 23)
        elif w == 'base_b64':          # the bench workload at full size (the oracle needs a few minutes of host time)
            run('base', 64, 26)
        elif w in ('base_token', 'base_dense', 'base_as', 'base_m75', 'large'):      # BASELINE configs[2] / [3] shapes
            run(w, 2, 25)
        elif w == 'video_micro':
            run_video('video_micro', 2, 31)
        elif w == 'video_base':
            run_video('video_base', 1, 33)
        elif w == 'video_long':       # micro widths on the full 784-token clip
            run_video('video_micro', int(os.environ.get('B', 1)), 34, video_size=(8, 224, 224), audio_size=(128, 192))
