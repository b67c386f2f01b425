#!/usr/bin/env python3
"""Race screen: the same forward + backward (fixed data, fixed masking noise) many times; every parameter gradient
must repeat up to summation-order noise.  Prints the parameters whose gradient deviates from the first run.
Usage: python tests/grad_repeat_probe.py [config] [iters] [batch]     env DAV_STREAMS=0 serialises the three branches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from oracle import avmae_oracle as O                                  # noqa: E402
from oracle.configs import CONFIGS as OC                              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'micro'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
model = build_avmae(CONFIGS[name]).cuda()
model.load_state_dict(O.closed_form_state(OC[name], 0), strict=True)
image, audio, ni, na = O.structured_batch(OC[name], B, seed=3)
image, audio, ni, na = image.cuda(), audio.cuda(), torch.from_numpy(ni).cuda(), torch.from_numpy(na).cuda()
params = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
ref, bad = None, {}
for it in range(iters):
    model.zero_grad(set_to_none=True)
    out = model(image, audio, ni, na)
    (out[0] + out[1]).backward()
    torch.cuda.synchronize()
    g = {n: p.grad.detach().clone() for n, p in params}
    if ref is None:
        ref = g
        continue
    for n, _ in params:
        d = float((g[n] - ref[n]).norm() / (ref[n].norm() + 1e-20))
        if d > 1e-5:
            bad.setdefault(n, []).append((it, d))
print(f'{name} B={B}: {iters} iterations, {len(bad)} parameters deviated (> 1e-5 relative) at least once')
for n, ev in sorted(bad.items(), key=lambda kv: -max(e[1] for e in kv[1]))[:25]:
    print(f'   {n:60s} {len(ev):4d} times, worst {max(e[1] for e in ev):.2e} (iters {[e[0] for e in ev][:6]})')
