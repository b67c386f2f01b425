#!/usr/bin/env python3
"""Race screen for the CAPTURED step: forward + backward with fixed data and fixed masking noise captured in one
hipGraph (the three branches of a layer are parallel graph branches there), replayed many times; the flat gradient
buffer must repeat up to summation-order noise.  Usage: python tests/graph_race_probe.py [config] [replays] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import autograd_bridge as bridge                # noqa: E402
from deepavfusion_amd import engine                                   # noqa: E402
from deepavfusion_amd.build_model import build_avmae                  # noqa: E402
from deepavfusion_amd.configs import CONFIGS                          # noqa: E402
from deepavfusion_amd.util import lr_sched                            # noqa: E402
from deepavfusion_amd.util.flat import FlatAdamW                      # noqa: E402
from oracle import avmae_oracle as O                                  # noqa: E402
from oracle.configs import CONFIGS as OC                              # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'micro'
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
model = build_avmae(CONFIGS[name]).cuda()
if name in OC and OC[name].embed_dim <= 192:
    model.load_state_dict(O.closed_form_state(OC[name], 0), strict=True)
nd = [n for n, p in model.named_parameters() if 'bias' in n or 'norm' in n]
groups = lr_sched.param_groups_pretrained(model, 0.05, no_weight_decay_list=nd, image_pt='', audio_pt='')
opt = FlatAdamW(groups, lr=1e-3, betas=(0.9, 0.95), model=model)
cfg = CONFIGS[name]
g = torch.Generator(device='cuda'); g.manual_seed(1)
image = torch.randn(B, 3, *cfg.image_size, device='cuda', generator=g)
audio = (torch.randn(B, 1, *cfg.audio_size, device='cuda', generator=g) * 2 - 3).clamp(-7, 4)
Li, La = cfg.image_grid[0] * cfg.image_grid[1], cfg.audio_grid[0] * cfg.audio_grid[1]
noise_i, noise_a = torch.rand(B, Li, device='cuda', generator=g), torch.rand(B, La, device='cuda', generator=g)
one = torch.ones((), device='cuda')


def fwd_bwd():
    outs, tape, _ = bridge.avmae_fwd(model, image, audio, noise_i, noise_a)
    bridge.avmae_bwd(model, tape, one, one)
    return outs[0], outs[1]


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
    li, la = fwd_bwd()
    graph.capture_end()
torch.cuda.current_stream().wait_stream(cap)
torch.cuda.synchronize()

ref, events = None, []
offs = list(zip(opt.flat.params, opt.flat.offsets))
names = {id(p): n for n, p in model.named_parameters()}
for it in range(replays):
    graph.replay()
    torch.cuda.synchronize()
    gcur = opt.flat.flat_g.clone()
    if ref is None:
        ref, lref = gcur, (float(li), float(la))
        continue
    if (float(li), float(la)) != lref:
        events.append((it, 'LOSS', abs(float(li) - lref[0]) + abs(float(la) - lref[1])))
    d = (gcur - ref).abs()
    if float(d.max()) > 1e-5 * float(ref.abs().max()):
        for p, o in offs:
            n = p.numel()
            rn = float(ref[o:o + n].norm())
            dd = float((gcur[o:o + n] - ref[o:o + n]).norm())
            if dd > 1e-5 * rn + 1e-9:
                events.append((it, names[id(p)], dd / (rn + 1e-20)))
# the captured graph against an eager pass over the same inputs (same kernels, launches serialised by the host)
gref = ref.clone()
opt.flat.zero_grad()
le = fwd_bwd()
torch.cuda.synchronize()
ge = opt.flat.flat_g
worst = max(float((ge[o:o + p.numel()] - gref[o:o + p.numel()]).norm() / (gref[o:o + p.numel()].norm() + 1e-20)) for p, o in offs)
print(f'graph vs eager: loss {lref} vs {(float(le[0]), float(le[1]))}, worst per-tensor gradient rel diff {worst:.2e}')
print(f'{name} B={B}: {replays} replays, {len(events)} deviation events')
seen = {}
for it, n, d in events:
    seen.setdefault(n, []).append((it, d))
import re
groups = {}
for n, ev in seen.items():
    groups.setdefault(re.sub(r'\.(weight|bias)$', '', n).rsplit('.', 2)[0] if n != 'LOSS' else 'LOSS', []).append((len(ev), max(e[1] for e in ev)))
print('   by module: ' + '; '.join(f'{k}: {len(v)} tensors, worst {max(x[1] for x in v):.1e}' for k, v in sorted(groups.items())))
for n, ev in sorted(seen.items(), key=lambda kv: kv[0])[:int(os.environ.get('TOP', 8))]:
    print(f'   {n:60s} {len(ev):4d} times, worst {max(e[1] for e in ev):.2e} (replays {[e[0] for e in ev][:6]})')
