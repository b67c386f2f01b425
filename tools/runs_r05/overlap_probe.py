#!/usr/bin/env python3
"""Does the optimizer pass hide under the gang-scheduled weight-gradient launch?  (dependencies ignored: timing only)
12 encoder layers' weight gradients (one gang launch) and dav_adamw_flat over 320 M parameters: one after the other vs on two streams."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: F401  (DAV_BENCH_LIB: an experiment build with DAV_TN_GANG_WGS = persistent workgroups of the gang launch)
from deepavfusion_amd import ops
dev = torch.device('cuda')
mix = json.load(open(os.path.join(ROOT, 'profiles', 'r02_step_launch_mix.json')))['tn']
launches = [(l[0] if isinstance(l[0][0], list) else l) for l in mix]
probs = []
for i, shapes in enumerate(launches[2:]):
    for (Mc, N, K) in shapes:
        probs.append(dict(A=torch.randn(Mc, N, device=dev).bfloat16(), B=torch.randn(Mc, K, device=dev).bfloat16(), Mc=Mc, N=N, K=K,
                          C=torch.zeros(N, K, device=dev), lda=N, ldb=K, ldc=K, bias_grad=None, overwrite=True))
n = 320 * 1024 * 1024
p, g, m, v = (torch.zeros(n, device=dev) for _ in range(4))
pb = torch.empty(n, device=dev, dtype=torch.bfloat16)
seg_end = torch.tensor([n], dtype=torch.int64, device=dev)
hyper = torch.tensor([[1e-4, 0.05]], device=dev)
bc = torch.ones(2, device=dev)
ss = torch.zeros(1, device=dev)
s2 = torch.cuda.Stream()

def adam():
    ops.adamw_flat(p, g, m, v, pb, seg_end, hyper, 1, 0.9, 0.95, 1e-8, bc, 1.0, sumsq_out=ss, zero_grad=True)

def seq():
    ops.gemm_tn_gang(probs); adam()

def par():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    ops.gemm_tn_gang(probs)
    with torch.cuda.stream(s2):
        adam()
    cur.wait_stream(s2)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for _ in range(2):
    print(f'gang alone {timed(lambda: ops.gemm_tn_gang(probs)):8.1f} us | AdamW alone {timed(adam):8.1f} us | one after the other {timed(seq):8.1f} us | two streams {timed(par):8.1f} us', flush=True)
