#!/usr/bin/env python3
"""Grouped weight-gradient launches of the step (profiles/r02_step_launch_mix.json 'tn': one entry per launch = list of
(Mc, N, K)) timed alone: python tools/tn_group_bench.py [launch indices...]   (env DAV_TN256=0/1 selects the kernel)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
mix = json.load(open(os.path.join(ROOT, 'profiles', 'r02_step_launch_mix.json')))['tn']
which = [int(a) for a in sys.argv[1:]] or [0, 1, len(mix) - 1]
overwrite = os.environ.get('TN_BENCH_OVERWRITE', '0') == '1'
for li in which:
    shapes = mix[li][0] if isinstance(mix[li][0][0], list) else mix[li]
    probs = []
    for (Mc, N, K) in shapes:
        A = torch.randn(Mc, N, device=dev).bfloat16()
        B = torch.randn(Mc, K, device=dev).bfloat16()
        C = torch.zeros(N, K, device=dev)
        d = dict(A=A, B=B, Mc=Mc, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, a_rowmap=None, b_rowmap=None, bias_grad=None)
        if overwrite:
            d['overwrite'] = True
        probs.append(d)
    fl = sum(2.0 * Mc * N * K for (Mc, N, K) in shapes)
    n_l = (len(probs) + 39) // 40
    chunks = [probs[i::n_l] for i in range(n_l)]

    def run():
        for c in chunks:
            ops.gemm_tn_grouped(c)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    print(f'launch {li}: {len(shapes)} problems, {fl / 1e9:.1f} GFLOP: {us:9.1f} us  {fl / us / 1e6:7.0f} TF  (DAV_TN256={os.environ.get("DAV_TN256", "1")}, overwrite={overwrite})', flush=True)
