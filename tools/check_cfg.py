#!/usr/bin/env python3
"""Numerical check of an explicit NT tile configuration against torch (fp32 matmul of the bf16 operands), NT and b_kn modes.
Usage: check_cfg.py CFG [M N K]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import ops  # noqa: E402

cfg = int(sys.argv[1])
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) >= 5 else (1000, 512, 512)
dev = 'cuda'
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).bfloat16()
for b_kn in (0, 1):
    W = (torch.randn(K, N, device=dev) * 0.05).bfloat16() if b_kn else (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(ldb=N, variant=(1 << 12) | (cfg << 4)) if b_kn else dict(variant=cfg << 4)
    ops.gemm_nt(A, W, M, N, K, C_out=C, c_bf16=True, **kw)
    ref = A.float() @ (W.float() if b_kn else W.float().t())
    err = float((C.float() - ref).norm() / ref.norm())
    print(f'cfg {cfg} b_kn={b_kn} {M}x{N}x{K}: rel err {err:.3e}', 'OK' if err < 5e-3 else 'FAIL')
