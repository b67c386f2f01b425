#!/usr/bin/env python3
"""Run one GEMM shape/variant a few times (for rocprofv3 --pmc runs). Usage: gemm_one.py nt|tn M N K variant [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepavfusion_amd import ops  # noqa: E402

kind, M, N, K, v = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device('cuda')
if kind == 'nt':
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(reps):
        ops.gemm_nt(A, B, M, N, K, C_out=C, c_bf16=True, variant=v)
else:
    A = torch.randn(M, N, device=dev).bfloat16(); B = (torch.randn(M, K, device=dev) * 0.05).bfloat16()
    C = torch.zeros(N, K, device=dev)
    for _ in range(reps):
        ops.gemm_tn(A, B, M, N, K, C, beta=1, variant=v)
torch.cuda.synchronize()
