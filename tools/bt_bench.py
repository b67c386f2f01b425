#!/usr/bin/env python3
"""Input-gradient (b_kn: B given as W[N_out, K_in], read through transposing LDS reads) and forward NT GEMMs of the step's big
Linears, graph-timed alone.  DAV_BENCH_LIB=<path of another libdavfusion_hip.so> times an older build with the same script
(the round-4 library: tools/runs_r05/lib_r04/, built from git by tools/runs_r05/build_r04_lib.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import _libsel  # noqa: E402,F401
from deepavfusion_amd import _lib  # noqa: E402
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
# (rows, out features, in features) of the Linears: forward = [rows, in] x W[out, in]^T, dgrad = [rows, out] x W[out, in]
LINEARS = [(5184, 2304, 768), (6080, 2304, 768), (3136, 3072, 768), (4032, 3072, 768), (3136, 768, 3072), (4032, 768, 3072),
           (3136, 768, 768), (14592, 1536, 512), (22528, 1536, 512), (22528, 2048, 512), (22528, 512, 2048), (14592, 512, 512),
           (7168, 768, 3072), (7168, 3072, 768), (11264, 2304, 768)]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


print(f'lib: {_lib.LIB_PATH}')
for (M, O, I) in LINEARS:
    x = torch.randn(M, I, device=dev).bfloat16()
    dy = torch.randn(M, O, device=dev).bfloat16()
    W = (torch.randn(O, I, device=dev) * 0.05).bfloat16()
    y = torch.empty(M, O, device=dev, dtype=torch.bfloat16)
    dx = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    fwd = lambda: ops.gemm_nt(x, W, M, O, I, C_out=y, c_bf16=True)
    bwd = lambda: ops.gemm_nt(dy, W, M, I, O, lda=O, ldb=I, C_out=dx, ldc=I, c_bf16=True, variant=1 << 12)
    bwd(); torch.cuda.synchronize()
    ref = dy.float() @ W.float()
    err = float((dx.float() - ref).norm() / ref.norm())
    tf, tb = timed(fwd), timed(bwd)
    fl = 2.0 * M * O * I
    print(f'rows {M:6d} out {O:5d} in {I:5d}: forward {tf:7.1f} us {fl / tf / 1e6:6.0f} TF | dgrad (b_kn) {tb:7.1f} us {fl / tb / 1e6:6.0f} TF  err {err:.1e}', flush=True)
