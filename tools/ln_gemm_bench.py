#!/usr/bin/env python3
"""LayerNorm-folded GEMMs against the plain forms on the step's shapes, alone on the GPU, rotating operand sets (cold operands).
  producer: fp32 C = A.W^T + b + res            vs  the same + bf16 twin + row-statistics partials (dav_gemm_nt_ln_bf16 producer side)
  consumer: bf16 C = LN(x).W^T + b (LN kernel output as A)  vs  raw twin x gamma-folded W with the statistics applied in the epilogue
DAV_BENCH_LIB=<old .so>: plain forms only (what the presence of the LayerNorm code costs a GEMM that does not use it).
Usage: python tools/ln_gemm_bench.py [cfg]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools._libsel  # noqa: E402,F401
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
OLD = bool(os.environ.get('DAV_BENCH_LIB'))
ROT = 6
PROD = [(3136, 768, 768), (4096, 768, 768), (3136, 768, 3072), (4096, 768, 3072), (22528, 512, 512), (22528, 512, 2048), (14592, 512, 2048), (1024, 768, 768)]
CONS = [(4160, 2304, 768, 0), (5120, 2304, 768, 0), (3136, 3072, 768, 1), (4096, 3072, 768, 1), (22528, 1536, 512, 0), (22528, 2048, 512, 1),
        (14592, 1536, 512, 0), (14592, 2048, 512, 1), (3136, 1536, 768, 0), (1024, 768, 768, 1)]


def timeit(fn, reps):
    for _ in range(ROT):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    v = cfg << 4
    print(f'# library: {"OLD " + os.environ["DAV_BENCH_LIB"] if OLD else "tree"}   cfg {cfg}   us per launch, {ROT} operand sets in turn')
    print('producer  M x N x K          plain      +twin+stats')
    for (M, N, K) in PROD:
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(ROT)]
        Ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(ROT)]
        bias = torch.randn(N, device=dev)
        Rs = [torch.randn(M, N, device=dev) for _ in range(ROT)]
        Cs = [torch.empty(M, N, device=dev) for _ in range(ROT)]
        Ts = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]
        Ss = [torch.empty(M, N // 64, 2, device=dev) for _ in range(ROT)]
        it = [0]

        def plain():
            i = it[0] % ROT; it[0] += 1
            ops.gemm_nt(As[i], Ws[i], M, N, K, bias=bias, res=Rs[i], ldres=N, C_out=Cs[i], variant=v)

        def prod():
            i = it[0] % ROT; it[0] += 1
            ops.gemm_nt_ln(As[i], Ws[i], M, N, K, prod=dict(stats_out=Ss[i], twin_out=Ts[i], ld_twin=N), bias=bias, res=Rs[i], ldres=N, C_out=Cs[i], variant=v)
        a = timeit(plain, 4 * ROT)
        b = float('nan') if OLD else timeit(prod, 4 * ROT)
        print(f'{M:>7}x{N:>5}x{K:>5}  {a:9.1f}  {b:9.1f}  {b / a:6.3f}', flush=True)
    print('consumer  M x N x K  act     plain      folded LN')
    for (M, N, K, act) in CONS:
        As = [torch.randn(M, K, device=dev).bfloat16() for _ in range(ROT)]
        Ws = [(torch.randn(N, K, device=dev) * 0.05).bfloat16() for _ in range(ROT)]
        bias, c = torch.randn(N, device=dev), torch.randn(N, device=dev)
        Ss = [torch.rand(M, K // 64, 2, device=dev) + 1 for _ in range(ROT)]
        Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]
        Zs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(ROT)]
        it = [0]

        def plain():
            i = it[0] % ROT; it[0] += 1
            ops.gemm_nt(As[i], Ws[i], M, N, K, bias=bias, act=act, C_out=Cs[i], c_bf16=True, C2=Zs[i] if act else None, ldc2=N, c2_mode=4 if act else 0, variant=v)

        def cons():
            i = it[0] % ROT; it[0] += 1
            ops.gemm_nt_ln(As[i], Ws[i], M, N, K, ln=dict(stats=Ss[i], ln_c=c, eps=1e-6), bias=bias, act=act, C_out=Cs[i], c_bf16=True,
                           C2=Zs[i] if act else None, ldc2=N, c2_mode=4 if act else 0, variant=v)
        a = timeit(plain, 4 * ROT)
        b = float('nan') if OLD else timeit(cons, 4 * ROT)
        print(f'{M:>7}x{N:>5}x{K:>5}  {act}  {a:9.1f}  {b:9.1f}  {b / a:6.3f}', flush=True)


if __name__ == '__main__':
    main()
