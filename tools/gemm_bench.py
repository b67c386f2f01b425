#!/usr/bin/env python3
"""Per-shape timing of the GEMM kernels on the shapes of the ViT-B pre-training step (B=64).
Usage: python tools/gemm_bench.py [nt|tn] [variants...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
NT_SHAPES = [  # (M, N, K, count per step)  forward + dgrad shapes
    (5184, 2304, 768, 12), (6080, 2304, 768, 12), (3136, 768, 768, 12), (4032, 768, 768, 12),
    (3136, 3072, 768, 12 + 12), (4032, 3072, 768, 12 + 12), (3136, 768, 3072, 12 + 12), (4032, 768, 3072, 12 + 12),
    (5184, 768, 2304, 12), (6080, 768, 2304, 12),
    (14592, 1536, 512, 8), (22528, 1536, 512, 8), (14592, 512, 512, 16), (22528, 512, 512, 16),
    (14592, 2048, 512, 16), (22528, 2048, 512, 16), (14592, 512, 2048, 16), (22528, 512, 2048, 16),
    (14592, 512, 1536, 8), (22528, 512, 1536, 8),
    (3136, 1536, 768, 12), (4032, 1536, 768, 12), (2048, 768, 768, 48), (512, 768, 768, 100), (1024, 192, 768, 24),
    (12544, 768, 512, 1), (20480, 256, 512, 1), (3136, 768, 768, 1), (4032, 768, 256, 1),
]
TN_SHAPES = [  # (Mc, N, K, count)
    (5184, 2304, 768, 12), (6080, 2304, 768, 12), (3136, 768, 768, 12), (4032, 768, 768, 12),
    (3136, 3072, 768, 12), (4032, 3072, 768, 12), (3136, 768, 3072, 12), (4032, 768, 3072, 12),
    (14592, 1536, 512, 8), (22528, 1536, 512, 8), (14592, 512, 512, 8), (22528, 512, 512, 8),
    (14592, 2048, 512, 8), (22528, 2048, 512, 8), (14592, 512, 2048, 8), (22528, 512, 2048, 8),
    (3136, 1536, 768, 12), (4032, 1536, 768, 12), (2048, 768, 768, 24), (512, 768, 768, 100), (1024, 192, 768, 12),
]


def timeit(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'nt'
    variants = [int(v) for v in sys.argv[2:]] or [0]
    shapes = NT_SHAPES if kind in ('nt', 'kn') else TN_SHAPES
    tot = {v: 0.0 for v in variants}
    totfl = 0.0
    print(f'{"shape":>22} {"cnt":>4} ' + ' '.join(f'{"v%d us" % v:>9} {"TF":>6}' for v in variants))
    for (M, N, K, cnt) in shapes:
        A = torch.randn(M, K if kind in ('nt', 'kn') else N, device=dev).bfloat16()
        Bm = (torch.randn(N if kind == 'nt' else (K if kind == 'kn' else M), K if kind != 'kn' else N, device=dev) * 0.05).bfloat16()
        if os.environ.get('GEMM_BENCH_UNIFORM') == '1':      # full-range uniform operands (what tools/micro/gemm256.hip times): the GEMMs are power-bound, the fill matters
            A = (torch.rand_like(A.float()) * 2 - 1).bfloat16()
            Bm = (torch.rand_like(Bm.float()) * 2 - 1).bfloat16()
        rot = int(os.environ.get('GEMM_BENCH_ROTATE', '0'))      # > 0: that many operand / output sets used in turn (cold operands, as in the step)
        if rot > 0 and kind in ('nt', 'kn'):
            As = [A.clone() for _ in range(rot)]
            Bs = [Bm.clone() for _ in range(rot)]
            Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(rot)]
        row = f'{M:>7}x{N:>5}x{K:>5} {cnt:>4} '
        fl = 2.0 * M * N * K
        totfl += fl * cnt
        for v in variants:
            if rot > 0 and kind in ('nt', 'kn'):
                it = [0]

                def fn():
                    i = it[0] % rot
                    it[0] += 1
                    if kind == 'nt':
                        ops.gemm_nt(As[i], Bs[i], M, N, K, C_out=Cs[i], c_bf16=True, variant=v)
                    else:
                        ops.gemm_nt(As[i], Bs[i], M, N, K, ldb=N, C_out=Cs[i], c_bf16=True, variant=v | (1 << 12))
                us = timeit(fn, reps=2 * rot)
            elif kind == 'nt':
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                us = timeit(lambda: ops.gemm_nt(A, Bm, M, N, K, C_out=C, c_bf16=True, variant=v))
            elif kind == 'kn':
                C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                us = timeit(lambda: ops.gemm_nt(A, Bm, M, N, K, ldb=N, C_out=C, c_bf16=True, variant=v | (1 << 12)))
            else:
                C = torch.zeros(N, K, device=dev)
                us = timeit(lambda: ops.gemm_tn(A, Bm, M, N, K, C, beta=1, variant=v))
            tot[v] += us * cnt
            row += f'{us:9.1f} {fl / us / 1e6:6.0f} '
        print(row, flush=True)
    print('per-step total (ms): ' + '  '.join(f'v{v}: {tot[v] / 1e3:.2f} ms ({totfl / tot[v] / 1e6:.0f} TF avg)' for v in variants))


if __name__ == '__main__':
    main()
