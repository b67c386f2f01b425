#!/usr/bin/env python3
"""Reference point only (NOT on the product path): what the vendor library (hipBLASLt behind torch.matmul) reaches on the
step's GEMM shapes, bf16 in / bf16 out, no epilogue — to judge how far the hand-written kernels are from what is known
to be achievable on this chip for these shapes."""
import torch

dev = 'cuda'
for (M, N, K) in [(5184, 2304, 768), (6080, 2304, 768), (4032, 3072, 768), (5184, 768, 2304), (14592, 1536, 512), (11264, 2304, 768), (7168, 3072, 768), (7168, 768, 3072), (7168, 768, 768), (22528, 2048, 512), (37120, 1536, 512), (4096, 4096, 4096)]:
    sets = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.05).bfloat16()) for _ in range(4)]
    outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(4)]
    it = [0]

    def fn():
        A, W = sets[it[0] % 4]
        torch.matmul(A, W.t(), out=outs[it[0] % 4])
        it[0] += 1
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'{M}x{N}x{K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF  (hipBLASLt via torch.matmul, bf16, no epilogue)')

# weight-gradient shapes (C[N, K] = A[Mc, N]^T . B[Mc, K]): what the vendor library reaches on single problems of the grouped launches
print('--- TN (weight gradient) shapes')
for (Mc, N, K) in [(22528, 2048, 512), (22528, 512, 2048), (22528, 1536, 512), (6080, 2304, 768), (4032, 768, 3072), (3136, 3072, 768), (5184, 768, 768), (4096, 4096, 4096), (8192, 4096, 4096)]:
    sets = [(torch.randn(Mc, N, device=dev).bfloat16(), torch.randn(Mc, K, device=dev).bfloat16()) for _ in range(4)]
    outs = [torch.empty(N, K, device=dev, dtype=torch.bfloat16) for _ in range(4)]
    it = [0]

    def fn():
        A, B = sets[it[0] % 4]
        torch.matmul(A.t(), B, out=outs[it[0] % 4])
        it[0] += 1
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f'Mc {Mc} -> {N}x{K}: {us:8.1f} us  {2.0 * Mc * N * K / us / 1e6:7.0f} TF  (hipBLASLt via torch.matmul(A.t(), B), bf16 out)')
