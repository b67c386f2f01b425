#!/usr/bin/env python3
"""Run every HIP kernel against a torch fp32 reference on the GPU box and print the errors.
Diagnostic companion of tests/ (never aborts on the first failure).  Usage: python tests/gpu_selfcheck.py [filter]"""
import math
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops  # noqa: E402

dev = torch.device('cuda')
BF16, F32 = torch.bfloat16, torch.float32
RESULTS = []


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def report(name, err, tol):
    ok = err <= tol and math.isfinite(err)
    RESULTS.append((name, err, tol, ok))
    print(f'{"PASS" if ok else "FAIL"} {name}: err={err:.3e} tol={tol:.1e}', flush=True)


def check(fn):
    def run():
        try:
            fn()
            torch.cuda.synchronize()
        except Exception:
            traceback.print_exc()
            RESULTS.append((fn.__name__, float('nan'), 0, False))
            print(f'FAIL {fn.__name__}: exception', flush=True)
    run.__name__ = fn.__name__
    return run


def rnd(*shape, dtype=F32, scale=1.0, seed=None):
    g = torch.Generator(device='cpu')
    g.manual_seed(seed if seed is not None else (hash(shape) & 0xffff))
    return (torch.randn(*shape, generator=g) * scale).to(device=dev, dtype=dtype)


@check
def gemm_nt():
    for (M, N, K) in [(162, 192, 64), (5184, 2304, 768), (130, 768, 3072), (64, 48, 192), (512, 512, 48), (4032, 768, 256), (37, 100, 136)]:
        for variant in (0, 1, 2, 3):
            A, Bm = rnd(M, K, dtype=BF16, seed=1), rnd(N, K, dtype=BF16, scale=0.05, seed=2)
            bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
            ref = A.float() @ Bm.float().t() + bias
            C = torch.empty(M, N, device=dev)
            ops.gemm_nt(A, Bm, M, N, K, bias=bias, C_out=C, variant=variant)
            report(f'gemm_nt {M}x{N}x{K} v{variant} bias', rel(C, ref), 1e-4)
            # gelu + preact twin, bf16 out
            U = torch.empty(M, N, device=dev, dtype=BF16)
            Z = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Bm, M, N, K, bias=bias, act=1, C_out=U, c_bf16=True, C2=Z, ldc2=N, c2_mode=1, variant=variant)
            report(f'gemm_nt {M}x{N}x{K} v{variant} gelu', rel(U, torch.nn.functional.gelu(ref)), 6e-3)
            report(f'gemm_nt {M}x{N}x{K} v{variant} preact', rel(Z, ref), 6e-3)
            # gelu fp32 out (the erf approximation itself: |err| <= 1.5e-7) + GELU' twin; then the multiply-only backward form
            U32 = torch.empty(M, N, device=dev)
            D = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Bm, M, N, K, bias=bias, act=1, C_out=U32, C2=D, ldc2=N, c2_mode=4, variant=variant)
            xg = ref.clone().requires_grad_(True)
            torch.nn.functional.gelu(xg).sum().backward()
            report(f'gemm_nt {M}x{N}x{K} v{variant} gelu32', float((U32 - torch.nn.functional.gelu(ref)).abs().max()), 2e-5)
            report(f"gemm_nt {M}x{N}x{K} v{variant} gelu'twin", float((D.float() - xg.grad).abs().max()), 5e-3)
            G = torch.empty(M, N, device=dev)
            ops.gemm_nt(A, Bm, M, N, K, act=3, aux=D, ldaux=N, C_out=G, variant=variant)
            report(f'gemm_nt {M}x{N}x{K} v{variant} act3', rel(G, (ref - bias) * D.float()), 1e-4)
            # residual + beta accumulate + bf16 twin of the final value
            C = torch.full((M, N), 0.5, device=dev)
            T = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Bm, M, N, K, res=res, ldres=N, C_out=C, beta=1, C2=T, ldc2=N, c2_mode=3, variant=variant)
            report(f'gemm_nt {M}x{N}x{K} v{variant} res+beta', rel(C, ref - bias + res + 0.5), 1e-4)
            report(f'gemm_nt {M}x{N}x{K} v{variant} twin', rel(T, ref - bias + res + 0.5), 6e-3)
    # every second-generation tile configuration (explicit cfg in variant bits 4-11)
    for (M, N, K) in [(5184, 2304, 768), (300, 200, 128), (4032, 768, 3072)]:
        A, Bm = rnd(M, K, dtype=BF16, seed=1), rnd(N, K, dtype=BF16, scale=0.05, seed=2)
        bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
        ref = A.float() @ Bm.float().t() + bias + res
        experimental = (1, 13, 15, 16, 17, 18, 19, 20, 21, 22) if ops._lib.load().dav_build_flags() & 1 else ()      # make EXPERIMENTAL=1
        for cfg in (3, 5, 7, 8, 60) + experimental:
            C = torch.empty(M, N, device=dev)
            ops.gemm_nt(A, Bm, M, N, K, bias=bias, res=res, ldres=N, C_out=C, variant=cfg << 4)
            report(f'gemm_nt {M}x{N}x{K} cfg{cfg}', rel(C, ref), 1e-4)
    # the big-tile / deep-ring configurations of round 2 (43 / 45 = 256x128, 44 / 46 = 128x256 on 32-deep rings, 7 = 64x64 on
    # four stages) with every epilogue kind the step gives them: staged bf16 output, the SPLIT staged epilogue with a GELU' twin,
    # multiply by aux, fp32 + residual; forward form and b_kn (32-deep stages in b_kn mode are new)
    for (M, N, K) in [(2100, 1024, 512), (1000, 768, 192), (300, 512, 64)]:
        A = rnd(M, K, dtype=BF16, seed=31)
        W_nk, W_kn = rnd(N, K, dtype=BF16, scale=0.05, seed=32), rnd(K, N, dtype=BF16, scale=0.05, seed=33)
        bias, res, aux = rnd(N, seed=34), rnd(M, N, seed=35), rnd(M, N, dtype=BF16, seed=36)
        for cfg in (43, 44, 45, 46, 7, 8, 3):
            for bt in (0, 1):
                Wm, kw = (W_kn, dict(ldb=N)) if bt else (W_nk, {})
                base = A.float() @ (W_kn.float() if bt else W_nk.float().t())
                var = (cfg << 4) | (bt << 12)
                tag = f'gemm_nt {M}x{N}x{K} cfg{cfg} b_kn{bt}'
                U, D = torch.empty(M, N, device=dev, dtype=BF16), torch.empty(M, N, device=dev, dtype=BF16)
                ops.gemm_nt(A, Wm, M, N, K, bias=bias, act=1, C_out=U, c_bf16=True, C2=D, ldc2=N, c2_mode=4, variant=var, **kw)
                xg = (base + bias).clone().requires_grad_(True)
                torch.nn.functional.gelu(xg).sum().backward()
                report(tag + ' gelu (staged)', rel(U, torch.nn.functional.gelu(base + bias)), 6e-3)
                report(tag + " gelu' twin (staged)", float((D.float() - xg.grad).abs().max()), 5e-3)
                G = torch.empty(M, N, device=dev, dtype=BF16)
                ops.gemm_nt(A, Wm, M, N, K, act=3, aux=aux, ldaux=N, C_out=G, c_bf16=True, variant=var, **kw)
                report(tag + ' act3 bf16', rel(G, base * aux.float()), 6e-3)
                C = torch.empty(M, N, device=dev)
                ops.gemm_nt(A, Wm, M, N, K, bias=bias, res=res, ldres=N, C_out=C, variant=var, **kw)
                report(tag + ' fp32 + res', rel(C, base + bias + res), 1e-4)
    # 256 x 256 tiles (configuration 60, csrc/gemm_nt256.h; K % 128 == 0): every epilogue kind in both operand modes, ragged M / N
    # edges (tiles of 256 on 2100 / 1000 / 300 rows, 200 .. 2304 columns), a K of a single pair of K-tiles, row maps, grouped form
    for (M, N, K) in [(2100, 1024, 512), (1000, 768, 256), (300, 512, 128), (5184, 2304, 768), (517, 200, 384)]:
        A = rnd(M, K, dtype=BF16, seed=41)
        W_nk, W_kn = rnd(N, K, dtype=BF16, scale=0.05, seed=42), rnd(K, N, dtype=BF16, scale=0.05, seed=43)
        bias, res, aux = rnd(N, seed=44), rnd(M, N, seed=45), rnd(M, N, dtype=BF16, seed=46)
        for bt in (0, 1):
            Wm, kw = (W_kn, dict(ldb=N)) if bt else (W_nk, {})
            base = A.float() @ (W_kn.float() if bt else W_nk.float().t())
            var = (60 << 4) | (bt << 12)
            tag = f'gemm_nt {M}x{N}x{K} cfg60 b_kn{bt}'
            U, D = torch.empty(M, N, device=dev, dtype=BF16), torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Wm, M, N, K, bias=bias, act=1, C_out=U, c_bf16=True, C2=D, ldc2=N, c2_mode=4, variant=var, **kw)
            xg = (base + bias).clone().requires_grad_(True)
            torch.nn.functional.gelu(xg).sum().backward()
            report(tag + ' gelu', rel(U, torch.nn.functional.gelu(base + bias)), 6e-3)
            report(tag + " gelu' twin", float((D.float() - xg.grad).abs().max()), 5e-3)
            Z = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Wm, M, N, K, bias=bias, act=1, C_out=U, c_bf16=True, C2=Z, ldc2=N, c2_mode=1, variant=var, **kw)
            report(tag + ' preact twin', rel(Z, base + bias), 6e-3)
            G = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Wm, M, N, K, act=3, aux=aux, ldaux=N, C_out=G, c_bf16=True, variant=var, **kw)
            report(tag + ' act3 bf16', rel(G, base * aux.float()), 6e-3)
            xa = aux.float().requires_grad_(True)
            torch.nn.functional.gelu(xa).sum().backward()
            ops.gemm_nt(A, Wm, M, N, K, act=2, aux=aux, ldaux=N, C_out=G, c_bf16=True, alpha=0.5, variant=var, **kw)
            report(tag + ' act2 bf16 alpha', rel(G, 0.5 * base * xa.grad), 6e-3)
            C = torch.full((M, N), 0.25, device=dev)
            T = torch.empty(M, N, device=dev, dtype=BF16)
            ops.gemm_nt(A, Wm, M, N, K, bias=bias, res=res, ldres=N, C_out=C, beta=1, C2=T, ldc2=N, c2_mode=3, variant=var, **kw)
            report(tag + ' fp32 + res + beta', rel(C, base + bias + res + 0.25), 1e-4)
            report(tag + ' final twin', rel(T, base + bias + res + 0.25), 6e-3)
            P = torch.full((M, N), 7.0, device=dev, dtype=BF16)
            ops.gemm_nt(A, Wm, M, N, K, C_out=P, c_bf16=True, variant=var, **kw)
            report(tag + ' plain bf16', rel(P, base), 6e-3)
    # (row maps: A rows 4.. of every batch, C rows 1..)
    Bsz, rpb, tot, N, K = 5, 70, 90, 512, 256
    M = Bsz * rpb
    Afull, Bm = rnd(Bsz * tot, K, dtype=BF16, seed=47), rnd(N, K, dtype=BF16, scale=0.1, seed=48)
    Asub = Afull.view(Bsz, tot, K)[:, 4:4 + rpb, :].reshape(M, K)
    Cfull = torch.zeros(Bsz * tot, N, device=dev, dtype=BF16)
    ops.gemm_nt(Afull, Bm, M, N, K, a_rowmap=(rpb, tot, 4), C_out=Cfull, c_bf16=True, c_rowmap=(rpb, tot, 1), variant=60 << 4)
    ref_full = torch.zeros(Bsz, tot, N, device=dev)
    ref_full[:, 1:1 + rpb] = (Asub.float() @ Bm.float().t()).view(Bsz, rpb, N)
    report('gemm_nt cfg60 rowmaps', rel(Cfull, ref_full.view(-1, N)), 6e-3)
    # act 2 (multiply by gelu'(aux)) and row maps
    M, N, K = 3 * 7, 128, 64
    Bsz, rpb, tot = 3, 7, 11
    Afull = rnd(Bsz * tot, K, dtype=BF16, seed=5)
    Bm = rnd(N, K, dtype=BF16, scale=0.1, seed=6)
    aux = rnd(M, N, dtype=BF16, seed=7)
    Asub = Afull.view(Bsz, tot, K)[:, 4:, :].reshape(M, K)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    ref = (Asub.float() @ Bm.float().t()) * x.grad
    Cfull = torch.zeros(Bsz * tot, N, device=dev)
    resfull = rnd(Bsz * tot, N, seed=8)
    ops.gemm_nt(Afull, Bm, M, N, K, a_rowmap=(rpb, tot, 4), act=2, aux=aux, ldaux=N, res=resfull, ldres=N, res_rowmap=(rpb, tot, 2),
                C_out=Cfull, c_rowmap=(rpb, tot, 1))
    ref_full = torch.zeros_like(Cfull).view(Bsz, tot, N)
    ref_full[:, 1:8] = ref.view(Bsz, rpb, N) + resfull.view(Bsz, tot, N)[:, 2:9]
    report('gemm_nt rowmaps+act2', rel(Cfull, ref_full.view(-1, N)), 1e-4)
    rows = torch.randint(0, 5, (M,), device=dev, dtype=torch.int32)
    pos = rnd(5, N, seed=9)
    C = torch.empty(M, N, device=dev)
    ops.gemm_nt(Asub.contiguous(), Bm, M, N, K, res=pos, ldres=N, res_rows=rows, C_out=C)
    report('gemm_nt res_rows', rel(C, Asub.float() @ Bm.float().t() + pos[rows.long()]), 1e-4)
    # B given as [K, N] (dgrad reading W itself), incl. column offset / ldb and all tile configs
    for (M2, N2, K2) in [(300, 768, 192), (5184, 768, 2304), (2048, 136, 64), (4032, 3072, 768)]:
        A2 = rnd(M2, K2, dtype=BF16, seed=15)
        Wkn = rnd(K2, 2 * N2, dtype=BF16, scale=0.05, seed=16)
        for cfg in (0, 3, 8, 5):
            C = torch.empty(M2, N2, device=dev)
            ops.gemm_nt(A2, Wkn.view(-1)[N2:], M2, N2, K2, ldb=2 * N2, C_out=C, variant=(1 << 12) | (cfg << 4))
            report(f'gemm_nt b_kn {M2}x{N2}x{K2} cfg{cfg}', rel(C, A2.float() @ Wkn[:, N2:].float()), 1e-4)
    # B sub-matrix (column offset, ldb)
    Bw = rnd(N, 2 * K, dtype=BF16, scale=0.1, seed=10)
    C = torch.empty(M, N, device=dev)
    ops.gemm_nt(Asub.contiguous(), Bw.view(-1)[K:], M, N, K, ldb=2 * K, C_out=C)
    report('gemm_nt ldb/offset', rel(C, Asub.float() @ Bw[:, K:].float().t()), 1e-4)


@check
def gemm_tn():
    for (Mc, N, K) in [(162, 192, 64), (5184, 768, 768), (100, 48, 192), (4032, 256, 768), (77, 136, 200), (6080, 3072, 768), (3136, 192, 768), (128, 72, 40)]:
        for variant in (0, 1, 2, 16, 32):
            A, Bm = rnd(Mc, N, dtype=BF16, seed=11), rnd(Mc, K, dtype=BF16, seed=12)
            ref = A.float().t() @ Bm.float()
            C = torch.zeros(N, K, device=dev)
            bg = torch.zeros(N, device=dev)
            ops.gemm_tn(A, Bm, Mc, N, K, C, beta=1, bias_grad=bg, variant=variant)
            report(f'gemm_tn {Mc}x{N}x{K} v{variant} acc', rel(C, ref), 2e-4)
            report(f'gemm_tn {Mc}x{N}x{K} v{variant} bias_grad', rel(bg, A.float().sum(0)), 2e-4)
            C = torch.full((N, K), 7.0, device=dev)
            ops.gemm_tn(A, Bm, Mc, N, K, C, beta=0, variant=variant)
            report(f'gemm_tn {Mc}x{N}x{K} v{variant} store', rel(C, ref), 2e-4)
    # grouped launch of several problems (deferred wgrads of a layer), incl. bias grads and the split-K path
    probs, refs = [], []
    for i, (Mc, N, K) in enumerate([(3136, 768, 768), (3136, 2304, 768), (4032, 768, 3072), (512, 192, 768), (2048, 72, 136), (6080, 3072, 768)]):
        A, Bm = rnd(Mc, N, dtype=BF16, seed=60 + i), rnd(Mc, K, dtype=BF16, seed=70 + i)
        C = torch.full((N, K), 0.25, device=dev)
        bg = torch.full((N,), 0.5, device=dev) if i % 2 == 0 else None
        probs.append(dict(A=A, B=Bm, Mc=Mc, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, bias_grad=bg))
        refs.append((C, 0.25 + A.float().t() @ Bm.float(), bg, None if bg is None else 0.5 + A.float().sum(0)))
    ops.gemm_tn_grouped(probs)
    for i, (C, rc, bg, rb) in enumerate(refs):
        report(f'gemm_tn grouped #{i} C', rel(C, rc), 2e-4)
        if bg is not None:
            report(f'gemm_tn grouped #{i} bias', rel(bg, rb), 2e-4)
    ops.gemm_tn_grouped(probs[:2])        # small group -> split-K atomics path
    report('gemm_tn grouped split acc', rel(refs[0][0], 2 * refs[0][1] - 0.25), 2e-4)
    # written (not accumulated) tiles: DavTnProblem.flags bit 0 — old contents (NaN here) are ignored, the bias gradient still
    # accumulates, never split over the contraction (a small group would otherwise take the atomics path); mixed with an accumulating problem
    for group in (probs[:2], probs):
        for i, pr in enumerate(group):
            pr['overwrite'] = i != 1
            pr['C'].fill_(float('nan') if pr['overwrite'] else 0.25)
            if pr['bias_grad'] is not None:
                pr['bias_grad'].fill_(0.5)
        ops.gemm_tn_grouped(group)
        for i, (C, rc, bg, rb) in enumerate(refs[:len(group)]):
            report(f'gemm_tn grouped[{len(group)}] #{i} {"written" if i != 1 else "accumulated"}', rel(C, rc - (0.25 if i != 1 else 0.0)), 2e-4)
            if bg is not None:
                report(f'gemm_tn grouped[{len(group)}] #{i} bias', rel(bg, rb), 2e-4)
    for pr in probs:
        pr.pop('overwrite')
    # row maps + ldc sub-block
    Bsz, rpb, tot, N, K = 3, 5, 9, 64, 128
    Af, Bf = rnd(Bsz * tot, N, dtype=BF16, seed=13), rnd(Bsz * tot, K, dtype=BF16, seed=14)
    As = Af.view(Bsz, tot, N)[:, 2:7].reshape(-1, N)
    Bs = Bf.view(Bsz, tot, K)[:, 4:9].reshape(-1, K)
    Cw = torch.zeros(N, 2 * K, device=dev)
    ops.gemm_tn(Af, Bf, Bsz * rpb, N, K, Cw.view(-1)[K:], ldc=2 * K, a_rowmap=(rpb, tot, 2), b_rowmap=(rpb, tot, 4), beta=1)
    ref = torch.zeros_like(Cw)
    ref[:, K:] = As.float().t() @ Bs.float()
    report('gemm_tn rowmaps+ldc', rel(Cw, ref), 2e-4)


@check
def gemm_tn_gang():
    """dav_gemm_tn_gang_bf16: the queued weight gradients of several layers as one persistent launch of 256 x 256 tiles."""
    # odd K-tile counts (Mc = 64 * odd), N / K below, across and beyond one tile, ragged last tiles, a long contraction
    shapes = [(512, 768, 768), (3136, 192, 768), (4032, 2304, 768), (3136, 768, 3072), (640, 8, 264), (14592, 512, 1536), (2048, 520, 776),
              (64, 256, 256), (192, 1032, 40), (5184, 2304, 768), (6080, 3072, 1024),
              (2592, 1024, 1024), (3040, 1024, 3072), (1568, 4096, 1024), (40, 264, 520), (200, 8, 8), (63 * 8, 768, 192)]      # ragged: Mc % 64 != 0
    for mode in ('accumulate', 'written', 'mixed'):
        probs, refs = [], []
        for i, (Mc, N, K) in enumerate(shapes):
            A, Bm = rnd(Mc, N, dtype=BF16, seed=160 + i), rnd(Mc, K, dtype=BF16, seed=170 + i)
            ow = mode == 'written' or (mode == 'mixed' and i % 2 == 0)
            C = torch.full((N, K), float('nan') if ow else 0.25, device=dev)
            bg = torch.full((N,), 0.5, device=dev) if i % 3 != 1 else None
            probs.append(dict(A=A, B=Bm, Mc=Mc, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, bias_grad=bg, overwrite=ow))
            refs.append((C, (0.0 if ow else 0.25) + A.float().t() @ Bm.float(), bg, None if bg is None else 0.5 + A.float().sum(0)))
        ops.gemm_tn_gang(probs)
        for i, (C, rc, bg, rb) in enumerate(refs):
            report(f'gemm_tn_gang {mode} #{i} {shapes[i]} C', rel(C, rc), 2e-4)
            if bg is not None:
                report(f'gemm_tn_gang {mode} #{i} bias', rel(bg, rb), 2e-4)
    # bit-repeatable whoever draws which ticket: the same launch twice, written tiles
    Cs = []
    for _ in range(2):
        for pr in probs:
            pr['overwrite'] = True
            pr['C'] = torch.empty_like(pr['C'])
        ops.gemm_tn_gang(probs)
        torch.cuda.synchronize()
        Cs.append([pr['C'].clone() for pr in probs])
    report('gemm_tn_gang bit-repeatable', float(max((a != b).sum() for a, b in zip(*Cs))), 0.0)
    # row maps on both operands + a column block of a wider gradient (ldc), rows-per-batch below and above 64
    for (Bsz, rpb, tot, offa, offb, N, K) in [(16, 12, 20, 3, 8, 320, 136), (8, 81, 95, 14, 0, 768, 512), (64, 8, 40, 32, 0, 264, 768)]:
        Af, Bf = rnd(Bsz * tot, N, dtype=BF16, seed=113), rnd(Bsz * tot, K, dtype=BF16, seed=114)
        Mc = Bsz * rpb                    # (8 x 81 = 648 rows: a ragged contraction THROUGH a row map — ViT-L's towers at small batch)
        As = Af.view(Bsz, tot, N)[:, offa:offa + rpb].reshape(-1, N)
        Bs = Bf.view(Bsz, tot, K)[:, offb:offb + rpb].reshape(-1, K)
        Cw = torch.zeros(N, 2 * K, device=dev)
        ops.gemm_tn_gang([dict(A=Af, B=Bf, Mc=Mc, N=N, K=K, C=Cw.view(-1)[K:], lda=N, ldb=K, ldc=2 * K, a_rowmap=(rpb, tot, offa),
                               b_rowmap=(rpb, tot, offb), bias_grad=None)])
        ref = torch.zeros_like(Cw)
        ref[:, K:] = As.float().t() @ Bs.float()
        report(f'gemm_tn_gang rowmaps+ldc rpb={rpb}', rel(Cw, ref), 2e-4)
    # a few hundred problems in one launch (several table-writer launches)
    many, refs = [], []
    for i in range(150):
        Mc, N, K = 64 * (1 + i % 5), 8 * (1 + (7 * i) % 60), 8 * (1 + (11 * i) % 70)
        A, Bm = rnd(Mc, N, dtype=BF16, seed=300 + i), rnd(Mc, K, dtype=BF16, seed=500 + i)
        C = torch.zeros(N, K, device=dev)
        many.append(dict(A=A, B=Bm, Mc=Mc, N=N, K=K, C=C, lda=N, ldb=K, ldc=K, bias_grad=None, overwrite=bool(i & 1)))
        refs.append(A.float().t() @ Bm.float())
    ops.gemm_tn_gang(many)
    report('gemm_tn_gang 150 problems', max(rel(pr['C'], r) for pr, r in zip(many, refs)), 2e-4)
    # one very wide weight: 50 x 50 tiles = 104 gangs, more than one table-writer launch carries (its tickets span two), beside a small one
    A, Bm = rnd(64, 12800, dtype=BF16, seed=700), rnd(64, 12800, dtype=BF16, seed=701)
    A2, B2 = rnd(100, 264, dtype=BF16, seed=702), rnd(100, 40, dtype=BF16, seed=703)
    Cbig, C2 = torch.empty(12800, 12800, device=dev), torch.zeros(264, 40, device=dev)
    ops.gemm_tn_gang([dict(A=A, B=Bm, Mc=64, N=12800, K=12800, C=Cbig, lda=12800, ldb=12800, ldc=12800, bias_grad=None, overwrite=True),
                      dict(A=A2, B=B2, Mc=100, N=264, K=40, C=C2, lda=264, ldb=40, ldc=40, bias_grad=None)])
    report('gemm_tn_gang 12800 x 12800 weight (104 gangs)', rel(Cbig, A.float().t() @ Bm.float()), 2e-4)
    report('gemm_tn_gang ... and the problem after it', rel(C2, A2.float().t() @ B2.float()), 2e-4)
    del Cbig


def ref_attn(q, k, v, scale):
    s = (q @ k.transpose(-2, -1)) * scale
    return s.softmax(-1) @ v


@check
def patch_gather3d():
    for (B, C, T, H, W, pt) in [(2, 3, 4, 48, 32, 2), (1, 3, 8, 64, 64, 2), (2, 1, 3, 32, 48, 1)]:
        x = rnd(B, C, T, H, W, seed=91)
        gt, gh, gw = T // pt, H // 16, W // 16
        L = gt * gh * gw
        cols = x.view(B, C, gt, pt, gh, 16, gw, 16).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, L, C * pt * 256)
        A = torch.empty(B * L, C * pt * 256, device=dev, dtype=BF16)
        ops.patch_gather(x, None, L, A, pt)
        report(f'patch_gather3d all {B}x{C}x{T}x{H}x{W}', rel(A.view(B, L, -1), cols.to(BF16)), 1e-6)
        nk = max(1, L // 3)
        ids = torch.stack([torch.randperm(L, device=dev)[:nk] for _ in range(B)]).to(torch.int32)
        A2 = torch.empty(B * nk, C * pt * 256, device=dev, dtype=BF16)
        ops.patch_gather(x, ids, nk, A2, pt)
        ref = torch.gather(cols, 1, ids.long().unsqueeze(-1).expand(-1, -1, cols.shape[-1]))
        report(f'patch_gather3d kept {B}x{C}x{T}x{H}x{W}', rel(A2.view(B, nk, -1), ref.to(BF16)), 1e-6)


@check
def attention():
    for (B, H, Nq, Nk, dqk, dv, off) in [(2, 3, 49, 81, 64, 64, 32), (3, 2, 8, 49, 64, 64, 0), (2, 16, 228, 228, 32, 32, 0),
                                        (2, 2, 352, 352, 32, 32, 0), (2, 12, 16, 64, 16, 64, 0), (3, 2, 4, 6, 16, 64, 0),
                                        (2, 2, 5, 13, 64, 64, 8), (1, 1, 1, 1, 32, 32, 0), (2, 12, 63, 95, 64, 64, 32),
                                        # long sequences: keys / queries stream through LDS in chunks
                                        (2, 3, 784, 816, 64, 64, 32), (2, 2, 352, 352, 64, 64, 0), (1, 2, 100, 1000, 64, 64, 0),
                                        (1, 2, 1000, 40, 64, 64, 0), (2, 2, 1300, 1300, 32, 32, 0), (1, 3, 17, 530, 16, 64, 0),
                                        (1, 2, 257, 257, 64, 64, 0),
                                        # q/k/v head width 16 (token / dense_mmi fusion archs at attn_ratio 0.25)
                                        (2, 12, 32, 112, 16, 16, 0), (1, 3, 32, 3087, 16, 16, 0), (2, 2, 9, 20, 16, 16, 0)]:
        scale = 0.125 if dqk == 16 else dqk ** -0.5
        # fused layout when dqk == dv: buffer [B, Nk, 3, H, d]; queries are rows off.. of the same buffer
        fused = dqk == dv and Nq <= Nk
        if fused:
            buf = rnd(B, Nk, 3, H, dqk, dtype=BF16, seed=21)
            assert Nq + off == Nk or off == 0
            qo = off if Nq + off == Nk else 0
            qt = (buf, qo * 3 * H * dqk); kt = (buf, H * dqk); vt = (buf, 2 * H * dqk)
            strides = (Nk * 3 * H * dqk, 3 * H * dqk) * 3
            q = buf[:, qo:qo + Nq, 0].permute(0, 2, 1, 3).float()
            k = buf[:, :, 1].permute(0, 2, 1, 3).float()
            v = buf[:, :, 2].permute(0, 2, 1, 3).float()
        else:
            qb, kb, vb = rnd(B, Nq, H, dqk, dtype=BF16, seed=22), rnd(B, Nk, H, dqk, dtype=BF16, seed=23), rnd(B, Nk, H, dv, dtype=BF16, seed=24)
            qt, kt, vt = (qb, 0), (kb, 0), (vb, 0)
            strides = (Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv)
            q, k, v = qb.permute(0, 2, 1, 3).float(), kb.permute(0, 2, 1, 3).float(), vb.permute(0, 2, 1, 3).float()
        q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
        ref = ref_attn(q, k, v, scale)
        O = torch.empty(B * Nq, H * dv, device=dev, dtype=BF16)
        LSE = torch.empty(B, H, Nq, device=dev)
        p = lambda t: t[0].data_ptr() + 2 * t[1]
        ops.attn_fwd(p(qt), p(kt), p(vt), O, LSE, B, H, Nq, Nk, dqk, dv, *strides, Nq * H * dv, H * dv, scale)
        tag = f'attn B{B} H{H} {Nq}x{Nk} d{dqk}/{dv}'
        report(tag + ' fwd', rel(O.view(B, Nq, H, dv).permute(0, 2, 1, 3), ref), 1e-2)
        lse_ref = torch.logsumexp((q @ k.transpose(-2, -1)) * scale, -1)
        report(tag + ' lse', rel(LSE, lse_ref), 1e-4)
        dO = rnd(B * Nq, H * dv, dtype=BF16, seed=25)
        ref.backward(dO.view(B, Nq, H, dv).permute(0, 2, 1, 3).float())
        if fused:
            dbuf = torch.zeros_like(buf)
            dqt = (dbuf, qt[1]); dkt = (dbuf, kt[1]); dvt = (dbuf, vt[1])
        else:
            dqb, dkb, dvb = torch.zeros_like(qb), torch.zeros_like(kb), torch.zeros_like(vb)
            dqt, dkt, dvt = (dqb, 0), (dkb, 0), (dvb, 0)
        Delta = torch.empty_like(LSE)
        ops.attn_bwd(p(qt), p(kt), p(vt), O, dO, LSE, Delta, p(dqt), p(dkt), p(dvt), B, H, Nq, Nk, dqk, dv, *strides,
                     Nq * H * dv, H * dv, Nq * H * dv, H * dv, *strides, scale)
        if fused:
            gq = dbuf[:, qo:qo + Nq, 0].permute(0, 2, 1, 3); gk = dbuf[:, :, 1].permute(0, 2, 1, 3); gv = dbuf[:, :, 2].permute(0, 2, 1, 3)
        else:
            gq, gk, gv = dqb.permute(0, 2, 1, 3), dkb.permute(0, 2, 1, 3), dvb.permute(0, 2, 1, 3)
        report(tag + ' dq', rel(gq, q.grad), 2e-2)
        report(tag + ' dk', rel(gk, k.grad), 2e-2)
        report(tag + ' dv', rel(gv, v.grad), 2e-2)
        if fused and qo > 0:
            # dav_attn_bwd_ctx: the dQ kernel zero-fills the q slots of the qo context-only rows in front of the queries — into a
            # buffer full of NaNs the whole fused gradient must come out equal to the one written into zeros above
            dbuf2 = torch.full_like(buf, float('nan'))
            ops.attn_bwd(p(qt), p(kt), p(vt), O, dO, LSE, Delta, dbuf2.data_ptr() + 2 * qt[1], dbuf2.data_ptr() + 2 * kt[1],
                         dbuf2.data_ptr() + 2 * vt[1], B, H, Nq, Nk, dqk, dv, *strides, Nq * H * dv, H * dv, Nq * H * dv, H * dv,
                         *strides, scale, dq_ctx_rows=qo)
            same = torch.equal(dbuf2, dbuf) and float(dbuf2[:, :qo, 0].abs().max()) == 0.0
            report(tag + f' ctx rows {qo}', 0.0 if same else 1.0, 1e-9)


@check
def dropout():
    """Attention dropout (dav_attn_drop_fwd / _bwd, bf16 and fp32 kernels) and dropout on activations (dav_dropout_rows) against
    torch with the SAME keep masks: the kernels are deterministic in the mask, the draw is the caller's."""
    for (B, H, Nq, Nk, dqk, dv, off, pdrop) in [(2, 3, 49, 81, 64, 64, 32, 0.1), (3, 2, 8, 49, 64, 64, 0, 0.5), (2, 16, 228, 228, 32, 32, 0, 0.1),
                                               (2, 12, 16, 64, 16, 64, 0, 0.2), (3, 2, 4, 6, 16, 64, 0, 0.3), (1, 1, 1, 1, 32, 32, 0, 0.5),
                                               (2, 12, 63, 95, 64, 64, 32, 0.1), (2, 2, 204, 204, 64, 64, 0, 0.1),
                                               (2, 12, 32, 112, 16, 16, 0, 0.2), (1, 3, 32, 3087, 16, 16, 0, 0.1), (2, 2, 9, 20, 16, 16, 0, 0.3),
                                               # chunked variants (keys / queries stream through LDS)
                                               (2, 3, 784, 816, 64, 64, 32, 0.1), (1, 2, 100, 1000, 64, 64, 0, 0.2), (1, 2, 1000, 40, 64, 64, 0, 0.1),
                                               (2, 2, 1300, 1300, 32, 32, 0, 0.1), (1, 3, 17, 530, 16, 64, 0, 0.25)]:
        scale = 0.125 if dqk == 16 else dqk ** -0.5
        keep_p = 1.0 - pdrop
        ld = (Nk + 31) // 32 * 32
        g = torch.Generator(device='cpu'); g.manual_seed(1000 + Nq * 7 + Nk)
        keep = (torch.rand(B, H, Nq, ld, generator=g) < keep_p).to(torch.uint8).to(dev)
        km = keep[..., :Nk].float() / keep_p
        for f32 in (False, True):
            dt = F32 if f32 else BF16
            es = 4 if f32 else 2
            fused = dqk == dv and Nq <= Nk
            if fused:
                buf = rnd(B, Nk, 3, H, dqk, dtype=dt, seed=21)
                qo = off if Nq + off == Nk else 0
                qt = (buf, qo * 3 * H * dqk); kt = (buf, H * dqk); vt = (buf, 2 * H * dqk)
                strides = (Nk * 3 * H * dqk, 3 * H * dqk) * 3
                q = buf[:, qo:qo + Nq, 0].permute(0, 2, 1, 3).float()
                k = buf[:, :, 1].permute(0, 2, 1, 3).float()
                v = buf[:, :, 2].permute(0, 2, 1, 3).float()
            else:
                qo = 0
                qb, kb, vb = rnd(B, Nq, H, dqk, dtype=dt, seed=22), rnd(B, Nk, H, dqk, dtype=dt, seed=23), rnd(B, Nk, H, dv, dtype=dt, seed=24)
                qt, kt, vt = (qb, 0), (kb, 0), (vb, 0)
                strides = (Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv)
                q, k, v = qb.permute(0, 2, 1, 3).float(), kb.permute(0, 2, 1, 3).float(), vb.permute(0, 2, 1, 3).float()
            q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
            ref = (((q @ k.transpose(-2, -1)) * scale).softmax(-1) * km) @ v
            O = torch.empty(B * Nq, H * dv, device=dev, dtype=dt)
            LSE = torch.empty(B, H, Nq, device=dev)
            p = lambda t: t[0].data_ptr() + es * t[1]
            ops.attn_drop_fwd(p(qt), p(kt), p(vt), O, LSE, B, H, Nq, Nk, dqk, dv, *strides, Nq * H * dv, H * dv, scale, keep, ld, 1.0 / keep_p)
            tag = f'attn_drop {"f32" if f32 else "bf16"} B{B} H{H} {Nq}x{Nk} d{dqk}/{dv} p{pdrop}'
            tf, tb = (2e-5, 5e-5) if f32 else (1e-2, 2e-2)
            report(tag + ' fwd', rel(O.view(B, Nq, H, dv).permute(0, 2, 1, 3), ref), tf)
            report(tag + ' lse', rel(LSE, torch.logsumexp((q @ k.transpose(-2, -1)) * scale, -1)), 1e-4)
            dO = rnd(B * Nq, H * dv, dtype=dt, seed=25)
            ref.backward(dO.view(B, Nq, H, dv).permute(0, 2, 1, 3).float())
            fill = float('nan') if (fused and qo > 0 and not f32) else 0.0       # bf16: the dQ kernel zero-fills the context rows' q slots
            if fused:
                dbuf = torch.full_like(buf, fill)
                dqt = (dbuf, qt[1]); dkt = (dbuf, kt[1]); dvt = (dbuf, vt[1])
            else:
                dqb, dkb, dvb = torch.zeros_like(qb), torch.zeros_like(kb), torch.zeros_like(vb)
                dqt, dkt, dvt = (dqb, 0), (dkb, 0), (dvb, 0)
            Delta = torch.empty_like(LSE)
            ops.attn_drop_bwd(p(qt), p(kt), p(vt), O, dO, LSE, Delta, p(dqt), p(dkt), p(dvt), B, H, Nq, Nk, dqk, dv, *strides,
                              Nq * H * dv, H * dv, Nq * H * dv, H * dv, *strides, scale, keep, ld, 1.0 / keep_p,
                              dq_ctx_rows=qo if not f32 else 0)
            if fused:
                gq = dbuf[:, qo:qo + Nq, 0].permute(0, 2, 1, 3); gk = dbuf[:, :, 1].permute(0, 2, 1, 3); gv = dbuf[:, :, 2].permute(0, 2, 1, 3)
                if fill != 0.0:
                    report(tag + f' ctx rows {qo}', float(dbuf[:, :qo, 0].abs().max().nan_to_num(nan=1.0)), 1e-9)
            else:
                gq, gk, gv = dqb.permute(0, 2, 1, 3), dkb.permute(0, 2, 1, 3), dvb.permute(0, 2, 1, 3)
            report(tag + ' dq', rel(gq, q.grad), tb)
            report(tag + ' dk', rel(gk, k.grad), tb)
            report(tag + ' dv', rel(gv, v.grad), tb)
    # dropout on activations, DropPath folded in
    for (B, rows, D) in [(4, 49, 768), (3, 7, 128), (2, 204, 3072), (1, 1, 4)]:
        g = torch.Generator(device='cpu'); g.manual_seed(7 + rows)
        keep = (torch.rand(B * rows, D, generator=g) < 0.8).to(torch.uint8).to(dev)
        rs = (torch.rand(B, generator=g) < 0.7).float().to(dev) / 0.7
        res = rnd(B * rows, D, seed=3)
        for din in (F32, BF16):
            for dout in (F32, BF16):
                x = rnd(B * rows, D, dtype=din, seed=5)
                for (kp, rsc, rr) in [(keep, None, None), (keep, rs, res), (None, rs, None), (keep, None, res)]:
                    ref = x.float()
                    if kp is not None:
                        ref = ref * kp.float() / 0.8
                    if rsc is not None:
                        ref = (ref.view(B, rows, D) * rsc.view(B, 1, 1)).reshape(B * rows, D)
                    if rr is not None:
                        ref = ref + rr
                    out = torch.empty(B * rows, D, device=dev, dtype=dout)
                    ops.dropout_rows(x, kp, 1.0 / 0.8, B, rows, D, out, res=rr, rowscale=rsc)
                    report(f'dropout_rows {B}x{rows}x{D} {str(din)[6:]}->{str(dout)[6:]} keep{kp is not None} rs{rsc is not None} res{rr is not None}',
                           rel(out, ref.to(dout)), 1e-6 if dout == F32 else 4e-3)
                if din == dout:         # in place
                    y = x.clone()
                    ops.dropout_rows(y, keep, 1.25, B, rows, D, y)
                    report(f'dropout_rows in place {B}x{rows}x{D} {str(din)[6:]}', rel(y, (x.float() * keep.float() * 1.25).to(din)), 1e-6 if din == F32 else 4e-3)


@check
def window_attention():
    """Swin decoder kernels (models/swin.py): attention with the relative-position bias + shift mask on the A x A corner of
    [A window tokens | nF fusion tokens] sequences (bias table per window, b % nb), its dS output and the table gradient
    reduced from it, and the unfold / fold row movers — against plain torch."""
    LOG2E = 1.4426950408889634
    for (B, nW, H, A, nF, d, masked) in [(2, 4, 2, 16, 9, 32, True), (3, 6, 2, 16, 9, 32, False), (2, 20, 16, 16, 32, 32, True),
                                          (1, 1, 2, 16, 3, 32, False), (2, 4, 2, 16, 16, 64, True)]:
        N, D = A + nF, H * d
        ld = (N + 31) // 32 * 32
        win = int(A ** 0.5)
        T = (2 * win - 1) ** 2
        table = rnd(T, H, seed=71, scale=0.7)
        coords = torch.stack(torch.meshgrid(torch.arange(win), torch.arange(win), indexing='ij')).flatten(1)
        relc = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0) + (win - 1)
        index = (relc[..., 0] * (2 * win - 1) + relc[..., 1]).to(dev)
        index32 = index.reshape(-1).to(torch.int32)
        mask = None
        if masked:
            mask = torch.where(rnd(nW, A, A, seed=72) > 0.3, torch.full((nW, A, A), -100.0, device=dev), torch.zeros(nW, A, A, device=dev))
            mask[:, torch.arange(A), torch.arange(A)] = 0.0            # a row never masks itself
        nb = nW if masked else 1
        bias2 = torch.empty(nb, H, N, ld, device=dev)
        ops.relpos_bias_build(table, index32, mask, nb, H, A, N, ld, LOG2E, bias2)
        full = torch.zeros(nb, H, N, N, device=dev)
        full[:, :, :A, :A] = table[index.reshape(-1)].view(A, A, H).permute(2, 0, 1)[None] + (mask[:, None] if masked else 0.0)
        report(f'relpos_bias_build nW{nW} H{H} N{N}', rel(bias2[..., :N] / LOG2E, full) + (float(bias2[..., N:].abs().max()) if ld > N else 0.0), 1e-6)
        buf = rnd(B * nW, N, 3, H, d, dtype=BF16, seed=73)
        q = buf[:, :, 0].permute(0, 2, 1, 3).float().requires_grad_(True)
        k = buf[:, :, 1].permute(0, 2, 1, 3).float().requires_grad_(True)
        v = buf[:, :, 2].permute(0, 2, 1, 3).float().requires_grad_(True)
        scale = d ** -0.5
        logits = (q @ k.transpose(-2, -1)) * scale + full.repeat(B * nW // nb, 1, 1, 1)
        logits.retain_grad()
        ref = logits.softmax(-1) @ v
        O = torch.empty(B * nW * N, D, device=dev, dtype=BF16)
        LSE = torch.empty(B * nW, H, N, device=dev)
        st = (N * 3 * D, 3 * D) * 3
        p0 = buf.data_ptr()
        ops.attn_bias_fwd(p0, p0 + 2 * D, p0 + 4 * D, O, LSE, B * nW, H, N, N, d, d, *st, N * D, D, scale, bias2, nb, ld)
        tag = f'window attn B{B} nW{nW} H{H} N{N} d{d} mask{int(masked)}'
        report(tag + ' fwd', rel(O.view(B * nW, N, H, d).permute(0, 2, 1, 3), ref), 1e-2)
        report(tag + ' lse', rel(LSE, torch.logsumexp(logits, -1)), 1e-4)
        dO = rnd(B * nW * N, D, dtype=BF16, seed=74)
        ref.backward(dO.view(B * nW, N, H, d).permute(0, 2, 1, 3).float())
        dbuf = torch.zeros_like(buf)
        dS = torch.zeros(B * nW, H, N, ld, device=dev)
        d0 = dbuf.data_ptr()
        ops.attn_bias_bwd(p0, p0 + 2 * D, p0 + 4 * D, O, dO, LSE, torch.empty_like(LSE), d0, d0 + 2 * D, d0 + 4 * D, B * nW, H, N, N, d, d,
                          *st, N * D, D, N * D, D, *st, scale, bias2, nb, ld, dS)
        report(tag + ' dq', rel(dbuf[:, :, 0].permute(0, 2, 1, 3), q.grad), 2e-2)
        report(tag + ' dk', rel(dbuf[:, :, 1].permute(0, 2, 1, 3), k.grad), 2e-2)
        report(tag + ' dv', rel(dbuf[:, :, 2].permute(0, 2, 1, 3), v.grad), 2e-2)
        report(tag + ' dS', rel(dS[..., :N], logits.grad), 2e-2)
        dtab = torch.zeros(T, H, device=dev)
        ops.relpos_bias_bwd(dS, index32, B * nW, H, A, N, ld, T, dtab)
        want = torch.zeros(T, H, device=dev)
        want.index_add_(0, index.reshape(-1), dS[:, :, :A, :A].sum(0).permute(1, 2, 0).reshape(A * A, H))
        report(tag + ' dtable (from the kernel dS)', rel(dtab, want), 1e-5)
        # ---- row movers: [B, nF + L, C] <-> [B * nW, A + nF, C] through a random token permutation
        L, C = nW * A, 96
        rows = torch.randperm(L, generator=torch.Generator().manual_seed(5)).to(dev)
        inv = torch.empty_like(rows); inv[rows] = torch.arange(L, device=dev)
        x = rnd(B, nF + L, C, seed=75)
        seq = torch.empty(B * nW * N, C, device=dev, dtype=BF16)
        ops.window_unfold(x, rows.to(torch.int32), B, nW, A, nF, L, C, 0.5, seq)
        want = torch.cat([x[:, nF:][:, rows].reshape(B * nW, A, C), (0.5 * x[:, None, :nF]).expand(B, nW, nF, C).reshape(B * nW, nF, C)], 1)
        report(f'window_unfold nW{nW} nF{nF}', rel(seq.view(B * nW, N, C), want), 4e-3)
        t = rnd(B * nW * N, C, seed=76)
        res = rnd(B, nF + L, C, seed=77)
        out = torch.empty(B, nF + L, C, device=dev)
        ops.window_fold(t, inv.to(torch.int32), res, B, nW, A, nF, L, C, 1.0 / nW, out)
        tv = t.view(B, nW, N, C)
        tok = torch.empty(B, L, C, device=dev)
        tok[:, rows] = tv[:, :, :A].reshape(B, L, C)
        want = res + torch.cat([tv[:, :, A:].mean(1), tok], 1)
        report(f'window_fold nW{nW} nF{nF}', rel(out, want), 1e-6)


@check
def layernorm():
    for (B, r0, r1, D) in [(3, 4, 9, 128), (2, 0, 81, 768), (64, 32, 49, 768), (2, 0, 228, 512), (3, 5, 0, 192), (2, 3, 3, 1024)]:
        x0 = rnd(B, max(r0, 1), D, seed=31)[:, :r0].contiguous() if r0 else None
        x1 = rnd(B, max(r1, 1), D, seed=32)[:, :r1].contiguous() if r1 else None
        g, bt = rnd(D, seed=33) * 0.1 + 1, rnd(D, seed=34) * 0.1
        xs = [t for t in (x0, x1) if t is not None]
        xc = torch.cat(xs, 1).clone().requires_grad_(True)
        gp, bp = g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(xc, (D,), gp, bp, 1e-6)
        R = r0 + r1
        y, y32 = torch.empty(B * R, D, device=dev, dtype=BF16), torch.empty(B * R, D, device=dev)
        mean, rstd = torch.empty(B * R, device=dev), torch.empty(B * R, device=dev)
        a0, a1 = (x0, x1) if x0 is not None else (x1, None)
        n0, n1 = (r0, r1) if x0 is not None else (r1, 0)
        ops.layernorm_fwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, g, bt, 1e-6, y, y32, mean, rstd)
        tag = f'ln B{B} {r0}+{r1} D{D}'
        report(tag + ' fwd32', rel(y32, ref.view(-1, D)), 1e-5)
        report(tag + ' fwd16', rel(y, ref.view(-1, D)), 5e-3)
        dy = rnd(B * R, D, dtype=BF16, seed=35)
        dy32 = rnd(B * R, D, seed=36)
        ref.backward((dy.float() + dy32).view(B, R, D))
        dx0 = torch.full((B, n0, D), 1.0, device=dev)
        res0 = rnd(B, n0, D, seed=37)
        tw0 = torch.empty(B, n0, D, device=dev, dtype=BF16)
        dx1 = torch.empty(B, max(n1, 1), D, device=dev)[:, :n1].contiguous() if n1 else None
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        ops.layernorm_bwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, dy, dy32, g, mean, rstd,
                          dx0, n0 * D, 1, res0, n0 * D, tw0, n0 * D, dx1, n1 * D, 0, None, 0, None, 0, dg, db)
        report(tag + ' dx0(acc+res)', rel(dx0, xc.grad[:, :n0] + 1.0 + res0), 1e-4)
        report(tag + ' dx0 twin', rel(tw0, xc.grad[:, :n0] + 1.0 + res0), 5e-3)
        if n1:
            report(tag + ' dx1', rel(dx1, xc.grad[:, n0:]), 1e-4)
        report(tag + ' dgamma', rel(dg, gp.grad), 1e-4)
        report(tag + ' dbeta', rel(db, bp.grad), 1e-4)


def _slot_stats(x):
    """[rows, D] fp32 -> [rows, D/64, 2] {sum, sum of squares} per 64-column slot (float64 reference)"""
    r, D = x.shape
    v = x.double().view(r, D // 64, 64)
    return torch.stack([v.sum(-1), (v * v).sum(-1)], -1)


@check
def ln_fused():
    """LayerNorm folded into the GEMMs either side of it (dav_gemm_nt_ln_bf16, dav_ln_fold_grouped, dav_rowstats_cast,
    dav_layernorm_bwd_twin) against torch fp32 / fp64 on the same operands."""
    ln = torch.nn.functional.layer_norm
    # 1. stand-alone row statistics + twin (incl. a broadcast source: batch stride 0)
    for (B, rows, D, bs) in [(3, 7, 128, None), (2, 81, 768, None), (4, 16, 512, 0), (2, 5, 1024, None), (64, 49, 768, None)]:
        x = rnd(B if bs is None else 1, rows, D, seed=41) + 0.3
        tw = torch.empty(B * rows, D, device=dev, dtype=BF16)
        st = torch.empty(B * rows, D // 64, 2, device=dev)
        ops.rowstats_cast(x, rows * D if bs is None else 0, B, rows, D, tw, st)
        xe = x.expand(B, rows, D).reshape(B * rows, D)
        report(f'rowstats B{B} r{rows} D{D} twin', float((tw.float() - xe.to(BF16).float()).abs().max()), 0.0)
        report(f'rowstats B{B} r{rows} D{D} sums', rel(st, _slot_stats(xe)), 2e-6)
    # 2. weight fold
    items, refs = [], []
    for (N, K, has_b) in [(2304, 768, True), (1536, 512, True), (96, 64, False), (3072, 1024, True), (40, 192, True)]:
        w = rnd(N, K, scale=0.05, seed=42)
        g, bt = rnd(K, seed=43) * 0.2 + 1, rnd(K, seed=44) * 0.2
        bias = rnd(N, seed=45) if has_b else None
        wl, c, d = torch.empty(N, K, device=dev, dtype=BF16), torch.empty(N, device=dev), torch.empty(N, device=dev)
        items.append((w, g, bt, bias, wl, c, d))
        wl_ref = (w * g).to(BF16)
        refs.append((wl_ref, wl_ref.double().sum(1), (w.double() * bt.double()).sum(1) + (bias.double() if has_b else 0)))
    ops.ln_fold_grouped(items)
    for (w, g, bt, bias, wl, c, d), (wl_ref, c_ref, d_ref) in zip(items, refs):
        tag = f'ln_fold {w.shape[0]}x{w.shape[1]}'
        report(tag + ' w_ln', float((wl.float() - wl_ref.float()).abs().max()), 0.0)
        report(tag + ' c', rel(c, c_ref), 2e-6)
        report(tag + ' d', rel(d, d_ref), 2e-6)
    # 3. producer side: statistics partials + twin of the fp32 result, through row maps, in every tile configuration that has them
    for (M, N, K, cfg, cmap) in [(3136, 768, 768, 0, None), (3136, 768, 768, 3, None), (1024, 768, 3072, 8, None), (130, 512, 512, 7, None),
                                 (260, 512, 2048, 5, None), (22528, 512, 2048, 0, None), (64 * 8, 768, 768, 0, (8, 16, 4)), (200, 192, 192, 0, None),
                                 (4096, 1024, 1024, 3, None)]:
        A, Bm = rnd(M, K, dtype=BF16, seed=46), rnd(N, K, dtype=BF16, scale=0.05, seed=47)
        bias = rnd(N, seed=48)
        rows_out = M if cmap is None else (M // cmap[0]) * cmap[1]
        res = rnd(rows_out, N, seed=49)
        C = torch.zeros(rows_out, N, device=dev)
        tw = torch.zeros(rows_out, N, device=dev, dtype=BF16)
        st = torch.zeros(rows_out, N // 64, 2, device=dev)
        ops.gemm_nt_ln(A, Bm, M, N, K, prod=dict(stats_out=st, twin_out=tw, ld_twin=N), bias=bias, res=res, ldres=N, res_rowmap=cmap,
                       C_out=C, c_rowmap=cmap, variant=cfg << 4)
        ref = A.float() @ Bm.float().t() + bias
        if cmap is None:
            want = ref + res
            got, gtw, gst = C, tw, st
        else:
            rpb, bs_, off = cmap
            idx = (torch.arange(M, device=dev) // rpb) * bs_ + off + torch.arange(M, device=dev) % rpb
            want = ref + res[idx]
            got, gtw, gst = C[idx], tw[idx], st[idx]
        tag = f'nt_ln producer {M}x{N}x{K} cfg{cfg}{" rowmap" if cmap else ""}'
        report(tag + ' C', rel(got, want), 1e-4)
        report(tag + ' twin', float((gtw.float() - got.to(BF16).float()).abs().max()), 0.0)
        report(tag + ' sums', rel(gst, _slot_stats(got)), 5e-6)
    # 4. consumer side: raw twin + partials + folded weight == LayerNorm -> Linear
    for (B, r0, r1, D, N, cfg, act, c_bf16) in [(64, 16, 49, 768, 2304, 0, 0, True), (64, 0, 49, 768, 3072, 0, 1, True), (8, 0, 352, 512, 1536, 44, 0, True),
                                                (8, 0, 352, 512, 2048, 44, 1, True), (4, 0, 320, 512, 256, 0, 0, False), (2, 5, 12, 192, 576, 0, 0, True),
                                                (64, 0, 49, 768, 1536, 45, 0, True), (3, 0, 50, 1024, 3072, 3, 0, True), (16, 16, 64, 768, 2304, 8, 0, True),
                                                (2, 0, 33, 128, 64, 7, 0, False)]:
        R = r0 + r1
        M = B * R
        eps = 1e-6 if D != 512 else 1e-5
        x0 = (rnd(B, max(r0, 1), D, seed=51) * 1.3 + 0.2)[:, :r0].contiguous() if r0 else None
        x1 = (rnd(B, r1, D, seed=52) * 0.7 - 0.1)
        w32 = rnd(N, D, scale=0.05, seed=53)            # fp32 master; the un-fused path contracts with its bf16 mirror
        w = w32.to(BF16)
        g, bt, bias = rnd(D, seed=54) * 0.2 + 1, rnd(D, seed=55) * 0.2, rnd(N, seed=56)
        wl, c, d = torch.empty_like(w), torch.empty(N, device=dev), torch.empty(N, device=dev)
        ops.ln_fold_grouped([(w32, g, bt, bias, wl, c, d)])
        segs = []
        for xs, r in ((x0, r0), (x1, r1)):
            if xs is None:
                continue
            tw = torch.empty(B * r, D, device=dev, dtype=BF16)
            st = torch.empty(B * r, D // 64, 2, device=dev)
            ops.rowstats_cast(xs, r * D, B, r, D, tw, st)
            segs.append((tw, st, r))
        xc = torch.cat([t for t in (x0, x1) if t is not None], 1)
        h = ln(xc, (D,), g, bt, eps).view(M, D)
        ref = h @ w32.t() + bias
        pre = ref
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        out = torch.empty(M, N, device=dev, dtype=BF16 if c_bf16 else F32)
        Z = torch.empty(M, N, device=dev, dtype=BF16) if act == 1 else None
        lnd = dict(stats=segs[0][1], ln_c=c, eps=eps)
        if len(segs) == 2:
            lnd.update(A2=segs[1][0], stats2=segs[1][1], a_r0=r0, a_r1=r1)
        ops.gemm_nt_ln(segs[0][0], wl, M, N, D, ln=lnd, bias=d, act=act, C_out=out, c_bf16=c_bf16, C2=Z, ldc2=N, c2_mode=4 if act == 1 else 0,
                       variant=cfg << 4)
        tag = f'nt_ln consumer B{B} {r0}+{r1} D{D} N{N} cfg{cfg} act{act}'
        report(tag, rel(out, ref), 6e-3)
        if Z is not None:
            xg = pre.clone().requires_grad_(True)
            torch.nn.functional.gelu(xg).sum().backward()
            report(tag + " gelu'twin", float((Z.float() - xg.grad).abs().max()), 2e-2)
        # against the un-fused product path on the same operands (LayerNorm kernel -> bf16 -> GEMM): both are bf16-operand results
        y = torch.empty(M, D, device=dev, dtype=BF16)
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        a0, a1 = (x0, x1) if x0 is not None else (x1, None)
        n0, n1 = (r0, r1) if x0 is not None else (r1, 0)
        ops.layernorm_fwd(a0, n0 * D, n0, a1, n1 * D if a1 is not None else 0, n1, B, D, g, bt, eps, y, None, mean, rstd)
        out2 = torch.empty(M, N, device=dev)
        ops.gemm_nt(y, w, M, N, D, bias=bias, act=act, C_out=out2)
        report(tag + ' vs unfused (fp32 ref distance ratio)', rel(out, ref) / max(rel(out2, ref), 1e-9), 2.0)
    # 4b. consumer through a row map (the decoder head reads x[:, nF:])
    B, nF, L, D, N = 4, 8, 96, 512, 256
    x = rnd(B, nF + L, D, seed=57)
    tw, st = torch.empty(B * (nF + L), D, device=dev, dtype=BF16), torch.empty(B * (nF + L), D // 64, 2, device=dev)
    ops.rowstats_cast(x, (nF + L) * D, B, nF + L, D, tw, st)
    w = rnd(N, D, scale=0.05, seed=58)
    g, bt, bias = rnd(D, seed=59) * 0.2 + 1, rnd(D, seed=60) * 0.2, rnd(N, seed=61)
    wl, c, d = torch.empty(N, D, device=dev, dtype=BF16), torch.empty(N, device=dev), torch.empty(N, device=dev)
    ops.ln_fold_grouped([(w, g, bt, bias, wl, c, d)])
    out = torch.empty(B * L, N, device=dev)
    ops.gemm_nt_ln(tw, wl, B * L, N, D, ln=dict(stats=st, ln_c=c, eps=1e-5), a_rowmap=(L, nF + L, nF), bias=d, C_out=out)
    ref = ln(x[:, nF:], (D,), g, bt, 1e-5).reshape(B * L, D) @ w.t() + bias
    report('nt_ln consumer rowmap', rel(out, ref), 6e-3)
    # 5. backward from the twin: exact against autograd ON THE TWIN'S VALUES, h_out = the LayerNorm output
    for (B, r0, r1, D) in [(3, 4, 9, 128), (2, 0, 81, 768), (64, 16, 49, 768), (2, 0, 228, 512), (2, 3, 3, 1024)]:
        R = r0 + r1
        eps = 1e-6
        xs = [(rnd(B, r, D, seed=62 + i) * (1 + i) + 0.1 * i) for i, r in enumerate((r0, r1)) if r]
        twins = []
        for xx in xs:
            r = xx.shape[1]
            tw, st = torch.empty(B * r, D, device=dev, dtype=BF16), torch.empty(B * r, D // 64, 2, device=dev)
            ops.rowstats_cast(xx, r * D, B, r, D, tw, st)
            twins.append((tw, st, r))
        g, bt = rnd(D, seed=33) * 0.1 + 1, rnd(D, seed=34) * 0.1
        xc = torch.cat([tw.float().view(B, r, D) for (tw, st, r) in twins], 1).clone().requires_grad_(True)
        gp, bp = g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        ref = ln(xc, (D,), gp, bp, eps)
        dy, dy32 = rnd(B * R, D, dtype=BF16, seed=35), rnd(B * R, D, seed=36)
        ref.backward((dy.float() + dy32).view(B, R, D))
        (t0, s0, n0) = twins[0]
        (t1, s1, n1) = twins[1] if len(twins) == 2 else (None, None, 0)
        dx0 = torch.full((B, n0, D), 1.0, device=dev)
        res0 = rnd(B, n0, D, seed=37)
        tw0 = torch.empty(B, n0, D, device=dev, dtype=BF16)
        dx1 = torch.empty(B, max(n1, 1), D, device=dev)[:, :n1].contiguous() if n1 else None
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        h = torch.empty(B * R, D, device=dev, dtype=BF16)
        ops.layernorm_bwd_twin(t0, n0 * D, s0, n0, t1, n1 * D, s1, n1, B, D, eps, dy, dy32, g, bt,
                               dx0, n0 * D, 1, res0, n0 * D, tw0, n0 * D, dx1, n1 * D, 0, None, 0, None, 0, h_out=h, dgamma=dg, dbeta=db)
        tag = f'ln_bwd_twin B{B} {r0}+{r1} D{D}'
        # (the statistics are those of the fp32 rows, x itself their bf16 rounding: mean / rstd differ from the twin's own by ~1e-3 relative)
        report(tag + ' dx0(acc+res)', rel(dx0, xc.grad[:, :n0] + 1.0 + res0), 3e-3)
        report(tag + ' dx0 twin', rel(tw0, xc.grad[:, :n0] + 1.0 + res0), 6e-3)
        if n1:
            report(tag + ' dx1', rel(dx1, xc.grad[:, n0:]), 3e-3)
        report(tag + ' dgamma', rel(dg, gp.grad), 3e-3)
        report(tag + ' dbeta', rel(db, bp.grad), 1e-4)
        report(tag + ' h_out', rel(h, ref.detach().view(-1, D)), 6e-3)


@check
def masking():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'masking.npz'))
    for t in sorted({k.split('.')[0] for k in g.files}):
        noise = torch.from_numpy(g[f'{t}.noise']).to(dev)
        lk = g[f'{t}.ids_keep'].shape[1]
        ik, mask, ir, ik32, ir32 = ops.mask_build(noise, lk)
        ok = (np.array_equal(ik.cpu().numpy(), g[f'{t}.ids_keep']) and np.array_equal(ir.cpu().numpy(), g[f'{t}.ids_restore'])
              and np.array_equal(mask.cpu().numpy(), g[f'{t}.mask']) and np.array_equal(ik32.cpu().numpy(), g[f'{t}.ids_keep'])
              and np.array_equal(ir32.cpu().numpy(), g[f'{t}.ids_restore']))
        report(f'mask_build {t} bit-exact', 0.0 if ok else 1.0, 0.0)
    noise = torch.rand(64, 320, device=dev)
    ik, mask, ir, _, _ = ops.mask_build(noise, 63)
    sh = torch.argsort(noise, dim=1)
    ok = torch.equal(ik, sh[:, :63]) and torch.equal(ir, torch.argsort(sh, dim=1))
    report('mask_build vs torch.argsort 64x320', 0.0 if ok else 1.0, 0.0)


@check
def misc_kernels():
    from oracle import avmae_oracle as O
    B, C, H, W, nk = 3, 3, 64, 96, 5
    img = rnd(B, C, H, W, seed=41)
    L = (H // 16) * (W // 16)
    ids = torch.stack([torch.randperm(L)[:nk] for _ in range(B)]).to(dev).to(torch.int32)
    A = torch.empty(B * nk, C * 256, device=dev, dtype=BF16)
    ops.patch_gather(img, ids, nk, A)
    cols = img.reshape(B, C, H // 16, 16, W // 16, 16).permute(0, 2, 4, 1, 3, 5).reshape(B, L, C * 256)
    ref = cols.gather(1, ids.long().unsqueeze(-1).expand(-1, -1, C * 256)).reshape(B * nk, -1)
    report('patch_gather', rel(A, ref), 4e-3)
    A2 = torch.empty(B * L, C * 256, device=dev, dtype=BF16)
    ops.patch_gather(img, None, L, A2)
    report('patch_gather all', rel(A2, cols.reshape(B * L, -1)), 4e-3)
    # unshuffle
    D, nF = 64, 3
    emb, mt, pos = rnd(B * nk, D, seed=42), rnd(D, seed=43), rnd(L, D, seed=44)
    restore = torch.stack([torch.randperm(L) for _ in range(B)]).to(dev)
    out = torch.zeros(B, nF + L, D, device=dev)
    ops.unshuffle_fwd(emb, mt, pos, restore.to(torch.int32), B, L, nk, D, out, (nF + L) * D, nF)
    full = torch.cat([emb.view(B, nk, D), mt.view(1, 1, D).expand(B, L - nk, D)], 1)
    ref = full.gather(1, restore.unsqueeze(-1).expand(-1, -1, D)) + pos
    report('unshuffle_fwd', rel(out[:, nF:], ref), 1e-6)
    gx = rnd(B, nF + L, D, seed=45)
    keep = torch.argsort(restore, dim=1)[:, :nk].to(torch.int32)
    o = torch.empty(B * nk, D, device=dev, dtype=BF16)
    ops.rows_gather_cast(gx, (nF + L) * D, nF, keep, B, nk, D, o)
    report('rows_gather_cast', rel(o.view(B, nk, D), gx[:, nF:].gather(1, keep.long().unsqueeze(-1).expand(-1, -1, D))), 4e-3)
    dpos, dmt = torch.zeros(L, D, device=dev), torch.zeros(D, device=dev)
    ops.unshuffle_bwd_reduce(gx, (nF + L) * D, nF, restore.to(torch.int32), B, L, nk, D, dpos, dmt)
    report('unshuffle dpos', rel(dpos, gx[:, nF:].sum(0)), 1e-5)
    msk = (restore >= nk).float().unsqueeze(-1)
    report('unshuffle dmask_token', rel(dmt, (gx[:, nF:] * msk).sum((0, 1))), 1e-5)
    # loss
    for Cc in (3, 1):
        im = rnd(B, Cc, H, W, seed=46)
        P = 256 * Cc
        pred = rnd(B, L, P, seed=47).requires_grad_(True)
        mask = (torch.rand(B, L, device=dev) > 0.3).float()
        for norm in (True, False):
            tgt = O.patchify(im, (16, 16))
            ref = O.forward_loss(tgt, pred, mask, norm)
            gsc = torch.tensor(0.7, device=dev)
            pred.grad = None
            (ref * gsc).backward()
            lp, tm, tr = torch.empty(B * L, device=dev), torch.empty(B * L, device=dev), torch.empty(B * L, device=dev)
            loss, ms = torch.empty(1, device=dev), torch.empty(1, device=dev)
            ops.patch_mse_fwd(im, pred.detach(), mask, norm, lp, tm, tr, loss, ms)
            report(f'patch_mse fwd C{Cc} norm{int(norm)}', abs(float(loss) - float(ref)) / abs(float(ref)), 1e-5)
            dp = torch.empty(B * L, P, device=dev, dtype=BF16)
            ops.patch_mse_bwd(im, pred.detach(), mask, tm, tr, ms, gsc, dp)
            report(f'patch_mse bwd C{Cc} norm{int(norm)}', rel(dp.view(B, L, P), pred.grad), 5e-3)
    # pairs
    nv, na, Wd = 3, 2, 64
    Pv, Pa = rnd(B * nv, Wd, seed=48), rnd(B * na, Wd, seed=49)
    out = torch.empty(B * nv * na, Wd, device=dev, dtype=BF16)
    ops.pair_expand(Pv, Pa, B, nv, na, Wd, out)
    ref = (Pv.view(B, nv, 1, Wd) + Pa.view(B, 1, na, Wd)).reshape(-1, Wd)
    report('pair_expand', rel(out, ref), 4e-3)
    d = rnd(B * nv * na, Wd, dtype=BF16, seed=50)
    dPv, dPa = torch.empty(B * nv, Wd, device=dev, dtype=BF16), torch.empty(B * na, Wd, device=dev, dtype=BF16)
    ops.pair_reduce(d, B, nv, na, Wd, dPv, dPa)
    d4 = d.float().view(B, nv, na, Wd)
    report('pair_reduce v', rel(dPv, d4.sum(2).reshape(-1, Wd)), 4e-3)
    report('pair_reduce a', rel(dPa, d4.sum(1).reshape(-1, Wd)), 4e-3)
    # DropPath row kernels
    Bq, rq, Dq = 5, 7, 192
    res, yb, sc = rnd(Bq * rq, Dq, seed=61), rnd(Bq * rq, Dq, seed=62), torch.tensor([0., 1.25, 1.25, 0., 1.25], device=dev)
    outp = torch.empty_like(res)
    ops.rows_axpy(res, yb, sc, Bq, rq, Dq, outp)
    refp = res + yb * sc.repeat_interleave(rq)[:, None]
    report('rows_axpy', rel(outp, refp), 1e-6)
    ops.rows_axpy(res, yb, sc, Bq, rq, Dq, yb)                      # in place over y
    report('rows_axpy in place', rel(yb, refp), 1e-6)
    gq = rnd(Bq * rq, Dq, seed=63)
    ob = torch.empty(Bq * rq, Dq, device=dev, dtype=BF16)
    ops.rows_scale_cast(gq, sc, Bq, rq, Dq, ob)
    report('rows_scale_cast exact', float((ob != (gq * sc.repeat_interleave(rq)[:, None]).to(BF16)).sum()), 0.0)
    # casts / norm / adamw
    x = rnd(1000, 77, seed=51)
    y = torch.empty(1000, 77, device=dev, dtype=BF16)
    ops.cast_bf16(x, y)
    report('cast_bf16 exact', float((y != x.to(BF16)).sum()), 0.0)
    for shape in [(64, 16, 768), (3, 5, 128), (1, 4)]:
        a, b = rnd(*shape, seed=91), rnd(*shape, seed=92)
        o32, ob = ops.add_cast(a, b)
        report(f'add_cast {shape}', float((o32 != a + b).sum()) + float((ob != (a + b).to(BF16)).sum()), 0.0)
    yt = torch.empty(77, 1000, device=dev, dtype=BF16)
    ops.cast_transpose_bf16(x, yt)
    report('cast_transpose exact', float((yt != x.t().to(BF16)).sum()), 0.0)
    flat = rnd(1234567, seed=52)
    out, ws = torch.empty(1, device=dev), torch.empty(1024, device=dev)
    ops.l2norm(flat, out, ws, 0.5)
    report('l2norm', abs(float(out) - 0.5 * float(flat.double().norm())) / float(flat.double().norm()), 1e-6)
    n = 100036                    # the flat buffers' contract: length and segment boundaries multiples of 4, 16-byte aligned
    p0, g0 = rnd(n, seed=53), rnd(n, seed=54)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([{'params': [pr], 'weight_decay': 0.05}], lr=1e-2, betas=(0.9, 0.95))
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pb = torch.empty(n, device=dev, dtype=BF16)
    seg = torch.tensor([40000, n], device=dev, dtype=torch.int64)
    hyper = torch.tensor([1e-2, 0.05, 1e-2, 0.05], device=dev)
    for step in range(1, 4):
        pr.grad = g0 * step
        opt.step()
        bc = torch.tensor([1 - 0.9 ** step, math.sqrt(1 - 0.95 ** step)], device=dev)
        gcur = (g0 * step).clone()
        ssq = torch.zeros(1, device=dev)
        keep = torch.tensor([0, 1 if step == 2 else 0], device=dev, dtype=torch.uint8)      # step 2: the second segment's gradient is kept
        ops.adamw_flat(p, gcur, m, v, pb, seg, hyper, 2, 0.9, 0.95, 1e-8, bc, sumsq_out=ssq, zero_grad=True, keep_grad=keep)
        report(f'adamw fused sumsq step{step}', abs(float(ssq) - float((g0 * step).double().pow(2).sum())) / float((g0 * step).double().pow(2).sum()), 1e-5)
        report(f'adamw fused zero_grad step{step}', float(gcur[:40000].abs().max()), 0.0)
        report(f'adamw fused zero_grad / keep_grad step{step}', float((gcur[40000:] - (g0 * step)[40000:] * (step == 2)).abs().max()), 0.0)
    report('adamw_flat vs torch.optim.AdamW', rel(p, pr.detach()), 1e-6)
    report('adamw bf16 mirror', rel(pb, p), 4e-3)
    # device-side step guard (dav_step_guard + dav_adamw_flat gscale_dev): clip factor as a device scalar == the same factor as
    # a host argument, bit for bit; scale 0 (non-finite loss / norm) leaves parameters, moments and mirror untouched, still
    # zero-fills the gradients and still reports sum(g^2)
    bc = torch.tensor([1 - 0.9 ** 4, math.sqrt(1 - 0.95 ** 4)], device=dev)
    one, nan, inf = torch.tensor([2.5], device=dev), torch.tensor([float('nan')], device=dev), torch.tensor([float('inf')], device=dev)
    gn, ws = torch.empty(1, device=dev), torch.empty(1024, device=dev)
    ops.l2norm(g0, gn, ws, 1.0)
    sc, bad = torch.empty(1, device=dev), torch.zeros(1, device=dev, dtype=torch.int32)
    clip = 0.25 * float(gn)
    ops.step_guard(one, one, gn, clip, 1.0, sc, bad)
    report('step_guard clip factor', abs(float(sc) - clip / (float(gn) + 1e-6)), 1e-7)
    ops.step_guard(one, None, gn, 10.0 * float(gn), 1.0, sc, bad)
    report('step_guard no clipping below the limit', abs(float(sc) - 1.0), 0.0)
    ops.step_guard(one, one, None, 0.0, 1.0, sc, bad)
    report('step_guard plain', abs(float(sc) - 1.0) + float(bad), 0.0)
    for tag, (la, lb, nrm) in dict(nan_loss=(nan, one, None), inf_loss=(one, inf, gn), nan_norm=(one, one, nan)).items():
        bad.zero_()
        ops.step_guard(la, lb, nrm, 1.0, 1.0, sc, bad)
        report(f'step_guard {tag} -> 0', abs(float(sc)) + abs(int(bad) - 1), 0.0)
    ops.step_guard(one, one, gn, clip, 1.0, sc, bad)
    pa, ma, va, pba, ga = p.clone(), m.clone(), v.clone(), pb.clone(), g0.clone()
    pc, mc, vc, pbc, gc = p.clone(), m.clone(), v.clone(), pb.clone(), g0.clone()
    ops.adamw_flat(pa, ga, ma, va, pba, seg, hyper, 2, 0.9, 0.95, 1e-8, bc, grad_scale=float(sc))
    ops.adamw_flat(pc, gc, mc, vc, pbc, seg, hyper, 2, 0.9, 0.95, 1e-8, bc, gscale_dev=sc)
    report('adamw gscale_dev == grad_scale (bit-equal)', float((pa != pc).sum() + (ma != mc).sum() + (va != vc).sum()), 0.0)
    sc.zero_()
    pz, mz, vz, pbz, gz, ssq = p.clone(), m.clone(), v.clone(), pb.clone(), g0.clone(), torch.zeros(1, device=dev)
    ops.adamw_flat(pz, gz, mz, vz, pbz, seg, hyper, 2, 0.9, 0.95, 1e-8, bc, sumsq_out=ssq, zero_grad=True, gscale_dev=sc)
    report('adamw skipped step: p, m, v, mirror untouched', float((pz != p).sum() + (mz != m).sum() + (vz != v).sum() + (pbz != pb).sum()), 0.0)
    report('adamw skipped step: gradients zero-filled, sumsq reported', float(gz.abs().max()) + abs(float(ssq) / float(g0.double().pow(2).sum()) - 1.0), 1e-5)


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    print('device:', torch.cuda.get_device_name(0), flush=True)
    for kv in os.environ.get('DAV_TUNE', '').split(','):      # same launch-geometry knobs as bench.py (e.g. DAV_TUNE=4:1)
        if ':' in kv:
            from deepavfusion_amd import _lib
            _lib.check(_lib.load().dav_tune(int(kv.split(':')[0]), int(kv.split(':')[1])), 'dav_tune')
    for fn in (gemm_nt, gemm_tn, gemm_tn_gang, attention, dropout, window_attention, layernorm, ln_fused, masking, misc_kernels, patch_gather3d):
        if flt in fn.__name__:
            fn()
    bad = [r for r in RESULTS if not r[3]]
    print(f'\n{len(RESULTS) - len(bad)}/{len(RESULTS)} checks passed')
    for r in bad:
        print('  FAILED:', r[0], r[1])
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
