#!/usr/bin/env python3
"""Randomised shape fuzz of the C-ABI kernels against torch fp32 references (companion of tests/gpu_selfcheck.py, which uses
fixed shape lists).  Usage: python tests/gpu_fuzz.py [seed] [n_gemm] [n_attn] [n_ln] [n_gang]"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepavfusion_amd import ops   # noqa: E402

dev = torch.device('cuda')
BF16, F32 = torch.bfloat16, torch.float32
FAILS = []


def rel(a, b):
    """relative L2 error; where the reference is (numerically) zero — e.g. dq / dk of a softmax over ONE key — the error
    is taken relative to unit scale instead"""
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    nb = float(b.norm())
    return float((a - b).norm() / (nb if nb > 1e-6 * max(1.0, b.numel() ** 0.5) else max(1.0, b.numel() ** 0.5)))


def check(tag, err, tol):
    if not (err <= tol):
        FAILS.append((tag, err, tol))
        print(f'FAIL {tag}: {err:.3e} > {tol:.1e}', flush=True)


def fuzz_gemm(rng, n):
    for it in range(n):
        M = rng.choice([1, 3, 17, 64, 100, 128, 129, 255, 300, 777, 1024, 2049, rng.randint(1, 3000)])
        N = 8 * rng.choice([1, 2, 3, 8, 9, 16, 24, 33, 64, 96, rng.randint(1, 200)])
        K = 8 * rng.choice([1, 2, 8, 9, 16, 24, 64, 96, rng.randint(1, 160)])
        kn = rng.random() < 0.3 and K % 64 == 0
        A = torch.randn(M, K, device=dev).to(BF16)
        W = (torch.randn(N, K, device=dev) * 0.1).to(BF16)
        Bm = W.t().contiguous() if kn else W
        ref = A.float() @ W.float().t()
        bias = torch.randn(N, device=dev) if rng.random() < 0.5 else None
        alpha = rng.choice([1.0, 0.5])
        act = rng.choice([0, 0, 1])
        res = torch.randn(M, N, device=dev) if rng.random() < 0.5 else None
        beta = 1 if (rng.random() < 0.3) else 0
        c_bf16 = (rng.random() < 0.5) and not beta
        c2_mode = rng.choice([0, 0, 1, 2, 3, 4])
        if c2_mode == 4 and act != 1:
            c2_mode = 0
        v = ref * alpha + (bias if bias is not None else 0)
        pre = v.clone()
        dgelu = None
        if act == 1:
            x = v.clone().requires_grad_(True)
            y = torch.nn.functional.gelu(x)
            y.sum().backward()
            v, dgelu = y.detach(), x.grad
        post = v.clone()
        if res is not None:
            v = v + res
        C0 = torch.randn(M, N, device=dev)
        if beta:
            v = v + C0
        C = C0.clone().to(BF16) if c_bf16 else C0.clone()
        C2 = torch.empty(M, N, device=dev, dtype=BF16) if c2_mode else None
        ops.gemm_nt(A, Bm, M, N, K, ldb=N if kn else K, bias=bias, act=act, res=res, ldres=N, C_out=C, c_bf16=c_bf16, beta=beta,
                    alpha=alpha, C2=C2, ldc2=N, c2_mode=c2_mode, variant=(1 << 12) if kn else 0)
        tag = f'gemm M{M} N{N} K{K} kn{int(kn)} act{act} res{int(res is not None)} beta{beta} bf{int(c_bf16)} c2{c2_mode} bias{int(bias is not None)}'
        check(tag, rel(C.float(), v), 8e-3 if c_bf16 else 2e-4)
        if c2_mode:
            want = {1: pre, 2: post, 3: v, 4: dgelu}[c2_mode]
            check(tag + ' C2', rel(C2.float(), want), 8e-3)
        # weight gradient of the same problem
        if N % 8 == 0 and K % 8 == 0:
            dY = torch.randn(M, N, device=dev).to(BF16)
            G0 = torch.randn(N, K, device=dev)
            G = G0.clone()
            bg0 = torch.randn(N, device=dev)
            bg = bg0.clone()
            ops.gemm_tn(dY, A, M, N, K, G, beta=1, bias_grad=bg)
            check(tag + ' wgrad', rel(G, G0 + dY.float().t() @ A.float()), 3e-4)
            check(tag + ' bgrad', rel(bg, bg0 + dY.float().sum(0)), 3e-4)


def fuzz_attn(rng, n):
    for it in range(n):
        dqk, dv = rng.choice([(64, 64), (32, 32), (16, 64), (16, 16)])
        B, H = rng.randint(1, 3), rng.randint(1, 4)
        Nk = rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 64, 100, 255, 256, 257, 320, 511, 513, 700, rng.randint(1, 1200)])
        Nq = rng.choice([1, 5, 16, 17, 33, 64, 129, rng.randint(1, 600)])
        q = torch.randn(B, Nq, H, dqk, device=dev).to(BF16)
        k = torch.randn(B, Nk, H, dqk, device=dev).to(BF16)
        v = torch.randn(B, Nk, H, dv, device=dev).to(BF16)
        scale = rng.choice([dqk ** -0.5, 0.125])
        qf, kf, vf = (t.float().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
        s = (qf @ kf.transpose(-2, -1)) * scale
        ref = s.softmax(-1) @ vf
        O = torch.empty(B * Nq, H * dv, device=dev, dtype=BF16)
        LSE = torch.empty(B, H, Nq, device=dev)
        st = (Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv)
        ops.attn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, LSE, B, H, Nq, Nk, dqk, dv, *st, Nq * H * dv, H * dv, scale)
        tag = f'attn B{B} H{H} {Nq}x{Nk} d{dqk}/{dv}'
        check(tag + ' fwd', rel(O.view(B, Nq, H, dv).permute(0, 2, 1, 3).float(), ref), 1.2e-2)
        check(tag + ' lse', rel(LSE, torch.logsumexp(s, -1)), 2e-4)
        dO = torch.randn(B * Nq, H * dv, device=dev).to(BF16)
        ref.backward(dO.view(B, Nq, H, dv).permute(0, 2, 1, 3).float())
        dq, dk, dvv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        Delta = torch.empty_like(LSE)
        ops.attn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, dO, LSE, Delta, dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), B, H, Nq, Nk,
                     dqk, dv, *st, Nq * H * dv, H * dv, Nq * H * dv, H * dv, *st, scale)
        check(tag + ' dq', rel(dq.permute(0, 2, 1, 3).float(), qf.grad), 2.5e-2)
        check(tag + ' dk', rel(dk.permute(0, 2, 1, 3).float(), kf.grad), 2.5e-2)
        check(tag + ' dv', rel(dvv.permute(0, 2, 1, 3).float(), vf.grad), 2.5e-2)


def fuzz_attn_drop(rng, n):
    """Attention dropout (dav_attn_drop_fwd / _bwd): random lengths / head widths / probabilities / mask row strides, against torch with the
    same keep mask."""
    for it in range(n):
        dqk, dv = rng.choice([(64, 64), (32, 32), (16, 64), (16, 16)])
        B, H = rng.randint(1, 3), rng.randint(1, 4)
        Nk = rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 64, 100, 255, 256, 257, 320, 511, 513, 700, rng.randint(1, 1200)])
        Nq = rng.choice([1, 5, 16, 17, 33, 64, 129, rng.randint(1, 600)])
        pd = rng.choice([0.05, 0.1, 0.25, 0.5, 0.9])
        ld = (Nk + 31) // 32 * 32 + 4 * rng.choice([0, 0, 1, 8])          # any multiple of 4 from Nk rounded up to 32
        keep = (torch.rand(B, H, Nq, ld, device=dev) >= pd).to(torch.uint8)
        km = keep[..., :Nk].float() / (1.0 - pd)
        q = torch.randn(B, Nq, H, dqk, device=dev).to(BF16)
        k = torch.randn(B, Nk, H, dqk, device=dev).to(BF16)
        v = torch.randn(B, Nk, H, dv, device=dev).to(BF16)
        scale = rng.choice([dqk ** -0.5, 0.125])
        qf, kf, vf = (t.float().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
        s = (qf @ kf.transpose(-2, -1)) * scale
        ref = (s.softmax(-1) * km) @ vf
        O = torch.empty(B * Nq, H * dv, device=dev, dtype=BF16)
        LSE = torch.empty(B, H, Nq, device=dev)
        st = (Nq * H * dqk, H * dqk, Nk * H * dqk, H * dqk, Nk * H * dv, H * dv)
        ops.attn_drop_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, LSE, B, H, Nq, Nk, dqk, dv, *st, Nq * H * dv, H * dv, scale,
                          keep, ld, 1.0 / (1.0 - pd))
        tag = f'attn_drop B{B} H{H} {Nq}x{Nk} d{dqk}/{dv} p{pd} ld{ld}'
        check(tag + ' fwd', rel(O.view(B, Nq, H, dv).permute(0, 2, 1, 3).float(), ref), 1.2e-2)
        check(tag + ' lse', rel(LSE, torch.logsumexp(s, -1)), 2e-4)
        dO = torch.randn(B * Nq, H * dv, device=dev).to(BF16)
        ref.backward(dO.view(B, Nq, H, dv).permute(0, 2, 1, 3).float())
        dq, dk, dvv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
        Delta = torch.empty_like(LSE)
        ops.attn_drop_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), O, dO, LSE, Delta, dq.data_ptr(), dk.data_ptr(), dvv.data_ptr(), B, H, Nq, Nk,
                          dqk, dv, *st, Nq * H * dv, H * dv, Nq * H * dv, H * dv, *st, scale, keep, ld, 1.0 / (1.0 - pd))
        check(tag + ' dq', rel(dq.permute(0, 2, 1, 3).float(), qf.grad), 2.5e-2)
        check(tag + ' dk', rel(dk.permute(0, 2, 1, 3).float(), kf.grad), 2.5e-2)
        check(tag + ' dv', rel(dvv.permute(0, 2, 1, 3).float(), vf.grad), 2.5e-2)


def fuzz_ln(rng, n):
    for it in range(n):
        B = rng.randint(1, 5)
        r0, r1 = rng.choice([0, 1, 3, 32]), rng.choice([1, 2, 7, 49, 196, rng.randint(1, 400)])
        D = 4 * rng.choice([1, 8, 32, 48, 96, 128, 192, 256, 320])
        x0 = torch.randn(B, r0, D, device=dev) if r0 else None
        x1 = torch.randn(B, r1, D, device=dev)
        g, bt = torch.randn(D, device=dev) * 0.2 + 1, torch.randn(D, device=dev) * 0.2
        xc = torch.cat([t for t in (x0, x1) if t is not None], 1).clone().requires_grad_(True)
        gp, bp = g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        eps = rng.choice([1e-5, 1e-6])
        ref = torch.nn.functional.layer_norm(xc, (D,), gp, bp, eps)
        R = r0 + r1
        y, y32 = torch.empty(B * R, D, device=dev, dtype=BF16), torch.empty(B * R, D, device=dev)
        mean, rstd = torch.empty(B * R, device=dev), torch.empty(B * R, device=dev)
        a0, a1, n0, n1 = (x0, x1, r0, r1) if x0 is not None else (x1, None, r1, 0)
        ops.layernorm_fwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, g, bt, eps, y, y32, mean, rstd)
        tag = f'ln B{B} {r0}+{r1} D{D}'
        check(tag + ' fwd', rel(y32, ref.view(-1, D)), 2e-5)
        dy = torch.randn(B * R, D, device=dev).to(BF16)
        ref.backward(dy.view(B, R, D).float())
        dx0 = torch.zeros(B, n0, D, device=dev)
        dx1 = torch.zeros(B, max(n1, 1), D, device=dev) if n1 else None
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        ops.layernorm_bwd(a0, n0 * D, n0, a1, n1 * D, n1, B, D, dy, None, g, mean, rstd, dx0=dx0, dx0_bs=n0 * D, dx1=dx1,
                          dx1_bs=n1 * D, dgamma=dg, dbeta=db)
        dxc = torch.cat([t for t in (dx0, dx1) if t is not None], 1)
        check(tag + ' dx', rel(dxc, xc.grad), 3e-5)
        check(tag + ' dgamma', rel(dg, gp.grad), 3e-4)
        check(tag + ' dbeta', rel(db, bp.grad), 3e-4)


def fuzz_ln_fused(rng, n):
    """The LayerNorm-folding chain on random shapes: a producer GEMM (fp32 result + residual through an optional row map, any tile
    configuration that has the statistics) writes twin + row-statistics partials; a consumer GEMM (optionally two twin sources: rows of a
    stand-alone dav_rowstats_cast in front of the producer's) contracts them with gamma-folded weights; dav_layernorm_bwd_twin re-makes
    the LayerNorm output.  Each against torch fp32 on the same operands."""
    ln = torch.nn.functional.layer_norm
    for it in range(n):
        D = 64 * rng.choice([1, 2, 3, 8, 12, 16])
        B = rng.randint(1, 6)
        r1 = rng.choice([1, 7, 16, 49, 64, 130, rng.randint(1, 300)])
        r0 = rng.choice([0, 0, 3, 16])
        Kp = 64 * rng.choice([1, 2, 4, 12])
        N2 = 8 * rng.choice([8, 24, 96, 288, rng.randint(1, 200)])
        M1 = B * r1
        cfg = rng.choice([0, 0, 3, 5, 7, 8])
        eps = rng.choice([1e-5, 1e-6])
        # producer: x1 = A . W^T + b + res  (+ twin + statistics)
        A = torch.randn(M1, Kp, device=dev).to(BF16)
        W = (torch.randn(D, Kp, device=dev) * 0.05).to(BF16)
        bias, res = torch.randn(D, device=dev), torch.randn(M1, D, device=dev) * rng.choice([0.1, 1.0, 3.0]) + rng.choice([0.0, 0.5])
        x1 = torch.empty(M1, D, device=dev)
        tw1, st1 = torch.empty(M1, D, device=dev, dtype=BF16), torch.empty(M1, D // 64, 2, device=dev)
        ops.gemm_nt_ln(A, W, M1, D, Kp, prod=dict(stats_out=st1, twin_out=tw1, ld_twin=D), bias=bias, res=res, ldres=D, C_out=x1, variant=cfg << 4)
        ref1 = A.float() @ W.float().t() + bias + res
        tag = f'ln_fused B{B} {r0}+{r1} D{D} Kp{Kp} N{N2} cfg{cfg}'
        check(tag + ' producer C', rel(x1, ref1), 1e-4)
        check(tag + ' twin', float((tw1.float() - x1.to(BF16).float()).abs().max()), 0.0)
        v = x1.double().view(M1, D // 64, 64)
        check(tag + ' sums', rel(st1, torch.stack([v.sum(-1), (v * v).sum(-1)], -1)), 5e-6)
        # optional first segment from the stand-alone kernel
        segs = []
        x0 = None
        if r0:
            x0 = torch.randn(B, r0, D, device=dev) * 1.5 + 0.3
            tw0, st0 = torch.empty(B * r0, D, device=dev, dtype=BF16), torch.empty(B * r0, D // 64, 2, device=dev)
            ops.rowstats_cast(x0, r0 * D, B, r0, D, tw0, st0)
            segs.append((tw0, st0, r0))
        segs.append((tw1, st1, r1))
        R, M = r0 + r1, B * (r0 + r1)
        w32 = torch.randn(N2, D, device=dev) * 0.05
        g, bt, b2 = torch.randn(D, device=dev) * 0.2 + 1, torch.randn(D, device=dev) * 0.2, torch.randn(N2, device=dev)
        wl, c, d = torch.empty(N2, D, device=dev, dtype=BF16), torch.empty(N2, device=dev), torch.empty(N2, device=dev)
        ops.ln_fold_grouped([(w32, g, bt, b2, wl, c, d)])
        xc = torch.cat(([x0] if r0 else []) + [x1.view(B, r1, D)], 1)
        ref2 = ln(xc, (D,), g, bt, eps).view(M, D) @ w32.t() + b2
        out = torch.empty(M, N2, device=dev, dtype=BF16 if rng.random() < 0.5 else torch.float32)
        lnd = dict(stats=segs[0][1], ln_c=c, eps=eps)
        if r0:
            lnd.update(A2=segs[1][0], stats2=segs[1][1], a_r0=r0, a_r1=r1)
        ops.gemm_nt_ln(segs[0][0], wl, M, N2, D, ln=lnd, bias=d, C_out=out, c_bf16=out.dtype == BF16, variant=rng.choice([0, 0, 3, 5, 8, 44]) << 4)
        check(tag + ' consumer', rel(out, ref2), 1.2e-2)
        # backward from the twins (reference: autograd on the twins' values)
        xt = torch.cat([tw.float().view(B, r, D) for (tw, st_, r) in segs], 1).clone().requires_grad_(True)
        gp, bp = g.clone().requires_grad_(True), bt.clone().requires_grad_(True)
        refl = ln(xt, (D,), gp, bp, eps)
        dy = torch.randn(M, D, device=dev).to(BF16)
        refl.backward(dy.float().view(B, R, D))
        (t0, s0, n0) = segs[0]
        (t1, s1, n1) = segs[1] if len(segs) == 2 else (None, None, 0)
        dx0 = torch.zeros(B, n0, D, device=dev)
        dx1 = torch.zeros(B, max(n1, 1), D, device=dev)[:, :n1].contiguous() if n1 else None
        dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        h = torch.empty(M, D, device=dev, dtype=BF16)
        ops.layernorm_bwd_twin(t0, n0 * D, s0, n0, t1, n1 * D, s1, n1, B, D, eps, dy, None, g, bt, dx0, n0 * D, 0, None, 0, None, 0,
                               dx1, n1 * D, 0, None, 0, None, 0, h_out=h, dgamma=dg, dbeta=db)
        dxc = torch.cat([t for t in (dx0, dx1) if t is not None], 1)
        check(tag + ' bwd dx', rel(dxc, xt.grad), 6e-3)           # (statistics of the fp32 rows, x their bf16 rounding)
        check(tag + ' bwd dgamma', rel(dg, gp.grad), 6e-3)
        check(tag + ' bwd dbeta', rel(db, bp.grad), 3e-4)
        check(tag + ' h_out', rel(h, refl.detach().view(M, D)), 8e-3)


def fuzz_gang(rng, n):
    """dav_gemm_tn_gang_bf16 on random problem LISTS: 1 .. 90 problems per launch, contraction lengths from one row to a few thousand
    (mostly ragged), N / K any multiple of 8 (below, across and beyond one 256 x 256 tile), written and accumulated tiles mixed, bias
    gradients on some, row maps (rows-per-batch windows of a taller tensor) on some operands, column blocks of a wider gradient."""
    for it in range(n):
        probs, refs = [], []
        for j in range(rng.choice([1, 1, 2, 3, 5, 8, 13, 24, 24, 45, 90])):      # (> 28 problems: several table-writer launches)
            N = 8 * rng.choice([1, 2, 31, 32, 33, 64, 96, 100, rng.randint(1, 130)])
            K = 8 * rng.choice([1, 3, 32, 33, 64, 65, 96, rng.randint(1, 130)])
            if rng.random() < 0.06:              # a weight of more than 32 tiles: the planner cuts its tile grid into several gangs
                N, K = 8 * rng.randint(260, 520), 8 * rng.randint(130, 400)
            mapped = rng.random() < 0.35
            if mapped:
                Bsz, tot = rng.choice([1, 2, 7, 16, 32]), rng.randint(2, 120)
                rpb = rng.randint(1, tot)
                offa, offb = rng.randint(0, tot - rpb), rng.randint(0, tot - rpb)
                Mc = Bsz * rpb
                Af, Bf = torch.randn(Bsz * tot, N, device=dev).to(BF16), torch.randn(Bsz * tot, K, device=dev).to(BF16)
                As = Af.view(Bsz, tot, N)[:, offa:offa + rpb].reshape(-1, N)
                Bs = Bf.view(Bsz, tot, K)[:, offb:offb + rpb].reshape(-1, K)
                maps = dict(a_rowmap=(rpb, tot, offa), b_rowmap=(rpb, tot, offb))
            else:
                Mc = rng.choice([1, 7, 63, 64, 65, 128, 200, 640, 1000, rng.randint(1, 3000)])
                Af = As = torch.randn(Mc, N, device=dev).to(BF16)
                Bf = Bs = torch.randn(Mc, K, device=dev).to(BF16)
                maps = dict(a_rowmap=None, b_rowmap=None)
            ow = rng.random() < 0.5
            wide = rng.random() < 0.25                      # the problem's gradient is a column block of a wider tensor
            ldc = K + (8 * rng.randint(1, 20) if wide else 0)
            Cw = torch.full((N, ldc), 0.25, device=dev)
            if ow:
                Cw[:, :K] = float('nan')
            bg = torch.full((N,), 0.5, device=dev) if rng.random() < 0.4 else None
            probs.append(dict(A=Af, B=Bf, Mc=Mc, N=N, K=K, C=Cw, lda=N, ldb=K, ldc=ldc, bias_grad=bg, overwrite=ow, **maps))
            ref = torch.full((N, ldc), 0.25, device=dev)
            ref[:, :K] = (0.0 if ow else 0.25) + As.float().t() @ Bs.float()
            refs.append((Cw, ref, bg, None if bg is None else 0.5 + As.float().sum(0), (Mc, N, K, mapped, ow, wide)))
        ops.gemm_tn_gang(probs)
        for j, (C, rc, bg, rb, what) in enumerate(refs):
            check(f'gang #{it}.{j} {what} C', rel(C, rc), 2e-4)
            if bg is not None:
                check(f'gang #{it}.{j} {what} bias', rel(bg, rb), 2e-4)


if __name__ == '__main__':
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ng, na, nl, ngg = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 150), (3, 60), (4, 60), (5, 60)))
    rng = random.Random(seed)
    torch.manual_seed(seed)
    fuzz_gemm(rng, ng)
    fuzz_attn(rng, na)
    fuzz_attn_drop(rng, na)
    fuzz_ln(rng, nl)
    fuzz_ln_fused(rng, nl)
    fuzz_gang(rng, ngg)
    print(f'fuzz seed {seed}: {len(FAILS)} failures')
    for f in FAILS[:20]:
        print('  ', f)
    sys.exit(1 if FAILS else 0)
