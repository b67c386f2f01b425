"""GPU: every HIP kernel, called through the C ABI, against a torch fp32 reference of the same op
(tolerances are in tests/gpu_selfcheck.py next to each check; index work is bit-exact)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['gemm_nt', 'gemm_tn', 'gemm_tn_gang', 'attention', 'dropout', 'window_attention', 'layernorm', 'ln_fused', 'masking', 'misc_kernels', 'patch_gather3d'])
def test_kernel_family(name):
    import gpu_selfcheck as sc
    sc.RESULTS.clear()
    getattr(sc, name)()
    bad = [r for r in sc.RESULTS if not r[3]]
    assert sc.RESULTS and not bad, bad[:10]


def test_kernel_shape_fuzz():
    """A fixed-seed slice of tests/gpu_fuzz.py: random GEMM shapes with random epilogue options (+ their weight gradients),
    random attention lengths / head widths (with and without dropout masks), random two-segment LayerNorms, random LayerNorm-folding chains (producer GEMM -> consumer GEMM -> backward from the twin), random weight-gradient problem lists through the gang launch —
    each against a torch fp32 reference."""
    import random

    import torch

    import gpu_fuzz as fz
    fz.FAILS.clear()
    rng = random.Random(7)
    torch.manual_seed(7)
    fz.fuzz_gemm(rng, 60)
    fz.fuzz_attn(rng, 25)
    fz.fuzz_attn_drop(rng, 15)
    fz.fuzz_ln(rng, 25)
    fz.fuzz_ln_fused(rng, 25)
    fz.fuzz_gang(rng, 25)
    assert not fz.FAILS, fz.FAILS[:10]


def test_batched_grouped_launches_equal_individual_launches():
    """csrc/batch.h: launches recorded in lanes and issued in lockstep — GEMMs / LayerNorms / attentions of equal rank as
    ONE grouped grid — must give bit-identical results to issuing the same launches one by one, and really group."""
    import torch

    from deepavfusion_amd import engine as E
    from deepavfusion_amd import ops
    torch.manual_seed(3)
    dev = 'cuda'
    bf, f32 = torch.bfloat16, torch.float32

    def chain(M, n_keys, B, H, tag):
        """LN -> qkv GEMM (bias) -> attention -> proj GEMM (+ residual, bf16 twin) -> LN backward -> b_kn dgrad GEMM:
        one lane.  Shapes differ per lane (M rows, tile counts, workgroup sizes)."""
        D, hd = 128, 64
        g = torch.Generator(device=dev).manual_seed(100 + M)
        R = M // B
        x = torch.randn(B, R, D, device=dev, generator=g)
        gamma, beta = torch.randn(D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
        Wqkv = (torch.randn(3 * D, D, device=dev, generator=g) * 0.1).to(bf)
        bqkv = torch.randn(3 * D, device=dev, generator=g)
        Wp = (torch.randn(D, D, device=dev, generator=g) * 0.1).to(bf)
        dy = torch.randn(M, D, device=dev, generator=g).to(bf)
        out = {}

        def run():
            y = torch.empty(M, D, dtype=bf, device=dev)
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            ops.layernorm_fwd(x, R * D, R, None, 0, 0, B, D, gamma, beta, 1e-6, y, None, mean, rstd)
            qkv = torch.empty(M, 3 * D, dtype=bf, device=dev)
            ops.gemm_nt(y, Wqkv, M, 3 * D, D, bias=bqkv, C_out=qkv, c_bf16=True)
            O = torch.empty(M, D, dtype=bf, device=dev)
            LSE = torch.empty(B, H, R, device=dev)
            ops.hold(qkv)
            ops.attn_fwd(qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, O, LSE, B, H, R, R, hd, hd,
                         R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * D, D, hd ** -0.5)
            x1 = torch.empty(M, D, device=dev)
            x1b = torch.empty(M, D, dtype=bf, device=dev)
            ops.gemm_nt(O, Wp, M, D, D, res=x.view(M, D), ldres=D, C_out=x1, C2=x1b, ldc2=D, c2_mode=3)
            dq = torch.zeros(M, 3 * D, dtype=bf, device=dev)
            Delta = torch.empty_like(LSE)
            dO = dy
            ops.hold(dq)
            ops.attn_bwd(qkv.data_ptr(), qkv.data_ptr() + 2 * D, qkv.data_ptr() + 4 * D, O, dO, LSE, Delta,
                         dq.data_ptr(), dq.data_ptr() + 2 * D, dq.data_ptr() + 4 * D, B, H, R, R, hd, hd,
                         R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * D, D, R * D, D,
                         R * 3 * D, 3 * D, R * 3 * D, 3 * D, R * 3 * D, 3 * D, hd ** -0.5)
            dh = torch.empty(M, D, dtype=bf, device=dev)
            ops.gemm_nt(dq, Wqkv, M, D, 3 * D, lda=3 * D, ldb=D, C_out=dh, c_bf16=True, variant=1 << 12)     # dgrad reading W itself
            dx = torch.empty(B, R, D, device=dev)
            dxb = torch.empty(M, D, dtype=bf, device=dev)
            dgam, dbet = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
            ops.layernorm_bwd(x, R * D, R, None, 0, 0, B, D, dh, None, gamma, mean, rstd,
                              dx1=None, dx0=dx, dx0_bs=R * D, dx0_bf16=dxb, dx0_bf_bs=R * D, dgamma=dgam, dbeta=dbet)
            out.update(y=y, qkv=qkv, O=O, LSE=LSE, x1=x1, x1b=x1b, dq=dq, dh=dh, dx=dx, dxb=dxb, dgam=dgam, dbet=dbet)
        return run, out

    lanes = [chain(64 * 49, 49, 64, 2, 'a'), chain(32 * 80, 80, 32, 2, 'b'), chain(4 * 16, 16, 4, 2, 'c')]
    for run, _ in lanes:
        run()
    torch.cuda.synchronize()
    ref = [{k: v.clone() for k, v in out.items()} for _, out in lanes]
    E.BATCH_STATS[:] = [0, 0]
    with E.batch() as bt:
        for run, _ in lanes:
            bt.lane()
            run()
    torch.cuda.synchronize()
    rec, iss = E.BATCH_STATS
    assert rec >= 3 * 9 and iss <= rec // 3 + 2, (rec, iss)          # three equal chains -> one grid per rank
    for (_, out), r in zip(lanes, ref):
        for k in r:
            assert torch.equal(out[k], r[k]), k
    # a region of mutually independent launches (auto lanes): three GEMMs of different shapes -> one grid
    E.BATCH_STATS[:] = [0, 0]
    A = [torch.randn(m, 256, device=dev).to(bf) for m in (520, 64, 1000)]
    W = [(torch.randn(n, 256, device=dev) * 0.1).to(bf) for n in (192, 768, 64)]
    single = []
    for a, w in zip(A, W):
        c = torch.empty(a.shape[0], w.shape[0], device=dev)
        ops.gemm_nt(a, w, a.shape[0], w.shape[0], 256, C_out=c)
        single.append(c)
    grouped = [torch.empty_like(c) for c in single]
    with E.batch(auto_lanes=True):
        for a, w, c in zip(A, W, grouped):
            ops.gemm_nt(a, w, a.shape[0], w.shape[0], 256, C_out=c)
    torch.cuda.synchronize()
    assert E.BATCH_STATS == [3, 1], E.BATCH_STATS
    for c, r in zip(grouped, single):
        assert torch.equal(c, r)
    # a region inside a lane: its launches form ONE step and merge with the other lane's launch of that step
    E.BATCH_STATS[:] = [0, 0]
    again = [torch.empty_like(c) for c in single]
    tail = torch.empty_like(single[0])
    with E.batch() as bt:
        bt.lane()
        with E.region():
            ops.gemm_nt(A[0], W[0], A[0].shape[0], W[0].shape[0], 256, C_out=again[0])
            ops.gemm_nt(A[1], W[1], A[1].shape[0], W[1].shape[0], 256, C_out=again[1])
        ops.gemm_nt(A[0], W[0], A[0].shape[0], W[0].shape[0], 256, C_out=tail, res=again[0], ldres=W[0].shape[0])   # depends on step 1
        bt.lane()
        ops.gemm_nt(A[2], W[2], A[2].shape[0], W[2].shape[0], 256, C_out=again[2])
    torch.cuda.synchronize()
    assert E.BATCH_STATS == [4, 2], E.BATCH_STATS
    for c, r in zip(again, single):
        assert torch.equal(c, r)
    assert torch.equal(tail, single[0] + single[0])


def test_tuned_table_picks_the_configuration_and_results_do_not_change():
    """dav_nt_tune_set: a recorded group whose signature is in the table is issued with the table's tile configuration (seen in
    the library's own issue log), any other group by the rules — and the numbers are the same either way."""
    import ctypes as C
    import torch
    from deepavfusion_amd import _lib, engine as E, ops
    lib = _lib.load()
    dev = 'cuda'
    torch.manual_seed(0)
    probs = [(1536, 1024, 256), (2048, 1024, 256)]
    ops_in = [(torch.randn(M, K, device=dev).bfloat16(), (torch.randn(N, K, device=dev) * 0.05).bfloat16(),
               torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for (M, N, K) in probs]

    def run():
        ops.nt_issue_log(True)
        with E.batch() as bt:
            for (M, N, K), (A, W, Cc) in zip(probs, ops_in):
                bt.lane()
                ops.gemm_nt(A, W, M, N, K, C_out=Cc, c_bf16=True)
        log = ops.nt_issue_log(with_flags=True)
        ops.nt_issue_log(False)
        torch.cuda.synchronize()
        return log, [c.clone() for _, _, c in ops_in]
    try:
        assert lib.dav_nt_tune_set(None, 0) == 0
        log0, out0 = run()
        assert len(log0) == 1 and len(log0[0][2]) == 2                  # ONE grouped launch of both problems
        rule_cfg, flags = log0[0][0], log0[0][3]
        want = 46 if rule_cfg != 46 else 44
        blob = [want, 0, 2]
        for (M, N, K), f in zip(probs, flags):
            blob += [M, N, K, f]
        assert lib.dav_nt_tune_set((C.c_int * len(blob))(*blob), len(blob)) == 1
        log1, out1 = run()
        assert log1[0][0] == want
        for a, b in zip(out0, out1):
            assert torch.equal(a, b)                                    # same k order per element: bit-identical
        # a different group (other M) is not in the table -> the rule's choice again
        ops_in[0] = (torch.randn(1664, 256, device=dev).bfloat16(), ops_in[0][1], torch.empty(1664, 1024, device=dev, dtype=torch.bfloat16))
        probs[0] = (1664, 1024, 256)
        log2, _ = run()
        assert log2[0][0] == rule_cfg
    finally:
        _lib.load_nt_tuning(_lib.NT_TUNING_PATH)


def test_two_grouped_weight_gradient_launches_recorded_in_one_batch():
    """dav_gemm_tn_grouped_bf16 staged its problem table in static storage and a recorded launch captured
    it by reference — two grouped calls inside one dav_batch_begin .. dav_batch_end both ran with the LAST table.  Each must
    keep its own: the results equal the two launches issued one by one."""
    import torch

    from deepavfusion_amd import ops
    dev, bf = 'cuda', torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(5)

    def problem(Mc, N, K):
        A = torch.randn(Mc, N, device=dev, generator=g).to(bf)
        B = torch.randn(Mc, K, device=dev, generator=g).to(bf)
        return dict(A=A, B=B, Mc=Mc, N=N, K=K, C=None, lda=N, ldb=K, ldc=K, a_rowmap=None, b_rowmap=None, bias_grad=None)
    sets = [[problem(256, 128, 192), problem(320, 64, 128)], [problem(192, 256, 64), problem(128, 192, 320), problem(64, 128, 128)]]
    outs = {}
    for mode in ('single', 'batched'):
        for ps in sets:
            for d in ps:
                d['C'] = torch.zeros(d['N'], d['K'], device=dev)
        if mode == 'batched':
            ops.batch_begin()
        for ps in sets:
            ops.gemm_tn_grouped(ps)
        if mode == 'batched':
            ops.batch_end()
        torch.cuda.synchronize()
        outs[mode] = [d['C'].clone() for ps in sets for d in ps]
    for a, b, d in zip(outs['single'], outs['batched'], [d for ps in sets for d in ps]):
        ref = d['A'].float().t() @ d['B'].float()
        assert float((a - ref).norm() / ref.norm()) < 1e-5
        assert torch.equal(a, b)
