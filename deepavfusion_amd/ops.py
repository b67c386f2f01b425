"""Thin torch-tensor wrappers over the C ABI (include/dav_kernels.h).

PyTorch is used here only for device memory and streams: every wrapper passes raw
device pointers plus torch's *current* HIP stream to the HIP library, so the calls
are ordered with surrounding torch work and can be captured in a hipGraph.
There is no fallback path: a missing library or a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


# While a launch batch is being recorded (engine.batch / dav_batch_begin) kernels run LATER than the Python code that
# "launches" them, so every tensor handed to the library is kept alive until the batch has been issued: torch's caching
# allocator would otherwise hand a temporary's block to a later allocation of the same recording, and with lanes issued
# in lockstep that later kernel may run BEFORE the temporary's last reader.
import threading

_TLS = threading.local()          # .hold: list while THIS thread records a batch (the library's batch state is per thread too:
                                  # autograd runs the backward — and its batches — on its own thread)


def hold(*tensors):
    h = getattr(_TLS, 'hold', None)
    if h is not None:
        h.extend(t for t in tensors if t is not None)


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('deepavfusion_amd kernels need tensors on an MI355X (cuda) device; there is no CPU fallback')
    h = getattr(_TLS, 'hold', None)
    if h is not None:
        h.append(t)
    return t.data_ptr()


def batch_begin(auto_lanes=False):
    _lib.check(_lib.load().dav_batch_begin(int(auto_lanes)), 'dav_batch_begin')
    _TLS.hold = []


def batch_lane():
    _lib.check(_lib.load().dav_batch_lane(), 'dav_batch_lane')


def batch_skip(steps):
    _lib.check(_lib.load().dav_batch_skip(int(steps)), 'dav_batch_skip')


def batch_region(begin):
    _lib.check(_lib.load().dav_batch_region(int(begin)), 'dav_batch_region')


class unbatched:
    """Launches inside go out at once even while a batch is being recorded (weight-cache refreshes: nothing recorded
    so far depends on them, the launches recorded next read their result)."""

    def __enter__(self):
        self.on = getattr(_TLS, 'hold', None) is not None
        if self.on:
            _lib.load().dav_batch_suspend(1)

    def __exit__(self, *exc):
        if self.on:
            _lib.load().dav_batch_suspend(0)
        return False


def batch_end(abort=False):
    """Issue (or drop) the recorded launches; returns (recorded, issued) launch counts."""
    lib = _lib.load()
    try:
        if abort:
            lib.dav_batch_abort()
            return 0, 0
        _lib.check(lib.dav_batch_end(), 'dav_batch_end')
        a, b = C.c_int(0), C.c_int(0)
        lib.dav_batch_stats(C.byref(a), C.byref(b))
        return a.value, b.value
    finally:
        _TLS.hold = None


def _rm(m: Optional[Sequence[int]]):
    return (C.c_int * 3)(*m) if m is not None else None


def gemm_nt(A, B, M, N, K, *, lda=None, ldb=None, a_rowmap=None, bias=None, act=0, aux=None, ldaux=0, res=None, ldres=0,
            res_rowmap=None, res_rows=None, C_out=None, ldc=None, c_bf16=False, c_rowmap=None, C2=None, ldc2=0, c2_mode=0,
            beta=0, alpha=1.0, variant=0):
    """C[M,N] = epi(A[M,K] . B[N,K]^T); see dav_gemm_nt_bf16 (fp32 operands: dav_gemm_nt_f32, csrc/f32_path.hip)."""
    lib = _lib.load()
    if A.dtype == F32:                  # (c_bf16 is meaningless here: every output of the fp32 path is fp32)
        _lib.check(lib.dav_gemm_nt_f32(_ptr(A), _ptr(B), M, N, K, lda if lda is not None else K, ldb if ldb is not None else K,
                                       _rm(a_rowmap), _ptr(bias), act, _ptr(aux), ldaux, _ptr(res), ldres, _rm(res_rowmap),
                                       _ptr(res_rows), _ptr(C_out), ldc if ldc is not None else N, _rm(c_rowmap), _ptr(C2), ldc2,
                                       c2_mode, beta, float(alpha), (variant >> 12) & 1, _stream()), 'dav_gemm_nt_f32')
        return
    _lib.check(lib.dav_gemm_nt_bf16(_ptr(A), _ptr(B), M, N, K, lda if lda is not None else K, ldb if ldb is not None else K,
                                    _rm(a_rowmap), _ptr(bias), act, _ptr(aux), ldaux, _ptr(res), ldres, _rm(res_rowmap),
                                    _ptr(res_rows), _ptr(C_out), ldc if ldc is not None else N, int(c_bf16), _rm(c_rowmap),
                                    _ptr(C2), ldc2, c2_mode, beta, float(alpha), variant, _stream()), 'dav_gemm_nt_bf16')


def gemm_nt_ln(A, B, M, N, K, *, ln=None, prod=None, lda=None, ldb=None, a_rowmap=None, bias=None, act=0, aux=None, ldaux=0, res=None,
               ldres=0, res_rowmap=None, res_rows=None, C_out=None, ldc=None, c_bf16=False, c_rowmap=None, C2=None, ldc2=0,
               c2_mode=0, beta=0, variant=0):
    """dav_gemm_nt_ln_bf16 (bf16 path only).  ``ln`` = dict(stats, ln_c, eps[, A2, stats2, a_r0, a_r1]): A holds RAW twin rows, B the
    gamma-folded weight, ``bias`` the folded bias (consumer side).  ``prod`` = dict(stats_out=..., twin_out=..., ld_twin=...):
    row-statistics partials + bf16 twin of the fp32 result (producer side)."""
    lib = _lib.load()
    q = _lib.DavNtLn()
    if ln is not None:
        q.stats, q.ln_c, q.eps = _ptr(ln['stats']), _ptr(ln['ln_c']), float(ln['eps'])
        q.stats2, q.A2 = _ptr(ln.get('stats2')), _ptr(ln.get('A2'))
        q.a_r0, q.a_r1 = int(ln.get('a_r0', 0)), int(ln.get('a_r1', 0))
    if prod is not None:
        q.stats_out, q.twin_out, q.ld_twin = _ptr(prod.get('stats_out')), _ptr(prod.get('twin_out')), int(prod.get('ld_twin', N))
    _lib.check(lib.dav_gemm_nt_ln_bf16(_ptr(A), _ptr(B), M, N, K, lda if lda is not None else K, ldb if ldb is not None else K,
                                       _rm(a_rowmap), _ptr(bias), act, _ptr(aux), ldaux, _ptr(res), ldres, _rm(res_rowmap),
                                       _ptr(res_rows), _ptr(C_out), ldc if ldc is not None else N, int(c_bf16), _rm(c_rowmap),
                                       _ptr(C2), ldc2, c2_mode, beta, 1.0, variant, C.byref(q), _stream()), 'dav_gemm_nt_ln_bf16')


LN_FOLD_MAX = 48          # dav_ln_fold_grouped: pairs per launch


def ln_fold_grouped(items):
    """items: [(w fp32 [N, K], gamma, beta, bias or None, w_ln_bf16, ln_c, ln_d), ...] (dav_ln_fold_grouped)."""
    lib = _lib.load()
    for i in range(0, len(items), LN_FOLD_MAX):
        chunk = items[i:i + LN_FOLD_MAX]
        arr = (_lib.DavLnFold * len(chunk))()
        for q, (w, g, b, bias, wl, c, d) in zip(arr, chunk):
            q.w, q.gamma, q.beta, q.bias, q.w_ln_bf16, q.ln_c, q.ln_d = _ptr(w), _ptr(g), _ptr(b), _ptr(bias), _ptr(wl), _ptr(c), _ptr(d)
            q.N, q.K = int(w.shape[0]), int(w.shape[1])
        _lib.check(lib.dav_ln_fold_grouped(arr, len(chunk), _stream()), 'dav_ln_fold_grouped')


def rowstats_cast(x, x_bs, B, rows, D, twin, stats):
    """bf16 twin + per-row {sum, sum of squares} partials over 64-column slots of fp32 rows (dav_rowstats_cast)."""
    _lib.check(_lib.load().dav_rowstats_cast(_ptr(x), x_bs, B, rows, D, _ptr(twin), _ptr(stats), _stream()), 'dav_rowstats_cast')


def layernorm_bwd_twin(xb0, xb0_bs, st0, r0, xb1, xb1_bs, st1, r1, B, D, eps, dy_bf16, dy_f32, gamma, beta,
                       dx0=None, dx0_bs=0, acc0=0, res0=None, res0_bs=0, dx0_bf16=None, dx0_bf_bs=0,
                       dx1=None, dx1_bs=0, acc1=0, res1=None, res1_bs=0, dx1_bf16=None, dx1_bf_bs=0, h_out=None, dgamma=None, dbeta=None,
                       defer=None):
    """dav_layernorm_bwd_twin; xb* / st* / dx* may be (tensor, element offset) pairs.  ``defer`` as layernorm_bwd."""
    lib = _lib.load()

    def po(t):
        if t is None:
            return None
        if isinstance(t, tuple):
            return _ptr(t[0]) + t[0].element_size() * t[1]
        return _ptr(t)
    ws = None
    if dgamma is not None:
        ws = torch.empty(lib.dav_layernorm_bwd_workspace_bytes(B * (r0 + r1), D) // 4, dtype=F32, device=dgamma.device)
        if defer is not None:
            defer.append((ws, dgamma, dbeta, B * (r0 + r1), D))
            dgamma = dbeta = None
    _lib.check(lib.dav_layernorm_bwd_twin(po(xb0), xb0_bs, po(st0), r0, po(xb1), xb1_bs, po(st1), r1, B, D, float(eps),
                                          _ptr(dy_bf16), _ptr(dy_f32), _ptr(gamma), _ptr(beta),
                                          po(dx0), dx0_bs, acc0, po(res0), res0_bs, po(dx0_bf16), dx0_bf_bs,
                                          po(dx1), dx1_bs, acc1, po(res1), res1_bs, po(dx1_bf16), dx1_bf_bs,
                                          _ptr(h_out), _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel() * 4 if ws is not None else 0,
                                          _stream()), 'dav_layernorm_bwd_twin')


def nt_issue_log(enable=None, with_flags=False):
    """enable True / False: start (clearing) / stop logging the NT launches the library issues; None: fetch the log as a
    list of (cfg, b_kn, [(M, N, K), ...]) — one entry per launch (grouped launches have several problems); with_flags adds a
    fourth element, the problems' epilogue flags (act | c_bf16 << 2 | has_res << 3 | c2_mode << 4 | beta << 8)."""
    lib = _lib.load()
    if enable is not None:
        lib.dav_nt_issue_log(int(enable), None, 0)
        return None
    n = -lib.dav_nt_issue_log(0, (C.c_int * 1)(), 0)
    if n <= 0:
        return []
    buf = (C.c_int * n)()
    lib.dav_nt_issue_log(0, buf, n)
    out, i = [], 0
    while i < n:
        cfg, bt, cnt = buf[i], buf[i + 1], buf[i + 2]
        probs = [(buf[i + 3 + 4 * j], buf[i + 4 + 4 * j], buf[i + 5 + 4 * j]) for j in range(cnt)]
        flags = [buf[i + 6 + 4 * j] for j in range(cnt)]
        out.append((cfg, bt, probs, flags) if with_flags else (cfg, bt, probs))
        i += 3 + 4 * cnt
    return out


def gemm_tn(A, B, Mc, N, K, C_out, *, lda=None, ldb=None, ldc=None, a_rowmap=None, b_rowmap=None, beta=1, bias_grad=None,
            variant=0):
    """C[N,K] (+)= A[Mc,N]^T . B[Mc,K]; see dav_gemm_tn_bf16 (fp32 operands: dav_gemm_tn_f32)."""
    lib = _lib.load()
    if A.dtype == F32:
        _lib.check(lib.dav_gemm_tn_f32(_ptr(A), _ptr(B), Mc, N, K, lda if lda is not None else N, ldb if ldb is not None else K,
                                       _rm(a_rowmap), _rm(b_rowmap), _ptr(C_out), ldc if ldc is not None else K, beta,
                                       _ptr(bias_grad), _stream()), 'dav_gemm_tn_f32')
        return
    _lib.check(lib.dav_gemm_tn_bf16(_ptr(A), _ptr(B), Mc, N, K, lda if lda is not None else N, ldb if ldb is not None else K,
                                    _rm(a_rowmap), _rm(b_rowmap), _ptr(C_out), ldc if ldc is not None else K, beta,
                                    _ptr(bias_grad), variant, _stream()), 'dav_gemm_tn_bf16')


TN_GROUP_MAX = 40          # dav_gemm_tn_grouped_bf16: problems per launch


def gemm_tn_grouped(problems):
    """problems: list of dicts(A, B, Mc, N, K, C, lda, ldb, ldc, a_rowmap, b_rowmap, bias_grad[, overwrite]); see
    dav_gemm_tn_grouped_bf16 (overwrite: C is written instead of accumulated, DavTnProblem.flags bit 0)."""
    lib = _lib.load()
    if problems and problems[0]['A'].dtype == F32:      # fp32 path: one launch per problem
        for d in problems:
            gemm_tn(d['A'], d['B'], d['Mc'], d['N'], d['K'], d['C'], lda=d['lda'], ldb=d['ldb'], ldc=d['ldc'],
                    a_rowmap=d.get('a_rowmap'), b_rowmap=d.get('b_rowmap'), beta=0 if d.get('overwrite') else 1,
                    bias_grad=d.get('bias_grad'))
        return
    # more than one launch (40 problems each at most): deal the list out in turns, so that every launch gets its share of the long
    # and of the short contractions — cut consecutively, 68 problems became 32 long + 32 medium + 4 small ones whose ~100 tiles
    # had the 512 workgroup slots to themselves for a whole contraction
    n_launch = (len(problems) + TN_GROUP_MAX - 1) // TN_GROUP_MAX
    for i in range(n_launch):
        chunk = problems[i::n_launch]
        _lib.check(lib.dav_gemm_tn_grouped_bf16(_tn_problem_array(chunk), len(chunk), _stream()), 'dav_gemm_tn_grouped_bf16')


def _tn_problem_array(problems):
    arr = (_lib.DavTnProblem * len(problems))()
    for q, d in zip(arr, problems):
        q.A, q.B, q.C, q.bias_grad = _ptr(d['A']), _ptr(d['B']), _ptr(d['C']), _ptr(d.get('bias_grad'))
        q.Mc, q.N, q.K, q.lda, q.ldb, q.ldc = d['Mc'], d['N'], d['K'], d['lda'], d['ldb'], d['ldc']
        q.a_rowmap[:] = d.get('a_rowmap') or (0, 0, 0)
        q.b_rowmap[:] = d.get('b_rowmap') or (0, 0, 0)
        q.flags = 1 if d.get('overwrite') else 0
    return arr


def gemm_tn_gang(problems):
    """The queued weight-gradient problems of several layers (dicts as for gemm_tn_grouped, no two with the same C) as ONE
    gang-scheduled launch of 256 x 256 tiles (dav_gemm_tn_gang_bf16).  The workspace is allocated here on the current stream."""
    if not problems:
        return
    if problems[0]['A'].dtype == F32:
        return gemm_tn_grouped(problems)
    lib = _lib.load()
    arr = _tn_problem_array(problems)
    nbytes = int(lib.dav_gemm_tn_gang_workspace_bytes(arr, len(problems)))
    if nbytes == 0:
        raise RuntimeError('dav_gemm_tn_gang_workspace_bytes: invalid weight-gradient problem (shape / alignment)')
    ws = torch.empty(nbytes, dtype=torch.uint8, device=problems[0]['A'].device)
    hold(ws)
    _lib.check(lib.dav_gemm_tn_gang_bf16(arr, len(problems), _ptr(ws), nbytes, _stream()), 'dav_gemm_tn_gang_bf16')


def attn_fwd(q_ptr, k_ptr, v_ptr, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale):
    """q / k / v are raw addresses (views into fused projection buffers) of the dtype of ``O``."""
    lib = _lib.load()
    if O.dtype == F32:
        _lib.check(lib.dav_attn_fwd_f32(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(LSE), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                                        v_bs, v_rs, o_bs, o_rs, float(scale), _stream()), 'dav_attn_fwd_f32')
        return
    _lib.check(lib.dav_attn_fwd(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(LSE), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                                v_bs, v_rs, o_bs, o_rs, float(scale), _stream()), 'dav_attn_fwd')


def attn_bias_fwd(q_ptr, k_ptr, v_ptr, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                  bias, bias_nb, bias_ld):
    """attn_fwd with an additive logit bias [bias_nb, H, Nq, bias_ld] (bf16 path: log2 units; fp32 path: natural units)."""
    lib = _lib.load()
    fn, name = (lib.dav_attn_bias_fwd_f32, 'dav_attn_bias_fwd_f32') if O.dtype == F32 else (lib.dav_attn_bias_fwd, 'dav_attn_bias_fwd')
    _lib.check(fn(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(LSE), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                  float(scale), _ptr(bias), bias_nb, bias_ld, _stream()), name)


def attn_bias_bwd(q_ptr, k_ptr, v_ptr, O, dO, LSE, Delta, dq_ptr, dk_ptr, dv_ptr, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                  v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, bias, bias_nb, bias_ld,
                  dS, part=3):
    """attn_bwd with the bias of the forward; dS [B, H, Nq, bias_ld] fp32 receives the gradient of the biased logits."""
    lib = _lib.load()
    fn, name = (lib.dav_attn_bias_bwd_f32, 'dav_attn_bias_bwd_f32') if O.dtype == F32 else (lib.dav_attn_bias_bwd, 'dav_attn_bias_bwd')
    _lib.check(fn(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr, B, H, Nq, Nk, dqk, dv,
                  q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs,
                  float(scale), _ptr(bias), bias_nb, bias_ld, _ptr(dS), part, _stream()), name)


def window_unfold(src, rows32, B, nW, A, nF, L, Cc, fusion_scale, out):
    """[B, nF + L, C] rows -> [B * nW, A + nF, C] window sequences (see dav_window_unfold)."""
    _lib.check(_lib.load().dav_window_unfold(_ptr(src), int(src.dtype == BF16), _ptr(rows32), B, nW, A, nF, L, Cc, float(fusion_scale),
                                             _ptr(out), int(out.dtype == BF16), _stream()), 'dav_window_unfold')


def window_fold(t, inv32, res, B, nW, A, nF, L, Cc, fusion_scale, out):
    """[B * nW, A + nF, C] fp32 sequences (+ res) -> [B, nF + L, C] fp32 rows (see dav_window_fold)."""
    _lib.check(_lib.load().dav_window_fold(_ptr(t), _ptr(inv32), _ptr(res), B, nW, A, nF, L, Cc, float(fusion_scale), _ptr(out),
                                           _stream()), 'dav_window_fold')


def relpos_bias_build(table, index32, mask, nb, H, A, N, ld, mul, out):
    _lib.check(_lib.load().dav_relpos_bias_build(_ptr(table), _ptr(index32), _ptr(mask), nb, H, A, N, ld, float(mul), _ptr(out),
                                                 _stream()), 'dav_relpos_bias_build')


def relpos_bias_bwd(dS, index32, Bw, H, A, N, ld, T, dtable):
    _lib.check(_lib.load().dav_relpos_bias_bwd(_ptr(dS), _ptr(index32), Bw, H, A, N, ld, T, _ptr(dtable), _stream()),
               'dav_relpos_bias_bwd')


def attn_bwd(q_ptr, k_ptr, v_ptr, O, dO, LSE, Delta, dq_ptr, dk_ptr, dv_ptr, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
             v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, part=3, dq_ctx_rows=0):
    """part 1: dQ (+ Delta) kernel only, 2: dK/dV kernel only (after part 1), 3: both.  dq_ctx_rows (bf16 path): rows in front of
    the first query row of every batch element of the dQ buffer whose dqk columns the dQ kernel zero-fills (dav_attn_bwd_ctx)."""
    lib = _lib.load()
    if O.dtype == F32:
        if dq_ctx_rows:
            raise RuntimeError('dq_ctx_rows: bf16 path only')
        _lib.check(lib.dav_attn_bwd_f32(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr,
                                        B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs,
                                        dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, float(scale), part, _stream()), 'dav_attn_bwd_f32')
        return
    if dq_ctx_rows:
        _lib.check(lib.dav_attn_bwd_ctx(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr,
                                        B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs,
                                        dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, float(scale), int(dq_ctx_rows), part, _stream()), 'dav_attn_bwd_ctx')
        return
    _lib.check(lib.dav_attn_bwd_part(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr,
                                     B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs,
                                     dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, float(scale), part, _stream()), 'dav_attn_bwd')


def layernorm_fwd(x0, x0_bs, r0, x1, x1_bs, r1, B, D, gamma, beta, eps, y_bf16, y_f32, mean, rstd):
    lib = _lib.load()
    if y_bf16 is not None and y_bf16.dtype == F32:      # fp32 path: the "operand copy" of the output is fp32 too
        if y_f32 is not None and y_f32.data_ptr() != y_bf16.data_ptr():
            raise RuntimeError('fp32 LayerNorm: pass one output buffer')
        y_bf16, y_f32 = None, y_bf16
    _lib.check(lib.dav_layernorm_fwd(_ptr(x0), x0_bs, r0, _ptr(x1), x1_bs, r1, B, D, _ptr(gamma), _ptr(beta), float(eps),
                                     _ptr(y_bf16), _ptr(y_f32), _ptr(mean), _ptr(rstd), _stream()), 'dav_layernorm_fwd')


def layernorm_bwd(x0, x0_bs, r0, x1, x1_bs, r1, B, D, dy_bf16, dy_f32, gamma, mean, rstd,
                  dx0=None, dx0_bs=0, acc0=0, res0=None, res0_bs=0, dx0_bf16=None, dx0_bf_bs=0,
                  dx1=None, dx1_bs=0, acc1=0, res1=None, res1_bs=0, dx1_bf16=None, dx1_bf_bs=0, dgamma=None, dbeta=None,
                  defer=None):
    """``defer``: a list; when given, only the partial dgamma/dbeta rows are produced and
    (workspace, dgamma, dbeta, rows, D) is appended for a later ``layernorm_bwd_reduce_grouped``."""
    lib = _lib.load()
    twins = []
    if dy_bf16 is not None and dy_bf16.dtype == F32:    # fp32 path: both gradient inputs are fp32 -> one (summed) fp32 input
        if dy_f32 is not None:
            tmp = torch.empty_like(dy_bf16)
            _lib.check(lib.dav_add_f32(_ptr(dy_bf16), _ptr(dy_f32), _ptr(tmp), tmp.numel(), _stream()), 'dav_add_f32')
            dy_f32 = tmp
        else:
            dy_f32 = dy_bf16
        dy_bf16 = None
    if dx0_bf16 is not None and dx0_bf16.dtype == F32:  # ... and the "bf16 twins" of the outputs are fp32 row copies
        twins.append((dx0, dx0_bs, r0, dx0_bf16, dx0_bf_bs))
        dx0_bf16 = None
    if dx1_bf16 is not None and dx1_bf16.dtype == F32:
        twins.append((dx1, dx1_bs, r1, dx1_bf16, dx1_bf_bs))
        dx1_bf16 = None
    ws = None
    if dgamma is not None:
        ws = torch.empty(lib.dav_layernorm_bwd_workspace_bytes(B * (r0 + r1), D) // 4, dtype=F32, device=dgamma.device)
        if defer is not None:
            defer.append((ws, dgamma, dbeta, B * (r0 + r1), D))
            dgamma = dbeta = None
    _lib.check(lib.dav_layernorm_bwd(_ptr(x0), x0_bs, r0, _ptr(x1), x1_bs, r1, B, D, _ptr(dy_bf16), _ptr(dy_f32),
                                     _ptr(gamma), _ptr(mean), _ptr(rstd),
                                     _ptr(dx0), dx0_bs, acc0, _ptr(res0), res0_bs, _ptr(dx0_bf16), dx0_bf_bs,
                                     _ptr(dx1), dx1_bs, acc1, _ptr(res1), res1_bs, _ptr(dx1_bf16), dx1_bf_bs,
                                     _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel() * 4 if ws is not None else 0, _stream()),
               'dav_layernorm_bwd')
    for (dx, bs, rows, twin, twin_bs) in twins:
        _lib.check(lib.dav_rows_gather_f32(_ptr(dx), bs, 0, None, B, rows, D, _ptr(twin), twin_bs, _stream()), 'dav_rows_gather_f32')


def layernorm_bwd_reduce_grouped(items):
    lib = _lib.load()
    for i in range(0, len(items), 64):
        chunk = items[i:i + 64]
        arr = (_lib.DavLnReduce * len(chunk))()
        for q, (ws, dg, db, rows, D) in zip(arr, chunk):
            q.workspace, q.dgamma, q.dbeta, q.rows, q.D = _ptr(ws), _ptr(dg), _ptr(db), rows, D
        _lib.check(lib.dav_layernorm_bwd_reduce_grouped(arr, len(chunk), _stream()), 'dav_layernorm_bwd_reduce_grouped')


def mask_build(noise: torch.Tensor, len_keep: int):
    """AVMAE.random_masking for given noise -> (ids_keep i64, mask f32, ids_restore i64, ids_keep i32, ids_restore i32)."""
    lib = _lib.load()
    N, L = noise.shape
    dev = noise.device
    ids_keep = torch.empty(N, len_keep, dtype=torch.int64, device=dev)
    ids_restore = torch.empty(N, L, dtype=torch.int64, device=dev)
    mask = torch.empty(N, L, dtype=F32, device=dev)
    ids_keep32 = torch.empty(N, len_keep, dtype=torch.int32, device=dev)
    ids_restore32 = torch.empty(N, L, dtype=torch.int32, device=dev)
    _lib.check(lib.dav_mask_build(_ptr(noise.contiguous()), N, L, len_keep, _ptr(ids_keep), _ptr(ids_restore), _ptr(mask),
                                  _ptr(ids_keep32), _ptr(ids_restore32), _stream()), 'dav_mask_build')
    return ids_keep, mask, ids_restore, ids_keep32, ids_restore32


def patch_gather(img, ids_keep32, nk, out, pt=1):
    """img [B,C,H,W] with 16x16 patches, or a clip [B,C,T,H,W] with (pt,16,16) tubelets."""
    lib = _lib.load()
    if out.dtype == F32:
        if img.dim() == 5:
            B, Cc, T, H, W = img.shape
        else:
            (B, Cc, H, W), T = img.shape, 1
        _lib.check(lib.dav_patch_gather_f32(_ptr(img), B, Cc, T, H, W, pt, _ptr(ids_keep32), nk, _ptr(out), _stream()), 'dav_patch_gather_f32')
        return
    if img.dim() == 5:
        B, Cc, T, H, W = img.shape
        _lib.check(lib.dav_patch_gather3d(_ptr(img), B, Cc, T, H, W, pt, _ptr(ids_keep32), nk, _ptr(out), _stream()),
                   'dav_patch_gather3d')
        return
    B, Cc, H, W = img.shape
    _lib.check(lib.dav_patch_gather(_ptr(img), B, Cc, H, W, _ptr(ids_keep32), nk, _ptr(out), _stream()), 'dav_patch_gather')


def unshuffle_fwd(emb, mask_token, pos, ids_restore32, B, L, nk, D, out, out_bs, out_row_off):
    lib = _lib.load()
    _lib.check(lib.dav_unshuffle_fwd(_ptr(emb), _ptr(mask_token), _ptr(pos), _ptr(ids_restore32), B, L, nk, D, _ptr(out),
                                     out_bs, out_row_off, _stream()), 'dav_unshuffle_fwd')


def rows_gather_cast(x, x_bs, row_off, ids32, B, n, D, out):
    lib = _lib.load()
    if out.dtype == F32:
        _lib.check(lib.dav_rows_gather_f32(_ptr(x), x_bs, row_off, _ptr(ids32), B, n, D, _ptr(out), 0, _stream()), 'dav_rows_gather_f32')
        return
    _lib.check(lib.dav_rows_gather_cast(_ptr(x), x_bs, row_off, _ptr(ids32), B, n, D, _ptr(out), _stream()), 'dav_rows_gather_cast')


def unshuffle_bwd_reduce(dx, dx_bs, row_off, ids_restore32, B, L, nk, D, dpos, dmask_token):
    lib = _lib.load()
    _lib.check(lib.dav_unshuffle_bwd_reduce(_ptr(dx), dx_bs, row_off, _ptr(ids_restore32), B, L, nk, D, _ptr(dpos),
                                            _ptr(dmask_token), _stream()), 'dav_unshuffle_bwd_reduce')


def patch_mse_fwd(img, pred, mask, norm_pix, loss_patch, tmean, trstd, loss, mask_sum):
    lib = _lib.load()
    B, Cc, H, W = img.shape
    _lib.check(lib.dav_patch_mse_fwd(_ptr(img), _ptr(pred), _ptr(mask), B, Cc, H, W, int(norm_pix), _ptr(loss_patch),
                                     _ptr(tmean), _ptr(trstd), _ptr(loss), _ptr(mask_sum), _stream()), 'dav_patch_mse_fwd')


def patch_mse_bwd(img, pred, mask, tmean, trstd, mask_sum, gout, dpred_bf16):
    lib = _lib.load()
    B, Cc, H, W = img.shape
    if dpred_bf16.dtype == F32:
        _lib.check(lib.dav_patch_mse_bwd_f32(_ptr(img), _ptr(pred), _ptr(mask), _ptr(tmean), _ptr(trstd), _ptr(mask_sum), _ptr(gout),
                                             B, Cc, H, W, _ptr(dpred_bf16), _stream()), 'dav_patch_mse_bwd_f32')
        return
    _lib.check(lib.dav_patch_mse_bwd(_ptr(img), _ptr(pred), _ptr(mask), _ptr(tmean), _ptr(trstd), _ptr(mask_sum), _ptr(gout),
                                     B, Cc, H, W, _ptr(dpred_bf16), _stream()), 'dav_patch_mse_bwd')


def pair_expand(Pv, Pa, B, nv, na, Wd, out):
    lib = _lib.load()
    if out.dtype == F32:
        _lib.check(lib.dav_pair_expand_f32(_ptr(Pv), _ptr(Pa), B, nv, na, Wd, _ptr(out), _stream()), 'dav_pair_expand_f32')
        return
    _lib.check(lib.dav_pair_expand(_ptr(Pv), _ptr(Pa), B, nv, na, Wd, _ptr(out), _stream()), 'dav_pair_expand')


def pair_reduce(d, B, nv, na, Wd, dPv, dPa):
    lib = _lib.load()
    if d.dtype == F32:
        _lib.check(lib.dav_pair_reduce_f32(_ptr(d), B, nv, na, Wd, _ptr(dPv), _ptr(dPa), _stream()), 'dav_pair_reduce_f32')
        return
    _lib.check(lib.dav_pair_reduce(_ptr(d), B, nv, na, Wd, _ptr(dPv), _ptr(dPa), _stream()), 'dav_pair_reduce')


def cast_bf16(x, y):
    lib = _lib.load()
    _lib.check(lib.dav_cast_bf16(_ptr(x), _ptr(y), x.numel(), _stream()), 'dav_cast_bf16')


def cast_transpose_bf16(x2d, y):
    lib = _lib.load()
    R, Cc = x2d.shape
    _lib.check(lib.dav_cast_transpose_bf16(_ptr(x2d), _ptr(y), R, Cc, _stream()), 'dav_cast_transpose_bf16')


def l2norm(x_flat, out, workspace, scale=1.0):
    lib = _lib.load()
    _lib.check(lib.dav_l2norm(_ptr(x_flat), x_flat.numel(), float(scale), _ptr(out), _ptr(workspace),
                              workspace.numel() * workspace.element_size(), _stream()), 'dav_l2norm')


def adamw_flat(p, g, m, v, p_bf16, seg_end, hyper, nseg, beta1, beta2, eps, bias_corr, grad_scale=1.0, sumsq_out=None,
               zero_grad=False, keep_grad=None, gscale_dev=None):
    """keep_grad: uint8 [nseg] or None — segments whose gradient is NOT zero-filled by zero_grad (see dav_adamw_flat).
    gscale_dev: float32 [1] device scalar from ``step_guard`` (clip factor; 0 = skip the update) or None."""
    lib = _lib.load()
    _lib.check(lib.dav_adamw_flat(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(p_bf16), p.numel(), _ptr(seg_end), _ptr(hyper), nseg,
                                  float(beta1), float(beta2), float(eps), _ptr(bias_corr), float(grad_scale), _ptr(sumsq_out),
                                  int(zero_grad), _ptr(keep_grad), _ptr(gscale_dev), _stream()), 'dav_adamw_flat')


def add_cast(a, b):
    """(a + b as fp32, the same as bf16) in one pass (dav_add_cast); a, b: contiguous fp32 of the same shape, numel % 4 == 0."""
    lib = _lib.load()
    out = torch.empty_like(a)
    out_b = torch.empty(a.shape, dtype=BF16, device=a.device)
    _lib.check(lib.dav_add_cast(_ptr(a), _ptr(b), _ptr(out), _ptr(out_b), a.numel(), _stream()), 'dav_add_cast')
    return out, out_b


def step_guard(loss_a, loss_b, gnorm, clip, grad_scale, out_scale, bad_count):
    """Device-side replacement of train.py:166-167 / util/misc.py:118-120 for a captured step (dav_step_guard)."""
    lib = _lib.load()
    _lib.check(lib.dav_step_guard(_ptr(loss_a), _ptr(loss_b), _ptr(gnorm), float(clip if clip else 0.0), float(grad_scale),
                                  _ptr(out_scale), _ptr(bad_count), _stream()), 'dav_step_guard')


def rows_axpy(res, y, scale, B, rows, D, out):
    """out[b, r] = res[b, r] + scale[b] * y[b, r]  (fp32 rows of D; DropPath forward)."""
    lib = _lib.load()
    _lib.check(lib.dav_rows_axpy(_ptr(res), _ptr(y), _ptr(scale), B, rows, D, _ptr(out), _stream()), 'dav_rows_axpy')


def rows_scale_cast(g, scale, B, rows, D, out_bf16):
    """out_bf16[b, r] = bf16(scale[b] * g[b, r])  (DropPath backward, branch side)."""
    lib = _lib.load()
    _lib.check(lib.dav_rows_scale_cast(_ptr(g), _ptr(scale), B, rows, D, _ptr(out_bf16), _stream()), 'dav_rows_scale_cast')


def attn_drop_fwd(q_ptr, k_ptr, v_ptr, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                  keep, keep_ld, keep_scale):
    """Attention with dropout on the softmax probabilities (dav_attn_drop_fwd / _f32): keep = bytes 0 / 1 [B, H, Nq, keep_ld]."""
    lib = _lib.load()
    fn, name = (lib.dav_attn_drop_fwd_f32, 'dav_attn_drop_fwd_f32') if O.dtype == F32 else (lib.dav_attn_drop_fwd, 'dav_attn_drop_fwd')
    _lib.check(fn(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(LSE), B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                  float(scale), _ptr(keep), keep_ld, float(keep_scale), _stream()), name)


def attn_drop_bwd(q_ptr, k_ptr, v_ptr, O, dO, LSE, Delta, dq_ptr, dk_ptr, dv_ptr, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs,
                  v_bs, v_rs, o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, keep, keep_ld, keep_scale,
                  part=3, dq_ctx_rows=0):
    lib = _lib.load()
    if O.dtype == F32:
        if dq_ctx_rows:
            raise ValueError('the fp32 attention kernels have no context-row entry')
        _lib.check(lib.dav_attn_drop_bwd_f32(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr,
                                             B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs,
                                             dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, float(scale), _ptr(keep), keep_ld,
                                             float(keep_scale), part, _stream()), 'dav_attn_drop_bwd_f32')
        return
    _lib.check(lib.dav_attn_drop_bwd(q_ptr, k_ptr, v_ptr, _ptr(O), _ptr(dO), _ptr(LSE), _ptr(Delta), dq_ptr, dk_ptr, dv_ptr,
                                     B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, do_bs, do_rs,
                                     dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, float(scale), _ptr(keep), keep_ld, float(keep_scale),
                                     dq_ctx_rows, part, _stream()), 'dav_attn_drop_bwd')


def dropout_rows(x, keep, keep_scale, B, rows, D, out, res=None, rowscale=None):
    """out[b, r] = (res[b, r] if res else 0) + (rowscale[b] if rowscale else 1) * keep[b, r] * keep_scale * x[b, r]
    (dav_dropout_rows; x / out bf16 or fp32, keep bytes [B * rows, D] or None; out may alias x or res)."""
    lib = _lib.load()
    _lib.check(lib.dav_dropout_rows(_ptr(x), int(x.dtype == F32), _ptr(res), _ptr(keep), float(keep_scale), _ptr(rowscale), B, rows, D,
                                    _ptr(out), int(out.dtype == F32), _stream()), 'dav_dropout_rows')


def cast_transpose_grouped(pairs):
    """pairs: [(x fp32 [R, C], y bf16 [C, R]), ...] -> every y = bf16(x^T) in one launch (dav_cast_transpose_grouped)."""
    if not pairs:
        return
    arr = (_lib.DavTranspose * len(pairs))()
    for q, (x, y) in zip(arr, pairs):
        q.x, q.y_bf16, q.R, q.C = _ptr(x), _ptr(y), int(x.shape[0]), int(x.shape[1])
    _lib.check(_lib.load().dav_cast_transpose_grouped(arr, len(pairs), _stream()), 'dav_cast_transpose_grouped')
