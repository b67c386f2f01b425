/* C ABI of libdavfusion_hip.so — the MI355X (gfx950) kernels behind the DeepAVFusion / AVMAE
 * pre-training step.
 *
 * The reference (stoneMo/DeepAVFusion) has no FFI of its own: every device op on this path is a
 * stock ATen call issued from Python.  This header is the boundary one level below the Python
 * modules that mirror the reference's classes: each entry point replaces the ATen/cuDNN/cuBLAS/SDPA
 * work behind the reference lines cited on it (paths relative to the reference repo).
 *
 * Conventions:
 *  - plain pointers and sizes only; the caller owns every buffer (inputs, outputs, workspaces);
 *    the library never allocates, frees or keeps a pointer past return;
 *  - all work is enqueued on `stream`; no host synchronisation, no host callbacks; safe to call from
 *    several host threads on distinct streams, and inside hipGraph stream capture;
 *  - return 0 on success, a negative DAV_ERR_* code otherwise (never aborts).  Arguments are validated BEFORE any HIP call:
 *    empty input (a size of 0) is a bad shape (-1) for every kernel entry point, an unsupported dtype combination -2, a
 *    workspace that is too small -3, a HIP launch failure -4 (text: dav_last_error_string), a misaligned pointer or stride
 *    -5; a refused call has enqueued nothing (tests/test_cabi_and_host.py checks every entry point, without a GPU);
 *  - "bf16" buffers are raw uint16 bfloat16; index tensors are int64 where the reference exposes
 *    them (ids_keep / ids_restore) with int32 twins for in-kernel use; row strides are in elements.
 *  - a "row map" is an int[3] {rows_per_batch, batch_stride_rows, row_offset} (NULL = identity):
 *    logical row m -> physical row (m / rpb) * bs + off + m % rpb.  It lets a GEMM read or write a
 *    sub-range of a [B, rows, D] activation without cat/split copies.
 */
#ifndef DAV_KERNELS_H_
#define DAV_KERNELS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
typedef struct ihipStream_t* hipStream_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define DAV_ABI_VERSION 9   /* 9: dropout (dav_attn_drop_fwd / _bwd (+ _f32), dav_dropout_rows); 8: LayerNorm folded into the GEMMs either side of it (dav_gemm_nt_ln_bf16, dav_ln_fold_grouped, dav_rowstats_cast, dav_layernorm_bwd_twin); 2: DavTnProblem.flags, dav_adamw_flat keep_grad + gscale_dev, dav_step_guard; 3: dav_attn_bwd_ctx, dav_add_cast; 4: fused fusion tails, grouped cast-transpose; 5: dav_gemm_tn_grouped_adamw_bf16; 6: dav_gemm_tn_gang_bf16; 7: the fused fusion tails and dav_gemm_tn_grouped_adamw_bf16 are gone (measured slower, DESIGN_HISTORY section 11) */
int dav_abi_version(void);
int dav_build_flags(void);   /* bit 0: experimental build (make EXPERIMENTAL=1): the rejected GEMM tile configurations exist */
/* text of the last HIP error latched by a kernel launch of the calling thread (diagnostics) */
const char* dav_last_error_string(void);
/* launch-geometry knobs for tuning experiments (1: LayerNorm-backward waves per workgroup {2,4,8}; 2: its grid cap;
 * 3: query tiles per wave of the attention forward, 0 = automatic, 1, 2) */
int dav_tune(int knob, int value);

/* ---- GEMM --------------------------------------------------------------------------------- */
/* C[M,N] = epi(alpha * A[M,K] . B[N,K]^T): every nn.Linear forward on the path (timm Attention.qkv/proj,
 * Mlp.fc1/fc2; models/fusion_blocks.py:41-44 q/kv/proj, :227-232 q/k/v/proj; models/avmae.py:31,59
 * decoder_embed, :60,88 decoder_pred; the patch-embed conv of models/vits.py:27 as a GEMM over
 * gathered patches) and, with B = W^T, their input gradients.
 * epilogue order: v = alpha*acc + bias[n]; [C2 mode 1]; act 1 = exact (erf) GELU [C2 mode 4: the twin is GELU'(v)
 * of the pre-activation v], act 2 = v *= GELU'(aux[m,n]), act 3 = v *= aux[m,n] (fc1 backward, with aux = the pre-activation
 * or the mode-4 twin of the forward); [C2 mode 2]; + fp32 residual (row map or explicit row list: the block residual adds
 * and "+ pos_embed[ids_keep]"); beta != 0 adds the old fp32 C; store C (fp32 or bf16, through
 * c_rowmap; may be NULL); [C2 mode 3].  C2 is a dense bf16 [M, ldc2] twin for the next GEMM.
 * variant bit12: B is given as [K, N] row-major (ldb = row stride; K % 64 == 0, N % 8 == 0) — the dgrad reads
 * W itself through transposing LDS reads, no W^T copy; bits 4-11: explicit tile configuration (benchmarks);
 * bit0: register-staged loads instead of global->LDS DMA; bit1: force 64x64 tiles (first-generation kernel). */
int dav_gemm_nt_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const int* a_rowmap,
                     const float* bias, int act, const void* aux, int ldaux, const float* res, int ldres,
                     const int* res_rowmap, const int* res_rows, void* C, int ldc, int c_is_bf16, const int* c_rowmap,
                     void* C2, int ldc2, int c2_mode, int beta, float alpha, int variant, hipStream_t stream);

/* dav_gemm_nt_bf16 with the LayerNorm between two GEMMs folded into them (ABI 8) — timm Block.norm1 / norm2 in front of attn.qkv /
 * mlp.fc1 (models/vits.py:32-34, models/avmae.py:53-57, 83-87), norm1_img / norm1_aud / norm2 of the fusion blocks
 * (models/fusion_blocks.py:281-288), decoder_norm in front of decoder_pred (models/avmae.py:59-60,177-178): no LayerNorm launch and no
 * normalised tensor in memory on the forward path.
 *  PRODUCER side (stats_out / twin_out; C must be fp32, N % 64 == 0): besides C the epilogue writes, for every row, the partial sums
 *   stats_out[crow * (N / 64) + s] = {sum, sum of squares} of the FINAL fp32 values of columns [64 s, 64 s + 64) (crow = the C row after
 *   c_rowmap) and the bf16 twin twin_out[crow * ld_twin + n] of the final value.  Fixed summation order, no atomics.
 *  CONSUMER side (stats): A holds RAW rows — such a twin — and B the weight with the LayerNorm's gamma folded in (dav_ln_fold_grouped:
 *   B[n, k] = bf16(gamma[k] W[n, k]) from the fp32 master W, ln_c[n] = sum_k B[n, k], bias := b[n] + sum_k beta[k] W[n, k]); the epilogue starts from
 *   v = rstd[m] * (acc[m, n] - mean[m] * ln_c[n]) + bias[n] with mean / rstd formed from the K / 64 partial sums of row m
 *   (mean = sum / K, var = sumsq / K - mean^2 clamped at 0, rstd = 1 / sqrt(var + eps)); K % 64 == 0, K <= 1024, alpha == 1.
 *   The statistics row of A row m is the physical A row (through a_rowmap).  a_r0 > 0: every batch element's rows are a_r0 rows of
 *   (A, stats) followed by a_r1 rows of (A2, stats2), both dense and with row stride lda — the cat((x_fusion, x_mod)) of
 *   models/deepavfusion.py:104-105; a_rowmap must be NULL and M a multiple of a_r0 + a_r1.
 * Everything else as dav_gemm_nt_bf16 (b_kn and explicit first-generation variants are refused). */
typedef struct DavNtLn {
  const float* stats; const float* stats2; const void* A2; const float* ln_c; float eps; int a_r0, a_r1;   /* consumer */
  float* stats_out; void* twin_out; int ld_twin;                                                          /* producer */
} DavNtLn;
int dav_gemm_nt_ln_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb, const int* a_rowmap,
                        const float* bias, int act, const void* aux, int ldaux, const float* res, int ldres,
                        const int* res_rowmap, const int* res_rows, void* C, int ldc, int c_is_bf16, const int* c_rowmap,
                        void* C2, int ldc2, int c2_mode, int beta, float alpha, int variant, const DavNtLn* ln, hipStream_t stream);

/* Diagnostics for benchmarks: log which launches the NT family issues (grouped or single).  enable 1 / 0 with out == NULL
 * starts (and clears) / stops logging (process-wide: autograd runs the backward on its own thread); with out != NULL the log is copied: entries of
 * {tile configuration, b_kn, n, n x (M, N, K, epilogue flags)}; returns the number of ints, or -needed when capacity is too small.
 * flags = act | c_is_bf16 << 2 | has_res << 3 | c2_mode << 4 | beta << 8. */
int dav_nt_issue_log(int enable, int* out, int capacity);

/* Tuned tile-configuration table for grouped NT launches (the role cuBLAS' / hipBLASLt's per-size solution tables play
 * behind torch's F.linear in the reference, models/vits.py:32-34): `blob` holds entries in dav_nt_issue_log's format; a
 * group recorded between dav_batch_begin / dav_batch_end whose b_kn + sorted (M, N, K, flags) list matches an entry is issued
 * with that entry's tile configuration, any other group by the built-in rules.  n_ints == 0 clears.  Returns the number of
 * entries held, or a negative DAV_ERR_*.  Written by tools/mix_sweep.py, loaded by deepavfusion_amd/_lib.py. */
int dav_nt_tune_set(const int* blob, int n_ints);

/* C[N,K] (+)= A[Mc,N]^T . B[Mc,K] in fp32: the weight gradient of every nn.Linear above (autograd of
 * F.linear).  beta != 0 accumulates into the live gradient (split over the contraction with fp32
 * atomics); bias_grad (optional) receives the column sums of A (atomic accumulate).
 * variant bit0: scalar LDS reads instead of ds_read_b64_tr_b16. */
int dav_gemm_tn_bf16(const void* A, const void* B, int Mc, int N, int K, int lda, int ldb, const int* a_rowmap,
                     const int* b_rowmap, float* C, int ldc, int beta, float* bias_grad, int variant, hipStream_t stream);

/* Several weight-gradient problems (C_i[N_i,K_i] += A_i^T . B_i, optional bias_grad_i) in ONE launch — the deferred
 * wgrads of a whole layer — so that every 128x128 output tile is owned by one workgroup over (most of) its
 * contraction: no or very few split-K atomics.  count <= 40; every Mc_i % 64 == 0.  Row maps as above
 * ({0,0,0} = identity).  The table is read on the host during the call only. */
typedef struct DavTnProblem {
  const void* A; const void* B;       /* bf16 [Mc, N] (lda), bf16 [Mc, K] (ldb) */
  float* C; float* bias_grad;         /* fp32 [N, K] (ldc), fp32 [N] or NULL; both accumulated */
  int Mc, N, K, lda, ldb, ldc;
  int a_rowmap[3], b_rowmap[3];
  int flags;                          /* bit 0: C is WRITTEN, its old contents ignored (the first contribution to a gradient
                                         that the optimizer pass did not zero-fill, see dav_adamw_flat keep_grad); bias_grad
                                         is accumulated regardless.  Such a problem is never split over the contraction. */
} DavTnProblem;
int dav_gemm_tn_grouped_bf16(const DavTnProblem* problems, int count, hipStream_t stream);

/* The same problems (any number up to 4096: the queued weight gradients of SEVERAL layers, autograd of every nn.Linear on the path —
 * models/fusion_blocks.py:41-44,227-232, timm Block of models/vits.py:32-34, models/avmae.py:53-60,83-88) as ONE persistent launch of
 * 256 x 256 tiles, one workgroup per CU: every problem's tile grid is cut into gangs of <= 32 tiles that share operand panels, the
 * gangs are dealt out to eight per-XCD ticket queues (longest contraction first) and the workgroups of an XCD draw the tiles of one gang
 * together, so that a panel crosses the fabric once per XCD instead of once per tile.  One owner per tile over the whole contraction:
 * no atomics on C, bit-repeatable.  Any Mc > 0 (a contraction that is not a multiple of 64 rows — B x tokens at batch 32 — is padded with
 * zeros on the fly).  flags bit 0 as above; other bits: DAV_ERR_SHAPE.  Two problems of one call must not
 * address the same C.  workspace: caller-owned device memory of dav_gemm_tn_gang_workspace_bytes(problems, count) bytes (0 = invalid
 * problems), 16-byte aligned, written and read on `stream` only (it must stay untouched until the launch has run). */
size_t dav_gemm_tn_gang_workspace_bytes(const DavTnProblem* problems, int count);
int dav_gemm_tn_gang_bf16(const DavTnProblem* problems, int count, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- attention ---------------------------------------------------------------------------- */
/* softmax(scale * Q K^T) V per (batch, head); element (b, n, h, d) of X is X[b*x_bs + n*x_rs + h*dX + d].
 * (dqk, dv) in {(64,64), (32,32), (16,64)}.  Replaces F.scaled_dot_product_attention inside timm
 * Attention (models/vits.py:32-34 blocks, models/avmae.py:53-55,83-85 decoder blocks), CrossAttention
 * (models/fusion_blocks.py:50-56) and the factorised pair attention (:250-258).
 * LSE [B,H,Nq] (log-sum-exp of the scaled scores) is saved for the backward. */
int dav_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int dqk,
                 int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                 hipStream_t stream);
/* Delta [B,H,Nq] is scratch written by the first kernel and read by the second. */
int dav_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Delta,
                 void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs,
                 int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs,
                 int dk_rs, long dv_bs, int dv_rs, float scale, hipStream_t stream);
/* the same in two calls: part 1 = the dQ (+ Delta) kernel, part 2 = the dK/dV kernel (reads Delta), part 3 = both */
int dav_attn_bwd_part(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Delta,
                 void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs,
                 int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs,
                 int dk_rs, long dv_bs, int dv_rs, float scale, int part, hipStream_t stream);

/* dav_attn_bwd_part for a query buffer that is the tail of a longer key / value buffer (the tower blocks' fused qkv: the fusion
 * tokens are context rows — keys and values only, models/deepavfusion.py:104-105): dQ has dq_ctx_rows rows BEFORE its first query
 * row in every batch element (dQ - dq_ctx_rows * dq_rs must be valid memory); the dQ kernel (part & 1) zero-fills the dqk columns
 * of every head in them, so that the fused gradient buffer can go into the qkv GEMMs without a fill pass of its own. */
int dav_attn_bwd_ctx(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Delta,
                 void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs,
                 int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs,
                 int dk_rs, long dv_bs, int dv_rs, float scale, int dq_ctx_rows, int part, hipStream_t stream);

/* Window attention of the Swin decoder blocks (models/swin.py:55-87; decoder_arch == 'swin', models/avmae.py:37-51): the same
 * kernels with an additive logit bias.  bias [bias_nb][H][Nq][bias_ld] fp32 in LOG2 units (natural value x log2 e), zero
 * padded to bias_ld >= Nk rounded up to 32; batch element b uses table b % bias_nb (the shift mask differs per window,
 * :75-78).  Backward: dS [B][H][Nq][bias_ld] (optional) receives the gradient of the biased logits in natural units — what
 * dav_relpos_bias_bwd reduces into the relative-position table's gradient. */
int dav_attn_bias_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int dqk,
                      int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                      const float* bias, int bias_nb, int bias_ld, hipStream_t stream);
int dav_attn_bias_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Delta,
                      void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs,
                      int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs,
                      int dk_rs, long dv_bs, int dv_rs, float scale, const float* bias, int bias_nb, int bias_ld, float* dS,
                      int part, hipStream_t stream);

/* Attention dropout (nn.Dropout on the softmax probabilities: timm Attention.attn_drop of the tower blocks, models/vits.py:33, and
 * models/fusion_blocks.py:14,25,42,54,99,112,164,181,231,256 — the fine-tuning constructors' attn_drop > 0; every pre-training
 * config has 0): O = (P .* keep * keep_scale) V with P the full softmax (LSE unchanged).  keep: bytes 0 / 1,
 * [B][H][Nq][keep_ld], keep_ld >= Nk rounded up to 32 and a multiple of 4, 4-byte aligned; keep_scale = 1 / (1 - p).  The draw
 * itself is the caller's (torch's generator): the kernels are deterministic in the mask.  Kernels of their own (the head widths
 * of dav_attn_fwd): the pre-training kernels carry no dropout code.  Backward: dP = keep * keep_scale * (dO V^T),
 * Delta = dO . O as without dropout; dq_ctx_rows as dav_attn_bwd_ctx, part as dav_attn_bwd_part. */
int dav_attn_drop_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int dqk,
                      int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                      const void* keep, int keep_ld, float keep_scale, hipStream_t stream);
int dav_attn_drop_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* Delta,
                      void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs,
                      int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs,
                      int dk_rs, long dv_bs, int dv_rs, float scale, const void* keep, int keep_ld, float keep_scale,
                      int dq_ctx_rows, int part, hipStream_t stream);

/* ---- Swin decoder data movers (csrc/swin.hip) --------------------------------------------- */
/* A decoder activation is [B][nF fusion rows | L = nW * A token rows][C]; a window sequence is [A window tokens | nF fusion
 * tokens], B * nW of them.  rows[w * A + i] = token (0-based, within the L token rows) at slot i of window w after the cyclic
 * shift (torch.roll + timm window_partition, models/swin.py:172-179); inv = its inverse permutation.
 * dav_window_unfold: sequences <- rows (the fusion rows repeated per window and scaled by fusion_scale; models/swin.py:183-185;
 *   with fusion_scale = 1 / nW also the backward of the fold's mean).  src fp32 or bf16, out bf16 or fp32.
 * dav_window_fold: rows <- sequences (+ res): token rows through inv (window_reverse + roll back, :191-197), fusion rows =
 *   fusion_scale x the sum over the windows (1 / nW: the mean of :199; 1: the backward of the unfold's repeat). */
int dav_window_unfold(const void* src, int src_is_bf16, const int* rows, int B, int nW, int A, int nF, int L, int C,
                      float fusion_scale, void* out, int out_is_bf16, hipStream_t stream);
int dav_window_fold(const float* t, const int* inv, const float* res, int B, int nW, int A, int nF, int L, int C,
                    float fusion_scale, float* out, hipStream_t stream);
/* out[w][h][q][k] = mul * (table[index[q * A + k]][h] + mask[w][q][k]) for q, k < A, 0 elsewhere ([nb][H][N][ld]; mask NULL
 * for unshifted blocks, then nb = 1): models/swin.py:49-53 + :66-78 (the reference pads both with zeros to the N = A + nF
 * sequence).  dav_relpos_bias_bwd: dtable[e][h] += sum over the Bw sequences and the (q, k) with index == e of dS. */
int dav_relpos_bias_build(const float* table, const int* index, const float* mask, int nb, int H, int A, int N, int ld, float mul,
                          float* out, hipStream_t stream);
int dav_relpos_bias_bwd(const float* dS, const int* index, int Bw, int H, int A, int N, int ld, int T, float* dtable,
                        hipStream_t stream);

/* ---- LayerNorm ---------------------------------------------------------------------------- */
/* nn.LayerNorm over D on rows taken from two fp32 sources per batch element (r0 rows of x0, then r1
 * rows of x1; r1 may be 0): folds torch.cat((x_fusion, x_mod), 1) of models/deepavfusion.py:104-105
 * into Block.norm1.  Writes bf16 and/or fp32 outputs ([B*(r0+r1), D]) and the row statistics. */
int dav_layernorm_fwd(const float* x0, long x0_bs, int r0, const float* x1, long x1_bs, int r1, int B, int D,
                      const float* gamma, const float* beta, float eps, void* y_bf16, float* y_f32, float* mean, float* rstd,
                      hipStream_t stream);
/* total dy = dy_bf16 + dy_f32 (either may be NULL).  For each source segment s: dx_s (=|+=) LN'(dy)
 * (+ res_s), optional bf16 copy; dx_s NULL skips the segment.  dgamma/dbeta are accumulated (+=) from
 * per-workgroup partial rows staged in `workspace` (two kernels, no atomics). */
int dav_layernorm_bwd(const float* x0, long x0_bs, int r0, const float* x1, long x1_bs, int r1, int B, int D,
                      const void* dy_bf16, const float* dy_f32, const float* gamma, const float* mean, const float* rstd,
                      float* dx0, long dx0_bs, int acc0, const float* res0, long res0_bs, void* dx0_bf16, long dx0_bf_bs,
                      float* dx1, long dx1_bs, int acc1, const float* res1, long res1_bs, void* dx1_bf16, long dx1_bf_bs,
                      float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream);
size_t dav_layernorm_bwd_workspace_bytes(int rows, int D);   /* rows = B * (r0 + r1) */
/* Deferred form: call dav_layernorm_bwd with dgamma = dbeta = NULL and a workspace (only the per-workgroup
 * partial rows are written), then reduce up to 64 such workspaces into their dgamma/dbeta (+=) in ONE launch. */
/* rows > 0: the LayerNorm backward ran over `rows` rows (its own grid decides how many partial rows the workspace holds);
 * rows < 0 (ABI 4): the workspace holds exactly -rows partial rows [2 D] */
typedef struct DavLnReduce { const void* workspace; float* dgamma; float* dbeta; int rows; int D; } DavLnReduce;
int dav_layernorm_bwd_reduce_grouped(const DavLnReduce* items, int count, hipStream_t stream);

/* ---- LayerNorm folded into its neighbour GEMMs (ABI 8; see dav_gemm_nt_ln_bf16) ------------------------------------------- */
/* Per (Linear, LayerNorm-in-front-of-it) pair, from the fp32 master weight w [N, K] (one rounding, as for the plain bf16 mirror):
 * w_ln[n, k] = bf16(gamma[k] * w[n, k]), ln_c[n] = sum_k float(w_ln[n, k]) (the ROUNDED values the GEMM contracts with: acc - mean * ln_c
 * is an exact centring), ln_d[n] = (bias ? bias[n] : 0) + sum_k beta[k] * w[n, k].  One launch for up to 48 pairs (one wave per output
 * row); the table is read on the host during the call only.  K % 8 == 0. */
typedef struct DavLnFold {
  const float* w; const float* gamma; const float* beta; const float* bias;   /* [N, K] fp32, [K], [K], [N] or NULL */
  void* w_ln_bf16; float* ln_c; float* ln_d;                                   /* [N, K] bf16, [N], [N] */
  int N, K;
} DavLnFold;
int dav_ln_fold_grouped(const DavLnFold* items, int count, hipStream_t stream);
/* Row statistics + bf16 twin of an fp32 activation that no GEMM epilogue produced (the decoder input after the un-shuffle, the
 * expanded fusion tokens, DropPath outputs): rows r of [B][rows][D] at x + b * x_bs + r * D (x_bs in elements, 0 = broadcast) ->
 * twin[(b * rows + r) * D + ...] bf16, stats[(b * rows + r) * (D / 64) + s] = {sum, sum of squares} over columns [64 s, 64 s + 64).
 * D % 64 == 0, D <= 1024. */
int dav_rowstats_cast(const float* x, long x_bs, int B, int rows, int D, void* twin_bf16, float* stats, hipStream_t stream);
/* dav_layernorm_bwd for a LayerNorm that was folded away on the forward path: x comes as the bf16 twin (two dense segments per batch
 * element, batch strides in elements as there) with its statistics partials (st*: row b * (st*_bs / D) + r), mean / rstd are re-formed
 * as dav_gemm_nt_ln_bf16 forms them, and h_out (optional, bf16 [B * (r0 + r1), D]) receives gamma * xhat + beta — the operand the weight
 * gradient of the consuming Linear contracts with, produced here instead of being kept from the forward.  Everything else as
 * dav_layernorm_bwd (dgamma / dbeta partial rows in `workspace`, reduced by dav_layernorm_bwd_reduce_grouped or at once). */
int dav_layernorm_bwd_twin(const void* xb0, long xb0_bs, const float* st0, int r0, const void* xb1, long xb1_bs, const float* st1, int r1,
                           int B, int D, float eps, const void* dy_bf16, const float* dy_f32, const float* gamma, const float* beta,
                           float* dx0, long dx0_bs, int acc0, const float* res0, long res0_bs, void* dx0_bf16, long dx0_bf_bs,
                           float* dx1, long dx1_bs, int acc1, const float* res1, long res1_bs, void* dx1_bf16, long dx1_bf_bs,
                           void* h_out_bf16, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- masking / gather / scatter ------------------------------------------------------------ */
/* AVMAE.random_masking (models/avmae.py:120-142) for given noise [N,L]: argsort twice, keep the first
 * len_keep, un-shuffled 0/1 mask.  Bit-exact with torch.argsort for tie-free noise. */
int dav_mask_build(const float* noise, int N, int L, int len_keep, int64_t* ids_keep, int64_t* ids_restore, float* mask,
                   int* ids_keep32, int* ids_restore32, hipStream_t stream);
/* im2row of the kept 16x16 patches only (timm PatchEmbed + the gather of models/vits.py:100):
 * A[b*nk+t, c*256+py*16+px] bf16; ids NULL = all patches in order. */
int dav_patch_gather(const float* img, int B, int C, int H, int W, const int* ids_keep32, int nk, void* A_bf16, hipStream_t stream);
/* the same for clips [B,C,T,H,W] and (pt,16,16) tubelets (PatchEmbed3D, util/pos_embed.py:123-146, used by
 * models/video_vits.py:218-220): token = (gt*gH + gy)*gW + gx, A[b*nk+t, ((c*pt + dt)*16 + py)*16 + px] */
int dav_patch_gather3d(const float* video, int B, int C, int T, int H, int W, int pt, const int* ids_keep32, int nk,
                       void* A_bf16, hipStream_t stream);
/* models/avmae.py:161-165: out[b, off+r] = (restore[b,r] < nk ? emb[b*nk+restore[b,r]] : mask_token) + pos[r] */
int dav_unshuffle_fwd(const float* emb, const float* mask_token, const float* pos, const int* ids_restore32, int B, int L, int nk,
                      int D, float* out, long out_bs, int out_row_off, hipStream_t stream);
/* out[b*n+t] = bf16(x[b, row_off + (ids ? ids[b*n+t] : t)]) — backward of the gather / of x[:, nF:] slices */
int dav_rows_gather_cast(const float* x, long x_bs, int row_off, const int* ids32, int B, int n, int D, void* out_bf16,
                         hipStream_t stream);
/* DropPath (timm; models/vits.py:32-34, models/fusion_blocks.py:69,130,199,278 when drop_path > 0 — fine-tuning) with a
 * per-sample scale s[b] in {0, 1/keep}: forward out[b,r,:] = res[b,r,:] + s[b]*y[b,r,:] (fp32; out may alias res); backward of
 * the branch side out_bf16[b,r,:] = bf16(s[b]*g[b,r,:]) (the residual side passes g through unchanged). */
int dav_rows_axpy(const float* res, const float* y, const float* scale, int B, int rows, int D, float* out, hipStream_t stream);
int dav_rows_scale_cast(const float* g, const float* scale, int B, int rows, int D, void* out_bf16, hipStream_t stream);
/* Dropout on activations (timm Mlp.drop1 / drop2, the attention modules' proj_drop; fine-tuning constructors' drop > 0), DropPath
 * folded in: out[b,r,:] = (res ? res[b,r,:] : 0) + (rowscale ? rowscale[b] : 1) * (keep[b,r,:] ? keep_scale : 0) * in[b,r,:].
 * keep: bytes 0 / 1 [B*rows][D] (null = all kept), in / out bf16 (in_f32 / out_f32 = 0) or fp32 (1), D % 4 == 0; out may alias
 * in or res.  The backward is the same call on the gradient (res = null). */
int dav_dropout_rows(const void* in, int in_f32, const float* res, const void* keep, float keep_scale, const float* rowscale,
                     int B, int rows, int D, void* out, int out_f32, hipStream_t stream);
/* dpos[r] += sum_b dx[b, off+r]; dmask_token += sum over masked (b, r) */
int dav_unshuffle_bwd_reduce(const float* dx, long dx_bs, int row_off, const int* ids_restore32, int B, int L, int nk, int D,
                             float* dpos, float* dmask_token, hipStream_t stream);

/* ---- loss --------------------------------------------------------------------------------- */
/* AVMAE.patchify + forward_loss (models/avmae.py:182-214) without a patchified target tensor:
 * per-patch MSE against the (optionally norm-pix, unbiased variance) target read from NCHW,
 * loss = sum(loss_patch * mask) / sum(mask). */
int dav_patch_mse_fwd(const float* img, const float* pred, const float* mask, int B, int C, int H, int W, int norm_pix,
                      float* loss_patch, float* tmean, float* trstd, float* loss, float* mask_sum, hipStream_t stream);
int dav_patch_mse_bwd(const float* img, const float* pred, const float* mask, const float* tmean, const float* trstd,
                      const float* mask_sum, const float* gout, int B, int C, int H, int W, void* dpred_bf16, hipStream_t stream);

/* ---- factorised (v,a) pairs (models/fusion_blocks.py:245-252) ------------------------------- */
int dav_pair_expand(const float* Pv, const float* Pa, int B, int nv, int na, int Wd, void* out_bf16, hipStream_t stream);
int dav_pair_reduce(const void* d_bf16, int B, int nv, int na, int Wd, void* dPv_bf16, void* dPa_bf16, hipStream_t stream);

/* ---- casts, grad norm, optimizer ----------------------------------------------------------- */
int dav_cast_bf16(const float* x, void* y_bf16, long n, hipStream_t stream);
/* out = a + b (fp32, n % 4 == 0) and out_bf16 = bf16(out) in one pass: the gradient w.r.t. the fusion tokens is the sum of what the
 * image tower and the audio tower return (models/deepavfusion.py:104-105 feeds the same tokens to both), and the next block's
 * backward wants it in both precisions.  (ABI 3) */
int dav_add_cast(const float* a, const float* b, float* out, void* out_bf16, long n, hipStream_t stream);
int dav_cast_transpose_bf16(const float* x, void* y_bf16, int R, int C, hipStream_t stream);   /* y[c,r] = x[r,c] */
/* the same for several matrices in ONE launch (ABI 4): y_bf16 [C, R] = bf16(x [R, C] fp32) per item */
typedef struct DavTranspose { const float* x; void* y_bf16; int R; int C; } DavTranspose;
int dav_cast_transpose_grouped(const DavTranspose* items, int count, hipStream_t stream);
/* get_grad_norm_ (util/misc.py:151-163) over one flat fp32 buffer: out = scale * ||x||_2 */
size_t dav_l2norm_workspace_bytes(long n);
int dav_l2norm(const float* x, long n, float scale, float* out, void* workspace, size_t workspace_bytes, hipStream_t stream);
/* torch.optim.AdamW(betas=(0.9,0.95)) step (train.py:93) over flat param/grad/state buffers with
 * per-segment {lr, weight_decay}; bias_corr = {1-beta1^t, sqrt(1-beta2^t)} in device memory.  The same pass can
 * (a) accumulate sum(g^2) of the unscaled gradients into *sumsq_out (zeroed first; the grad norm of
 * util/misc.py:151-163 is its square root), (b) rewrite the bf16 weight mirror p_bf16, (c) zero the gradients
 * (Trainer.zero_grad) — one trip over the buffers instead of three.  keep_grad (NULL or one byte per segment): segments
 * with a non-zero byte are NOT zero-filled — their next gradient will be written, not accumulated (DavTnProblem.flags bit 0),
 * which saves the fill here and the read there (values other than 0 / 1 are not defined).  All accesses are 16-byte ones: the buffers must be
 * 16-byte aligned (p_bf16 8-byte), n and every seg_end a multiple of 4 (FlatParams aligns segments to 64 elements).
 * gscale_dev (optional): device scalar from dav_step_guard, see there. */
int dav_adamw_flat(float* p, float* g, float* m, float* v, void* p_bf16, long n, const long* seg_end, const float* hyper,
                   int nseg, float beta1, float beta2, float eps, const float* bias_corr, float grad_scale, float* sumsq_out,
                   int zero_grad, const unsigned char* keep_grad, const float* gscale_dev, hipStream_t stream);

/* Device-side guard of a captured (hipGraph) step, replaces the host checks of train.py:166-167 (non-finite loss aborts before the
 * optimizer step) and util/misc.py:118-120 (clip_grad_norm_) that a replayed graph cannot run:
 *   out_scale[0] = 0                                    if loss_a + loss_b (either may be NULL) or gnorm[0] is not finite,
 *                = min(1, clip / (gnorm[0] * grad_scale + 1e-6))   if gnorm != NULL (the global gradient norm, dav_l2norm) and
 *                                                                   clip > 0,   = 1 otherwise;
 * bad_count[0] (optional) is incremented when the scale is 0.  dav_adamw_flat(gscale_dev = out_scale) multiplies the gradients
 * by it and, when it is 0, leaves parameters, moments and the bf16 mirror untouched (gradients are still zero-filled). */
int dav_step_guard(const float* loss_a, const float* loss_b, const float* gnorm, float clip, float grad_scale, float* out_scale,
                   int* bad_count, hipStream_t stream);

/* ---- log-mel front-end (csrc/mel.hip) ------------------------------ */
/* out[b, m, t] = log10(mel_m(|STFT_t(wave[b])|^2) + eps): the reference's audio transform
 * aT.MelSpectrogram(sample_rate, n_fft, hop_length, n_mels) -> aT.Log() (train.py:50-54, util/audio_transforms.py:29-35) and
 * the [:, :, :-1] of datasets.py:242, computed on the device from raw waveforms [B, S] instead of in CPU loader workers.
 * Semantics of torchaudio's Spectrogram + MelScale defaults: centre padding by reflection, the caller's window of n_fft
 * samples (periodic Hann), one-sided power spectrum, dense filterbank fbank[n_fft/2+1, n_mels] with each filter's non-zero bin
 * range [band_lo[m], band_hi[m]); cos_tab / sin_tab[n] = cos / sin(2 pi n / n_fft).  frames = S / hop + 1; drop_last removes
 * the final frame (out is [B, n_mels, frames - drop_last]); apply_log = 0 returns the mel power itself. */
int dav_logmel(const float* wave, int B, int S, int n_fft, int hop, int n_mels, const float* window, const float* cos_tab,
               const float* sin_tab, const float* fbank, const int* band_lo, const int* band_hi, float eps, int apply_log,
               int drop_last, float* out, hipStream_t stream);
int dav_log10_eps(const float* x, float eps, long n, float* y, hipStream_t stream);      /* aT.Log: y = log10(x + eps) */

/* ---- fp32-operand twins (csrc/f32_path.hip) ------------------------------------------------ */
/* The same contracts as the bf16 entry points above with every operand, second output and intermediate in fp32 (plain
 * one-thread-per-output kernels, untuned).  They exist so that the whole hand-written forward / backward can run in fp32
 * (engine.set_precision('fp32')) and be compared with the oracle and the reference's fixtures at 1e-4 instead of bf16's
 * 2e-2 — SURVEY.md section 8(b) "fp32-everything variants for tight parity tests".  b_kn as variant bit 12 above; part as
 * dav_attn_bwd_part.  LayerNorm, the loss forward, unshuffle and the optimizer are fp32 already. */
int dav_gemm_nt_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, const int* a_rowmap,
                    const float* bias, int act, const float* aux, int ldaux, const float* res, int ldres,
                    const int* res_rowmap, const int* res_rows, float* C, int ldc, const int* c_rowmap, float* C2, int ldc2,
                    int c2_mode, int beta, float alpha, int b_kn, hipStream_t stream);
int dav_gemm_tn_f32(const float* A, const float* B, int Mc, int N, int K, int lda, int ldb, const int* a_rowmap,
                    const int* b_rowmap, float* C, int ldc, int beta, float* bias_grad, hipStream_t stream);
int dav_attn_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk, int dqk,
                     int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                     hipStream_t stream);
int dav_attn_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                     float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs,
                     int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs, long dq_bs,
                     int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale, int part, hipStream_t stream);
/* fp32 twins of dav_attn_bias_fwd / dav_attn_bias_bwd; bias in NATURAL units here, bias_ld >= Nk */
int dav_attn_bias_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                          int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                          float scale, const float* bias, int bias_nb, int bias_ld, hipStream_t stream);
int dav_attn_bias_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                          float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs,
                          int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs,
                          long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale, const float* bias,
                          int bias_nb, int bias_ld, float* dS, int part, hipStream_t stream);
int dav_attn_drop_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                          int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                          float scale, const void* keep, int keep_ld, float keep_scale, hipStream_t stream);   /* keep_ld >= Nk */
int dav_attn_drop_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                          float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv, long q_bs,
                          int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs, int do_rs,
                          long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale, const void* keep,
                          int keep_ld, float keep_scale, int part, hipStream_t stream);
int dav_patch_gather_f32(const float* img, int B, int C, int T, int H, int W, int pt, const int* ids_keep32, int nk, float* A,
                         hipStream_t stream);
int dav_rows_gather_f32(const float* x, long x_bs, int row_off, const int* ids32, int B, int n, int D, float* out, long out_bs,
                        hipStream_t stream);      /* out_bs: batch stride of out in elements (0 = dense) */
int dav_pair_expand_f32(const float* Pv, const float* Pa, int B, int nv, int na, int Wd, float* out, hipStream_t stream);
int dav_pair_reduce_f32(const float* d, int B, int nv, int na, int Wd, float* dPv, float* dPa, hipStream_t stream);
int dav_patch_mse_bwd_f32(const float* img, const float* pred, const float* mask, const float* tmean, const float* trstd,
                          const float* mask_sum, const float* gout, int B, int C, int H, int W, float* dpred, hipStream_t stream);
int dav_add_f32(const float* a, const float* b, float* out, long n, hipStream_t stream);   /* out = a + b */

/* ---- launch batching ---------------------------------------------------------------------- */
/* The reference runs the image block, the audio block and the fusion block of a layer one after the other although they
 * only depend on the layer's inputs (models/deepavfusion.py:104-107), and likewise the two MAE decoders
 * (models/avmae.py:147-180).  Between dav_batch_begin() and dav_batch_end() every entry point of this library RECORDS its
 * kernel launches (per calling thread) instead of issuing them; dav_batch_lane() starts a new LANE — a sequence of
 * launches the caller declares independent of the other lanes.  dav_batch_end() issues the lanes in lockstep: the k-th
 * launches of all lanes go out together, those of one kernel family (NT GEMM, LayerNorm forward / backward, attention
 * forward / dQ / dK-dV) as ONE grouped grid whose workgroups look their problem up in a table passed by value, the rest
 * individually; the order inside a lane is kept.  With auto_lanes != 0 every recorded launch is its own lane (a region of
 * mutually independent launches).  Requirements on the caller: lanes must not depend on each other; every buffer a recorded
 * launch touches must stay allocated until dav_batch_end() returns; all launches of a batch use the stream they were
 * recorded with.  No nesting.  dav_batch_stats reports the last batch: launches recorded / launches issued.
 * dav_batch_region(1) .. dav_batch_region(0): the launches recorded in between are independent of each other and form ONE
 * step of the current lane (e.g. the q / kv projections of both aggregation cross-attentions of a fusion block). */
int dav_batch_begin(int auto_lanes);
int dav_batch_lane(void);
int dav_batch_region(int begin);
/* the current lane idles for `steps` steps (aligns its later launches with equal launches of a longer lane) */
int dav_batch_skip(int steps);
/* on != 0: launches are issued at once although a batch is open (for work that nothing recorded so far depends on and
 * that later recorded launches need, e.g. refreshing a bf16 weight copy); on == 0: recording resumes. */
int dav_batch_suspend(int on);
int dav_batch_end(void);
int dav_batch_abort(void);
int dav_batch_stats(int* recorded_ops, int* issued_launches);

#ifdef __cplusplus
}
#endif
#endif /* DAV_KERNELS_H_ */
