// fp32-operand twins of the bf16 kernels: the SAME entry-point contracts (row maps, strides, fused epilogue order, second
// outputs, accumulate flags) with every operand and every intermediate in fp32.
//
// Purpose ("fp32-everything variants for tight parity tests"): with bf16 operands the engine can
// only be compared with the fp32 oracle at ~1e-2, which would hide an indexing / transposition / ordering mistake worth
// 1 %.  Running the whole hand-written forward + backward (engine.set_precision('fp32')) on these kernels pins the engine
// itself — tapes, row maps, gradient formulas, accumulation order — against the oracle and the reference's fixtures at
// 1e-4 .. 1e-5.  They are plain one-thread-per-output kernels (fp32 FMA chains, fp32 softmax): correct and simple, not
// tuned — the bf16 MFMA kernels are the product path, these are its high-precision cross-check on the same GPU.
#include "common.h"
#include "dav_kernels.h"

namespace {

struct RowMapF { int rpb, bs, off; };
__device__ __forceinline__ long mrow(int m, const RowMapF& r) {
  return r.rpb > 0 ? (long)(m / r.rpb) * r.bs + r.off + (m % r.rpb) : (long)m;
}
RowMapF mkf(const int* m) { return m ? RowMapF{m[0], m[1], m[2]} : RowMapF{0, 0, 0}; }

struct NTF {
  const float *A, *B;
  int M, N, K, lda, ldb;
  RowMapF amap;
  const float* bias;
  int act;
  const float* aux; int ldaux;
  const float* res; int ldres; RowMapF rmap; const int* res_rows;
  float* C; int ldc; RowMapF cmap;
  int b_kn;
  float* C2; int ldc2; int c2_mode;
  int beta;
  float alpha;
};

// same epilogue order as dav_gemm_nt_bf16 (include/dav_kernels.h)
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(NTF p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.M * p.N) return;
  const int m = (int)(idx / p.N), n = (int)(idx % p.N);
  const float* a = p.A + mrow(m, p.amap) * p.lda;
  float s = 0.f;
  if (!p.b_kn) {
    const float* b = p.B + (long)n * p.ldb;
    for (int k = 0; k < p.K; ++k) s = fmaf(a[k], b[k], s);
  } else {
    for (int k = 0; k < p.K; ++k) s = fmaf(a[k], p.B[(long)k * p.ldb + n], s);
  }
  float v = s * p.alpha;
  if (p.bias) v += p.bias[n];
  if (p.c2_mode == 1) p.C2[(long)m * p.ldc2 + n] = v;
  if (p.act == 1) {
    float u, d;
    gelu_pair_f(v, u, d);
    if (p.c2_mode == 4) p.C2[(long)m * p.ldc2 + n] = d;
    v = u;
  } else if (p.act == 2) {
    v *= gelu_grad_f(p.aux[(long)m * p.ldaux + n]);
  } else if (p.act == 3) {
    v *= p.aux[(long)m * p.ldaux + n];
  }
  if (p.c2_mode == 2) p.C2[(long)m * p.ldc2 + n] = v;
  if (p.res) {
    const long rrow = p.res_rows ? (long)p.res_rows[m] : mrow(m, p.rmap);
    v += p.res[rrow * p.ldres + n];
  }
  if (p.C) {
    float* c = p.C + mrow(m, p.cmap) * p.ldc + n;
    if (p.beta) v += *c;
    *c = v;
  }
  if (p.c2_mode == 3) p.C2[(long)m * p.ldc2 + n] = v;
}

struct TNF {
  const float *A, *B;
  int Mc, N, K, lda, ldb;
  RowMapF amap, bmap;
  float* C; int ldc; int beta;
  float* bias_grad;
};

// C[n, k] (+)= sum_m A[m, n] B[m, k]; bias_grad[n] += sum_m A[m, n] (thread k == 0)
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(TNF p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.N * p.K) return;
  const int n = (int)(idx / p.K), k = (int)(idx % p.K);
  float s = 0.f, sb = 0.f;
  for (int m = 0; m < p.Mc; ++m) {
    const float av = p.A[mrow(m, p.amap) * p.lda + n];
    s = fmaf(av, p.B[mrow(m, p.bmap) * p.ldb + k], s);
    sb += av;
  }
  float* c = p.C + (long)n * p.ldc + k;
  *c = p.beta ? *c + s : s;
  if (p.bias_grad && k == 0) p.bias_grad[n] += sb;
}

struct AttnF {
  const float *Q, *K, *V;
  float* O; float* LSE;
  int B, H, Nq, Nk, dqk, dv;
  long q_bs, k_bs, v_bs, o_bs;
  int q_rs, k_rs, v_rs, o_rs;
  float scale;
  const float* dO; long do_bs; int do_rs;
  float* Delta;
  float *dQ, *dK, *dV;
  long dq_bs, dk_bs, dv_bs;
  int dq_rs, dk_rs, dv_rs;
  const float* bias; int bias_nb, bias_ld;    // additive logit bias [b % bias_nb][h][q][bias_ld] in NATURAL units; dS as in attention.hip
  float* dS;
  const unsigned char* keep; int keep_ld; float keep_scale;   // attention dropout: bytes 0/1 [b][h][q][keep_ld], kept probabilities * keep_scale
};
__device__ __forceinline__ float attnf_keep(const AttnF& p, int b, int h, int q, int j) {
  return !p.keep ? 1.f : (p.keep[(((long)b * p.H + h) * p.Nq + q) * p.keep_ld + j] ? p.keep_scale : 0.f);
}
__device__ __forceinline__ float attnf_bias(const AttnF& p, int b, int h, int q, int j) {
  return p.bias ? p.bias[(((long)(b % p.bias_nb) * p.H + h) * p.Nq + q) * p.bias_ld + j] : 0.f;
}

// one thread per (b, h, query): two passes over the keys (max, then exp-sum and P.V)
__global__ __launch_bounds__(128) void attn_fwd_f32_kernel(AttnF p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.B * p.H * p.Nq) return;
  const int q = (int)(idx % p.Nq), h = (int)((idx / p.Nq) % p.H), b = (int)(idx / ((long)p.Nq * p.H));
  const float* qr = p.Q + b * p.q_bs + (long)q * p.q_rs + h * p.dqk;
  float mx = -1e30f;
  for (int j = 0; j < p.Nk; ++j) {
    const float* kr = p.K + b * p.k_bs + (long)j * p.k_rs + h * p.dqk;
    float s = 0.f;
    for (int d = 0; d < p.dqk; ++d) s = fmaf(qr[d], kr[d], s);
    mx = fmaxf(mx, s * p.scale + attnf_bias(p, b, h, q, j));
  }
  float l = 0.f;
  float* o = p.O + b * p.o_bs + (long)q * p.o_rs + h * p.dv;
  for (int d = 0; d < p.dv; ++d) o[d] = 0.f;
  for (int j = 0; j < p.Nk; ++j) {
    const float* kr = p.K + b * p.k_bs + (long)j * p.k_rs + h * p.dqk;
    const float* vr = p.V + b * p.v_bs + (long)j * p.v_rs + h * p.dv;
    float s = 0.f;
    for (int d = 0; d < p.dqk; ++d) s = fmaf(qr[d], kr[d], s);
    const float e = expf(s * p.scale + attnf_bias(p, b, h, q, j) - mx);
    l += e;
    const float ek = e * attnf_keep(p, b, h, q, j);     // the normaliser sums ALL probabilities; the dropped ones leave only the output
    for (int d = 0; d < p.dv; ++d) o[d] = fmaf(ek, vr[d], o[d]);
  }
  const float inv = 1.f / l;
  for (int d = 0; d < p.dv; ++d) o[d] *= inv;
  if (p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q] = mx + logf(l);
}

// backward, query side: Delta = dO . O, dQ = scale * sum_j P_ij (dP_ij - Delta_i) K_j
__global__ __launch_bounds__(128) void attn_bwd_dq_f32_kernel(AttnF p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.B * p.H * p.Nq) return;
  const int q = (int)(idx % p.Nq), h = (int)((idx / p.Nq) % p.H), b = (int)(idx / ((long)p.Nq * p.H));
  const float* qr = p.Q + b * p.q_bs + (long)q * p.q_rs + h * p.dqk;
  const float* o = p.O + b * p.o_bs + (long)q * p.o_rs + h * p.dv;
  const float* dor = p.dO + b * p.do_bs + (long)q * p.do_rs + h * p.dv;
  const float lse = p.LSE[((long)b * p.H + h) * p.Nq + q];
  float delta = 0.f;
  for (int d = 0; d < p.dv; ++d) delta = fmaf(dor[d], o[d], delta);
  p.Delta[((long)b * p.H + h) * p.Nq + q] = delta;
  float* dq = p.dQ + b * p.dq_bs + (long)q * p.dq_rs + h * p.dqk;
  for (int d = 0; d < p.dqk; ++d) dq[d] = 0.f;
  for (int j = 0; j < p.Nk; ++j) {
    const float* kr = p.K + b * p.k_bs + (long)j * p.k_rs + h * p.dqk;
    const float* vr = p.V + b * p.v_bs + (long)j * p.v_rs + h * p.dv;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < p.dqk; ++d) s = fmaf(qr[d], kr[d], s);
    for (int d = 0; d < p.dv; ++d) dp = fmaf(dor[d], vr[d], dp);
    const float dsn = expf(s * p.scale + attnf_bias(p, b, h, q, j) - lse) * (dp * attnf_keep(p, b, h, q, j) - delta);     // d(biased logit)
    if (p.dS) p.dS[(((long)b * p.H + h) * p.Nq + q) * p.bias_ld + j] = dsn;
    const float ds = dsn * p.scale;
    for (int d = 0; d < p.dqk; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
  }
}

// backward, key side: one thread per (b, h, key): dV_j = sum_i P_ij dO_i, dK_j = scale * sum_i P_ij (dP_ij - Delta_i) Q_i
__global__ __launch_bounds__(128) void attn_bwd_dkv_f32_kernel(AttnF p) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)p.B * p.H * p.Nk) return;
  const int j = (int)(idx % p.Nk), h = (int)((idx / p.Nk) % p.H), b = (int)(idx / ((long)p.Nk * p.H));
  const float* kr = p.K + b * p.k_bs + (long)j * p.k_rs + h * p.dqk;
  const float* vr = p.V + b * p.v_bs + (long)j * p.v_rs + h * p.dv;
  float* dk = p.dK + b * p.dk_bs + (long)j * p.dk_rs + h * p.dqk;
  float* dvv = p.dV + b * p.dv_bs + (long)j * p.dv_rs + h * p.dv;
  for (int d = 0; d < p.dqk; ++d) dk[d] = 0.f;
  for (int d = 0; d < p.dv; ++d) dvv[d] = 0.f;
  for (int q = 0; q < p.Nq; ++q) {
    const float* qr = p.Q + b * p.q_bs + (long)q * p.q_rs + h * p.dqk;
    const float* dor = p.dO + b * p.do_bs + (long)q * p.do_rs + h * p.dv;
    const long si = ((long)b * p.H + h) * p.Nq + q;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < p.dqk; ++d) s = fmaf(qr[d], kr[d], s);
    for (int d = 0; d < p.dv; ++d) dp = fmaf(dor[d], vr[d], dp);
    const float pr = expf(s * p.scale + attnf_bias(p, b, h, q, j) - p.LSE[si]);
    const float km = attnf_keep(p, b, h, q, j);
    const float ds = pr * (dp * km - p.Delta[si]) * p.scale;
    for (int d = 0; d < p.dv; ++d) dvv[d] = fmaf(pr * km, dor[d], dvv[d]);
    for (int d = 0; d < p.dqk; ++d) dk[d] = fmaf(ds, qr[d], dk[d]);
  }
}

// ---- data movers with fp32 outputs (twins of patch_gather / rows_gather_cast / pair_expand / pair_reduce) ---------------
__global__ __launch_bounds__(256) void patch_gather_f32_kernel(const float* img, int B, int C, int T, int PT, int H, int W,
                                                               const int* ids, int nk, float* A) {
  const int gW = W >> 4, gH = H >> 4, K = C * PT * 256;
  const long total = (long)B * nk * K;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int col = (int)(e % K);
    const long row = e / K;
    const int b = (int)(row / nk), t = (int)(row % nk);
    const int pidx = ids ? ids[row] : t;
    const int gx = pidx % gW, gy = (pidx / gW) % gH, gt = pidx / (gW * gH);
    const int cd = col >> 8, py = (col >> 4) & 15, px = col & 15;
    const int c = cd / PT, dt = cd % PT;
    A[e] = img[((((long)b * C + c) * T + gt * PT + dt) * H + gy * 16 + py) * W + gx * 16 + px];
  }
}

__global__ __launch_bounds__(256) void rows_gather_f32_kernel(const float* x, long x_bs, int row_off, const int* ids, int B, int n,
                                                              int D, float* out, long out_bs) {
  const long total = (long)B * n * D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const long row = e / D;
    const int b = (int)(row / n), t = (int)(row % n);
    const int j = ids ? ids[row] : t;
    out[b * out_bs + (long)t * D + d] = x[b * x_bs + (long)(row_off + j) * D + d];
  }
}

// out[b, i*na + j, :] = Pv[b, i, :] + Pa[b, j, :]
__global__ __launch_bounds__(256) void pair_expand_f32_kernel(const float* Pv, const float* Pa, int B, int nv, int na, int Wd, float* out) {
  const long total = (long)B * nv * na * Wd;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % Wd);
    const long pr = e / Wd;
    const int j = (int)(pr % na), i = (int)((pr / na) % nv), b = (int)(pr / ((long)na * nv));
    out[e] = Pv[((long)b * nv + i) * Wd + d] + Pa[((long)b * na + j) * Wd + d];
  }
}

// dPv[b, i, :] = sum_j d[b, i*na + j, :],  dPa[b, j, :] = sum_i d[b, i*na + j, :]
__global__ __launch_bounds__(256) void pair_reduce_f32_kernel(const float* dd, int B, int nv, int na, int Wd, float* dPv, float* dPa) {
  const long total = (long)B * (nv + na) * Wd;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e % Wd);
    const long r = e / Wd;
    const int t = (int)(r % (nv + na)), b = (int)(r / (nv + na));
    float s = 0.f;
    if (t < nv) {
      for (int j = 0; j < na; ++j) s += dd[(((long)b * nv + t) * na + j) * Wd + d];
      dPv[((long)b * nv + t) * Wd + d] = s;
    } else {
      const int j = t - nv;
      for (int i = 0; i < nv; ++i) s += dd[(((long)b * nv + i) * na + j) * Wd + d];
      dPa[((long)b * na + j) * Wd + d] = s;
    }
  }
}

__global__ __launch_bounds__(256) void add_f32_kernel(const float* a, const float* b, float* out, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) out[e] = a[e] + b[e];
}

int grid_for(long n) { long g = (n + 255) / 256; return (int)(g > 65535 ? 65535 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int dav_gemm_nt_f32(const float* A, const float* B, int M, int N, int K, int lda, int ldb, const int* a_rowmap,
                               const float* bias, int act, const float* aux, int ldaux, const float* res, int ldres,
                               const int* res_rowmap, const int* res_rows, float* C, int ldc, const int* c_rowmap, float* C2,
                               int ldc2, int c2_mode, int beta, float alpha, int b_kn, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return DAV_ERR_SHAPE;
  if (!C && !C2) return DAV_ERR_SHAPE;
  if (C2 && (c2_mode < 1 || c2_mode > 4)) return DAV_ERR_SHAPE;
  if ((act == 2 || act == 3) && !aux) return DAV_ERR_SHAPE;
  NTF p;
  p.A = A; p.B = B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.amap = mkf(a_rowmap); p.bias = bias; p.act = act;
  p.aux = aux; p.ldaux = ldaux; p.res = res; p.ldres = ldres; p.rmap = mkf(res_rowmap); p.res_rows = res_rows;
  p.C = C; p.ldc = ldc; p.cmap = mkf(c_rowmap); p.b_kn = b_kn; p.C2 = C2; p.ldc2 = ldc2; p.c2_mode = C2 ? c2_mode : 0;
  p.beta = beta; p.alpha = alpha;
  const long total = (long)M * N;
  DAV_LAUNCH(gemm_nt_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  return dav_launch_status();
}

extern "C" int dav_gemm_tn_f32(const float* A, const float* B, int Mc, int N, int K, int lda, int ldb, const int* a_rowmap,
                               const int* b_rowmap, float* C, int ldc, int beta, float* bias_grad, hipStream_t stream) {
  if (Mc <= 0 || N <= 0 || K <= 0) return DAV_ERR_SHAPE;
  TNF p;
  p.A = A; p.B = B; p.Mc = Mc; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.amap = mkf(a_rowmap); p.bmap = mkf(b_rowmap);
  p.C = C; p.ldc = ldc; p.beta = beta; p.bias_grad = bias_grad;
  const long total = (long)N * K;
  DAV_LAUNCH(gemm_tn_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p);
  return dav_launch_status();
}

static int attnf_fwd(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                     int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs,
                     int o_rs, float scale, const float* bias, int bias_nb, int bias_ld, const void* keep, int keep_ld,
                     float keep_scale, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || dqk <= 0 || dv <= 0) return DAV_ERR_SHAPE;
  if (bias && (bias_nb <= 0 || B % bias_nb || bias_ld < Nk)) return DAV_ERR_SHAPE;
  if (keep && (keep_ld < Nk || !(keep_scale > 0.f))) return DAV_ERR_SHAPE;
  AttnF p = {};
  p.keep = (const unsigned char*)keep; p.keep_ld = keep_ld; p.keep_scale = keep_scale;
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.LSE = LSE; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.dqk = dqk; p.dv = dv;
  p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs; p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.o_rs = o_rs; p.scale = scale;
  p.bias = bias; p.bias_nb = bias_nb; p.bias_ld = bias_ld;
  const long total = (long)B * H * Nq;
  DAV_LAUNCH(attn_fwd_f32_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, stream, p);
  return dav_launch_status();
}

extern "C" int dav_attn_bias_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                                     int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs,
                                     int o_rs, float scale, const float* bias, int bias_nb, int bias_ld, hipStream_t stream) {
  return attnf_fwd(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale, bias, bias_nb, bias_ld,
                   nullptr, 0, 0.f, stream);
}

extern "C" int dav_attn_drop_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                                     int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs,
                                     int o_rs, float scale, const void* keep, int keep_ld, float keep_scale, hipStream_t stream) {
  if (!keep) return DAV_ERR_SHAPE;
  return attnf_fwd(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale, nullptr, 0, 0,
                   keep, keep_ld, keep_scale, stream);
}

extern "C" int dav_attn_fwd_f32(const float* Q, const float* K, const float* V, float* O, float* LSE, int B, int H, int Nq, int Nk,
                                int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs,
                                int o_rs, float scale, hipStream_t stream) {
  return dav_attn_bias_fwd_f32(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                               nullptr, 0, 0, stream);
}

static int attnf_bwd(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                     float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                     long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs,
                     int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale,
                     const float* bias, int bias_nb, int bias_ld, float* dS, const void* keep, int keep_ld, float keep_scale,
                     int part, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || part < 1 || part > 3) return DAV_ERR_SHAPE;
  if (bias ? (bias_nb <= 0 || B % bias_nb || bias_ld < Nk) : dS != nullptr) return DAV_ERR_SHAPE;
  if (keep && (keep_ld < Nk || !(keep_scale > 0.f))) return DAV_ERR_SHAPE;
  AttnF p = {};
  p.keep = (const unsigned char*)keep; p.keep_ld = keep_ld; p.keep_scale = keep_scale;
  p.Q = Q; p.K = K; p.V = V; p.O = const_cast<float*>(O); p.LSE = const_cast<float*>(LSE); p.dO = dO; p.Delta = Delta;
  p.dQ = dQ; p.dK = dK; p.dV = dV; p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.dqk = dqk; p.dv = dv;
  p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs; p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.o_rs = o_rs;
  p.do_bs = do_bs; p.do_rs = do_rs; p.dq_bs = dq_bs; p.dk_bs = dk_bs; p.dv_bs = dv_bs; p.dq_rs = dq_rs; p.dk_rs = dk_rs; p.dv_rs = dv_rs;
  p.scale = scale;
  p.bias = bias; p.bias_nb = bias_nb; p.bias_ld = bias_ld; p.dS = dS;
  if (part & 1) {
    const long total = (long)B * H * Nq;
    DAV_LAUNCH(attn_bwd_dq_f32_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, stream, p);
  }
  if (part & 2) {
    const long total = (long)B * H * Nk;
    DAV_LAUNCH(attn_bwd_dkv_f32_kernel, dim3((unsigned)((total + 127) / 128)), dim3(128), 0, stream, p);
  }
  return dav_launch_status();
}

extern "C" int dav_attn_bias_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                                     float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                     long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs,
                                     int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale,
                                     const float* bias, int bias_nb, int bias_ld, float* dS, int part, hipStream_t stream) {
  return attnf_bwd(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                   do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, bias, bias_nb, bias_ld, dS, nullptr, 0, 0.f, part, stream);
}

extern "C" int dav_attn_drop_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                                     float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                     long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs,
                                     int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale,
                                     const void* keep, int keep_ld, float keep_scale, int part, hipStream_t stream) {
  if (!keep) return DAV_ERR_SHAPE;
  return attnf_bwd(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                   do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, nullptr, 0, 0, nullptr, keep, keep_ld, keep_scale, part, stream);
}

extern "C" int dav_attn_bwd_f32(const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* LSE,
                                float* Delta, float* dQ, float* dK, float* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, long do_bs,
                                int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs, float scale,
                                int part, hipStream_t stream) {
  return dav_attn_bias_bwd_f32(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                               o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, nullptr, 0, 0, nullptr,
                               part, stream);
}

extern "C" int dav_patch_gather_f32(const float* img, int B, int C, int T, int H, int W, int pt, const int* ids_keep32, int nk,
                                    float* A, hipStream_t stream) {
  if (B <= 0 || nk <= 0 || (H & 15) || (W & 15) || pt <= 0 || T % pt) return DAV_ERR_SHAPE;
  DAV_LAUNCH(patch_gather_f32_kernel, dim3(grid_for((long)B * nk * C * pt * 256)), dim3(256), 0, stream, img, B, C, T, pt, H, W, ids_keep32, nk, A);
  return dav_launch_status();
}

extern "C" int dav_rows_gather_f32(const float* x, long x_bs, int row_off, const int* ids32, int B, int n, int D, float* out,
                                   long out_bs, hipStream_t stream) {
  if (B <= 0 || n <= 0 || D <= 0) return DAV_ERR_SHAPE;
  if (out_bs == 0) out_bs = (long)n * D;      // dense [B*n, D]
  DAV_LAUNCH(rows_gather_f32_kernel, dim3(grid_for((long)B * n * D)), dim3(256), 0, stream, x, x_bs, row_off, ids32, B, n, D, out, out_bs);
  return dav_launch_status();
}

extern "C" int dav_pair_expand_f32(const float* Pv, const float* Pa, int B, int nv, int na, int Wd, float* out, hipStream_t stream) {
  if (B <= 0 || nv <= 0 || na <= 0 || Wd <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(pair_expand_f32_kernel, dim3(grid_for((long)B * nv * na * Wd)), dim3(256), 0, stream, Pv, Pa, B, nv, na, Wd, out);
  return dav_launch_status();
}

extern "C" int dav_pair_reduce_f32(const float* d, int B, int nv, int na, int Wd, float* dPv, float* dPa, hipStream_t stream) {
  if (B <= 0 || nv <= 0 || na <= 0 || Wd <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(pair_reduce_f32_kernel, dim3(grid_for((long)B * (nv + na) * Wd)), dim3(256), 0, stream, d, B, nv, na, Wd, dPv, dPa);
  return dav_launch_status();
}

extern "C" int dav_add_f32(const float* a, const float* b, float* out, long n, hipStream_t stream) {
  if (n <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(add_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, out, n);
  return dav_launch_status();
}
