// HBM-bound kernels of the AVMAE step for gfx950: random-masking index build, kept-patch gather,
// decoder un-shuffle, patchify + norm-pix MSE, factorised-pair expand/reduce, dtype casts,
// flat grad-norm and flat AdamW.  One 64-lane wave per row wherever a row reduction is needed.
#include "common.h"
#include "dav_kernels.h"

namespace {

// ------------------------------------------------------------------------------------------------
// random masking (models/avmae.py:120-142): per-row argsort(noise) by bitonic sort in LDS.
// Keys are (noise, index) pairs compared lexicographically -> a total order, so the permutation is
// unique and equals torch.argsort's for tie-free noise (bit-exact), stable for ties.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mask_build_kernel(const float* noise, int L, int P, int len_keep, int64_t* ids_keep,
                                                          int64_t* ids_restore, float* mask, int* ids_keep32,
                                                          int* ids_restore32) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* key = reinterpret_cast<float*>(smem);
  int* idx = reinterpret_cast<int*>(smem + (size_t)P * 4);
  const int row = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < P; i += nt) {
    key[i] = i < L ? noise[(long)row * L + i] : __int_as_float(0x7f800000);
    idx[i] = i;
  }
  __syncthreads();
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += nt) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const float a = key[i], b = key[ixj];
          const int ia = idx[i], ib = idx[ixj];
          const bool a_gt_b = (a > b) || (a == b && ia > ib);
          const bool up = (i & k) == 0;
          if (a_gt_b == up) { key[i] = b; key[ixj] = a; idx[i] = ib; idx[ixj] = ia; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < L; i += nt) {
    const int src = idx[i];               // ids_shuffle[i]
    ids_restore[(long)row * L + src] = i;
    ids_restore32[(long)row * L + src] = i;
    mask[(long)row * L + src] = i < len_keep ? 0.f : 1.f;
    if (i < len_keep) {
      ids_keep[(long)row * len_keep + i] = src;
      ids_keep32[(long)row * len_keep + i] = src;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// kept-patch gather (timm PatchEmbed im2row restricted to ids_keep; models/vits.py:93,100):
// A[b*nk + t, c*256 + py*16 + px] = bf16(img[b, c, gy*16+py, gx*16+px]);  patch = 16 fixed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* img, int B, int C, int T, int PT, int H, int W,
                                                           const int* ids, int nk, bf16_t* A) {
  // img [B, C, T, H, W] (T = PT = 1 for images); patch (PT, 16, 16); token = (gt*gH + gy)*gW + gx;
  // A[row, ((c*PT + dt)*16 + py)*16 + px] — the flattening of the Conv2d / Conv3d weight
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int gW = W >> 4, gH = H >> 4, rows = B * nk, K = C * PT * 256;
  for (int row = gw; row < rows; row += nwaves) {
    const int b = row / nk, t = row % nk;
    const int pidx = ids ? ids[row] : t;
    const int gx = pidx % gW, gy = (pidx / gW) % gH, gt = pidx / (gW * gH);
    for (int ch = lane; ch < C * PT * 32; ch += 64) {     // chunk = 8 consecutive px of one patch row
      const int cd = ch >> 5, py = (ch >> 1) & 15, half = ch & 1;
      const int c = cd / PT, dt = cd % PT;
      const float* src = img + ((((long)b * C + c) * T + gt * PT + dt) * H + gy * 16 + py) * W + gx * 16 + half * 8;
      const float4 a = reinterpret_cast<const float4*>(src)[0], d = reinterpret_cast<const float4*>(src)[1];
      uint4 w;
      w.x = pack2bf(a.x, a.y); w.y = pack2bf(a.z, a.w); w.z = pack2bf(d.x, d.y); w.w = pack2bf(d.z, d.w);
      *reinterpret_cast<uint4*>(A + (long)row * K + cd * 256 + py * 16 + half * 8) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// decoder un-shuffle (models/avmae.py:161-165): out[b, r] = (restore[b,r] < nk ? emb[b, restore] : mask_token) + pos[r]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unshuffle_fwd_kernel(const float* emb, const float* mask_token, const float* pos,
                                                            const int* restore, int B, int L, int nk, int D, float* out,
                                                            long out_bs, int out_row_off) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = D >> 2;
  for (int row = gw; row < B * L; row += nwaves) {
    const int b = row / L, r = row % L;
    const int j = restore[row];
    const float4* src = reinterpret_cast<const float4*>(j < nk ? emb + ((long)b * nk + j) * D : mask_token);
    const float4* pe = reinterpret_cast<const float4*>(pos + (long)r * D);
    float4* dst = reinterpret_cast<float4*>(out + b * out_bs + (long)(out_row_off + r) * D);
    for (int c = lane; c < nch; c += 64) {
      const float4 a = src[c], p = pe[c];
      dst[c] = float4{a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w};
    }
  }
}

// gather rows of an fp32 [B, R, D] tensor into a bf16 [B*n, D] matrix: out[b*n+t] = x[b, row_off + (ids ? ids[b*n+t] : t)]
__global__ __launch_bounds__(256) void rows_gather_cast_kernel(const float* x, long x_bs, int row_off, const int* ids, int B,
                                                               int n, int D, bf16_t* out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = D >> 2;
  for (int row = gw; row < B * n; row += nwaves) {
    const int b = row / n, t = row % n;
    const int j = ids ? ids[row] : t;
    const float4* src = reinterpret_cast<const float4*>(x + b * x_bs + (long)(row_off + j) * D);
    uint2* dst = reinterpret_cast<uint2*>(out + (long)row * D);
    for (int c = lane; c < nch; c += 64) {
      const float4 a = src[c];
      uint2 w; w.x = pack2bf(a.x, a.y); w.y = pack2bf(a.z, a.w);
      dst[c] = w;
    }
  }
}

// DropPath (stochastic depth, timm DropPath as used at models/vits.py:32-34 and models/fusion_blocks.py:69,78-79,130,199,
// 278): per-sample scale s[b] in {0, 1/keep}.  out[b, r] = res[b, r] + s[b] * y[b, r]  (fp32, rows of D; out may alias res)
__global__ __launch_bounds__(256) void rows_axpy_kernel(const float* res, const float* y, const float* scale, int B, int rows,
                                                        int D, float* out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = D >> 2;
  for (int row = gw; row < B * rows; row += nwaves) {
    const float sb = scale[row / rows];
    const float4* r4 = reinterpret_cast<const float4*>(res + (long)row * D);
    const float4* y4 = reinterpret_cast<const float4*>(y + (long)row * D);
    float4* o4 = reinterpret_cast<float4*>(out + (long)row * D);
    for (int c = lane; c < nch; c += 64) {
      const float4 a = r4[c], b = y4[c];
      o4[c] = float4{a.x + sb * b.x, a.y + sb * b.y, a.z + sb * b.z, a.w + sb * b.w};
    }
  }
}

// its backward on the branch side: out_bf16[b, r] = bf16(s[b] * g[b, r])  (the residual side passes g through unchanged)
__global__ __launch_bounds__(256) void rows_scale_cast_kernel(const float* g, const float* scale, int B, int rows, int D,
                                                              bf16_t* out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = D >> 2;
  for (int row = gw; row < B * rows; row += nwaves) {
    const float sb = scale[row / rows];
    const float4* g4 = reinterpret_cast<const float4*>(g + (long)row * D);
    uint2* o = reinterpret_cast<uint2*>(out + (long)row * D);
    for (int c = lane; c < nch; c += 64) {
      const float4 a = g4[c];
      uint2 w; w.x = pack2bf(sb * a.x, sb * a.y); w.y = pack2bf(sb * a.z, sb * a.w);
      o[c] = w;
    }
  }
}

// Dropout on activations (nn.Dropout of timm Mlp.drop1 / drop2 and of the attention modules' proj_drop: models/fusion_blocks.py:16,29,
// 44,58,101,116,166,185,233,260; fine-tuning constructors only — every pre-training config has drop = 0), with DropPath folded in:
//   out[b, r, :] = (res ? res[b, r, :] : 0) + (rowscale ? rowscale[b] : 1) * (keep[b, r, :] ? keep_scale : 0) * in[b, r, :]
// keep: bytes 0 / 1, [B * rows, D] (null = all kept); in / out bf16 or fp32; out may alias in or res.  The backward of the same
// step is the same kernel on the gradient (res = null).
template <bool IN_F32, bool OUT_F32>
__global__ __launch_bounds__(256) void dropout_rows_kernel(const void* in, const float* res, const unsigned char* keep, float keep_scale,
                                                           const float* rowscale, int B, int rows, int D, void* out) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int nch = D >> 2;
  for (int row = gw; row < B * rows; row += nwaves) {
    const float sb = rowscale ? rowscale[row / rows] : 1.f;
    const long base = (long)row * D;
    for (int c = lane; c < nch; c += 64) {
      float4 a;
      if (IN_F32) {
        a = reinterpret_cast<const float4*>((const float*)in + base)[c];
      } else {
        const uint2 w = reinterpret_cast<const uint2*>((const bf16_t*)in + base)[c];
        a = float4{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
      }
      float4 k = float4{sb, sb, sb, sb};
      if (keep) {
        const uchar4 m = reinterpret_cast<const uchar4*>(keep + base)[c];
        const float ks = sb * keep_scale;
        k = float4{m.x ? ks : 0.f, m.y ? ks : 0.f, m.z ? ks : 0.f, m.w ? ks : 0.f};
      }
      float4 r = float4{0.f, 0.f, 0.f, 0.f};
      if (res) r = reinterpret_cast<const float4*>(res + base)[c];
      const float4 o = float4{r.x + k.x * a.x, r.y + k.y * a.y, r.z + k.z * a.z, r.w + k.w * a.w};
      if (OUT_F32) {
        reinterpret_cast<float4*>((float*)out + base)[c] = o;
      } else {
        uint2 w; w.x = pack2bf(o.x, o.y); w.y = pack2bf(o.z, o.w);
        reinterpret_cast<uint2*>((bf16_t*)out + base)[c] = w;
      }
    }
  }
}

// batch reductions of the un-shuffle backward: dpos[r] += sum_b dx[b, off+r];  dmask_token += sum over masked rows
__global__ __launch_bounds__(256) void unshuffle_bwd_reduce_kernel(const float* dx, long dx_bs, int row_off, const int* restore,
                                                                   int B, int L, int nk, int D, float* dpos, float* dmask_token) {
  const int r = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float sp = 0.f, sm = 0.f;
    for (int b = 0; b < B; ++b) {
      const float v = dx[b * dx_bs + (long)(row_off + r) * D + d];
      sp += v;
      if (restore[(long)b * L + r] >= nk) sm += v;
    }
    dpos[(long)r * D + d] += sp;
    unsafeAtomicAdd(dmask_token + d, sm);
  }
}

// ------------------------------------------------------------------------------------------------
// patchify + norm-pix + per-patch MSE (models/avmae.py:182-214), target read straight from NCHW.
// patch vector element e = (py*16 + px)*C + c   ('nchpwq->nhwpqc': channel fastest)
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void patch_mse_fwd_kernel(const float* img, const float* pred, int B, int H, int W,
                                                            int norm_pix, float* loss_patch, float* tmean, float* trstd) {
  constexpr int P = 256 * C, PER = P / 64;
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int gW = W >> 4, L = (H >> 4) * gW;
  for (int row = gw; row < B * L; row += nwaves) {
    const int b = row / L, l = row % L, gy = l / gW, gx = l % gW;
    float t[PER];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = lane + 64 * k, pix = e / C, c = e % C;
      t[k] = img[(((long)b * C + c) * H + gy * 16 + (pix >> 4)) * W + gx * 16 + (pix & 15)];
      s += t[k];
    }
    float mean = 0.f, rstd = 1.f;
    if (norm_pix) {
      mean = wave_sum(s) / P;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < PER; ++k) q += (t[k] - mean) * (t[k] - mean);
      rstd = 1.f / sqrtf(wave_sum(q) / (P - 1) + 1.e-6f);     // unbiased variance (models/avmae.py:191)
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const float d = pred[(long)row * P + lane + 64 * k] - (t[k] - mean) * rstd;
      acc += d * d;
    }
    acc = wave_sum(acc) / P;
    if (lane == 0) { loss_patch[row] = acc; tmean[row] = mean; trstd[row] = rstd; }
  }
}

template <int C>
__global__ __launch_bounds__(256) void patch_mse_bwd_kernel(const float* img, const float* pred, const float* mask,
                                                            const float* tmean, const float* trstd, const float* mask_sum,
                                                            const float* gout, int B, int H, int W, bf16_t* dpred, float* dpred32) {
  constexpr int P = 256 * C, PER = P / 64;
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int gW = W >> 4, L = (H >> 4) * gW;
  const float coef = (gout ? gout[0] : 1.f) * 2.f / (P * mask_sum[0]);
  for (int row = gw; row < B * L; row += nwaves) {
    const int b = row / L, l = row % L, gy = l / gW, gx = l % gW;
    const float m = mask[row] * coef, mean = tmean[row], rstd = trstd[row];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int e = lane + 64 * k, pix = e / C, c = e % C;
      float g = 0.f;
      if (m != 0.f) {
        const float t = img[(((long)b * C + c) * H + gy * 16 + (pix >> 4)) * W + gx * 16 + (pix & 15)];
        g = m * (pred[(long)row * P + e] - (t - mean) * rstd);
      }
      if (dpred32) dpred32[(long)row * P + e] = g;      // fp32 path (f32_path.hip)
      else dpred[(long)row * P + e] = f2bf(g);
    }
  }
}

// loss = sum(loss_patch * mask) / sum(mask)   (models/avmae.py:197); also exports sum(mask)
__global__ __launch_bounds__(1024) void masked_mean_kernel(const float* v, const float* mask, int n, float* out, float* mask_sum) {
  __shared__ float sa[16], sb[16];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { a += v[i] * mask[i]; b += mask[i]; }
  a = wave_sum(a); b = wave_sum(b);
  if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { ta += sa[i]; tb += sb[i]; }
    out[0] = ta / tb;
    mask_sum[0] = tb;
  }
}

// ------------------------------------------------------------------------------------------------
// factorised (v, a) pairs (models/fusion_blocks.py:245-252) without materialising cat(x_v[i], x_a[j]):
// Linear(cat(xv_i, xa_j)) = W[:, :D] xv_i + W[:, D:] xa_j + b  ->  out[b, i*na+j] = Pv[b,i] + Pa[b,j]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_expand_kernel(const float* Pv, const float* Pa, int B, int nv, int na, int Wd,
                                                          bf16_t* out) {
  const long total = (long)B * nv * na * (Wd >> 2);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % (Wd >> 2);
    const long pr = i / (Wd >> 2);
    const int j = pr % na, ii = (pr / na) % nv, b = pr / ((long)na * nv);
    const float4 x = reinterpret_cast<const float4*>(Pv + ((long)b * nv + ii) * Wd)[c];
    const float4 y = reinterpret_cast<const float4*>(Pa + ((long)b * na + j) * Wd)[c];
    uint2 w; w.x = pack2bf(x.x + y.x, x.y + y.y); w.y = pack2bf(x.z + y.z, x.w + y.w);
    reinterpret_cast<uint2*>(out + pr * Wd)[c] = w;
  }
}

// dPv[b,i] = sum_j d[b, i*na+j],  dPa[b,j] = sum_i d[b, i*na+j]   (bf16 in, bf16 out, fp32 sums)
__global__ __launch_bounds__(256) void pair_reduce_kernel(const bf16_t* d, int B, int nv, int na, int Wd, bf16_t* dPv, bf16_t* dPa) {
  const long total = (long)B * (nv + na) * Wd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % Wd;
    const long rr = i / Wd;
    const int t = rr % (nv + na), b = rr / (nv + na);
    float s = 0.f;
    if (t < nv) {
      for (int j = 0; j < na; ++j) s += bf2f(d[(((long)b * nv + t) * na + j) * Wd + c]);
      dPv[((long)b * nv + t) * Wd + c] = f2bf(s);
    } else {
      const int j = t - nv;
      for (int ii = 0; ii < nv; ++ii) s += bf2f(d[(((long)b * nv + ii) * na + j) * Wd + c]);
      dPa[((long)b * na + j) * Wd + c] = f2bf(s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// casts
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* x, bf16_t* y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    uint2 w; w.x = pack2bf(a.x, a.y); w.y = pack2bf(a.z, a.w);
    reinterpret_cast<uint2*>(y)[i] = w;
  }
}
// out = a + b (fp32) together with its bf16 copy: the sum of the fusion tokens' two gradient streams at a layer boundary and the
// GEMM operand of the next block's backward in one pass (was: a torch add, then dav_cast_bf16)
__global__ __launch_bounds__(256) void add_cast_kernel(const float* a, const float* b, float* out, bf16_t* out_bf16, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    const float4 r = float4{x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
    reinterpret_cast<float4*>(out)[i] = r;
    uint2 w; w.x = pack2bf(r.x, r.y); w.y = pack2bf(r.z, r.w);
    reinterpret_cast<uint2*>(out_bf16)[i] = w;
  }
}
__global__ __launch_bounds__(256) void cast_bf16_tail_kernel(const float* x, bf16_t* y, long start, long n) {
  const long i = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = f2bf(x[i]);
}

// y[c, r] = bf16(x[r, c]) for x [R, Ccols] fp32 (weight transpose for the dgrad GEMM)
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* x, bf16_t* y, int R, int Cc) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? x[(long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) y[(long)c * R + r] = f2bf(tile[tx][i]);
  }
}

// grouped form: up to TR_GROUP_MAX transposes in ONE launch (the fused fusion-block tails read seven transposed weights per
// layer in the backward: 84 launches of 15 us each per step otherwise); problem table by value in the kernel arguments
constexpr int TR_GROUP_MAX = 96;
struct TrGroup {
  const float* x[TR_GROUP_MAX];
  bf16_t* y[TR_GROUP_MAX];
  int R[TR_GROUP_MAX], Cc[TR_GROUP_MAX];
  int first_block[TR_GROUP_MAX + 1];
  int count;
};
__global__ __launch_bounds__(256) void cast_transpose_grouped_kernel(const TrGroup g) {
  __shared__ float tile[64][65];
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  const float* x = g.x[pi];
  bf16_t* y = g.y[pi];
  const int R = g.R[pi], Cc = g.Cc[pi], lb = (int)blockIdx.x - g.first_block[pi], nbx = (Cc + 63) / 64;
  const int r0 = (lb / nbx) * 64, c0 = (lb % nbx) * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < R && c < Cc) ? x[(long)r * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < Cc && r < R) y[(long)c * R + r] = f2bf(tile[tx][i]);
  }
}

// ------------------------------------------------------------------------------------------------
// flat fp32 reductions / optimizer
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* x, long n, float* partial) {
  __shared__ float sw[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    s += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}
__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* partial, int n, float scale, float* out) {
  __shared__ double sw[16];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)partial[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += sw[i];
    out[0] = (float)(sqrt(t) * scale);
  }
}

// streaming accesses of the optimizer pass: 30 bytes per parameter that nothing reads again before the next step — marked
// non-temporal so that they do not push the backward's activations out of L2 / MALL when the pass runs beside it
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt4(const float* q) {
  const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(q));
  return float4{t[0], t[1], t[2], t[3]};
}
__device__ __forceinline__ void stnt4(float* q, const float4& x) {
  nt_f4 t; t[0] = x.x; t[1] = x.y; t[2] = x.z; t[3] = x.w;
  __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(q));
}
// AdamW over flat buffers.  seg_* describe contiguous segments (param groups laid out back to back):
// element i belongs to the segment s with seg_end[s-1] <= i < seg_end[s]; hyper[s] = {lr, weight_decay}.
// step-dependent bias corrections are read from device memory so a captured graph can be replayed.
// Segments start on 64-element boundaries (FlatParams.ALIGN), so a lane's 4 consecutive elements share one segment and
// every access is a 16-byte one; n is a multiple of 4.
__global__ __launch_bounds__(256) void adamw_flat_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, bf16_t* __restrict__ p_bf16, long n,
                                                         const long* __restrict__ seg_end, const float* __restrict__ hyper, int nseg, float beta1,
                                                         float beta2, float eps, const float* bias_corr, float grad_scale,
                                                         float* sumsq_out, int zero_grad, const unsigned char* __restrict__ keep_grad,
                                                         const float* __restrict__ gscale_dev) {
  constexpr long CHUNK = 256 * 4 * 4;     // contiguous elements per workgroup iteration: 4 float4 per lane
  __shared__ float sw[4];
  const float bc1 = bias_corr[0], bc2_sqrt = bias_corr[1];
  const float ob1 = 1.f - beta1, ob2 = 1.f - beta2;
  // device-side step guard (dav_step_guard): the multiplier a captured step cannot know at capture time — min(1, clip / norm) —
  // and 0 for "non-finite loss or gradient norm: leave parameters and moments alone" (the reference raises before its
  // optimizer step, train.py:166-167; a replayed graph cannot, so the update is skipped and the host raises when it looks)
  const float gsd = gscale_dev ? gscale_dev[0] : 1.f;
  const bool skip = gscale_dev != nullptr && !(gsd > 0.f);
  grad_scale *= skip ? 0.f : gsd;
  float ss = 0.f;                         // sum of squared (unscaled) gradients seen by this thread
  for (long base = (long)blockIdx.x * CHUNK; base < n; base += (long)gridDim.x * CHUNK) {
    long i = base + threadIdx.x * 4;
    int lo = 0, hi = nseg - 1;            // first segment with seg_end > i
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (i >= seg_end[mid]) lo = mid + 1; else hi = mid; }
    int s = lo;
#pragma unroll
    for (int it = 0; it < 4; ++it, i += 1024) {
      if (i >= n) break;
      while (s + 1 < nseg && i >= seg_end[s]) ++s;
      const unsigned char kg = keep_grad ? keep_grad[s] : 0;            // strictly 0 / 1 (fetched with the segment's hyper-parameters)
      const float lr = hyper[2 * s], wd = hyper[2 * s + 1];
      const bool fill = zero_grad && !kg;
      const float decay = 1.f - lr * wd, step = lr / bc1;
      const float4 gr = ldnt4(g + i);
      if (skip) {
        ss += gr.x * gr.x + gr.y * gr.y + gr.z * gr.z + gr.w * gr.w;
        if (fill) stnt4(g + i, float4{0.f, 0.f, 0.f, 0.f});
        continue;
      }
      float4 pi = ldnt4(p + i);
      float4 mi = ldnt4(m + i);
      float4 vi = ldnt4(v + i);
      ss += gr.x * gr.x + gr.y * gr.y + gr.z * gr.z + gr.w * gr.w;
      const float gx = gr.x * grad_scale, gy = gr.y * grad_scale, gz = gr.z * grad_scale, gw = gr.w * grad_scale;
      mi.x = beta1 * mi.x + ob1 * gx; mi.y = beta1 * mi.y + ob1 * gy; mi.z = beta1 * mi.z + ob1 * gz; mi.w = beta1 * mi.w + ob1 * gw;
      vi.x = beta2 * vi.x + ob2 * gx * gx; vi.y = beta2 * vi.y + ob2 * gy * gy;
      vi.z = beta2 * vi.z + ob2 * gz * gz; vi.w = beta2 * vi.w + ob2 * gw * gw;
      pi.x = pi.x * decay - step * mi.x / (sqrtf(vi.x) / bc2_sqrt + eps);
      pi.y = pi.y * decay - step * mi.y / (sqrtf(vi.y) / bc2_sqrt + eps);
      pi.z = pi.z * decay - step * mi.z / (sqrtf(vi.z) / bc2_sqrt + eps);
      pi.w = pi.w * decay - step * mi.w / (sqrtf(vi.w) / bc2_sqrt + eps);
      stnt4(p + i, pi);
      stnt4(m + i, mi);
      stnt4(v + i, vi);
      if (p_bf16) {
        uint2 w; w.x = pack2bf(pi.x, pi.y); w.y = pack2bf(pi.z, pi.w);
        *reinterpret_cast<uint2*>(p_bf16 + i) = w;
      }
      if (fill) stnt4(g + i, float4{0.f, 0.f, 0.f, 0.f});
    }
  }
  if (sumsq_out) {                        // one atomic per workgroup (<= 16384 distinct-time adds on one word)
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(sumsq_out, sw[0] + sw[1] + sw[2] + sw[3]);
  }
}

// one thread: the scale of this step's update, see adamw_flat_kernel
__global__ void step_guard_kernel(const float* loss_a, const float* loss_b, const float* gnorm, float clip, float grad_scale,
                                  float* out_scale, int* bad_count) {
  float s = 1.f;
  if (gnorm) {
    const float norm = gnorm[0] * grad_scale;
    if (!isfinite(norm)) s = 0.f;
    else if (clip > 0.f) s = fminf(1.f, clip / (norm + 1e-6f));
  }
  float l = 0.f;
  if (loss_a) l += loss_a[0];
  if (loss_b) l += loss_b[0];
  if (!isfinite(l)) s = 0.f;
  out_scale[0] = s;
  if (s == 0.f && bad_count) bad_count[0] += 1;
}

int wave_grid(long rows) {
  long g = (rows + 3) / 4;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int dav_mask_build(const float* noise, int N, int L, int len_keep, int64_t* ids_keep, int64_t* ids_restore,
                              float* mask, int* ids_keep32, int* ids_restore32, hipStream_t stream) {
  if (N <= 0 || L <= 0 || L > 4096 || len_keep < 0 || len_keep > L) return DAV_ERR_SHAPE;
  int P = 1;
  while (P < L) P <<= 1;
  int nt = P / 2 < 64 ? 64 : (P / 2 > 1024 ? 1024 : P / 2);
  DAV_LAUNCH(mask_build_kernel, dim3(N), dim3(nt), (size_t)P * 8, stream, noise, L, P, len_keep, ids_keep,
                     ids_restore, mask, ids_keep32, ids_restore32);
  return dav_launch_status();
}

extern "C" int dav_patch_gather(const float* img, int B, int C, int H, int W, const int* ids_keep32, int nk, void* A,
                                hipStream_t stream) {
  if (B <= 0 || C <= 0 || (H & 15) || (W & 15) || nk <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(patch_gather_kernel, dim3(wave_grid((long)B * nk)), dim3(256), 0, stream, img, B, C, 1, 1, H, W, ids_keep32,
                     nk, (bf16_t*)A);
  return dav_launch_status();
}

extern "C" int dav_patch_gather3d(const float* video, int B, int C, int T, int H, int W, int pt, const int* ids_keep32,
                                  int nk, void* A, hipStream_t stream) {
  if (B <= 0 || C <= 0 || pt <= 0 || T <= 0 || (T % pt) || (H & 15) || (W & 15) || nk <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(patch_gather_kernel, dim3(wave_grid((long)B * nk)), dim3(256), 0, stream, video, B, C, T, pt, H, W,
                     ids_keep32, nk, (bf16_t*)A);
  return dav_launch_status();
}

extern "C" int dav_unshuffle_fwd(const float* emb, const float* mask_token, const float* pos, const int* ids_restore32, int B,
                                 int L, int nk, int D, float* out, long out_bs, int out_row_off, hipStream_t stream) {
  if (B <= 0 || L <= 0 || (D & 3)) return DAV_ERR_SHAPE;
  DAV_LAUNCH(unshuffle_fwd_kernel, dim3(wave_grid((long)B * L)), dim3(256), 0, stream, emb, mask_token, pos,
                     ids_restore32, B, L, nk, D, out, out_bs, out_row_off);
  return dav_launch_status();
}

extern "C" int dav_rows_gather_cast(const float* x, long x_bs, int row_off, const int* ids32, int B, int n, int D, void* out,
                                    hipStream_t stream) {
  if (B <= 0 || n <= 0 || (D & 3)) return DAV_ERR_SHAPE;
  DAV_LAUNCH(rows_gather_cast_kernel, dim3(wave_grid((long)B * n)), dim3(256), 0, stream, x, x_bs, row_off, ids32, B, n,
                     D, (bf16_t*)out);
  return dav_launch_status();
}

extern "C" int dav_rows_axpy(const float* res, const float* y, const float* scale, int B, int rows, int D, float* out,
                             hipStream_t stream) {
  if (B <= 0 || rows <= 0 || D <= 0 || (D & 3)) return DAV_ERR_SHAPE;
  DAV_LAUNCH(rows_axpy_kernel, dim3(wave_grid((long)B * rows)), dim3(256), 0, stream, res, y, scale, B, rows, D, out);
  return dav_launch_status();
}

extern "C" int dav_rows_scale_cast(const float* g, const float* scale, int B, int rows, int D, void* out_bf16,
                                   hipStream_t stream) {
  if (B <= 0 || rows <= 0 || D <= 0 || (D & 3)) return DAV_ERR_SHAPE;
  DAV_LAUNCH(rows_scale_cast_kernel, dim3(wave_grid((long)B * rows)), dim3(256), 0, stream, g, scale, B, rows, D,
                     (bf16_t*)out_bf16);
  return dav_launch_status();
}

extern "C" int dav_dropout_rows(const void* in, int in_f32, const float* res, const void* keep, float keep_scale, const float* rowscale,
                                int B, int rows, int D, void* out, int out_f32, hipStream_t stream) {
  if (B <= 0 || rows <= 0 || D <= 0 || (D & 3)) return DAV_ERR_SHAPE;
  if (!in || !out) return DAV_ERR_SHAPE;
  if (keep && !(keep_scale > 0.f)) return DAV_ERR_SHAPE;
  if (((uintptr_t)in | (uintptr_t)out | (uintptr_t)res) & 7 || ((uintptr_t)keep & 3)) return DAV_ERR_ALIGN;
  const dim3 grid(wave_grid((long)B * rows)), block(256);
  const unsigned char* kp = (const unsigned char*)keep;
  if (in_f32 && out_f32) { DAV_LAUNCH((dropout_rows_kernel<true, true>), grid, block, 0, stream, in, res, kp, keep_scale, rowscale, B, rows, D, out); }
  else if (in_f32) { DAV_LAUNCH((dropout_rows_kernel<true, false>), grid, block, 0, stream, in, res, kp, keep_scale, rowscale, B, rows, D, out); }
  else if (out_f32) { DAV_LAUNCH((dropout_rows_kernel<false, true>), grid, block, 0, stream, in, res, kp, keep_scale, rowscale, B, rows, D, out); }
  else { DAV_LAUNCH((dropout_rows_kernel<false, false>), grid, block, 0, stream, in, res, kp, keep_scale, rowscale, B, rows, D, out); }
  return dav_launch_status();
}

extern "C" int dav_unshuffle_bwd_reduce(const float* dx, long dx_bs, int row_off, const int* ids_restore32, int B, int L, int nk,
                                        int D, float* dpos, float* dmask_token, hipStream_t stream) {
  if (B <= 0 || L <= 0 || D <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(unshuffle_bwd_reduce_kernel, dim3(L), dim3(256), 0, stream, dx, dx_bs, row_off, ids_restore32, B, L, nk, D,
                     dpos, dmask_token);
  return dav_launch_status();
}

extern "C" int dav_patch_mse_fwd(const float* img, const float* pred, const float* mask, int B, int C, int H, int W,
                                 int norm_pix, float* loss_patch, float* tmean, float* trstd, float* loss, float* mask_sum,
                                 hipStream_t stream) {
  if (B <= 0 || (H & 15) || (W & 15) || (C != 1 && C != 3)) return DAV_ERR_SHAPE;
  const long rows = (long)B * (H >> 4) * (W >> 4);
  if (C == 3) DAV_LAUNCH(patch_mse_fwd_kernel<3>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, B, H, W, norm_pix, loss_patch, tmean, trstd);
  else DAV_LAUNCH(patch_mse_fwd_kernel<1>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, B, H, W, norm_pix, loss_patch, tmean, trstd);
  DAV_LAUNCH(masked_mean_kernel, dim3(1), dim3(1024), 0, stream, loss_patch, mask, (int)rows, loss, mask_sum);
  return dav_launch_status();
}

extern "C" int dav_patch_mse_bwd(const float* img, const float* pred, const float* mask, const float* tmean, const float* trstd,
                                 const float* mask_sum, const float* gout, int B, int C, int H, int W, void* dpred_bf16,
                                 hipStream_t stream) {
  if (B <= 0 || (H & 15) || (W & 15) || (C != 1 && C != 3)) return DAV_ERR_SHAPE;
  const long rows = (long)B * (H >> 4) * (W >> 4);
  if (C == 3) DAV_LAUNCH(patch_mse_bwd_kernel<3>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, mask, tmean, trstd, mask_sum, gout, B, H, W, (bf16_t*)dpred_bf16, (float*)nullptr);
  else DAV_LAUNCH(patch_mse_bwd_kernel<1>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, mask, tmean, trstd, mask_sum, gout, B, H, W, (bf16_t*)dpred_bf16, (float*)nullptr);
  return dav_launch_status();
}

extern "C" int dav_patch_mse_bwd_f32(const float* img, const float* pred, const float* mask, const float* tmean, const float* trstd,
                                     const float* mask_sum, const float* gout, int B, int C, int H, int W, float* dpred, hipStream_t stream) {
  if ((C != 1 && C != 3) || (H & 15) || (W & 15)) return DAV_ERR_SHAPE;
  const long rows = (long)B * (H >> 4) * (W >> 4);
  if (C == 3) DAV_LAUNCH(patch_mse_bwd_kernel<3>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, mask, tmean, trstd, mask_sum, gout, B, H, W, (bf16_t*)nullptr, dpred);
  else DAV_LAUNCH(patch_mse_bwd_kernel<1>, dim3(wave_grid(rows)), dim3(256), 0, stream, img, pred, mask, tmean, trstd, mask_sum, gout, B, H, W, (bf16_t*)nullptr, dpred);
  return dav_launch_status();
}

extern "C" int dav_pair_expand(const float* Pv, const float* Pa, int B, int nv, int na, int Wd, void* out_bf16, hipStream_t stream) {
  if (B <= 0 || nv <= 0 || na <= 0 || (Wd & 3)) return DAV_ERR_SHAPE;
  const long total = (long)B * nv * na * (Wd >> 2);
  long g = (total + 255) / 256; g = g > 4096 ? 4096 : g;
  DAV_LAUNCH(pair_expand_kernel, dim3((int)g), dim3(256), 0, stream, Pv, Pa, B, nv, na, Wd, (bf16_t*)out_bf16);
  return dav_launch_status();
}

extern "C" int dav_pair_reduce(const void* d_bf16, int B, int nv, int na, int Wd, void* dPv_bf16, void* dPa_bf16, hipStream_t stream) {
  if (B <= 0 || nv <= 0 || na <= 0 || Wd <= 0) return DAV_ERR_SHAPE;
  const long total = (long)B * (nv + na) * Wd;
  long g = (total + 255) / 256; g = g > 4096 ? 4096 : g;
  DAV_LAUNCH(pair_reduce_kernel, dim3((int)g), dim3(256), 0, stream, (const bf16_t*)d_bf16, B, nv, na, Wd, (bf16_t*)dPv_bf16, (bf16_t*)dPa_bf16);
  return dav_launch_status();
}

extern "C" int dav_cast_bf16(const float* x, void* y_bf16, long n, hipStream_t stream) {
  if (n <= 0) return DAV_ERR_SHAPE;
  const long n4 = n >> 2;
  if (n4 > 0) {
    long g = (n4 + 255) / 256; g = g > 8192 ? 8192 : g;
    DAV_LAUNCH(cast_bf16_kernel, dim3((int)g), dim3(256), 0, stream, x, (bf16_t*)y_bf16, n4);
  }
  if (n & 3) DAV_LAUNCH(cast_bf16_tail_kernel, dim3(1), dim3(64), 0, stream, x, (bf16_t*)y_bf16, n4 << 2, n);
  return dav_launch_status();
}

extern "C" int dav_add_cast(const float* a, const float* b, float* out, void* out_bf16, long n, hipStream_t stream) {
  if (n <= 0 || (n & 3)) return DAV_ERR_SHAPE;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15 || ((uintptr_t)out_bf16 & 7)) return DAV_ERR_ALIGN;
  const long n4 = n >> 2;
  long g = (n4 + 255) / 256; g = g > 8192 ? 8192 : g;
  DAV_LAUNCH(add_cast_kernel, dim3((int)g), dim3(256), 0, stream, a, b, out, (bf16_t*)out_bf16, n4);
  return dav_launch_status();
}

extern "C" int dav_cast_transpose_bf16(const float* x, void* y_bf16, int R, int C, hipStream_t stream) {
  if (R <= 0 || C <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(cast_transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, x, (bf16_t*)y_bf16, R, C);
  return dav_launch_status();
}

extern "C" int dav_cast_transpose_grouped(const DavTranspose* items, int count, hipStream_t stream) {
  if (count <= 0 || !items) return DAV_ERR_SHAPE;
  for (int i0 = 0; i0 < count; i0 += TR_GROUP_MAX) {
    TrGroup g;
    const int n = count - i0 < TR_GROUP_MAX ? count - i0 : TR_GROUP_MAX;
    int first = 0;
    for (int i = 0; i < n; ++i) {
      const DavTranspose& it = items[i0 + i];
      if (it.R <= 0 || it.C <= 0 || !it.x || !it.y_bf16) return DAV_ERR_SHAPE;
      g.x[i] = it.x; g.y[i] = (bf16_t*)it.y_bf16; g.R[i] = it.R; g.Cc[i] = it.C;
      g.first_block[i] = first;
      first += ((it.C + 63) / 64) * ((it.R + 63) / 64);
    }
    g.first_block[n] = first;
    g.count = n;
    const TrGroup gl = g;
    DAV_LAUNCH(cast_transpose_grouped_kernel, dim3(first), dim3(256), 0, stream, gl);
  }
  return dav_launch_status();
}

extern "C" size_t dav_l2norm_workspace_bytes(long n) { (void)n; return 1024 * sizeof(float); }

extern "C" int dav_l2norm(const float* x, long n, float scale, float* out, void* workspace, size_t workspace_bytes,
                          hipStream_t stream) {
  if (n <= 0) return DAV_ERR_SHAPE;
  if (workspace_bytes < 1024 * sizeof(float)) return DAV_ERR_WORKSPACE;
  long g = ((n >> 2) + 255) / 256; g = g > 1024 ? 1024 : (g < 1 ? 1 : g);
  DAV_LAUNCH(sumsq_partial_kernel, dim3((int)g), dim3(256), 0, stream, x, n, (float*)workspace);
  DAV_LAUNCH(sumsq_final_kernel, dim3(1), dim3(1024), 0, stream, (const float*)workspace, (int)g, scale, out);
  return dav_launch_status();
}

extern "C" int dav_step_guard(const float* loss_a, const float* loss_b, const float* gnorm, float clip, float grad_scale,
                              float* out_scale, int* bad_count, hipStream_t stream) {
  if (!out_scale) return DAV_ERR_SHAPE;
  DAV_LAUNCH(step_guard_kernel, dim3(1), dim3(1), 0, stream, loss_a, loss_b, gnorm, clip, grad_scale, out_scale, bad_count);
  return dav_launch_status();
}

extern "C" int dav_adamw_flat(float* p, float* g, float* m, float* v, void* p_bf16, long n, const long* seg_end,
                              const float* hyper, int nseg, float beta1, float beta2, float eps, const float* bias_corr,
                              float grad_scale, float* sumsq_out, int zero_grad, const unsigned char* keep_grad,
                              const float* gscale_dev, hipStream_t stream) {
  if (n <= 0 || nseg <= 0 || (n & 3)) return DAV_ERR_SHAPE;
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15 || ((uintptr_t)p_bf16 & 7)) return DAV_ERR_ALIGN;
  if (sumsq_out) HIP_CHECK_RET(hipMemsetAsync(sumsq_out, 0, sizeof(float), stream));
  long g2 = (n + 4095) / 4096; g2 = g2 > 16384 ? 16384 : g2;
  DAV_LAUNCH(adamw_flat_kernel, dim3((int)g2), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n, seg_end, hyper, nseg,
                     beta1, beta2, eps, bias_corr, grad_scale, sumsq_out, zero_grad, keep_grad, gscale_dev);
  return dav_launch_status();
}
