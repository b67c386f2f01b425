// LayerNorm forward / backward for gfx950: one 64-lane wave per row, float4 loads, shuffle
// reductions, fp32 statistics.  HBM-bound.
//
// The forward can read its rows from TWO fp32 sources per batch element (segment 0 = r0 rows,
// segment 1 = r1 rows) — this is the `torch.cat((x_fusion, x_image), dim=1)` of
// models/deepavfusion.py:104-105 folded into norm1 of the modality block (timm Block.norm1), with
// batch stride 0 allowed for the `fusion_tokens.expand(B, -1, -1)` of :97.  The backward scatters
// the row gradients back to the two sources (store or accumulate, optional fp32 residual-gradient
// add and bf16 copy for the next dgrad GEMM) and accumulates dgamma / dbeta with one atomic per
// column per workgroup.
//
// eps differs by call site (1e-6 encoder blocks, 1e-5 fusion blocks / decoders): SURVEY Appendix A.1.
#include "common.h"
#include "dav_kernels.h"

namespace {

constexpr int MAXC_LIMIT = 8;   // float4 chunks per lane -> D <= 2048 (kernels are instantiated for 1, 2, 3, 4, 8)

struct LNSeg {
  const float* x; long bs; int rows;      // source rows of this segment (batch stride in elements)
};

struct LNFwd {
  LNSeg s0, s1;
  int B, D;
  const float *gamma, *beta;
  float eps;
  bf16_t* y;        // [B*(r0+r1), D] bf16 (may be null)
  float* y32;       // same rows, fp32 (may be null)
  float *mean, *rstd;
};

struct LNDst {
  float* dx; long bs; int rows;   // destination gradient rows of this segment
  int accumulate;                 // 1: dx += ..., 0: dx = ...
  const float* res; long res_bs;  // optional fp32 residual gradient added on top (same row layout)
  bf16_t* dx_bf16; long bf_bs;    // optional bf16 copy of the written value
};

struct LNBwd {
  LNSeg s0, s1;
  LNDst d0, d1;
  int B, D;
  const bf16_t* dy;     // [B*R, D] bf16 (may be null)
  const float* dy32;    // [B*R, D] fp32 (may be null); total dy = dy + dy32
  const float *gamma, *mean, *rstd;
  float *dgamma, *dbeta;   // accumulated
  float* partial;          // workspace [gridDim.x][2*D]
  // LayerNorm folded away on the forward path (dav_layernorm_bwd_twin): x as bf16 twin + statistics partials, h_out = gamma xhat + beta
  const bf16_t *xb0, *xb1;   // twin rows of the two segments (batch strides s0.bs / s1.bs); null = the fp32 sources above
  const float *st0, *st1;    // {sum, sum of squares} per 64-column slot, row (b * bs / D + r)
  const float* beta;
  bf16_t* h_out;             // [B*R, D] or null
  float eps;
};

template <int MAXC>
__device__ __forceinline__ void ln_fwd_body(const LNFwd& p, const int vbid, const int vgrid) {
  const int lane = threadIdx.x & 63;
  const int gw = (vbid * blockDim.x + threadIdx.x) >> 6, nwaves = (vgrid * blockDim.x) >> 6;
  const int R = p.s0.rows + p.s1.rows, rows = p.B * R, nch = p.D >> 2;
  float4 gmv[MAXC], btv[MAXC];          // the affine parameters are the same for every row of the wave: read once
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    gmv[i] = c < nch ? reinterpret_cast<const float4*>(p.gamma)[c] : float4{0.f, 0.f, 0.f, 0.f};
    btv[i] = c < nch ? reinterpret_cast<const float4*>(p.beta)[c] : float4{0.f, 0.f, 0.f, 0.f};
  }
  for (int row = gw; row < rows; row += nwaves) {
    const int b = row / R, j = row % R;
    const float* x = j < p.s0.rows ? p.s0.x + b * p.s0.bs + (long)j * p.D : p.s1.x + b * p.s1.bs + (long)(j - p.s0.rows) * p.D;
    float4 v[MAXC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        v[i] = reinterpret_cast<const float4*>(x)[c];
        s += v[i].x + v[i].y + v[i].z + v[i].w;
      }
    }
    const float mean = wave_sum(s) / p.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += a * a + bb * bb + cc * cc + d * d;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / p.D + p.eps);
    if (lane == 0) { p.mean[row] = mean; p.rstd[row] = rstd; }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 gm = gmv[i], bt = btv[i];
        float4 o;
        o.x = (v[i].x - mean) * rstd * gm.x + bt.x;
        o.y = (v[i].y - mean) * rstd * gm.y + bt.y;
        o.z = (v[i].z - mean) * rstd * gm.z + bt.z;
        o.w = (v[i].w - mean) * rstd * gm.w + bt.w;
        if (p.y) {
          uint2 w; w.x = pack2bf(o.x, o.y); w.y = pack2bf(o.z, o.w);
          reinterpret_cast<uint2*>(p.y + (long)row * p.D)[c] = w;
        }
        if (p.y32) reinterpret_cast<float4*>(p.y32 + (long)row * p.D)[c] = o;
      }
    }
  }
}

template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LNFwd p) {
  ln_fwd_body<MAXC>(p, blockIdx.x, gridDim.x);
}

// grouped launches (batch.h): several independent LayerNorms (same float4-chunk count) in one grid — norm1 of the image
// and the audio block of a layer, the three input norms of a fusion block, ...
constexpr int LN_BATCH_MAX = 8;
struct LNFwdGroup {
  LNFwd prob[LN_BATCH_MAX];
  int first_block[LN_BATCH_MAX + 1];
  int count;
};
template <int MAXC>
__global__ __launch_bounds__(256) void ln_fwd_grouped_kernel(const LNFwdGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  const LNFwd p = g.prob[pi];          // (registers, not a reference into the by-value table: no scalar reloads inside the row loop)
  ln_fwd_body<MAXC>(p, (int)blockIdx.x - g.first_block[pi], g.first_block[pi + 1] - g.first_block[pi]);
}

// 8 waves per workgroup (512 workgroups -> 16 waves/CU in flight).  dgamma/dbeta: the waves of a workgroup combine their partial sums in LDS and write ONE partial row
// [2*D] per workgroup to the caller's workspace; ln_bwd_reduce_kernel then sums the rows (no same-address
// atomics, which serialise in L2 when hundreds of workgroups hit the same 2*D words).
template <int MAXC>
__device__ __forceinline__ void ln_bwd_body(const LNBwd& p, const int vbid, const int vgrid) {
  extern __shared__ __attribute__((aligned(16))) float lds_red[];   // [waves][2][D]
  const int lane = threadIdx.x & 63;
  const int gw = (vbid * blockDim.x + threadIdx.x) >> 6, nwaves = (vgrid * blockDim.x) >> 6;
  const int R = p.s0.rows + p.s1.rows, rows = p.B * R, nch = p.D >> 2;
  float4 dg[MAXC], db[MAXC], gmv[MAXC];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    dg[i] = float4{0.f, 0.f, 0.f, 0.f}; db[i] = float4{0.f, 0.f, 0.f, 0.f};
    const int c = lane + 64 * i;
    gmv[i] = c < nch ? reinterpret_cast<const float4*>(p.gamma)[c] : float4{0.f, 0.f, 0.f, 0.f};      // same for every row: read once
  }

  for (int row = gw; row < rows; row += nwaves) {
    const int b = row / R, j = row % R;
    const bool seg0 = j < p.s0.rows;
    const int jj = seg0 ? j : j - p.s0.rows;
    const LNSeg& sg = seg0 ? p.s0 : p.s1;
    const LNDst& d = seg0 ? p.d0 : p.d1;
    const bool tw = p.xb0 != nullptr;            // x as bf16 twin + statistics partials (the LayerNorm was folded away on the forward path)
    const float* x = tw ? nullptr : sg.x + b * sg.bs + (long)jj * p.D;
    const bf16_t* xb = tw ? (seg0 ? p.xb0 : p.xb1) + b * sg.bs + (long)jj * p.D : nullptr;
    // every global read of the row — statistics, x, dy, the residual gradient — is requested before the first one is used: one memory
    // latency per row
    float2 stp[4];
    float mean = 0.f, rstd = 0.f;
    const int ns = p.D >> 6;
    if (tw) {
      const float2* sp = reinterpret_cast<const float2*>(seg0 ? p.st0 : p.st1) + ((long)b * (sg.bs / p.D) + jj) * ns;
#pragma unroll
      for (int i = 0; i < 4; ++i) stp[i] = (lane & 3) + 4 * i < ns ? sp[(lane & 3) + 4 * i] : float2{0.f, 0.f};
    } else {
      mean = p.mean[row]; rstd = p.rstd[row];
    }
    float4 xh[MAXC], gy[MAXC], rv[MAXC];
    const float* rr = (d.dx && d.res) ? d.res + b * d.res_bs + (long)jj * p.D : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        rv[i] = rr ? reinterpret_cast<const float4*>(rr)[c] : float4{0.f, 0.f, 0.f, 0.f};
        if (tw) {
          const uint2 w = reinterpret_cast<const uint2*>(xb)[c];
          xh[i].x = __uint_as_float(w.x << 16); xh[i].y = __uint_as_float(w.x & 0xffff0000u);
          xh[i].z = __uint_as_float(w.y << 16); xh[i].w = __uint_as_float(w.y & 0xffff0000u);
        } else {
          xh[i] = reinterpret_cast<const float4*>(x)[c];
        }
        float4 dyv = float4{0.f, 0.f, 0.f, 0.f};
        if (p.dy) {
          const uint2 w = reinterpret_cast<const uint2*>(p.dy + (long)row * p.D)[c];
          dyv.x = __uint_as_float(w.x << 16); dyv.y = __uint_as_float(w.x & 0xffff0000u);
          dyv.z = __uint_as_float(w.y << 16); dyv.w = __uint_as_float(w.y & 0xffff0000u);
        }
        if (p.dy32) {
          const float4 t = reinterpret_cast<const float4*>(p.dy32 + (long)row * p.D)[c];
          dyv.x += t.x; dyv.y += t.y; dyv.z += t.z; dyv.w += t.w;
        }
        gy[i] = dyv;
      }
    }
    if (tw) {      // the same arithmetic, in the same order, as the consuming GEMM (dav_ln_row_stats): four lanes per row there, every group of four here
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { a1 += stp[i].x; a2 += stp[i].y; }
      a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
      a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
      const float inv = 1.0f / (float)p.D;
      mean = a1 * inv;
      rstd = rsqrtf(fmaxf(a2 * inv - mean * mean, 0.f) + p.eps);
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        const float4 gm = gmv[i], dyv = gy[i];
        xh[i].x = (xh[i].x - mean) * rstd; xh[i].y = (xh[i].y - mean) * rstd;
        xh[i].z = (xh[i].z - mean) * rstd; xh[i].w = (xh[i].w - mean) * rstd;
        dg[i].x += dyv.x * xh[i].x; dg[i].y += dyv.y * xh[i].y; dg[i].z += dyv.z * xh[i].z; dg[i].w += dyv.w * xh[i].w;
        db[i].x += dyv.x; db[i].y += dyv.y; db[i].z += dyv.z; db[i].w += dyv.w;
        gy[i].x = dyv.x * gm.x; gy[i].y = dyv.y * gm.y; gy[i].z = dyv.z * gm.z; gy[i].w = dyv.w * gm.w;
        s1 += gy[i].x + gy[i].y + gy[i].z + gy[i].w;
        s2 += gy[i].x * xh[i].x + gy[i].y * xh[i].y + gy[i].z * xh[i].z + gy[i].w * xh[i].w;
      }
    }
    s1 = wave_sum(s1) / p.D;
    s2 = wave_sum(s2) / p.D;
    if (p.h_out) {          // the consuming Linear's weight-gradient operand: LayerNorm output, re-made here
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
          const float4 gm = gmv[i], bt = reinterpret_cast<const float4*>(p.beta)[c];
          uint2 w;
          w.x = pack2bf(xh[i].x * gm.x + bt.x, xh[i].y * gm.y + bt.y);
          w.y = pack2bf(xh[i].z * gm.z + bt.z, xh[i].w * gm.w + bt.w);
          reinterpret_cast<uint2*>(p.h_out + (long)row * p.D)[c] = w;
        }
      }
    }
    if (d.dx == nullptr) continue;   // caller does not need this segment's input gradient
    float* dxr = d.dx + b * d.bs + (long)jj * p.D;
    bf16_t* br = d.dx_bf16 ? d.dx_bf16 + b * d.bf_bs + (long)jj * p.D : nullptr;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c < nch) {
        float4 o;
        o.x = rstd * (gy[i].x - s1 - xh[i].x * s2);
        o.y = rstd * (gy[i].y - s1 - xh[i].y * s2);
        o.z = rstd * (gy[i].z - s1 - xh[i].z * s2);
        o.w = rstd * (gy[i].w - s1 - xh[i].w * s2);
        o.x += rv[i].x; o.y += rv[i].y; o.z += rv[i].z; o.w += rv[i].w;
        if (d.accumulate) { const float4 t = reinterpret_cast<const float4*>(dxr)[c]; o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
        reinterpret_cast<float4*>(dxr)[c] = o;
        if (br) { uint2 w; w.x = pack2bf(o.x, o.y); w.y = pack2bf(o.z, o.w); reinterpret_cast<uint2*>(br)[c] = w; }
      }
    }
  }

  if (p.partial == nullptr) return;
  // per-wave slices [wave][2][D] written as float4 (conflict-free), then summed over the waves -> one partial row
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float* rg = lds_red + (size_t)wave * 2 * p.D;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c < nch) {
      reinterpret_cast<float4*>(rg)[c] = dg[i];
      reinterpret_cast<float4*>(rg + p.D)[c] = db[i];
    }
  }
  __syncthreads();
  float* part = p.partial + (size_t)vbid * 2 * p.D;
  for (int c = threadIdx.x; c < 2 * p.D; c += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += lds_red[(size_t)w * 2 * p.D + c];
    part[c] = s;
  }
}

template <int MAXC>
__global__ __launch_bounds__(512) void ln_bwd_kernel(LNBwd p) {
  ln_bwd_body<MAXC>(p, blockIdx.x, gridDim.x);
}

struct LNBwdGroup {
  LNBwd prob[LN_BATCH_MAX];
  int first_block[LN_BATCH_MAX + 1];
  int count;
};
template <int MAXC>
__global__ __launch_bounds__(512) void ln_bwd_grouped_kernel(const LNBwdGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  const LNBwd p = g.prob[pi];
  ln_bwd_body<MAXC>(p, (int)blockIdx.x - g.first_block[pi], g.first_block[pi + 1] - g.first_block[pi]);
}

// 64 columns x 16 row-lanes per workgroup: every thread sums its share of the partial rows with 4 independent
// accumulators (loads in flight), the 16 row-lanes are combined through LDS.
__global__ __launch_bounds__(1024) void ln_bwd_reduce_kernel(const float* partial, int nrows, int D, float* dgamma, float* dbeta) {
  __shared__ float red[16][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    int r = ty;
    for (; r + 48 < nrows; r += 64) {
      s0 += partial[(size_t)r * 2 * D + c];
      s1 += partial[(size_t)(r + 16) * 2 * D + c];
      s2 += partial[(size_t)(r + 32) * 2 * D + c];
      s3 += partial[(size_t)(r + 48) * 2 * D + c];
    }
    for (; r < nrows; r += 16) s0 += partial[(size_t)r * 2 * D + c];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < 2 * D) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    if (c < D) dgamma[c] += s; else dbeta[c - D] += s;
  }
}

// grouped form: up to LN_GROUP_MAX deferred (partial, nrows, D, dgamma, dbeta) reductions in one launch
constexpr int LN_GROUP_MAX = 64;
struct LNReduceGroup {
  const float* partial[LN_GROUP_MAX];
  float* dgamma[LN_GROUP_MAX];
  float* dbeta[LN_GROUP_MAX];
  int nrows[LN_GROUP_MAX], D[LN_GROUP_MAX], first_block[LN_GROUP_MAX + 1];
  int count;
};

__global__ __launch_bounds__(1024) void ln_bwd_reduce_grouped_kernel(const LNReduceGroup g) {
  __shared__ float red[16][65];
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  const float* partial = g.partial[pi];
  const int nrows = g.nrows[pi], D = g.D[pi];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = (blockIdx.x - g.first_block[pi]) * 64 + tx;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < 2 * D) {
    int r = ty;
    for (; r + 48 < nrows; r += 64) {
      s0 += partial[(size_t)r * 2 * D + c];
      s1 += partial[(size_t)(r + 16) * 2 * D + c];
      s2 += partial[(size_t)(r + 32) * 2 * D + c];
      s3 += partial[(size_t)(r + 48) * 2 * D + c];
    }
    for (; r < nrows; r += 16) s0 += partial[(size_t)r * 2 * D + c];
  }
  red[ty][tx] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ty == 0 && c < 2 * D) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    if (c < D) g.dgamma[pi][c] += s; else g.dbeta[pi][c - D] += s;
  }
}

int g_ln_bwd_waves = 8, g_ln_bwd_cap = 512;     // best of tools/ln_bench.py on MI355X; knobs: dav_tune 1 / 2

int ln_bwd_grid(int rows) {
  int g = (rows + g_ln_bwd_waves - 1) / g_ln_bwd_waves;
  return g > g_ln_bwd_cap ? g_ln_bwd_cap : (g < 1 ? 1 : g);
}

int ln_grid(int rows) {
  int g = (rows + 3) / 4;
  return g > 1024 ? 1024 : (g < 1 ? 1 : g);
}

template <int MAXC>
void ln_fwd_issue(const void* const* params, int n, hipStream_t stream) {
  for (int base = 0; base < n; base += LN_BATCH_MAX) {
    const int cnt = n - base < LN_BATCH_MAX ? n - base : LN_BATCH_MAX;
    if (cnt == 1) {
      const LNFwd& p = *(const LNFwd*)params[base];
      DAV_LAUNCH_NOW(ln_fwd_kernel<MAXC>, dim3(ln_grid(p.B * (p.s0.rows + p.s1.rows))), dim3(256), 0, stream, p);
      continue;
    }
    LNFwdGroup g;
    int first = 0;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const LNFwd*)params[base + i];
      g.first_block[i] = first;
      first += ln_grid(g.prob[i].B * (g.prob[i].s0.rows + g.prob[i].s1.rows));
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    DAV_LAUNCH_NOW(ln_fwd_grouped_kernel<MAXC>, dim3(first), dim3(256), 0, stream, g);
  }
}

// the per-problem grid (= number of partial rows the dgamma/dbeta reduction reads) is the one the single launch uses
template <int MAXC>
void ln_bwd_issue(const void* const* params, int n, hipStream_t stream) {
  for (int base = 0; base < n; base += LN_BATCH_MAX) {
    const int cnt = n - base < LN_BATCH_MAX ? n - base : LN_BATCH_MAX;
    size_t lds = 0;
    LNBwdGroup g;
    int first = 0;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const LNBwd*)params[base + i];
      const size_t l = (size_t)g_ln_bwd_waves * 2 * g.prob[i].D * sizeof(float);
      lds = l > lds ? l : lds;
      g.first_block[i] = first;
      first += ln_bwd_grid(g.prob[i].B * (g.prob[i].s0.rows + g.prob[i].s1.rows));
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    if (cnt == 1) DAV_LAUNCH_NOW(ln_bwd_kernel<MAXC>, dim3(first), dim3(64 * g_ln_bwd_waves), lds, stream, g.prob[0]);
    else DAV_LAUNCH_NOW(ln_bwd_grouped_kernel<MAXC>, dim3(first), dim3(64 * g_ln_bwd_waves), lds, stream, g);
  }
}

template <typename P, davb::GroupFn F>
void ln_dispatch(const P& p, hipStream_t stream) {
  if (davb::recording()) {
    davb::push_typed(F, &p, sizeof(p), stream);
    return;
  }
  const void* one = &p;
  F(&one, 1, stream);
}

}  // namespace

extern "C" int dav_layernorm_fwd(const float* x0, long x0_bs, int r0, const float* x1, long x1_bs, int r1, int B, int D,
                                 const float* gamma, const float* beta, float eps, void* y_bf16, float* y_f32,
                                 float* mean, float* rstd, hipStream_t stream) {
  if (B <= 0 || D <= 0 || (D & 3) || D > MAXC_LIMIT * 256 || r0 < 0 || r1 < 0 || r0 + r1 <= 0) return DAV_ERR_SHAPE;
  LNFwd p;
  p.s0 = LNSeg{x0, x0_bs, r0}; p.s1 = LNSeg{x1, x1_bs, r1}; p.B = B; p.D = D; p.gamma = gamma; p.beta = beta;
  p.eps = eps; p.y = (bf16_t*)y_bf16; p.y32 = y_f32; p.mean = mean; p.rstd = rstd;
  const int nch = (D / 4 + 63) / 64;
  if (nch <= 1) ln_dispatch<LNFwd, ln_fwd_issue<1>>(p, stream);
  else if (nch == 2) ln_dispatch<LNFwd, ln_fwd_issue<2>>(p, stream);
  else if (nch == 3) ln_dispatch<LNFwd, ln_fwd_issue<3>>(p, stream);
  else if (nch == 4) ln_dispatch<LNFwd, ln_fwd_issue<4>>(p, stream);
  else ln_dispatch<LNFwd, ln_fwd_issue<8>>(p, stream);
  return dav_launch_status();
}

extern "C" int dav_layernorm_bwd_reduce_grouped(const DavLnReduce* items, int count, hipStream_t stream) {
  if (count <= 0 || count > LN_GROUP_MAX) return DAV_ERR_SHAPE;
  static thread_local LNReduceGroup g;   // host staging, one per calling thread (the library is re-entrant across threads)
  int first = 0;
  for (int i = 0; i < count; ++i) {
    if (items[i].D <= 0 || items[i].rows == 0) return DAV_ERR_SHAPE;
    g.partial[i] = (const float*)items[i].workspace; g.dgamma[i] = items[i].dgamma; g.dbeta[i] = items[i].dbeta;
    g.nrows[i] = items[i].rows < 0 ? -items[i].rows : ln_bwd_grid(items[i].rows); g.D[i] = items[i].D;     // rows < 0: explicit partial-row count
    g.first_block[i] = first;
    first += (2 * items[i].D + 63) / 64;
  }
  g.first_block[count] = first;
  g.count = count;
  const LNReduceGroup gl = g;      // automatic copy: a recorded launch captures by [=], which does not copy static storage (see gemm.hip)
  DAV_LAUNCH(ln_bwd_reduce_grouped_kernel, dim3(first), dim3(1024), 0, stream, gl);
  return dav_launch_status();
}

extern int dav_attn_qt;      // attention.hip

extern "C" int dav_tune(int knob, int value) {
  if (knob == 3 && value >= 0 && value <= 2) { dav_attn_qt = value; return DAV_OK; }
  if (knob == 1 && (value == 2 || value == 4 || value == 8)) { g_ln_bwd_waves = value; return DAV_OK; }
  if (knob == 2 && value >= 1 && value <= 4096) { g_ln_bwd_cap = value; return DAV_OK; }
  return DAV_ERR_SHAPE;
}

extern "C" size_t dav_layernorm_bwd_workspace_bytes(int rows, int D) {
  return (size_t)ln_bwd_grid(rows) * 2 * D * sizeof(float);
}

extern "C" int dav_layernorm_bwd(const float* x0, long x0_bs, int r0, const float* x1, long x1_bs, int r1, int B, int D,
                                 const void* dy_bf16, const float* dy_f32, const float* gamma, const float* mean,
                                 const float* rstd,
                                 float* dx0, long dx0_bs, int acc0, const float* res0, long res0_bs, void* dx0_bf16, long dx0_bf_bs,
                                 float* dx1, long dx1_bs, int acc1, const float* res1, long res1_bs, void* dx1_bf16, long dx1_bf_bs,
                                 float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (B <= 0 || D <= 0 || (D & 3) || D > MAXC_LIMIT * 256 || r0 < 0 || r1 < 0 || r0 + r1 <= 0) return DAV_ERR_SHAPE;
  if (!dy_bf16 && !dy_f32) return DAV_ERR_SHAPE;
  if (workspace && workspace_bytes < dav_layernorm_bwd_workspace_bytes(B * (r0 + r1), D)) return DAV_ERR_WORKSPACE;
  if (dgamma && !workspace) return DAV_ERR_WORKSPACE;
  LNBwd p;
  p.s0 = LNSeg{x0, x0_bs, r0}; p.s1 = LNSeg{x1, x1_bs, r1}; p.B = B; p.D = D;
  p.d0 = LNDst{dx0, dx0_bs, r0, acc0, res0, res0_bs, (bf16_t*)dx0_bf16, dx0_bf_bs};
  p.d1 = LNDst{dx1, dx1_bs, r1, acc1, res1, res1_bs, (bf16_t*)dx1_bf16, dx1_bf_bs};
  p.dy = (const bf16_t*)dy_bf16; p.dy32 = dy_f32; p.gamma = gamma; p.mean = mean; p.rstd = rstd;
  p.dgamma = dgamma; p.dbeta = dbeta;
  p.partial = (float*)workspace;
  p.xb0 = p.xb1 = nullptr; p.st0 = p.st1 = nullptr; p.beta = nullptr; p.h_out = nullptr; p.eps = 0.f;
  const int grid = ln_bwd_grid(B * (r0 + r1));
  const int nch = (D / 4 + 63) / 64;
  if (nch <= 1) ln_dispatch<LNBwd, ln_bwd_issue<1>>(p, stream);
  else if (nch == 2) ln_dispatch<LNBwd, ln_bwd_issue<2>>(p, stream);
  else if (nch == 3) ln_dispatch<LNBwd, ln_bwd_issue<3>>(p, stream);
  else if (nch == 4) ln_dispatch<LNBwd, ln_bwd_issue<4>>(p, stream);
  else ln_dispatch<LNBwd, ln_bwd_issue<8>>(p, stream);
  if (dgamma) DAV_LAUNCH(ln_bwd_reduce_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, stream, (const float*)workspace, grid, D, dgamma, dbeta);
  return dav_launch_status();
}

extern "C" int dav_layernorm_bwd_twin(const void* xb0, long xb0_bs, const float* st0, int r0, const void* xb1, long xb1_bs, const float* st1,
                                      int r1, int B, int D, float eps, const void* dy_bf16, const float* dy_f32, const float* gamma,
                                      const float* beta,
                                      float* dx0, long dx0_bs, int acc0, const float* res0, long res0_bs, void* dx0_bf16, long dx0_bf_bs,
                                      float* dx1, long dx1_bs, int acc1, const float* res1, long res1_bs, void* dx1_bf16, long dx1_bf_bs,
                                      void* h_out_bf16, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                      hipStream_t stream) {
  if (B <= 0 || D <= 0 || (D & 63) || D > 1024 || r0 <= 0 || r1 < 0 || !(eps > 0.f)) return DAV_ERR_SHAPE;
  if (!xb0 || !st0 || (r1 > 0 && (!xb1 || !st1)) || !gamma) return DAV_ERR_SHAPE;
  if (!dy_bf16 && !dy_f32) return DAV_ERR_SHAPE;
  if (h_out_bf16 && !beta) return DAV_ERR_SHAPE;
  if ((xb0_bs % D) || (r1 > 0 && (xb1_bs % D))) return DAV_ERR_SHAPE;          // the statistics rows follow the twin's row numbering
  if (((uintptr_t)xb0 | (uintptr_t)xb1 | (uintptr_t)st0 | (uintptr_t)st1 | (uintptr_t)h_out_bf16) & 7) return DAV_ERR_ALIGN;
  if (workspace && workspace_bytes < dav_layernorm_bwd_workspace_bytes(B * (r0 + r1), D)) return DAV_ERR_WORKSPACE;
  if (dgamma && !workspace) return DAV_ERR_WORKSPACE;
  LNBwd p;
  p.s0 = LNSeg{nullptr, xb0_bs, r0}; p.s1 = LNSeg{nullptr, xb1_bs, r1}; p.B = B; p.D = D;
  p.d0 = LNDst{dx0, dx0_bs, r0, acc0, res0, res0_bs, (bf16_t*)dx0_bf16, dx0_bf_bs};
  p.d1 = LNDst{dx1, dx1_bs, r1, acc1, res1, res1_bs, (bf16_t*)dx1_bf16, dx1_bf_bs};
  p.dy = (const bf16_t*)dy_bf16; p.dy32 = dy_f32; p.gamma = gamma; p.mean = nullptr; p.rstd = nullptr;
  p.dgamma = dgamma; p.dbeta = dbeta;
  p.partial = (float*)workspace;
  p.xb0 = (const bf16_t*)xb0; p.xb1 = (const bf16_t*)xb1; p.st0 = st0; p.st1 = st1; p.beta = beta; p.h_out = (bf16_t*)h_out_bf16; p.eps = eps;
  const int grid = ln_bwd_grid(B * (r0 + r1));
  const int nch = (D / 4 + 63) / 64;
  if (nch <= 1) ln_dispatch<LNBwd, ln_bwd_issue<1>>(p, stream);
  else if (nch == 2) ln_dispatch<LNBwd, ln_bwd_issue<2>>(p, stream);
  else if (nch == 3) ln_dispatch<LNBwd, ln_bwd_issue<3>>(p, stream);
  else ln_dispatch<LNBwd, ln_bwd_issue<4>>(p, stream);
  if (dgamma) DAV_LAUNCH(ln_bwd_reduce_kernel, dim3((2 * D + 63) / 64), dim3(1024), 0, stream, (const float*)workspace, grid, D, dgamma, dbeta);
  return dav_launch_status();
}

// ---- LayerNorm folded into its neighbour GEMMs: weight fold and stand-alone row statistics ------------------------------------
namespace {

constexpr int FOLD_MAX = 48;
struct FoldItem { const float* w; const float *gamma, *beta, *bias; bf16_t* wl; float *c, *d; int N, K; };
struct FoldGroup { FoldItem it[FOLD_MAX]; int first_row[FOLD_MAX + 1]; int count; };
static_assert(sizeof(FoldGroup) <= 4096, "kernel argument block");

// one wave per output row n (w = the fp32 master: ONE rounding, like the plain bf16 mirror): w_ln[n, :] = bf16(gamma * w[n, :]), c[n] = sum of the ROUNDED products (what the GEMM really contracts
// with, so that acc - mean * c is an exact centring), d[n] = bias[n] + sum beta * w[n, :]
__global__ __launch_bounds__(256) void ln_fold_kernel(const FoldGroup g) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= g.first_row[g.count]) return;
  int pi = 0;
  while (pi + 1 < g.count && row >= g.first_row[pi + 1]) ++pi;
  const FoldItem it = g.it[pi];
  const int n = row - g.first_row[pi];
  const float* w = it.w + (long)n * it.K;
  bf16_t* wl = it.wl + (long)n * it.K;
  float c = 0.f, d = 0.f;
  for (int k = lane * 8; k < it.K; k += 512) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + k), w1 = *reinterpret_cast<const float4*>(w + k + 4);
    const float4 g0 = *reinterpret_cast<const float4*>(it.gamma + k), g1 = *reinterpret_cast<const float4*>(it.gamma + k + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(it.beta + k), b1 = *reinterpret_cast<const float4*>(it.beta + k + 4);
    const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    uint4 o;
    o.x = pack2bf(wv[0] * gv[0], wv[1] * gv[1]); o.y = pack2bf(wv[2] * gv[2], wv[3] * gv[3]);
    o.z = pack2bf(wv[4] * gv[4], wv[5] * gv[5]); o.w = pack2bf(wv[6] * gv[6], wv[7] * gv[7]);
    *reinterpret_cast<uint4*>(wl + k) = o;
    c += (__uint_as_float(o.x << 16) + __uint_as_float(o.x & 0xffff0000u)) + (__uint_as_float(o.y << 16) + __uint_as_float(o.y & 0xffff0000u)) +
         (__uint_as_float(o.z << 16) + __uint_as_float(o.z & 0xffff0000u)) + (__uint_as_float(o.w << 16) + __uint_as_float(o.w & 0xffff0000u));
#pragma unroll
    for (int e = 0; e < 8; ++e) d += wv[e] * bv[e];
  }
  c = wave_sum(c);
  d = wave_sum(d);
  if (lane == 0) {
    it.c[n] = c;
    it.d[n] = d + (it.bias ? it.bias[n] : 0.f);
  }
}

struct RowStat { const float* x; long bs; int B, rows, D; bf16_t* tw; float* st; };

// one wave per row: lane l holds columns [4 l + 256 i, 4 l + 256 i + 4); a 64-column slot = 16 consecutive lanes of one chunk i
template <int MAXC>
__global__ __launch_bounds__(256) void rowstats_cast_kernel(const RowStat p) {
  const int lane = threadIdx.x & 63;
  const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  const int total = p.B * p.rows, nch = p.D >> 2, ns = p.D >> 6;
  for (int row = gw; row < total; row += nwaves) {
    const int b = row / p.rows, r = row % p.rows;
    const float* x = p.x + b * p.bs + (long)r * p.D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      float s1 = 0.f, s2 = 0.f;
      if (c < nch) {
        const float4 v = reinterpret_cast<const float4*>(x)[c];
        uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
        reinterpret_cast<uint2*>(p.tw + (long)row * p.D)[c] = w;
        s1 = (v.x + v.y) + (v.z + v.w);
        s2 = (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      const int slot = 4 * i + (lane >> 4);
      if ((lane & 15) == 0 && slot < ns) reinterpret_cast<float2*>(p.st)[(long)row * ns + slot] = float2{s1, s2};
    }
  }
}

}  // namespace

extern "C" int dav_ln_fold_grouped(const DavLnFold* items, int count, hipStream_t stream) {
  if (count <= 0 || count > FOLD_MAX || !items) return DAV_ERR_SHAPE;
  FoldGroup g;
  int first = 0;
  for (int i = 0; i < count; ++i) {
    const DavLnFold& q = items[i];
    if (q.N <= 0 || q.K <= 0 || (q.K & 7) || !q.w || !q.gamma || !q.beta || !q.w_ln_bf16 || !q.ln_c || !q.ln_d) return DAV_ERR_SHAPE;
    if (((uintptr_t)q.w | (uintptr_t)q.w_ln_bf16 | (uintptr_t)q.gamma | (uintptr_t)q.beta) & 15) return DAV_ERR_ALIGN;
    g.it[i] = FoldItem{q.w, q.gamma, q.beta, q.bias, (bf16_t*)q.w_ln_bf16, q.ln_c, q.ln_d, q.N, q.K};
    g.first_row[i] = first;
    first += q.N;
  }
  for (int i = count; i <= FOLD_MAX; ++i) g.first_row[i] = first;
  g.count = count;
  const FoldGroup gl = g;
  DAV_LAUNCH(ln_fold_kernel, dim3((first + 3) / 4), dim3(256), 0, stream, gl);
  return dav_launch_status();
}

extern "C" int dav_rowstats_cast(const float* x, long x_bs, int B, int rows, int D, void* twin_bf16, float* stats, hipStream_t stream) {
  if (B <= 0 || rows <= 0 || D <= 0 || (D & 63) || D > 1024 || !x || !twin_bf16 || !stats) return DAV_ERR_SHAPE;
  if (((uintptr_t)x & 15) || ((uintptr_t)twin_bf16 & 7) || ((uintptr_t)stats & 7) || (x_bs & 3)) return DAV_ERR_ALIGN;
  const RowStat p{x, x_bs, B, rows, D, (bf16_t*)twin_bf16, stats};
  const int grid = ln_grid(B * rows);
  const int nch = (D / 4 + 63) / 64;
  if (nch <= 1) DAV_LAUNCH(rowstats_cast_kernel<1>, dim3(grid), dim3(256), 0, stream, p);
  else if (nch == 2) DAV_LAUNCH(rowstats_cast_kernel<2>, dim3(grid), dim3(256), 0, stream, p);
  else if (nch == 3) DAV_LAUNCH(rowstats_cast_kernel<3>, dim3(grid), dim3(256), 0, stream, p);
  else DAV_LAUNCH(rowstats_cast_kernel<4>, dim3(grid), dim3(256), 0, stream, p);
  return dav_launch_status();
}
