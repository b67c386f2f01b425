// bf16 MFMA GEMMs for gfx950 (MI355X): fp32 accumulate, fused epilogues.
//
//   gemm_nt : C[M,N] = epi( A[M,K] . B[N,K]^T )        forward linears and dgrad (with a W^T copy)
//   gemm_tn : C[N,K] (+)= A[Mc,N]^T . B[Mc,K]          weight gradients (contraction over token rows)
//
// Both tile 128x128 (or 64-wide variants) per 256-thread workgroup = 4 waves in 2x2, each wave
// computing 16x16x32 bf16 MFMA fragments out of XOR-swizzled LDS tiles (conflict-free ds_read_b128 /
// ds_read_b64_tr_b16).  K step is 64.  Row maps let A / C / the residual address sub-ranges of
// [B, rows, D] activations directly (no cat/split/gather copies).
//
// Replaces the cuBLAS/rocBLAS calls behind F.linear in timm Block / Mlp / Attention and
// models/fusion_blocks.py:41-44,227-232; models/avmae.py:31,59-60,88.
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "dav_kernels.h"

namespace {

struct RowMap {
  int rpb, bs, off;   // rpb == 0: identity; else row(m) = (m / rpb) * bs + off + m % rpb
};
__device__ __forceinline__ long map_row(int m, const RowMap& r) {
  return r.rpb > 0 ? (long)(m / r.rpb) * r.bs + r.off + (m % r.rpb) : (long)m;
}

struct NTParams {
  const bf16_t* A; const bf16_t* B;
  int M, N, K, lda, ldb;
  RowMap amap;
  const float* bias;
  int act;                       // 0 none, 1 gelu (pre-activation optionally to C2), 2 multiply by gelu'(aux)
  const bf16_t* aux; int ldaux;  // act == 2
  const float* res; int ldres; RowMap rmap; const int* res_rows;   // fp32 residual (optional row gather list)
  void* C; int ldc; int c_bf16; RowMap cmap;
  int b_kn;                      // B is given as [K, N] row-major (ldb = its row stride) instead of [N, K]
  bf16_t* C2; int ldc2; int c2_mode;   // second bf16 output [M, ldc2]: 1 pre-activation, 2 post-activation/pre-residual, 3 final value
  int beta;                      // C (fp32) += result
  float alpha;
  int debug;                     // timing experiments only (DAV_NT_DEBUG): 1 = epilogue without global stores, 2 = no epilogue, 4 = no fragment reads / MFMAs
  int force_cfg;                 // host side only: tile configuration asked for explicitly (0 = chosen per group at issue time)
  // ---- LayerNorm folded into the GEMM (dav_gemm_nt_ln_bf16, DavNtLn) ----
  // consumer: A holds the RAW rows (bf16 twin of the fp32 residual stream), B = gamma-scaled weights; the epilogue turns the
  // accumulator into rstd[m] * (acc - mean[m] * ln_c[n]) + bias[n].  Row statistics come as per-row partial sums over 64-column
  // slots, ln_st[row * (K / 64) + s] = {sum x, sum x^2}; a_r0 > 0: the rows of a batch element are a_r0 rows of (A, ln_st)
  // followed by a_r1 rows of (A2, ln_st2), both dense.
  const float* ln_st; const float* ln_st2; const bf16_t* A2; const float* ln_c; float ln_eps; int a_r0, a_r1;
  // producer: per-row partial sums of the final fp32 value over 64-column slots (row index = the C row) and its bf16 twin
  float* st_out; bf16_t* tw_out; int ldtw;
};
// row of the A operand / of its statistics partials (NS = K / 64 float2 per row)
__device__ __forceinline__ const bf16_t* nt_a_row(const NTParams& p, int gm) {
  if (p.a_r0 > 0) {
    const int R = p.a_r0 + p.a_r1, b = gm / R, i = gm - b * R;
    return i < p.a_r0 ? p.A + ((long)b * p.a_r0 + i) * p.lda : p.A2 + ((long)b * p.a_r1 + (i - p.a_r0)) * p.lda;
  }
  return p.A + map_row(gm, p.amap) * p.lda;
}
__device__ __forceinline__ const float2* nt_stat_row(const NTParams& p, int gm, int ns) {
  if (p.a_r0 > 0) {
    const int R = p.a_r0 + p.a_r1, b = gm / R, i = gm - b * R;
    return reinterpret_cast<const float2*>(i < p.a_r0 ? p.ln_st + ((long)b * p.a_r0 + i) * ns * 2 : p.ln_st2 + ((long)b * p.a_r1 + (i - p.a_r0)) * ns * 2);
  }
  return reinterpret_cast<const float2*>(p.ln_st + map_row(gm, p.amap) * ns * 2);
}

struct TNParams {                     // (pointers first: 96 bytes, 40 of them fit the grouped kernels' by-value table)
  const bf16_t* A; const bf16_t* B;   // A[Mc, N] (lda), B[Mc, K] (ldb)
  float* C;                           // fp32 [N, K] (ldc)
  float* bias_grad;                   // optional: column sums of A accumulated into [N]
  int Mc, N, K, lda, ldb, ldc;
  RowMap amap, bmap;
  int beta;                           // 0: overwrite (only legal with splits == 1), 1: accumulate
  int splits;                         // split of the contraction over blockIdx.y (atomic accumulate)
  int debug_plain_store;              // timing experiments only: 1 (variant bit 8) plain stores instead of atomics; DAV_TN_DEBUG: 2 no epilogue, 4 no reads / MFMAs
};
static_assert(sizeof(TNParams) == 96, "TNParams layout");

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(NTParams p) {
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                 // [2][BM][128B]
  char* Bs = smem + 2 * A_BYTES;   // [2][BN][128B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tiles_m = (p.M + BM - 1) / BM;
  // XCD-aware remap: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a
  // contiguous chunk of the tile space (neighbouring tiles share A rows / B rows in that XCD's L2).
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = bid / tiles_n, bn = bid % tiles_n;
  const int m0 = bm * BM, n0 = bn * BN;

  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
  int a_k[A_CH], b_k[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ls = (c & 7) ^ ((row >> 1) & 7);
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = p.A + map_row(gm, p.amap) * p.lda + ls * 8;
    a_k[i] = ls * 8;
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ls = (c & 7) ^ ((row >> 1) & 7);
    int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
    b_src[i] = p.B + (long)gn * p.ldb + ls * 8;
    b_k[i] = ls * 8;
  }

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + 63) >> 6;
  uint4 ra[A_CH], rb[B_CH];

  auto load_regs = [&](int kt) {
    const int k0 = kt << 6;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      ra[i] = (k0 + a_k[i] < p.K) ? *reinterpret_cast<const uint4*>(a_src[i] + k0) : uint4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      rb[i] = (k0 + b_k[i] < p.K) ? *reinterpret_cast<const uint4*>(b_src[i] + k0) : uint4{0, 0, 0, 0};
  };
  auto store_regs = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) *reinterpret_cast<uint4*>(As + buf * A_BYTES + (tid + 256 * i) * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) *reinterpret_cast<uint4*>(Bs + buf * B_BYTES + (tid + 256 * i) * 16) = rb[i];
  };
  auto dma_tile = [&](int kt, int buf) {   // global -> LDS DMA, 16 B per lane, LDS image lane-linear
    const int k0 = kt << 6;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, a_src[i] + k0),
                                       LDS_PTR(void, As + buf * A_BYTES + (wave * 64 + 256 * i) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, b_src[i] + k0),
                                       LDS_PTR(void, Bs + buf * B_BYTES + (wave * 64 + 256 * i) * 16), 16, 0, 0);
  };

  if (GLDS) {
    dma_tile(0, 0);
  } else {
    load_regs(0);
    store_regs(0);
  }
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      if (GLDS) dma_tile(kt + 1, buf ^ 1); else load_regs(kt + 1);
    }
    const char* Ab = As + buf * A_BYTES;
    const char* Bb = Bs + buf * B_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WM + i * 16 + fr;
        const int ps = (kk * 4 + fg) ^ ((row >> 1) & 7);
        af[i] = *reinterpret_cast<const bf16x8*>(Ab + row * 128 + ps * 16);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int row = wn * WN + j * 16 + fr;
        const int ps = (kk * 4 + fg) ^ ((row >> 1) & 7);
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + row * 128 + ps * 16);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more && !GLDS) store_regs(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * WM + i * 16 + fg * 4 + r;
      if (m >= p.M) continue;
      const long crow = map_row(m, p.cmap);
      long rrow = 0;
      if (p.res) rrow = p.res_rows ? (long)p.res_rows[m] : map_row(m, p.rmap);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WN + j * 16 + fr;
        if (n >= p.N) continue;
        float v = acc[i][j][r] * p.alpha;
        if (p.bias) v += p.bias[n];
        if (p.c2_mode == 1) p.C2[(long)m * p.ldc2 + n] = f2bf(v);
        if (p.act == 1) {
          float u, d;
          gelu_pair_f(v, u, d);
          if (p.c2_mode == 4) p.C2[(long)m * p.ldc2 + n] = f2bf(d);
          v = u;
        } else if (p.act == 2) {
          v *= gelu_grad_f(bf2f(p.aux[(long)m * p.ldaux + n]));
        } else if (p.act == 3) {
          v *= bf2f(p.aux[(long)m * p.ldaux + n]);
        }
        if (p.c2_mode == 2) p.C2[(long)m * p.ldc2 + n] = f2bf(v);
        if (p.res) v += p.res[rrow * p.ldres + n];
        if (p.C) {
          if (p.c_bf16) {
            reinterpret_cast<bf16_t*>(p.C)[crow * p.ldc + n] = f2bf(v);
          } else {
            float* c = reinterpret_cast<float*>(p.C) + crow * p.ldc + n;
            if (p.beta) v += *c;
            *c = v;
          }
        }
        if (p.c2_mode == 3) p.C2[(long)m * p.ldc2 + n] = f2bf(v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// NT kernel, second generation: STAGES-deep LDS ring filled by global->LDS DMA that stays in flight
// across k-steps (counted s_waitcnt vmcnt + raw s_barrier, one barrier per k-step), WM_ x WN_ waves,
// and an epilogue staged through LDS so that every global access of C / residual / C2 is a full
// 16-byte-per-lane row segment.  Requires K % 64 == 0 and N % 4 == 0.
// ------------------------------------------------------------------------------------------------
template <int RB> __device__ __forceinline__ int tn2_swz(int row) {
  return RB == 256 ? ((row & 3) | (((row >> 3) & 1) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 1) << 1));
}
template <int RB>
__device__ __forceinline__ bf16x8 tn2_frag(const char* tile, int mk, int colbase, int lane) {
  const int g = lane >> 4, li = lane & 15;
  union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = mk + 8 * g + 4 * h + (li >> 2);
    const int colb = (colbase + 4 * (li & 3)) * 2;
    const int addr = row * RB + ((((colb >> 5) ^ tn2_swz<RB>(row)) << 5) | (colb & 31));
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + addr));
  }
  return u.v;
}

// The same fragment through inline asm.  hipcc puts an s_waitcnt vmcnt(0) in front of every __builtin_amdgcn_ds_read_tr16_b64 that
// is issued while an LDS-DMA is in flight (the builtin carries no memory operand that its wait-count pass could tell apart from the
// DMA's LDS write): in the two-stage rings below that wait sat right behind the issue of the NEXT stage's DMA, so a workgroup never
// computed under its own loads (tools/isa_events.py shows it: "D4 [v0] t8 ...").  An asm read is invisible to that pass — and to its
// lgkmcnt bookkeeping: the callers wait with wait_lgkmcnt<N>() + sched_barrier before the first use ("memory" keeps the compiler's
// own LDS loads on their side of the statement, so the counted waits see the issue order written in the source).
//   tn2_frag_base: per-lane LDS byte offset of the fragment at contraction row 0 (the swizzle does not depend on mk % 32 == 0 / h)
//   tn2_tr<OFF, RB>(addr): the two reads at addr + OFF and addr + OFF + 4 * RB
template <int RB> __device__ __forceinline__ uint32_t tn2_frag_base(int colbase, int lane) {
  const int g = lane >> 4, li = lane & 15;
  const int row = 8 * g + (li >> 2);
  const int colb = (colbase + 4 * (li & 3)) * 2;
  return (uint32_t)(row * RB + ((((colb >> 5) ^ tn2_swz<RB>(row)) << 5) | (colb & 31)));
}
template <int OFF, int RB> __device__ __forceinline__ bf16x8 tn2_tr(uint32_t addr) {
  static_assert(OFF >= 0 && OFF + 4 * RB < 65536, "ds offset field");
  union { s16x4 h[2]; bf16x8 v; } u;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(u.h[0]), "=&v"(u.h[1]) : "v"(addr), "i"(OFF), "i"(OFF + 4 * RB) : "memory");
  return u.v;
}
template <int V> struct IntC { static constexpr int value = V; };

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_lgkmcnt() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");
}

// Epilogue shared by the NT kernels: 16 tile rows of a wave at a time: registers -> (alpha, bias) -> the wave's small LDS
// patch -> coalesced row segments (16 bytes per lane).  The patch (16 x (WTN + 4) floats per wave) is smaller than one
// ring stage, so the ring alone decides how many workgroups share a CU.
template <int FM, int FN, int WTM, int WTN>
__device__ __forceinline__ void nt_epilogue(const NTParams& p, f32x4 (&acc)[FM][FN], char* patch_base, int m0, int n0, int wave,
                                            int wm, int wn, int lane) {
  if (p.debug & 2) return;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int LDW = WTN + 4;              // floats per patch row (+4: at most 2-way ds_write conflicts)
  float* patch = reinterpret_cast<float*>(patch_base) + wave * (16 * LDW);
  constexpr int CPR = WTN / 4;              // float4 chunks per patch row
  constexpr int RPI = 64 / CPR;             // rows per wave-instruction
  const int cc = lane % CPR, rr = lane / CPR;
  const int n = n0 + wn * WTN + cc * 4;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nn = n0 + wn * WTN + j * 16 + fr;
      const float bv = (p.bias && nn < p.N) ? p.bias[nn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(fg * 4 + r) * LDW + j * 16 + fr] = acc[i][j][r] * p.alpha + bv;
    }
    if (n < p.N && rr < RPI) {
#pragma unroll
      for (int it = 0; it < (16 + RPI - 1) / RPI; ++it) {
        const int lr = it * RPI + rr;
        const int m = m0 + wm * WTM + i * 16 + lr;
        if (lr >= 16 || m >= p.M) continue;
        float4 v = *reinterpret_cast<const float4*>(patch + lr * LDW + cc * 4);
        if (p.debug & 1) { if (v.x == 123.456f) reinterpret_cast<float*>(p.C)[0] = v.y; continue; }
        if (p.c2_mode == 1) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
        if (p.act == 1) {
          float4 d;
          gelu_pair_f(v.x, v.x, d.x); gelu_pair_f(v.y, v.y, d.y); gelu_pair_f(v.z, v.z, d.z); gelu_pair_f(v.w, v.w, d.w);
          if (p.c2_mode == 4) {          // bf16 twin = GELU'(pre-activation): the fc2 input gradient only multiplies by it
            uint2 w; w.x = pack2bf(d.x, d.y); w.y = pack2bf(d.z, d.w);
            *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
          }
        } else if (p.act == 2) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (long)m * p.ldaux + n);
          v.x *= gelu_grad_f(__uint_as_float(a.x << 16)); v.y *= gelu_grad_f(__uint_as_float(a.x & 0xffff0000u));
          v.z *= gelu_grad_f(__uint_as_float(a.y << 16)); v.w *= gelu_grad_f(__uint_as_float(a.y & 0xffff0000u));
        } else if (p.act == 3) {
          const uint2 a = *reinterpret_cast<const uint2*>(p.aux + (long)m * p.ldaux + n);
          v.x *= __uint_as_float(a.x << 16); v.y *= __uint_as_float(a.x & 0xffff0000u);
          v.z *= __uint_as_float(a.y << 16); v.w *= __uint_as_float(a.y & 0xffff0000u);
        }
        if (p.c2_mode == 2) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
        if (p.res) {
          const long rrow = p.res_rows ? (long)p.res_rows[m] : map_row(m, p.rmap);
          const float4 t = *reinterpret_cast<const float4*>(p.res + rrow * p.ldres + n);
          v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (p.C) {
          const long crow = map_row(m, p.cmap);
          if (p.c_bf16) {
            uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + n) = w;
          } else {
            float4* c = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + crow * p.ldc + n);
            if (p.beta) { const float4 o = *c; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *c = v;
          }
        }
        if (p.c2_mode == 3) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
      }
    }
  }
}

// Epilogue for TRANSPOSED accumulators.  The k-loops of the second / third generation kernels issue their MFMAs with the
// operands swapped (B fragment first): the 16 x 16 result block then comes out transposed, i.e. lane (fr = lane & 15,
// fg = lane >> 4) holds C[m = fr][n = 4 fg .. 4 fg + 3] — FOUR CONSECUTIVE COLUMNS OF ONE ROW — instead of four rows of one
// column.  Every epilogue access (bias, residual, aux, C, C2) is then a 16-byte (fp32) / 8-byte (bf16) per-lane piece of a
// row straight from registers: no LDS round trip, no waits.  Measured on the LDS-staged epilogue above (DAV_NT_DEBUG):
// 31 % of the 11264 x 2304 x 768 GEMM and 45 % of the K = 512 decoder GEMMs were epilogue, more than half of it the
// registers -> LDS -> registers transposition.
// LayerNorm consumer: v = rstd[m] * (alpha acc - mean[m] * c[n]) for one fragment.  mean / rstd (ln_mr[BM]) and c (the tile's BN
// values behind them) come from LDS at the point of use; without a LayerNorm the operands are the identity (0, 1, 0) — a select on six
// scalars instead of control flow around the 64 accumulator registers (which made hipcc keep two copies of them: 230 bytes of scratch).
template <int BM>
__device__ __forceinline__ float4 nt_ln_frag(const f32x4& a, float alpha, bool lnc, const float2* ln_mr, int ml, int nl) {
  const float2 mr = lnc ? ln_mr[ml] : float2{0.f, 1.f};
  const float4 c = lnc ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ln_mr + BM) + nl) : float4{0.f, 0.f, 0.f, 0.f};
  float4 v;
  v.x = mr.y * (a[0] * alpha - mr.x * c.x); v.y = mr.y * (a[1] * alpha - mr.x * c.y);
  v.z = mr.y * (a[2] * alpha - mr.x * c.z); v.w = mr.y * (a[3] * alpha - mr.x * c.w);
  return v;
}

// LayerNorm folded in (DavNtLn): consumer — ln_mr[tile row] = {mean, rstd} (LDS), the accumulator becomes rstd * (acc - mean * c[n]);
// producer — per-row partial sums {sum v, sum v^2} of the FINAL value over 64-column slots (p.st_out, row = the C row) and its bf16
// twin (p.tw_out): what the next GEMM consumes instead of a LayerNorm output.  A wave tile narrower than a slot (WTN == 32)
// combines the waves of a slot through LDS in a fixed order (no atomics: bit-repeatable).
template <int FM, int FN, int WTM, int WTN, int BM = 0, int BN = 0, int WN_ = 1>
__device__ __forceinline__ void nt_epilogue_t(const NTParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn, int lane,
                                              const float2* ln_mr = nullptr, char* smem = nullptr, int tid = 0) {
  if (p.debug & 2) return;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr bool PROD = BM > 0 && FM * FN <= 8;      // producer side: wave tiles up to 32 x 64 (the 64 x 64 ones have no register to spare: nt_ln_producer_cfg)
  const bool lnp = PROD && p.st_out != nullptr;
  const bool lnc = BM > 0 && p.ln_st != nullptr;
  // Every global READ of the epilogue (bias, residual, aux) is requested for a batch of fragments (2 x FN, or FN for the
  // 64-wide wave tiles) before the first one is used: written fragment by fragment the compiler has to wait for each load
  // right where it stands (a store to C may alias the next residual), i.e. one memory latency per read — sixteen of them in
  // a row for an fp32 + residual tile, about as long as a K = 768 k-loop.
  float4 bv[FN];
  bool nok[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * 16 + fg * 4;
    nok[j] = n < p.N;
    bv[j] = (p.bias && nok[j]) ? *reinterpret_cast<const float4*>(p.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int IH = (FM >= 2 && FN <= 2) ? 2 : 1;          // fragment rows per batch (about 40 registers of staging)
#pragma unroll
  for (int i0 = 0; i0 < FM; i0 += IH) {
    float4 rv[IH][FN];
    uint2 av[IH][FN];
    long crow[IH], rrow[IH];
    bool mok[IH];
#pragma unroll
    for (int ii = 0; ii < IH; ++ii) {
      const int m = m0 + wm * WTM + (i0 + ii) * 16 + fr;
      mok[ii] = m < p.M;
      const int mc = mok[ii] ? m : p.M - 1;
      crow[ii] = p.C ? map_row(mc, p.cmap) : 0;
      rrow[ii] = p.res ? (p.res_rows ? (long)p.res_rows[mc] : map_row(mc, p.rmap)) : 0;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fg * 4;
        const bool ok = mok[ii] && nok[j];
        rv[ii][j] = (p.res && ok) ? *reinterpret_cast<const float4*>(p.res + rrow[ii] * p.ldres + n) : float4{0.f, 0.f, 0.f, 0.f};
        av[ii][j] = ((p.act == 2 || p.act == 3) && ok) ? *reinterpret_cast<const uint2*>(p.aux + (long)mc * p.ldaux + n) : uint2{0, 0};
      }
    }
#pragma unroll
    for (int ii = 0; ii < IH; ++ii) {
      const int i = i0 + ii;
      const int m = m0 + wm * WTM + i * 16 + fr;
      if constexpr (BM == 0) { if (!mok[ii]) continue; }
      const bool rowok = mok[ii];            // (BM > 0: no early exit from the row — the twin patch below is a whole-wave affair)
      float ps1 = 0.f, ps2 = 0.f;
      uint2 twv[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int n = n0 + wn * WTN + j * 16 + fg * 4;
        twv[j] = uint2{0, 0};
        if (!nok[j] || !rowok) continue;
        float4 v;
        if constexpr (BM > 0) {
          v = nt_ln_frag<BM>(acc[i][j], p.alpha, lnc, ln_mr, wm * WTM + i * 16 + fr, wn * WTN + j * 16 + fg * 4);
          v.x += bv[j].x; v.y += bv[j].y; v.z += bv[j].z; v.w += bv[j].w;
        } else {
          v.x = acc[i][j][0] * p.alpha + bv[j].x; v.y = acc[i][j][1] * p.alpha + bv[j].y;
          v.z = acc[i][j][2] * p.alpha + bv[j].z; v.w = acc[i][j][3] * p.alpha + bv[j].w;
        }
        if (p.debug & 1) { if (v.x == 123.456f) reinterpret_cast<float*>(p.C)[0] = v.y; continue; }
        if (p.c2_mode == 1) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
        if (p.act == 1) {
          float4 d;
          gelu_pair_f(v.x, v.x, d.x); gelu_pair_f(v.y, v.y, d.y); gelu_pair_f(v.z, v.z, d.z); gelu_pair_f(v.w, v.w, d.w);
          if (p.c2_mode == 4) {          // bf16 twin = GELU'(pre-activation): the fc2 input gradient only multiplies by it
            uint2 w; w.x = pack2bf(d.x, d.y); w.y = pack2bf(d.z, d.w);
            *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
          }
        } else if (p.act == 2) {
          const uint2 a = av[ii][j];
          v.x *= gelu_grad_f(__uint_as_float(a.x << 16)); v.y *= gelu_grad_f(__uint_as_float(a.x & 0xffff0000u));
          v.z *= gelu_grad_f(__uint_as_float(a.y << 16)); v.w *= gelu_grad_f(__uint_as_float(a.y & 0xffff0000u));
        } else if (p.act == 3) {
          const uint2 a = av[ii][j];
          v.x *= __uint_as_float(a.x << 16); v.y *= __uint_as_float(a.x & 0xffff0000u);
          v.z *= __uint_as_float(a.y << 16); v.w *= __uint_as_float(a.y & 0xffff0000u);
        }
        if (p.c2_mode == 2) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
        if (p.res) { const float4 t = rv[ii][j]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        if (p.C) {
          if (p.c_bf16) {
            uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + crow[ii] * p.ldc + n) = w;
          } else {
            float4* c = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + crow[ii] * p.ldc + n);
            if (p.beta) { const float4 o = *c; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }      // (accumulating GEMMs are the fusion block's small ones)
            *c = v;
          }
        }
        if (p.c2_mode == 3) {
          uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
          *reinterpret_cast<uint2*>(p.C2 + (long)m * p.ldc2 + n) = w;
        }
        if constexpr (PROD) {
          if (p.tw_out) {
            uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
            if constexpr (FN == 4) twv[j] = w;          // written below as whole 128-byte row segments
            else *reinterpret_cast<uint2*>(p.tw_out + crow[ii] * p.ldtw + n) = w;
          }
        }
        if (lnp) {
          ps1 += (v.x + v.y) + (v.z + v.w);
          ps2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
      }
      if constexpr (PROD && FN == 4) {
        // twin of a 64-column wave tile: lane (fr, fg) holds 4 bf16 of each of the four 16-column fragments — stored as they stand
        // that is 32 contiguous bytes per row and instruction (partial-line writes: +25 % on the K = 512 GEMMs).  The 16 x 64 block goes
        // through a wave-private LDS patch (the ring is free by now; 144-byte rows) and leaves as whole 128-byte row segments, eight
        // lanes per row.
        if (p.tw_out) {                      // (uniform: every lane of the wave gets here — rows beyond M are masked at the store)
          char* patch = smem + (tid >> 6) * (16 * 144);
#pragma unroll
          for (int j = 0; j < FN; ++j) *reinterpret_cast<uint2*>(patch + fr * 144 + j * 32 + fg * 8) = twv[j];
          const int pr = lane >> 3, pc = lane & 7;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int mr_ = m0 + wm * WTM + i * 16 + pr + 8 * h;
            const uint4 w = *reinterpret_cast<const uint4*>(patch + (pr + 8 * h) * 144 + pc * 16);
            if (mr_ < p.M && n0 + wn * WTN < p.N) *reinterpret_cast<uint4*>(p.tw_out + map_row(mr_, p.cmap) * p.ldtw + n0 + wn * WTN + pc * 8) = w;      // (N % 64 == 0: a wave tile's 64 columns are in or out together)
          }
        }
      }
      if (lnp) {       // the four lanes (fg = 0..3) that hold a row's columns of this wave tile
        ps1 += __shfl_xor(ps1, 16, 64); ps2 += __shfl_xor(ps2, 16, 64);
        ps1 += __shfl_xor(ps1, 32, 64); ps2 += __shfl_xor(ps2, 32, 64);
        if (fg == 0 && rowok && n0 + wn * WTN < p.N) {
          if constexpr (WTN >= 64) {
            static_assert(WTN == 64 || BM == 0, "one 64-column slot per wave tile");
            reinterpret_cast<float2*>(p.st_out)[crow[ii] * (p.N >> 6) + ((n0 + wn * WTN) >> 6)] = float2{ps1, ps2};
          } else {
            reinterpret_cast<float2*>(smem)[(wm * WTM + i * 16 + fr) * WN_ + wn] = float2{ps1, ps2};
          }
        }
      }
    }
  }
  if constexpr (PROD && WTN < 64) {
    static_assert(WTN == 32 || BM == 0, "two wave tiles per 64-column slot");
    if (lnp) {                                         // (uniform per launch: every wave of the workgroup gets here)
      __syncthreads();
      constexpr int SPT = BN / 64;                     // slots per tile
      if (tid < BM * SPT) {
        const int ml = tid % BM, sl = tid / BM, m = m0 + ml;
        if (m < p.M && n0 + 64 * sl < p.N) {
          const float2 a = reinterpret_cast<const float2*>(smem)[ml * WN_ + 2 * sl], b = reinterpret_cast<const float2*>(smem)[ml * WN_ + 2 * sl + 1];
          reinterpret_cast<float2*>(p.st_out)[map_row(m, p.cmap) * (p.N >> 6) + (n0 >> 6) + sl] = float2{a.x + b.x, a.y + b.y};
        }
      }
    }
  }
}

// Staged epilogue for the bf16-output GEMMs (qkv / fc1 forward, every input-gradient GEMM): the whole workgroup tile goes
// through LDS ONCE as bf16 so that the global stores are full rows — 16 lanes x 16 bytes = the tile row's whole
// BN * 2 bytes (two 128-byte lines for BN = 128) per instruction — instead of the 32-byte pieces the transposed accumulators
// give directly (measured slower than even the old LDS-staged 64-byte pieces: partial-line writes).  Elementwise work
// (alpha, bias, GELU / GELU' twin, multiply by aux) happens on the registers, four consecutive columns at a time; the tile
// image has 16 bytes of padding per row (the 16 rows of a ds_write_b64 lane group land on distinct banks).  One or two outputs (C and
// the bf16 twin C2) are staged side by side.  Eligible: bf16 C, no residual, N % 8 == 0, ldc % 8 == 0 (nt_staged_ok).
template <int BM, int BN>
constexpr int nt_stage_bytes() { return BM * (BN * 2 + 16); }
// C and its bf16 twin side by side, or (tiles above 128 x 128) one image at a time inside the ring's footprint
template <int BM, int BN> constexpr bool nt_epi_split() { return BM * BN > 128 * 128; }

__device__ __forceinline__ bool nt_staged_ok(const NTParams& p) {
  return p.C && p.c_bf16 && !p.res && !p.beta && !(p.N & 7) && !(p.ldc & 7) && (p.c2_mode == 0 || (p.c2_mode != 3 && !(p.ldc2 & 7))) &&
         !(p.debug & 3);
}

// SPLIT: the ring only holds ONE tile image (256-row tiles at two workgroups per CU) -> C and the twin go through it one
// after the other, the twin's packed values waiting in registers.
template <int BM, int BN, int NTHREADS, int FM, int FN, int WTM, int WTN, bool SPLIT = false>
__device__ __forceinline__ void nt_epilogue_s(const NTParams& p, f32x4 (&acc)[FM][FN], char* lds, int m0, int n0, int wm, int wn,
                                              int lane, int tid, const float2* ln_mr = nullptr) {      // (ln_mr: a compile-time null in the kernels without the LayerNorm code)
  constexpr int RB = BN * 2 + 16;                         // padded tile row (16: keeps the b128 row reads aligned)
  const int fr = lane & 15, fg = lane >> 4;
  const bool lnc = ln_mr != nullptr && p.ln_st != nullptr;      // LayerNorm consumer (see nt_epilogue_t)      // LayerNorm consumer (see nt_epilogue_t)
  char* img2 = SPLIT ? lds : lds + BM * RB;
  uint2 w2r[SPLIT ? FM : 1][SPLIT ? FN : 1];
  // all global reads of the elementwise part (bias per fragment column, aux per fragment) go out before the first use: one
  // memory latency per tile instead of one per fragment (see nt_epilogue_t)
  float4 bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = n0 + wn * WTN + j * 16 + fg * 4;
    bv[j] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : float4{0.f, 0.f, 0.f, 0.f};
  }
  constexpr int AH = FN <= 2 ? FM : 1;       // fragment rows whose aux reads are in flight together (8-16 registers)
  uint2 av[AH][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    if (i % AH == 0) {
#pragma unroll
      for (int ii = 0; ii < AH; ++ii)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int m = m0 + wm * WTM + (i + ii) * 16 + fr, n = n0 + wn * WTN + j * 16 + fg * 4;
          av[ii][j] = ((p.act == 2 || p.act == 3) && i + ii < FM && m < p.M && n < p.N) ? *reinterpret_cast<const uint2*>(p.aux + (long)m * p.ldaux + n) : uint2{0, 0};
        }
    }
    const int ml = wm * WTM + i * 16 + fr;
    const int m = m0 + ml;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int nl = wn * WTN + j * 16 + fg * 4;
      const int n = n0 + nl;
      float4 v;
      const bool ok = m < p.M && n < p.N;
      if (ln_mr != nullptr) {
        v = nt_ln_frag<BM>(acc[i][j], p.alpha, lnc, ln_mr, ml, nl);
        v.x += bv[j].x; v.y += bv[j].y; v.z += bv[j].z; v.w += bv[j].w;
      } else {
        v.x = acc[i][j][0] * p.alpha + bv[j].x; v.y = acc[i][j][1] * p.alpha + bv[j].y;
        v.z = acc[i][j][2] * p.alpha + bv[j].z; v.w = acc[i][j][3] * p.alpha + bv[j].w;
      }
      uint2 w2 = uint2{0, 0};
      if (p.c2_mode == 1) { w2.x = pack2bf(v.x, v.y); w2.y = pack2bf(v.z, v.w); }
      if (p.act == 1) {
        float4 d;
        gelu_pair_f(v.x, v.x, d.x); gelu_pair_f(v.y, v.y, d.y); gelu_pair_f(v.z, v.z, d.z); gelu_pair_f(v.w, v.w, d.w);
        if (p.c2_mode == 4) { w2.x = pack2bf(d.x, d.y); w2.y = pack2bf(d.z, d.w); }
      } else if ((p.act == 2 || p.act == 3) && ok) {
        const uint2 a = av[i % AH][j];
        float4 g;
        g.x = __uint_as_float(a.x << 16); g.y = __uint_as_float(a.x & 0xffff0000u);
        g.z = __uint_as_float(a.y << 16); g.w = __uint_as_float(a.y & 0xffff0000u);
        if (p.act == 2) { g.x = gelu_grad_f(g.x); g.y = gelu_grad_f(g.y); g.z = gelu_grad_f(g.z); g.w = gelu_grad_f(g.w); }
        v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
      }
      if (p.c2_mode == 2) { w2.x = pack2bf(v.x, v.y); w2.y = pack2bf(v.z, v.w); }
      uint2 w; w.x = pack2bf(v.x, v.y); w.y = pack2bf(v.z, v.w);
      *reinterpret_cast<uint2*>(lds + ml * RB + nl * 2) = w;
      if (SPLIT) w2r[SPLIT ? i : 0][SPLIT ? j : 0] = w2;
      else if (p.c2_mode) *reinterpret_cast<uint2*>(img2 + ml * RB + nl * 2) = w2;
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;                             // 16-byte chunks per tile row
  constexpr int RPI = NTHREADS / CPR;                     // rows per pass
  const int ch = tid % CPR, r0 = tid / CPR;
  const int n = n0 + ch * 8;
  if (n < p.N) {
#pragma unroll
    for (int it = 0; it < BM / RPI; ++it) {
      const int ml = it * RPI + r0;
      const int m = m0 + ml;
      if (m >= p.M) continue;
      const uint4 w = *reinterpret_cast<const uint4*>(lds + ml * RB + ch * 16);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + map_row(m, p.cmap) * p.ldc + n) = w;
      if (!SPLIT && p.c2_mode) {
        const uint4 w2 = *reinterpret_cast<const uint4*>(img2 + ml * RB + ch * 16);
        *reinterpret_cast<uint4*>(p.C2 + (long)m * p.ldc2 + n) = w2;
      }
    }
  }
  if (SPLIT && p.c2_mode) {
    __syncthreads();                                      // the C image has been read
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        *reinterpret_cast<uint2*>(lds + (wm * WTM + i * 16 + fr) * RB + (wn * WTN + j * 16 + fg * 4) * 2) = w2r[SPLIT ? i : 0][SPLIT ? j : 0];
    __syncthreads();
    if (n < p.N) {
#pragma unroll
      for (int it = 0; it < BM / RPI; ++it) {
        const int ml = it * RPI + r0;
        const int m = m0 + ml;
        if (m >= p.M) continue;
        *reinterpret_cast<uint4*>(p.C2 + (long)m * p.ldc2 + n) = *reinterpret_cast<const uint4*>(lds + ml * RB + ch * 16);
      }
    }
  }
}

// ring / epilogue image of a workgroup; the BM row statistics (float2 mean, rstd) of a LayerNorm-consuming GEMM sit behind it
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK>
constexpr size_t nt2_lds_bytes() {
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * BK * 2;
  constexpr size_t epi = (size_t)(nt_epi_split<BM, BN>() ? 1 : 2) * nt_stage_bytes<BM, BN>();
  return ring > epi ? ring : epi;
}
// LayerNorm consumer: BM x {mean, rstd} + BN x c.  Normally behind the ring / image and filled in the prologue (latency under the first
// tiles' flight); where those bytes would cost a workgroup per CU (64 x 64 tiles: five 32 KB rings fill the 160 KB exactly) they go
// into the part of the ring the epilogue image leaves free and are filled when the k-loop is over (LN_LATE).
template <int BM, int BN>
constexpr size_t nt2_ln_bytes() { return (size_t)BM * 8 + (size_t)BN * 4; }
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK>
constexpr bool nt2_ln_late() {
  constexpr size_t base = nt2_lds_bytes<BM, BN, WM_, WN_, STAGES, BT, BK>();
  constexpr size_t ring = (size_t)STAGES * (BM + BN) * BK * 2;
  constexpr size_t epi = (size_t)(nt_epi_split<BM, BN>() ? 1 : 2) * nt_stage_bytes<BM, BN>();
  return (160 * 1024) / base != (160 * 1024) / (base + nt2_ln_bytes<BM, BN>()) && ring >= epi + nt2_ln_bytes<BM, BN>();
}
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK>
constexpr size_t nt2_ln_offset() {
  constexpr size_t epi = (size_t)(nt_epi_split<BM, BN>() ? 1 : 2) * nt_stage_bytes<BM, BN>();
  return nt2_ln_late<BM, BN, WM_, WN_, STAGES, BT, BK>() ? epi : nt2_lds_bytes<BM, BN, WM_, WN_, STAGES, BT, BK>();
}
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK, bool LNK>
constexpr size_t nt2_lds_alloc() {
  return nt2_lds_bytes<BM, BN, WM_, WN_, STAGES, BT, BK>() + (nt2_ln_late<BM, BN, WM_, WN_, STAGES, BT, BK>() || !LNK ? 0 : nt2_ln_bytes<BM, BN>());
}

#include "gemm_nt256.h"
#include "gemm_tn_gang.h"

// PIPE == 2: two extra LOADER waves per workgroup issue every global -> LDS piece of the ring; the WM_ x WN_ compute waves only wait
// at the per-step barrier, read fragments and issue MFMAs.  A wave that issues LDS-DMA pieces is blocked by the vector-memory path's
// back-pressure for ~700 cycles per k-step (profiles/r02_gemm_phase_profile.txt), and in-order issue puts its MFMAs behind that
// stall: with the roles split the stall overlaps the other waves' MFMAs (step = max(stream, compute) instead of their sum).
template <int BM, int BN, int NTC, bool BT, int BK, int NL>
__device__ __forceinline__ void nt2_loader(const NTParams& p, int m0, int n0, int nk, char* smem) {
  constexpr int ARB = BK * 2, ACPR = BK / 8, A_CH = BM * ACPR / NL, B_CH = BN * ACPR / NL;      // NL loader threads
  constexpr int A_BYTES = BM * ARB, STAGE_BYTES = (BM + BN) * ARB, BRB = BN * 2, BCPR = BN / 8;
  static_assert(BK == 64 && BM * ACPR % NL == 0 && BN * ACPR % NL == 0, "loader tile");
  const int ltid = threadIdx.x - NTC, lw = __builtin_amdgcn_readfirstlane(ltid >> 6);
  auto rswz = [](int row) { return (row >> 1) & 7; };
  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = ltid + NL * i, row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = p.A + map_row(gm, p.amap) * p.lda + ls * 8;
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = ltid + NL * i;
    if (!BT) {
      const int row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
      int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
      b_src[i] = p.B + (long)gn * p.ldb + ls * 8;
    } else {
      const int row = c / BCPR, pc = c % BCPR;
      const int lc = (((pc >> 1) ^ tn2_swz<BRB>(row)) << 1) | (pc & 1);
      int gc = n0 + lc * 8; gc = gc < p.N ? gc : p.N - 8;
      b_src[i] = p.B + (long)row * p.ldb + gc;
    }
  }
  auto dma_tile = [&](int kt) {
    const int k0 = kt * BK;
    char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, a_src[i] + k0), LDS_PTR(void, st + (lw * 64 + NL * i) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, BT ? b_src[i] + (long)k0 * p.ldb : b_src[i] + k0),
                                       LDS_PTR(void, st + A_BYTES + (lw * 64 + NL * i) * 16), 16, 0, 0);
  };
  dma_tile(0);
  for (int kt = 0; kt < nk; ++kt) {
    wait_vmcnt<0>();                       // tile kt has landed
    __builtin_amdgcn_s_barrier();          // ... and the compute waves are done with tile kt - 1: its stage is free
    if (kt + 1 < nk) dma_tile(kt + 1);
  }
  // the loaders leave here; the barriers of the epilogue count the surviving (compute) waves only
}

// LNK: the instantiation that carries the LayerNorm-folding code (DavNtLn producer / consumer); every launch without such a problem runs
// the LNK = false kernels, which are instruction for instruction what they were before the folding existed.
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK, bool PROF = false, int PIPE = 0, bool LNK = false>
__device__ __forceinline__ void nt2_body(const NTParams& p, int bid, int tid_in = -1) {
  // PROF (tile configuration 30, tools/gemm_phase_prof.py only): per-wave shader-cycle sums of the k-loop phases —
  // [wait for the DMA, barrier, DMA issue, fragment reads + MFMAs, prologue, epilogue] — written to the int64 buffer
  // passed in place of res_rows (s_memtime; the instrumentation itself costs ~10 %).
  long long pt[6] = {0, 0, 0, 0, 0, 0};
  long long tk0 = 0, tstart = 0;
  if (PROF) tstart = tk0 = __builtin_readcyclecounter();
  // (a register: in the grouped kernels p points into the by-value table in the kernarg segment, and read through p inside the
  // k-loop this flag was a scalar load + s_waitcnt lgkmcnt(0) per k-step — the "memory" clobbers of the counted waits forbid hoisting it)
  const bool dbg4 = (p.debug & 4) != 0;
  constexpr int NT = WM_ * WN_ * 64;
  constexpr int WTM = BM / WM_, WTN = BN / WN_, FM = WTM / 16, FN = WTN / 16;
  constexpr int ARB = BK * 2, ACPR = BK / 8;               // LDS row bytes / 16-byte chunks per row of A (and NT-mode B) tiles
  constexpr int A_CH = BM * ACPR / NT, B_CH = BN * ACPR / NT, LPT = A_CH + B_CH;
  constexpr int A_BYTES = BM * ARB, STAGE_BYTES = (BM + BN) * ARB;
  static_assert(BM * ACPR % NT == 0 && BN * ACPR % NT == 0, "tile/threads mismatch");
  static_assert(BK == 64 || BK == 32, "BK");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // (tid_in: the persistent walk passes the lane id through an opaque asm per tile, so that nothing derived from it is hoisted out of
  // the tile loop and kept in registers across the whole body: +35 registers, one workgroup per CU instead of two)
  const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  {
    const int nwg = tiles_m * tiles_n;
    if (bid >= nwg) return;                 // grouped launches pad every problem's block range to a multiple of 8
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Each XCD owns a contiguous range of virtual tile ids.  Tiles are ordered panel by panel (a panel = up to
  // ~8 tile columns), row-major inside a panel, so an XCD's ~T/8 tiles form a compact rows x panel-width
  // block whose A rows and B columns (a few MB) stay resident in that XCD's 4 MB L2.
  int bm, bn;
  {
    const int npan = (tiles_n + 7) >> 3;
    const int wn_ = (tiles_n + npan - 1) / npan;
    const int per = tiles_m * wn_;
    const int pnl = bid / per, rem = bid - pnl * per;
    const int wp = (tiles_n - pnl * wn_) < wn_ ? (tiles_n - pnl * wn_) : wn_;
    bm = rem / wp;
    bn = pnl * wn_ + rem % wp;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  if constexpr (PIPE >= 2) {
    if (wave >= WM_ * WN_) {                 // PIPE loader waves (two-stage ring of 64-deep stages only)
      nt2_loader<BM, BN, NT, BT, BK, PIPE * 64>(p, m0, n0, p.K / BK, smem);
      return;
    }
  }
  // 16-byte-slot XOR swizzle of the row-major tiles: 128-byte rows (BK 64) / 64-byte rows (BK 32)
  auto rswz = [](int row) { return BK == 64 ? ((row >> 1) & 7) : ((0 - (row >> 2)) & 3); };

  static_assert(!LNK || (!BT && PIPE == 0 && !PROF), "LayerNorm folding: forward [N, K] kernels only");
  float2* ln_mr = reinterpret_cast<float2*>(smem + nt2_ln_offset<BM, BN, WM_, WN_, STAGES, BT, BK>());      // LayerNorm consumer: see below
  constexpr bool LN_LATE = nt2_ln_late<BM, BN, WM_, WN_, STAGES, BT, BK>();
  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
#pragma unroll
  for (int i = 0; i < (PIPE >= 2 ? 0 : A_CH); ++i) {
    const int c = tid + NT * i, row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = (LNK ? nt_a_row(p, gm) : p.A + map_row(gm, p.amap) * p.lda) + ls * 8;
  }
  // B tile: NT mode = [BN rows][BK k] (as A); BT mode (B given as [K, N], the dgrad reading W itself) =
  // [64 k rows][BN cols] (BN*2-byte rows) read back with the transposing ds_read_b64_tr_b16.
  constexpr int BRB = BN * 2, BCPR = BN / 8;
#pragma unroll
  for (int i = 0; i < (PIPE >= 2 ? 0 : B_CH); ++i) {
    const int c = tid + NT * i;
    if (!BT) {
      const int row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
      int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
      b_src[i] = p.B + (long)gn * p.ldb + ls * 8;
    } else {
      const int row = c / BCPR, pc = c % BCPR;
      const int lc = (((pc >> 1) ^ tn2_swz<BRB>(row)) << 1) | (pc & 1);
      int gc = n0 + lc * 8; gc = gc < p.N ? gc : p.N - 8;
      b_src[i] = p.B + (long)row * p.ldb + gc;
    }
  }
  auto dma_tile = [&](int kt) {
    const int k0 = kt * BK;
    char* st = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, a_src[i] + k0), LDS_PTR(void, st + (wave * 64 + NT * i) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, BT ? b_src[i] + (long)k0 * p.ldb : b_src[i] + k0),
                                       LDS_PTR(void, st + A_BYTES + (wave * 64 + NT * i) * 16), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  if (PIPE < 2) {
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
      if (s < nk) dma_tile(s);
  }
  // LayerNorm consumer (p.ln_st): the row statistics of this tile's BM rows — mean, rstd out of the K / 64 partial sums per row —
  // are formed once per workgroup into LDS behind the ring / epilogue image, by four lanes per row (dav_ln_row_stats: the same
  // summation order as the LayerNorm backward), and the tile's BN values of c behind them; the epilogue reads both after the barrier
  // that ends the k-loop.  The loads go out BEHIND the ring's first DMA pieces (their latency lies under the first tiles' flight; the
  // compiler's wait for them drains the DMA queue once, where the k-loop's first wait would drain most of it anyway).
  auto ln_stats_to_lds = [&]() {
    const int ns = p.K >> 6;
    for (int r = tid >> 2; r < BM; r += NT >> 2) {
      int gm = m0 + r; gm = gm < p.M ? gm : p.M - 1;
      const float2 mr = dav_ln_row_stats(nt_stat_row(p, gm, ns), ns, tid & 3, p.K, p.ln_eps);
      if ((tid & 3) == 0) ln_mr[r] = mr;
    }
    float* ln_cv = reinterpret_cast<float*>(ln_mr + BM);
    for (int c = tid; c < BN; c += NT) ln_cv[c] = n0 + c < p.N ? p.ln_c[n0 + c] : 0.f;
  };
  if (LNK && !LN_LATE && p.ln_st != nullptr) ln_stats_to_lds();

  const int fr = lane & 15, fg = lane >> 4;
  uint32_t bt_base[FN];
  if constexpr (BT) {
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
#pragma unroll
    for (int j = 0; j < FN; ++j) bt_base[j] = lds0 + tn2_frag_base<BRB>(wn * WTN + j * 16, lane);
  }
  if (PROF) { const long long t = __builtin_readcyclecounter(); pt[4] = t - tk0; tk0 = t; }
  if constexpr (PIPE == 1) {
    // Software-pipelined k-loop (2-stage ring, BK = 64).  Measured on the plain loop below (tools/gemm_phase_prof.py): per
    // k-step a wave spends 28 % of its time issuing its 4 DMA pieces (all 16 waves of the CU hit the one 64 B/clk
    // address path at once, right after the barrier) and 43 % in "read fragments -> wait -> 4 MFMAs" rounds.  Here the two
    // 32-deep halves of a k-step are double-buffered in registers: the fragment reads of the next half are in flight
    // under the 8 MFMAs of the current one, the DMA pieces of the tile after next are issued one per two MFMAs, and
    // the one barrier per k-step sits between the halves, where 8 MFMAs are already queued.
    static_assert(STAGES == 2 && BK == 64, "pipelined loop: 2 stages, BK 64");
    auto load_frags = [&](const char* Ab, int kk, bf16x8 (&af)[FM], bf16x8 (&bfr)[FN]) {
      const char* Bb = Ab + A_BYTES;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(Ab + row * ARB + (((kk * 4 + fg) ^ rswz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!BT) {
          const int row = wn * WTN + j * 16 + fr;
          bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + row * ARB + (((kk * 4 + fg) ^ rswz(row)) << 4));
        } else {
          bfr[j] = tn2_frag<BRB>(Bb, kk * 32, wn * WTN + j * 16, lane);
        }
      }
    };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // wave-uniform in an SGPR: the DMA's LDS base goes through M0
    auto dma_piece = [&](int kt, int piece) {
      const int k0 = kt * BK;
      char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < A_CH; ++i)
        if (piece == i)
          __builtin_amdgcn_global_load_lds(GLB_PTR(void, a_src[i] + k0), LDS_PTR(void, st + (wave_u * 64 + NT * i) * 16), 16, 0, 0);
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        if (piece == A_CH + i)
          __builtin_amdgcn_global_load_lds(GLB_PTR(void, BT ? b_src[i] + (long)k0 * p.ldb : b_src[i] + k0),
                                           LDS_PTR(void, st + A_BYTES + (wave_u * 64 + NT * i) * 16), 16, 0, 0);
    };
    constexpr int NMFMA = FM * FN;
    constexpr int EVERY = NMFMA / LPT > 0 ? NMFMA / LPT : 1;      // one DMA piece per EVERY MFMAs
    bf16x8 a0[FM], b0[FN], a1[FM], b1[FN];
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                                 // tile 0 is in LDS
    if (1 < nk) dma_tile(1);
    load_frags(smem, 0, a0, b0);
    for (int kt = 0; kt < nk; ++kt) {
      const char* Ab = smem + (kt & 1) * STAGE_BYTES;
      // first-half MFMAs; the second half's fragment reads go out behind the first row of them (issued in front, the
      // compiler's in-order lgkmcnt wait for the first half's fragments would also wait for these)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[0], acc[0][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(Ab, 1, a1, b1);                                  // second half of tile kt: in flight under the MFMAs below
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 1; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j], a0[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kt + 1 < nk) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // my pieces of tile kt+1 landed; my reads of tile kt are done
        __builtin_amdgcn_s_barrier();                             // tile kt+1 visible to all, stage kt & 1 free for tile kt+2
        const bool more = kt + 2 < nk;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
            const int idx = i * FN + j;
            if (idx == FN - 1) {                                    // behind the first row of MFMAs (see above)
              __builtin_amdgcn_sched_barrier(0);
              load_frags(smem + ((kt + 1) & 1) * STAGE_BYTES, 0, a0, b0);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (idx % EVERY == EVERY - 1 && idx / EVERY < LPT) {
              if (more) dma_piece(kt + 2, idx / EVERY);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        if (LPT > NMFMA / EVERY) {                                // more pieces than slots (does not happen for the shipped tiles)
#pragma unroll
          for (int q = NMFMA / EVERY; q < LPT; ++q)
            if (more) dma_piece(kt + 2, q);
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j], a1[i], acc[i][j], 0, 0, 0);
      }
    }
  } else
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt must have landed; up to STAGES-2 younger tiles may stay in flight
    const int younger = nk - 1 - kt;
    if (PIPE >= 2) {}                      // the loader waves wait for their pieces (nt2_loader); here only the barrier
    else if (STAGES >= 3 && younger >= STAGES - 2) wait_vmcnt<(STAGES - 2) * LPT>();
    else if (STAGES >= 4 && younger == 1) wait_vmcnt<LPT>();
    else wait_vmcnt<0>();
    if (PROF) { const long long t = __builtin_readcyclecounter(); pt[0] += t - tk0; tk0 = t; }
    __builtin_amdgcn_s_barrier();          // everyone's pieces of tile kt are in LDS; stage (kt-1)%STAGES is free
    if (PROF) { const long long t = __builtin_readcyclecounter(); pt[1] += t - tk0; tk0 = t; }
    if (PIPE < 2 && kt + STAGES - 1 < nk) dma_tile(kt + STAGES - 1);
    if (PROF) { const long long t = __builtin_readcyclecounter(); pt[2] += t - tk0; tk0 = t; }
    const char* Ab = smem + (kt % STAGES) * STAGE_BYTES;
    const char* Bb = Ab + A_BYTES;
    if (dbg4) continue;                    // timing experiment: the global -> LDS stream alone
    if constexpr (BT) {
      // b_kn operand: its fragments come from transposing reads, issued as asm (tn2_tr: a builtin read would make hipcc drain the
      // DMA issued just above).  All reads of the stage go out first, each 32-deep half is multiplied when ITS reads are back.
      constexpr int KK = BK / 32, RPK = FM + 2 * FN;          // ds_read instructions per half: FM b128 + 2 FN tr-b64
      static_assert((KK - 1) * RPK <= 15 || KK == 1, "lgkmcnt field");
      const uint32_t so = (uint32_t)((kt % STAGES) * STAGE_BYTES);
      bf16x8 af[KK][FM], bfr[KK][FN];
      auto rd = [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = wm * WTM + i * 16 + fr;
          af[kk][i] = *reinterpret_cast<const bf16x8*>(Ab + row * ARB + (((kk * 4 + fg) ^ rswz(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[kk][j] = tn2_tr<A_BYTES + kk * 32 * BRB, BRB>(bt_base[j] + so);
      };
      auto mm = [&](auto kc) {
        constexpr int kk = decltype(kc)::value;
        wait_lgkmcnt<(KK - 1 - kk) * RPK>();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      };
      rd(IntC<0>{});
      if constexpr (KK == 2) rd(IntC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      mm(IntC<0>{});
      if constexpr (KK == 2) mm(IntC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      if (PROF) {
        asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[FM - 1][FN - 1]));
        const long long t = __builtin_readcyclecounter(); pt[3] += t - tk0; tk0 = t;
      }
      continue;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int row = wm * WTM + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(Ab + row * ARB + (((kk * 4 + fg) ^ rswz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        if (!BT) {
          const int row = wn * WTN + j * 16 + fr;
          bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + row * ARB + (((kk * 4 + fg) ^ rswz(row)) << 4));
        } else {
          bfr[j] = tn2_frag<BRB>(Bb, kk * 32, wn * WTN + j * 16, lane);
        }
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (PROF) {
      asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[FM - 1][FN - 1]));      // the MFMAs have been issued before the stamp
      const long long t = __builtin_readcyclecounter(); pt[3] += t - tk0; tk0 = t;
    }
  }

  __syncthreads();                          // all waves finished reading the last stage
  if (PROF) {
    NTParams q = p;
    q.res_rows = nullptr;
    nt_epilogue_t<FM, FN, WTM, WTN>(q, acc, m0, n0, wm, wn, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t = __builtin_readcyclecounter(); pt[5] = t - tk0;
    if (lane == 0) {
      long long* out = reinterpret_cast<long long*>(const_cast<int*>(p.res_rows)) + ((long)blockIdx.x * (WM_ * WN_) + wave) * 10;
#pragma unroll
      for (int i = 0; i < 6; ++i) out[i] = pt[i];
      out[6] = tstart;
      out[7] = t;
      out[8] = __builtin_amdgcn_s_getreg(6164);     // HW_REG_XCC_ID (id 20), offset 0, size 4: ((4-1) << 11) | 20
      out[9] = __builtin_amdgcn_s_getreg(63492);    // HW_REG_HW_ID (id 4), all 32 bits: ((32-1) << 11) | 4
    }
    return;
  }
  if (LNK && LN_LATE && p.ln_st != nullptr) {      // (uniform per launch)
    ln_stats_to_lds();
    __syncthreads();
  }
  if (nt_staged_ok(p)) nt_epilogue_s<BM, BN, NT, FM, FN, WTM, WTN, nt_epi_split<BM, BN>()>(p, acc, smem, m0, n0, wm, wn, lane, tid, LNK ? ln_mr : nullptr);
  else if constexpr (!LNK) nt_epilogue_t<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, lane);
  else nt_epilogue_t<FM, FN, WTM, WTN, BM, BN, WN_>(p, acc, m0, n0, wm, wn, lane, ln_mr, smem, tid);
}

#ifdef DAV_EXPERIMENTAL
__global__ __launch_bounds__(512) void gemm_nt2_prof_kernel(NTParams p) {
  nt2_body<128, 128, 2, 4, 2, false, 64, true>(p, blockIdx.x);
}
#endif

// ------------------------------------------------------------------------------------------------
// NT kernel, third generation ("staggered halves"): 256 x 128 tile, 8 waves of 64 x 64, six 24 KB ring stages of
// 32 contraction rows each (144 KB, one workgroup per CU).
//
// Why: the 128 x 128 kernel above needs 32 KB of operands per 2.1 MFLOP k-step — at the CU's ~64 B/clk operand path
// that is as long as the k-step's MFMAs, and its LDS traffic is as long again, so three equally loaded units would have
// to overlap perfectly; measured (tools/gemm_phase_prof.py) its waves spend 28 % of a k-step queueing DMA pieces, 43 % in
// read -> wait -> MFMA rounds, 24 % of a tile in the epilogue.  A 256 x 128 tile halves the operand and LDS bytes per
// flop, and the overlap is built in rather than hoped for: the workgroup's two halves (waves 0-3 / 4-7: one wave of each
// on every SIMD) run the same phase sequence ONE BARRIER APART, so that whenever one half issues the 16 MFMAs of a stage
// the other half reads the fragments of its next stage and issues its DMA pieces.  Per stage and half:
// L (4 A + 4 B fragment reads, 3 DMA pieces, counted vmcnt) | M (16 MFMAs) |, '|' = s_barrier.
//
// Ring protocol (t = stage index along K, ring slot t % 6; H0's L(t) / M(t) run in barrier intervals 2t+1 / 2t+2, H1's one
// later):
//  * stage u's three DMA pieces per wave are issued in L(u-4): the slot held stage u-6, whose last reads (H1's, retired by
//    the lgkmcnt(0) of its M(u-6)) finished several barriers earlier; a DMA is in flight for three stages (~6 phases)
//    before anyone waits for it;
//  * the counted wait at the end of L(t) — vmcnt(9): everything but the pieces of stages t+2 .. t+4 — retires this wave's
//    pieces of stage t+1; the barriers behind it (H1 runs it one interval before H0 first reads stage t+1) publish them.
// ------------------------------------------------------------------------------------------------
// ABL (profiling builds only): 1 = no DMA inside the loop, 2 = no fragment reads inside the loop, 3 = no s_setprio,
// 4 = no MFMAs
template <bool BT, bool PROF = false, int ABL = 0>
__device__ __forceinline__ void nt3_body(const NTParams& p, int bid) {
  constexpr int BM = 256, BN = 128, BK = 32, NT = 512, NS = 6, D = 4;   // D = stages a DMA runs ahead
  long long pt[6] = {0, 0, 0, 0, 0, 0};      // PROF (configuration 33): cycles in L / barrier / M / barrier, prologue, epilogue
  long long tk0 = 0, tstart = 0;
  if (PROF) tstart = tk0 = __builtin_readcyclecounter();
#define NT3_STAMP(i) if (PROF) { const long long t_ = __builtin_readcyclecounter(); pt[i] += t_ - tk0; tk0 = t_; }
  constexpr int WTM = 64, WTN = 64, FM = 4, FN = 4;
  constexpr int ARB = BK * 2, ACPR = BK / 8;
  constexpr int A_CH = BM * ACPR / NT, B_CH = BN * ACPR / NT, LPT = A_CH + B_CH;   // 2 + 1 DMA pieces per wave and stage
  constexpr int A_BYTES = BM * ARB, STAGE_BYTES = (BM + BN) * ARB;      // 24 KB
  constexpr int BRB = BN * 2, BCPR = BN / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave >> 2, wr = wave & 3;
  const int wm = (wr >> 1) + 2 * half, wn = wr & 1;                     // 4 x 2 waves
  const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
  {
    const int nwg = tiles_m * tiles_n;
    if (bid >= nwg) return;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int bm, bn;
  {
    const int npan = (tiles_n + 7) >> 3;
    const int wn_ = (tiles_n + npan - 1) / npan;
    const int per = tiles_m * wn_;
    const int pnl = bid / per, rem = bid - pnl * per;
    const int wp = (tiles_n - pnl * wn_) < wn_ ? (tiles_n - pnl * wn_) : wn_;
    bm = rem / wp;
    bn = pnl * wn_ + rem % wp;
  }
  const int m0 = bm * BM, n0 = bn * BN;
  auto rswz = [](int row) { return (0 - (row >> 2)) & 3; };           // 64-byte rows: 4 slots, 4 rows per bank row

  const bf16_t* a_src[A_CH];
  const bf16_t* b_src[B_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int c = tid + NT * i, row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
    int gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = p.A + map_row(gm, p.amap) * p.lda + ls * 8;
  }
#pragma unroll
  for (int i = 0; i < B_CH; ++i) {
    const int c = tid + NT * i;
    if (!BT) {
      const int row = c / ACPR, ls = (c % ACPR) ^ rswz(row);
      int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
      b_src[i] = p.B + (long)gn * p.ldb + ls * 8;
    } else {
      const int row = c / BCPR, pc = c % BCPR;
      const int lc = (((pc >> 1) ^ tn2_swz<BRB>(row)) << 1) | (pc & 1);
      int gc = n0 + lc * 8; gc = gc < p.N ? gc : p.N - 8;
      b_src[i] = p.B + (long)row * p.ldb + gc;
    }
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_stage = [&](int t) {
    const int k0 = t * BK;
    char* st = smem + (t % NS) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < A_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, a_src[i] + k0), LDS_PTR(void, st + (wave_u * 64 + NT * i) * 16), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, BT ? b_src[i] + (long)k0 * p.ldb : b_src[i] + k0),
                                       LDS_PTR(void, st + A_BYTES + (wave_u * 64 + NT * i) * 16), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  const int fr = lane & 15, fg = lane >> 4;
  // prologue: stages 0 .. D-1
#pragma unroll
  for (int t = 0; t < D; ++t)
    if (t < nk) dma_stage(t);
  // stage 0 landed: everything but the (up to) D-1 younger stages
  if (nk >= D) wait_vmcnt<(D - 1) * LPT>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();                       // stage 0 published
  if (half == 1) __builtin_amdgcn_s_barrier();        // the second half runs one barrier behind the first
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 af[FM], bfr[FN];
  NT3_STAMP(4)
  for (int t = 0; t < nk; ++t) {
    const char* Ab = smem + (t % NS) * STAGE_BYTES;
    const char* Bb = Ab + A_BYTES;
    // ---- L(t): fragments of stage t; DMA of stage t+D; retire this wave's pieces of stage t+1
    if (ABL != 2 || t == 0) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int row = wm * WTM + i * 16 + fr;
      af[i] = *reinterpret_cast<const bf16x8*>(Ab + row * ARB + ((fg ^ rswz(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      if (!BT) {
        const int row = wn * WTN + j * 16 + fr;
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + row * ARB + ((fg ^ rswz(row)) << 4));
      } else {
        bfr[j] = tn2_frag<BRB>(Bb, 0, wn * WTN + j * 16, lane);
      }
    }
    }
    if (t + D < nk) {
      if (ABL != 1) dma_stage(t + D);
      wait_vmcnt<(D - 1) * LPT>();                   // younger: stages t+2 .. t+D
    } else {
      // tail: fewer younger stages are in flight; the exact count is (nk-1) - (t+1) stages
      const int younger = nk - 2 - t;
      if (younger >= 2) wait_vmcnt<2 * LPT>();
      else if (younger == 1) wait_vmcnt<LPT>();
      else wait_vmcnt<0>();
    }
    __builtin_amdgcn_sched_barrier(0);
    NT3_STAMP(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    NT3_STAMP(1)
    // ---- M(t)
    if (ABL != 3) __builtin_amdgcn_s_setprio(1);
    if (ABL != 4) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::"v"(af[0]), "v"(bfr[0]), "v"(af[FM - 1]), "v"(bfr[FN - 1]));
    }
    if (ABL != 3) __builtin_amdgcn_s_setprio(0);
    if (PROF) asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc[FM - 1][FN - 1]));
    __builtin_amdgcn_sched_barrier(0);
    NT3_STAMP(2)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    NT3_STAMP(3)
  }
  if (half == 0) __builtin_amdgcn_s_barrier();        // match the second half's extra barrier
  __syncthreads();                                    // every wave is done with the ring: the epilogue reuses it
  if (PROF) {
    NTParams q = p;
    q.res_rows = nullptr;
    nt_epilogue_t<FM, FN, WTM, WTN>(q, acc, m0, n0, wm, wn, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t = __builtin_readcyclecounter();
    pt[5] = t - tk0;
    if (lane == 0) {
      long long* out = reinterpret_cast<long long*>(const_cast<int*>(p.res_rows)) + ((long)blockIdx.x * 8 + wave) * 10;
#pragma unroll
      for (int i = 0; i < 6; ++i) out[i] = pt[i];
      out[6] = tstart; out[7] = t;
      out[8] = __builtin_amdgcn_s_getreg(6164);
      out[9] = __builtin_amdgcn_s_getreg(63492);
    }
    return;
  }
  if (nt_staged_ok(p)) nt_epilogue_s<BM, BN, NT, FM, FN, WTM, WTN>(p, acc, smem, m0, n0, wm, wn, lane, tid);
  else nt_epilogue_t<FM, FN, WTM, WTN>(p, acc, m0, n0, wm, wn, lane);
#undef NT3_STAMP
}

template <int ABL>
__global__ __launch_bounds__(512) void gemm_nt3_prof_kernel(NTParams p) {
  nt3_body<false, true, ABL>(p, blockIdx.x);
}

template <bool BT>
__global__ __launch_bounds__(512) void gemm_nt3_kernel(NTParams p) {
  nt3_body<BT>(p, blockIdx.x);
}
// PIPE: two 512-thread workgroups per CU = 4 waves per SIMD -> at most 128 VGPRs (2nd launch-bounds argument = waves per SIMD)
template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK = 64, int PIPE = 0, bool LNK = false>
// (4-wave tiles ask for >= 2 waves per SIMD: with one, the register budget is 512 = VGPRs + AGPRs, the compiler puts the accumulators
// into AGPRs and copies all of them to VGPRs and back around every k-step — 32-64 v_accvgpr moves against 8-16 MFMAs)
__global__ __launch_bounds__((WM_* WN_ + (PIPE >= 2 ? PIPE : 0)) * 64, PIPE == 4 ? 6 : PIPE == 2 ? 5 : ((PIPE || (WM_ * WN_ == 8 && BM * BN > 128 * 128)) ? 4 : (WM_ * WN_ <= 4 ? 2 : 1))) void gemm_nt2_kernel(NTParams p) {
  nt2_body<BM, BN, WM_, WN_, STAGES, BT, BK, false, PIPE, LNK>(p, blockIdx.x);
}

// Grouped launch: up to NT_GROUP_MAX independent problems (any M / N / K / epilogue, one tile configuration) in ONE grid —
// e.g. the qkv GEMMs of the image and the audio tower of a layer (738 + 864 tiles = 3.1 rounds of 512 workgroup slots
// instead of 1.44 -> 2 and 1.69 -> 2).  Each problem's block range starts at a multiple of 8 so that blockIdx & 7 (the XCD
// a workgroup lands on) is the same in the local numbering the panel order is built on.
constexpr int NT_GROUP_MAX = 8;
struct NTGroup {
  NTParams prob[NT_GROUP_MAX];
  int first_block[NT_GROUP_MAX + 1];
  int count;
};

template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK = 64, int PIPE = 0, bool LNK = false>
__global__ __launch_bounds__((WM_* WN_ + (PIPE >= 2 ? PIPE : 0)) * 64, PIPE == 4 ? 6 : PIPE == 2 ? 5 : ((PIPE || (WM_ * WN_ == 8 && BM * BN > 128 * 128)) ? 4 : (WM_ * WN_ <= 4 ? 2 : 1))) void gemm_nt2_grouped_kernel(const NTGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  nt2_body<BM, BN, WM_, WN_, STAGES, BT, BK, false, PIPE, LNK>(g.prob[pi], (int)blockIdx.x - g.first_block[pi]);
}

template <bool BT, int EK>
__global__ __launch_bounds__(512) void gemm_nt256_grouped_kernel(const NTGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  nt256_body<BT, EK>(g.prob[pi], (int)blockIdx.x - g.first_block[pi]);
}

// 256 x 256 tiles (configuration 60): n >= 1 recorded problems (davb::GroupFn).  One kernel form (problem table by value, also
// for a single problem); the epilogue specialisation is the kind all problems of the launch share, else the generic one.
template <bool BT, int EK>
void nt256_launch(const NTGroup& g, int grid, hipStream_t stream) {
  static bool big = false;
  if (!big) {
    (void)hipFuncSetAttribute((const void*)gemm_nt256_grouped_kernel<BT, EK>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    big = true;
  }
  DAV_LAUNCH_NOW((gemm_nt256_grouped_kernel<BT, EK>), dim3(grid), dim3(512), NT256_LDS, stream, g);
}
template <bool BT>
void nt256_issue(const void* const* params, int n, hipStream_t stream) {
  for (int base = 0; base < n; base += NT_GROUP_MAX) {
    const int cnt = n - base < NT_GROUP_MAX ? n - base : NT_GROUP_MAX;
    NTGroup g;
    int first = 0, kind = -1;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const NTParams*)params[base + i];
      g.first_block[i] = first;
      const int tiles = ((g.prob[i].M + 255) / 256) * ((g.prob[i].N + 255) / 256);
      first += cnt == 1 ? tiles : (tiles + 7) & ~7;
      const int k = nt256_kind(g.prob[i]);
      kind = (kind < 0 || kind == k) ? k : 0;
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    switch (kind) {
      case 1: nt256_launch<BT, 1>(g, first, stream); break;
      case 2: nt256_launch<BT, 2>(g, first, stream); break;
      case 3: nt256_launch<BT, 3>(g, first, stream); break;
      default: nt256_launch<BT, 0>(g, first, stream); break;
    }
  }
}
template <bool BT>
void launch_nt256(const NTParams& p, hipStream_t stream) {
  if (davb::recording()) {
    davb::push_typed(nt256_issue<BT>, &p, sizeof(p), stream);
    return;
  }
  const void* one = &p;
  nt256_issue<BT>(&one, 1, stream);
}

template <bool BT>
__global__ __launch_bounds__(512) void gemm_nt3_grouped_kernel(const NTGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  nt3_body<BT>(g.prob[pi], (int)blockIdx.x - g.first_block[pi]);
}

// staggered-halves kernel: n >= 1 recorded problems (davb::GroupFn)
template <bool BT>
void nt3_issue(const void* const* params, int n, hipStream_t stream) {
  constexpr int BM = 256, BN = 128;
  constexpr size_t lds = 6 * (BM + BN) * 64;
  static bool big = false;
  if (!big) {
    (void)hipFuncSetAttribute((const void*)gemm_nt3_kernel<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gemm_nt3_grouped_kernel<BT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    big = true;
  }
  for (int base = 0; base < n; base += NT_GROUP_MAX) {
    const int cnt = n - base < NT_GROUP_MAX ? n - base : NT_GROUP_MAX;
    if (cnt == 1) {
      const NTParams& p = *(const NTParams*)params[base];
      const int grid = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
      DAV_LAUNCH_NOW(gemm_nt3_kernel<BT>, dim3(grid), dim3(512), lds, stream, p);
      continue;
    }
    NTGroup g;
    int first = 0;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const NTParams*)params[base + i];
      g.first_block[i] = first;
      first += (((g.prob[i].M + BM - 1) / BM) * ((g.prob[i].N + BN - 1) / BN) + 7) & ~7;
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    DAV_LAUNCH_NOW(gemm_nt3_grouped_kernel<BT>, dim3(first), dim3(512), lds, stream, g);
  }
}
template <bool BT>
void launch_nt3(const NTParams& p, hipStream_t stream) {
  if (davb::recording()) {
    davb::push_typed(nt3_issue<BT>, &p, sizeof(p), stream);
    return;
  }
  const void* one = &p;
  nt3_issue<BT>(&one, 1, stream);
}

// launches n >= 1 recorded problems of this tile configuration (davb::GroupFn)
static bool nt_has_ln(const void* const* params, int n) {
  for (int i = 0; i < n; ++i) {
    const NTParams& p = *(const NTParams*)params[i];
    if (p.ln_st || p.st_out || p.tw_out) return true;
  }
  return false;
}

template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT, int BK, int PIPE = 0, bool LNK = false>
void nt2_issue(const void* const* params, int n, hipStream_t stream) {
  if constexpr (!LNK && !BT && PIPE == 0) {      // a problem with a LayerNorm folded in: the kernels that carry that code
    if (nt_has_ln(params, n)) { nt2_issue<BM, BN, WM_, WN_, STAGES, BT, BK, PIPE, true>(params, n, stream); return; }
  }
  constexpr int NT = (WM_ * WN_ + (PIPE >= 2 ? PIPE : 0)) * 64;      // + the loader waves
  constexpr size_t lds = nt2_lds_alloc<BM, BN, WM_, WN_, STAGES, BT, BK, LNK>();
  auto kern = gemm_nt2_kernel<BM, BN, WM_, WN_, STAGES, BT, BK, PIPE, LNK>;
  auto gkern = gemm_nt2_grouped_kernel<BM, BN, WM_, WN_, STAGES, BT, BK, PIPE, LNK>;
  static bool big = false;
  if (lds > 64 * 1024 && !big) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)gkern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    big = true;
  }
  for (int base = 0; base < n; base += NT_GROUP_MAX) {
    const int cnt = n - base < NT_GROUP_MAX ? n - base : NT_GROUP_MAX;
    if (cnt == 1) {
      const NTParams& p = *(const NTParams*)params[base];
      const int grid = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
      DAV_LAUNCH_NOW(kern, dim3(grid), dim3(NT), lds, stream, p);
      continue;
    }
    NTGroup g;
    int first = 0;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const NTParams*)params[base + i];
      g.first_block[i] = first;
      first += (((g.prob[i].M + BM - 1) / BM) * ((g.prob[i].N + BN - 1) / BN) + 7) & ~7;
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    DAV_LAUNCH_NOW(gkern, dim3(first), dim3(NT), lds, stream, g);
  }
}

template <int BM, int BN, int WM_, int WN_, int STAGES, bool BT = false, int BK = 64, int PIPE = 0>
void launch_nt2(const NTParams& p, hipStream_t stream) {
  if (davb::recording()) {
    davb::push_typed(nt2_issue<BM, BN, WM_, WN_, STAGES, BT, BK, PIPE>, &p, sizeof(p), stream);
    return;
  }
  const void* one = &p;
  nt2_issue<BM, BN, WM_, WN_, STAGES, BT, BK, PIPE>(&one, 1, stream);
}

int nt_auto_config_tiles(long t128, bool narrow);

// Issue log (bench.py's roofline leg): which grouped / single launches the NT family really issued, so that the launch mix
// of a step can be replayed kernel for kernel.  Entry: cfg, b_kn, n, then n x (M, N, K, epilogue flags).
// process-wide (autograd runs the backward on its own thread), guarded by a mutex
std::atomic<bool> nt_log_on{false};
std::vector<int> nt_log;
std::mutex nt_log_mu;
// epilogue kind of a problem (issue log, tuned-table signature)
static int nt_flags(const NTParams& p) { return p.act | (p.c_bf16 ? 4 : 0) | (p.res ? 8 : 0) | (p.c2_mode << 4) | (p.beta ? 256 : 0); }
void nt_log_issue(int cfg, bool bt, const void* const* params, int n) {
  if (!nt_log_on.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(nt_log_mu);
  nt_log.push_back(cfg); nt_log.push_back(bt ? 1 : 0); nt_log.push_back(n);
  for (int i = 0; i < n; ++i) {
    const NTParams& p = *(const NTParams*)params[i];
    nt_log.push_back(p.M); nt_log.push_back(p.N); nt_log.push_back(p.K);
    nt_log.push_back(nt_flags(p));   // epilogue kind
  }
}

// Tuned table: group signature -> tile configuration, measured offline on the target by tools/mix_sweep.py (every distinct
// grouped launch of a training step timed under each candidate configuration) and handed over once by the host
// (dav_nt_tune_set).  Signature = b_kn, then the problems' (M, N, K, epilogue flags) sorted.  The rule-based choice below
// remains the fallback for every group the table does not hold.
std::map<std::vector<int>, int> nt_tuned;
std::mutex nt_tuned_mu;
static int nt_tuned_lookup(bool bt, const void* const* params, int n) {
  if (nt_tuned.empty()) return 0;
  std::vector<std::array<int, 4>> v(n);
  for (int i = 0; i < n; ++i) {
    const NTParams& p = *(const NTParams*)params[i];
    v[i] = {p.M, p.N, p.K, nt_flags(p)};
  }
  std::sort(v.begin(), v.end());
  std::vector<int> key;
  key.reserve(1 + 4 * n);
  key.push_back(bt ? 1 : 0);
  for (auto& q : v) key.insert(key.end(), q.begin(), q.end());
  std::lock_guard<std::mutex> lk(nt_tuned_mu);
  auto it = nt_tuned.find(key);
  return it == nt_tuned.end() ? 0 : it->second;
}

// 256 x 256 tiles (configuration 60): an explicit / tunable configuration only.  Alone on the GPU it is 7-25 % faster than 128 x 128 /
// 128 x 256 on every wide shape of the step, but as a rule it made the step 0.25 ms SLOWER in every schedule (round 3, same-box
// alternation, profiles/r03_nt256_instep_ab.txt): a workgroup that owns 128 KB of LDS keeps the other streams' kernels off its CU.

static bool nt_ln_producer_cfg(int cfg) { return cfg == 3 || cfg == 5 || cfg == 7 || cfg == 8; }

// Recorded with the tile configuration left open: chosen at issue time from the tile count of the WHOLE group.
template <bool BT>
void nt2_issue_auto(const void* const* params_in, int n, hipStream_t stream) {
  // longest k-loops first: workgroups are dispatched in block order as slots free up, so the short tiles fill the tail
  std::vector<const void*> sorted(params_in, params_in + n);
  std::stable_sort(sorted.begin(), sorted.end(), [](const void* a, const void* b) { return ((const NTParams*)a)->K > ((const NTParams*)b)->K; });
  const void* const* params = sorted.data();
  long t128 = 0, t256 = 0;
  bool wide = true;
  bool narrow = true;
  bool all256 = true;                                     // every problem can go through the 256 x 256 body and is wide enough for it
  bool ln_prod = false;                                   // a problem of the group writes row statistics / a twin (DavNtLn producer)
  for (int i = 0; i < n; ++i) {
    const NTParams& p = *(const NTParams*)params[i];
    ln_prod = ln_prod || p.st_out || p.tw_out;
    all256 = all256 && nt256_ok(p);
    t128 += (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    t256 += (long)((p.M + 127) / 128) * ((p.N + 255) / 256);
    narrow = narrow && p.N <= 64;
    wide = wide && p.K <= 512 && !(p.N & 255) && p.N >= 1024;     // (512-wide outputs, the decoder proj: 5 % slower with it)
  }
  int cfg = nt_auto_config_tiles(t128, narrow);
  // short contractions into wide outputs (the decoders' qkv / fc1, K = 512): a 128 x 256 tile on a 3 x 24 KB ring of 32-deep
  // stages moves 3/4 of the operand bytes per flop and spends half as many tile prologues / epilogues: +14 % on both
  // (tools/group_bench.py cfg 44 vs 3); longer contractions and 768-wide outputs measured equal or worse.
  int forced = ((const NTParams*)params[0])->force_cfg;          // an explicit configuration shared by the whole group wins
  for (int i = 1; i < n; ++i) if (((const NTParams*)params[i])->force_cfg != forced) forced = 0;
  const int tuned = forced ? 0 : nt_tuned_lookup(BT, params, n);
  if (forced) cfg = forced;
  else if (tuned) cfg = tuned;
  else if (cfg == 3 && wide && t256 >= 512) cfg = 44;
#ifdef DAV_EXPERIMENTAL
  // (A/B of round 5, profiles/r05_experiments.txt: configurations 31 — software-pipelined k-loop — and 51 — loader waves — in place of 3
  // are 6-26 % faster alone and +0.6 / +1.9 ms in the step; DAV_NT_ALT=31 / 51 reproduces it in an EXPERIMENTAL build)
  static const int nt_alt = getenv("DAV_NT_ALT") ? atoi(getenv("DAV_NT_ALT")) : 0;
  if (cfg == 3 && (nt_alt == 31 || nt_alt == 51)) cfg = nt_alt;
#endif
  if (cfg == 60 && !all256) cfg = 3;                     // (an explicit or tuned 60 on a group the 256 x 256 body cannot take)
  if (ln_prod && !nt_ln_producer_cfg(cfg)) cfg = 3;      // row statistics come out of the 32 x 64 / 32 x 32 wave tiles only
  nt_log_issue(cfg, BT, params, n);
  switch (cfg) {
    case 60: nt256_issue<BT>(params, n, stream); break;
    case 44: nt2_issue<128, 256, 2, 4, 3, BT, 32>(params, n, stream); break;
    case 45: nt2_issue<256, 128, 4, 2, 2, BT, 32>(params, n, stream); break;
    case 43: nt2_issue<256, 128, 4, 2, 3, BT, 32>(params, n, stream); break;      // (43 / 46: tools/mix_sweep.py candidates)
    case 46: nt2_issue<128, 256, 2, 4, 2, BT, 32>(params, n, stream); break;
#ifdef DAV_EXPERIMENTAL
    case 31: nt2_issue<128, 128, 2, 4, 2, BT, 64, 1>(params, n, stream); break;      // software-pipelined k-loop
    case 50: nt2_issue<128, 128, 4, 2, 2, BT, 64, 2>(params, n, stream); break;      // + two loader waves
    case 51: nt2_issue<128, 128, 4, 2, 2, BT, 64, 4>(params, n, stream); break;      // + four
#endif
    case 3: nt2_issue<128, 128, 4, 2, 2, BT, 64>(params, n, stream); break;      // 4 x 2 waves (32 x 64 wave tiles): +0.8 % over 2 x 4 in the step
    case 8: nt2_issue<128, 64, 2, 2, 2, BT, 64>(params, n, stream); break;
    case 7: nt2_issue<64, 64, 2, 2, 4, BT, 64>(params, n, stream); break;
    default: nt2_issue<64, 64, 2, 2, 2, BT, 64>(params, n, stream); break;
  }
}

// ------------------------------------------------------------------------------------------------
// TN kernel (weight gradient).  LDS tiles are [64 contraction rows][128 cols] bf16 (256 B rows);
// MFMA fragments want 8 consecutive contraction rows per lane -> ds_read_b64_tr_b16 (hardware
// 4x4 transpose), two per fragment.  32-byte granules of a row are XOR-swizzled by
// f(row) = (row & 3) | ((row >> 3) & 1) << 2 so the 8 rows a 32-lane half touches hit distinct banks.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tn_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <bool TR>
__device__ __forceinline__ bf16x8 tn_frag(const char* tile, int mk, int colbase, int lane) {
  const int g = lane >> 4, li = lane & 15;
  if (TR) {
    union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = mk + 8 * g + 4 * h + (li >> 2);
      const int colb = (colbase + 4 * (li & 3)) * 2;
      const int addr = row * 256 + ((((colb >> 5) ^ tn_swz(row)) << 5) | (colb & 31));
      u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + addr));
    }
    return u.v;
  } else {
    union { bf16_t e[8]; bf16x8 v; } u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int row = mk + 8 * g + e;
      const int colb = (colbase + li) * 2;
      const int addr = row * 256 + ((((colb >> 5) ^ tn_swz(row)) << 5) | (colb & 31));
      u.e[e] = *reinterpret_cast<const bf16_t*>(tile + addr);
    }
    return u.v;
  }
}

template <bool TR>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(TNParams p) {
  constexpr int TILE_BYTES = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                    // [2][64][256B]
  char* Bs = smem + 2 * TILE_BYTES;   // [2][64][256B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = (p.K + 127) / 128;
  const int bn = blockIdx.x / tiles_k, bk = blockIdx.x % tiles_k;
  const int n0 = bn * 128, k0 = bk * 128;

  // contraction range of this split (multiples of 64 rows)
  const int steps_total = (p.Mc + 63) >> 6;
  const int steps_per = (steps_total + p.splits - 1) / p.splits;
  const int s_begin = blockIdx.y * steps_per;
  int s_end = s_begin + steps_per; s_end = s_end < steps_total ? s_end : steps_total;
  if (s_begin >= s_end) return;

  const int c16 = tid & 15, rbase = tid >> 4;   // chunk column (8 elems), rows rbase + 16*i
  const bool a_ok = n0 + c16 * 8 < p.N, b_ok = k0 + c16 * 8 < p.K;
  uint4 ra[4], rb[4];
  auto load_regs = [&](int s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = (s << 6) + rbase + 16 * i;
      const bool ok = m < p.Mc;
      ra[i] = (ok && a_ok) ? *reinterpret_cast<const uint4*>(p.A + map_row(m, p.amap) * p.lda + n0 + c16 * 8) : uint4{0, 0, 0, 0};
      rb[i] = (ok && b_ok) ? *reinterpret_cast<const uint4*>(p.B + map_row(m, p.bmap) * p.ldb + k0 + c16 * 8) : uint4{0, 0, 0, 0};
    }
  };
  auto store_regs = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rbase + 16 * i;
      const int addr = row * 256 + ((((c16 >> 1) ^ tn_swz(row)) << 5) | ((c16 & 1) << 4));
      *reinterpret_cast<uint4*>(As + buf * TILE_BYTES + addr) = ra[i];
      *reinterpret_cast<uint4*>(Bs + buf * TILE_BYTES + addr) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.bias_grad != nullptr && bk == 0;

  load_regs(s_begin);
  store_regs(0);
  __syncthreads();
  for (int s = s_begin; s < s_end; ++s) {
    const int buf = (s - s_begin) & 1;
    const bool more = s + 1 < s_end;
    if (do_bias) {   // column sums of the A tile this thread staged (its 4 rows x 8 columns)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t w[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bsum[2 * e] += __uint_as_float(w[e] << 16);
          bsum[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
      }
    }
    if (more) load_regs(s + 1);
    const char* Ab = As + buf * TILE_BYTES;
    const char* Bb = Bs + buf * TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tn_frag<TR>(Ab, kk * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tn_frag<TR>(Bb, kk * 32, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_regs(buf ^ 1);
    __syncthreads();
  }

  const int fr = lane & 15, fg = lane >> 4;
  const bool atomic = p.splits > 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + wm * 64 + i * 16 + fg * 4 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wn * 64 + j * 16 + fr;
        if (k >= p.K) continue;
        float* c = p.C + (long)n * p.ldc + k;
        if (atomic) unsafeAtomicAdd(c, acc[i][j][r]);
        else if (p.beta) *c += acc[i][j][r];
        else *c = acc[i][j][r];
      }
    }

  if (do_bias) {
    // reduce the 16 row-groups (threads with equal c16) through LDS, then one atomic per column
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [16 rowgroups][128 cols]
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rbase * 128 + c16 * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) s += red[g * 128 + tid];
      unsafeAtomicAdd(p.bias_grad + n0 + tid, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// TN kernel, second generation (contraction rows Mc % 64 == 0): 2-stage LDS ring filled by
// global->LDS DMA (source-side swizzle), T x T output tiles (T = 128 with 8 waves, T = 64 with 4),
// tr-read fragments, bias gradient as one extra MFMA against a fragment of ones, split-K with fp32
// atomics into the live gradient.
// ------------------------------------------------------------------------------------------------
// T x TK output tiles (T rows of the gradient = columns of A, TK columns = columns of B; TK = T unless given: round 4 adds 256 x 128)
template <int T, int WM_, int WN_, int RS = 64, int NST = 2, int TK = T>
__device__ __forceinline__ void tn2_body(const TNParams& p, const int tile_idx, const int split_idx) {
  // RS contraction rows per ring stage, NST stages (NST - 1 stages of loads in flight while one is consumed)
  constexpr int NT = WM_ * WN_ * 64, RB = T * 2, CPRW = T / 8;      // 16-byte chunks per tile row (A side)
  constexpr int RBK = TK * 2, CPRWK = TK / 8;                       // ... (B side)
  constexpr int TILE_BYTES = RS * RB, TILE_BYTES_B = RS * RBK, STAGE_BYTES = TILE_BYTES + TILE_BYTES_B;
  constexpr int CH = RS * CPRW / NT, CHB = RS * CPRWK / NT;         // chunks per thread per operand
  constexpr int WTN = T / WM_, WTK = TK / WN_, FM = WTN / 16, FN = WTK / 16;
  static_assert(RS * CPRW % NT == 0 && RS * CPRWK % NT == 0 && (RS == 32 || RS == 64) && NST >= 2, "tile/threads mismatch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int dbg = p.debug_plain_store & 255;                        // (a register: read through p inside the loop it was a scalar load + wait per stage)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int tiles_k = (p.K + TK - 1) / TK, tiles_n = (p.N + T - 1) / T;
  // consecutive tile ids walk the SHORTER side of the tile grid first: a run of ids (what one XCD owns, see the grouped kernel)
  // then covers a few whole rows / columns of it and touches few distinct operand panels
  const bool k_minor = tiles_k <= tiles_n;
  const int bn = k_minor ? tile_idx / tiles_k : tile_idx % tiles_n, bk = k_minor ? tile_idx % tiles_k : tile_idx / tiles_n;
  const int n0 = bn * T, k0 = bk * TK;
  const int steps_total = p.Mc >> 6;                                // the split of the contraction is in 64-row units
  const int steps_per = (steps_total + p.splits - 1) / p.splits;
  int s_begin = split_idx * steps_per;
  int s_end = s_begin + steps_per; s_end = s_end < steps_total ? s_end : steps_total;
  if (s_begin >= s_end) return;
  s_begin *= 64 / RS; s_end *= 64 / RS;                             // ... the ring's steps in RS-row units

  // Source pointers of this thread's chunks, carried from stage to stage (the stages are issued in contraction order): one 64-bit
  // add per chunk and stage.  Re-deriving them per stage (row map division, two 32-bit multiplies and a 64-bit multiply-add per
  // chunk, all quarter rate) was ~100 VALU cycles per chunk against the 256 MFMA cycles of a wave's whole stage.
  const bf16_t* ap[CH]; const bf16_t* bp[CHB];
  int ar[CH], br[CHB];                                     // row index inside the row map's period
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int c = tid + NT * i, row = c / CPRW, pc = c % CPRW;
    const int lc = (((pc >> 1) ^ tn2_swz<RB>(row)) << 1) | (pc & 1);
    int ac = n0 + lc * 8; ac = ac < p.N ? ac : p.N - 8;     // out-of-range output columns: any valid data (discarded)
    const int m = s_begin * RS + row;
    ap[i] = p.A + map_row(m, p.amap) * p.lda + ac;
    ar[i] = p.amap.rpb > 0 ? m % p.amap.rpb : 0;
  }
#pragma unroll
  for (int i = 0; i < CHB; ++i) {
    const int c = tid + NT * i, row = c / CPRWK, pc = c % CPRWK;
    const int lc = (((pc >> 1) ^ tn2_swz<RBK>(row)) << 1) | (pc & 1);
    int bc = k0 + lc * 8; bc = bc < p.K ? bc : p.K - 8;
    const int m = s_begin * RS + row;
    bp[i] = p.B + map_row(m, p.bmap) * p.ldb + bc;
    br[i] = p.bmap.rpb > 0 ? m % p.bmap.rpb : 0;
  }
  const long a_step = (long)RS * p.lda, b_step = (long)RS * p.ldb;
  const long a_wrap = (long)(p.amap.bs - p.amap.rpb) * p.lda, b_wrap = (long)(p.bmap.bs - p.bmap.rpb) * p.ldb;
  auto dma_tile = [&](int stage) {                         // the next RS contraction rows
    char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, ap[i]), LDS_PTR(void, st + (wave * 64 + NT * i) * 16), 16, 0, 0);
      ap[i] += a_step;
    }
#pragma unroll
    for (int i = 0; i < CHB; ++i) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, bp[i]), LDS_PTR(void, st + TILE_BYTES + (wave * 64 + NT * i) * 16), 16, 0, 0);
      bp[i] += b_step;
    }
    if (p.amap.rpb > 0) {                                  // (uniform) a mapped operand: period boundaries crossed by this advance
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        ar[i] += RS;
        while (ar[i] >= p.amap.rpb) { ar[i] -= p.amap.rpb; ap[i] += a_wrap; }
      }
    }
    if (p.bmap.rpb > 0) {
#pragma unroll
      for (int i = 0; i < CHB; ++i) {
        br[i] += RS;
        while (br[i] >= p.bmap.rpb) { br[i] -= p.bmap.rpb; bp[i] += b_wrap; }
      }
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  uint32_t abase[FM], bbase[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) abase[i] = lds0 + tn2_frag_base<RB>(wm * WTN + i * 16, lane);
#pragma unroll
  for (int j = 0; j < FN; ++j) bbase[j] = lds0 + tn2_frag_base<RBK>(wn * WTK + j * 16, lane);
  f32x4 acc[FM][FN], accb[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = p.bias_grad != nullptr && bk == 0 && wn == 0;
  union { uint32_t w[4]; bf16x8 v; } ones;
  ones.w[0] = ones.w[1] = ones.w[2] = ones.w[3] = 0x3f803f80u;

#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (s_begin + st < s_end) dma_tile(st);
  int stage = 0;
  for (int s = s_begin; s < s_end; ++s) {
    // the loads of step s must have landed; those of the NST - 2 steps after it may still be in flight (each step is 2 CH loads
    // per wave) — once the tail stops issuing, drain everything
    if (NST > 2 && s + NST - 2 < s_end) wait_vmcnt<(NST - 2) * (CH + CHB)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (s + NST - 1 < s_end) dma_tile(stage == 0 ? NST - 1 : stage - 1);      // the stage step s - 1 has just released
    const uint32_t so = (uint32_t)(stage * STAGE_BYTES);
    stage = stage + 1 == NST ? 0 : stage + 1;
    if (dbg & 4) continue;      // timing experiment (DAV_TN_DEBUG): the global -> LDS stream alone
    // every fragment of the stage is requested up front (asm reads, see tn2_tr: the DMA issued above stays in flight), then each
    // 32-row half is multiplied as soon as ITS reads are back (LDS returns in order: a counted lgkmcnt)
    constexpr int KK = RS / 32, RPK = 2 * (FM + FN);          // ds_read instructions per 32-row half
    bf16x8 af[KK][FM], bfr[KK][FN];
    auto rd = [&](auto kc) {
      constexpr int kk = decltype(kc)::value;
#pragma unroll
      for (int i = 0; i < FM; ++i) af[kk][i] = tn2_tr<kk * 32 * RB, RB>(abase[i] + so);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[kk][j] = tn2_tr<TILE_BYTES + kk * 32 * RBK, RBK>(bbase[j] + so);
    };
    auto mm = [&](auto kc) {
      constexpr int kk = decltype(kc)::value;
      wait_lgkmcnt<(KK - 1 - kk) * RPK>();
      __builtin_amdgcn_sched_barrier(0);
      // operands swapped (B fragment first): the 16 x 16 block comes out TRANSPOSED — lane (fr, fg) holds C[n = fr][k = 4 fg .. + 3],
      // four consecutive columns of one row — so the read-modify-write of the gradient tile is one 16-byte access per lane and
      // fragment instead of four scalar ones (as the NT kernels' epilogue)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < FM; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], ones.v, accb[i], 0, 0, 0);
      }
    };
    static_assert((KK - 1) * RPK <= 15 || KK == 1, "lgkmcnt field");
    rd(IntC<0>{});
    if constexpr (KK == 2) rd(IntC<1>{});
    __builtin_amdgcn_sched_barrier(0);
    mm(IntC<0>{});
    if constexpr (KK == 2) mm(IntC<1>{});
    __builtin_amdgcn_sched_barrier(0);
  }

  if (dbg & 2) return;          // timing experiment: no epilogue
  const int fr = lane & 15, fg = lane >> 4;
  {
  // splits > 1: partial sums meet through fp32 atomics; splits == 1: this workgroup owns the tile (plain
  // read-modify-write when accumulating)
  const bool atomic = p.splits > 1 && !(dbg & 1);
  if (atomic) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int n = n0 + wm * WTN + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int k = k0 + wn * WTK + j * 16 + fg * 4;
        if (n >= p.N || k >= p.K) continue;               // K is a multiple of 8: the four columns are in or out together
        float* c = p.C + (long)n * p.ldc + k;
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c + r, acc[i][j][r]);
      }
    }
  } else {
    // the tile's old values are fetched in ONE batch (FM * FN independent 16-byte loads in flight) before the first store: a
    // load -> add -> store chain per fragment is one exposed memory latency each (8 per wave, 25 % of the kernel's time)
    float4 old[FM][FN];
    if (p.beta) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int n = n0 + wm * WTN + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int k = k0 + wn * WTK + j * 16 + fg * 4;
          old[i][j] = (n < p.N && k < p.K) ? *reinterpret_cast<const float4*>(p.C + (long)n * p.ldc + k) : float4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int n = n0 + wm * WTN + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int k = k0 + wn * WTK + j * 16 + fg * 4;
        if (n >= p.N || k >= p.K) continue;
        float4 v = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.beta) { v.x += old[i][j].x; v.y += old[i][j].y; v.z += old[i][j].z; v.w += old[i][j].w; }
        *reinterpret_cast<float4*>(p.C + (long)n * p.ldc + k) = v;
      }
    }
  }
  }
  if (do_bias && fr == 0) {                               // the bias-gradient accumulators keep the untransposed layout; one
#pragma unroll                                            // contribution per element and launch: a no-return atomic costs no latency
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nb = n0 + wm * WTN + i * 16 + fg * 4 + r;
        if (nb < p.N) unsafeAtomicAdd(p.bias_grad + nb, accb[i][r]);
      }
  }
}

template <int T, int WM_, int WN_>
__global__ __launch_bounds__(WM_* WN_ * 64, WM_ * WN_ <= 4 ? 2 : 1) void gemm_tn2_kernel(TNParams p) {
  tn2_body<T, WM_, WN_>(p, blockIdx.x, blockIdx.y);
}

// grouped launch: up to TN_GROUP_MAX independent weight-gradient problems in ONE grid (the deferred wgrads of a
// whole layer), each workgroup looks its (problem, tile, split) up from the by-value table
constexpr int TN_GROUP_MAX = 40;      // (the table travels by value in the kernel arguments: 40 x 96 bytes + bookkeeping stays below the 4 KB they may take)
struct TNGroup {
  TNParams prob[TN_GROUP_MAX];
  int first_block[TN_GROUP_MAX + 1];
  int count;                          // bits 0..15: problems; bit 30 (TN_XCD_RUNS): every XCD owns one contiguous run of each problem's (split, tile) units
};
constexpr int TN_XCD_RUNS = 1 << 30;
static_assert(sizeof(TNGroup) <= 4096, "kernel argument block");

template <int T, int WM_, int WN_, int RS = 64, int NST = 2, int MINW = 1, int TK = T>
__global__ __launch_bounds__(WM_* WN_ * 64, MINW) void gemm_tn_grouped_kernel(const TNGroup g) {
  int pi = 0;
  const int count = g.count & 0xffff;
  while (pi + 1 < count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  const TNParams& p = g.prob[pi];
  const int local = blockIdx.x - g.first_block[pi];
  const int tiles = ((p.N + T - 1) / T) * ((p.K + TK - 1) / TK);
  int unit = local;
  if (g.count & TN_XCD_RUNS) {
    // workgroup ids go round the 8 XCDs (private L2s): the ids with the same residue form one XCD's share of this problem.  Give
    // that share a contiguous run of units — a few whole rows / columns of the tile grid, streaming the contraction side by
    // side — so that the XCD pulls few distinct operand panels through the fabric instead of nearly all of them
    // (profiles/r03_step_traffic.txt: the weight gradients fetched 33 GB per step, 6 x their operands)
    const int total = g.first_block[pi + 1] - g.first_block[pi];
    const int r = local & 7, j = local >> 3, q = total >> 3, rem = total & 7;
    unit = r * q + (r < rem ? r : rem) + j;
  }
  tn2_body<T, WM_, WN_, RS, NST, TK>(p, unit % tiles, unit / tiles);
}

template <int T, int WM_, int WN_>
void launch_tn2(const TNParams& p, int tiles, hipStream_t stream) {
  DAV_LAUNCH((gemm_tn2_kernel<T, WM_, WN_>), dim3(tiles, p.splits), dim3(WM_ * WN_ * 64), (size_t)2 * 2 * 64 * T * 2, stream, p);
}

RowMap mk(const int* m) { return m ? RowMap{m[0], m[1], m[2]} : RowMap{0, 0, 0}; }

// tile configuration heuristic for the second-generation NT kernel (see launch_nt2 cases)
int nt_auto_config_tiles(long t128, bool narrow) {
  // measured on MI355X over the ViT-B step's shapes (tools/gemm_bench.py): occupancy beats pipeline depth,
  // so 2-stage rings everywhere; 8 waves on 128x128 when there are enough tiles to fill 2 blocks per CU.
  // 64x64, 4 waves.  A few dozen tiles alone on the GPU (the fusion block's projections: weights last touched a step ago)
  // get the FOUR-stage ring — three k-steps of prefetch cover the HBM miss a two-stage ring exposes at every k-step (15 us in
  // the step vs 7 us with warm operands); from ~250 tiles of 64x64 on the 64 KB ring costs occupancy instead (3136x768x3072
  // alone: 40.5 vs 32.7 us).
  constexpr int small_cfg = 5, t5 = 100, t8 = 400;      // (round 3, stream schedule: the two-stage ring for the small launches is -0.18 ms per step; 7 = the four-stage ring)
  if (narrow || t128 < t5) return (small_cfg == 5 || t128 > 64) ? 5 : 7;
  if (t128 < t8) return 8;                   // 128x64, 4 waves
  return 3;                                   // 128x128, 8 waves (2 x 4)
}
int nt_auto_config(int M, int N, int K) {
  (void)K;
  return nt_auto_config_tiles((long)((M + 127) / 128) * ((N + 127) / 128), N <= 64);
}

}  // namespace

static int nt_entry(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                    const int* a_rowmap, const float* bias, int act, const void* aux, int ldaux,
                    const float* res, int ldres, const int* res_rowmap, const int* res_rows,
                    void* C, int ldc, int c_is_bf16, const int* c_rowmap, void* C2, int ldc2, int c2_mode,
                    int beta, float alpha, int variant, const DavNtLn* ln, hipStream_t stream);

extern "C" int dav_gemm_nt_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                                const int* a_rowmap, const float* bias, int act, const void* aux, int ldaux,
                                const float* res, int ldres, const int* res_rowmap, const int* res_rows,
                                void* C, int ldc, int c_is_bf16, const int* c_rowmap, void* C2, int ldc2, int c2_mode,
                                int beta, float alpha, int variant, hipStream_t stream) {
  return nt_entry(A, B, M, N, K, lda, ldb, a_rowmap, bias, act, aux, ldaux, res, ldres, res_rowmap, res_rows, C, ldc, c_is_bf16, c_rowmap,
                  C2, ldc2, c2_mode, beta, alpha, variant, nullptr, stream);
}

extern "C" int dav_gemm_nt_ln_bf16(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                                   const int* a_rowmap, const float* bias, int act, const void* aux, int ldaux,
                                   const float* res, int ldres, const int* res_rowmap, const int* res_rows,
                                   void* C, int ldc, int c_is_bf16, const int* c_rowmap, void* C2, int ldc2, int c2_mode,
                                   int beta, float alpha, int variant, const DavNtLn* ln, hipStream_t stream) {
  if (!ln) return DAV_ERR_SHAPE;
  return nt_entry(A, B, M, N, K, lda, ldb, a_rowmap, bias, act, aux, ldaux, res, ldres, res_rowmap, res_rows, C, ldc, c_is_bf16, c_rowmap,
                  C2, ldc2, c2_mode, beta, alpha, variant, ln, stream);
}

static int nt_entry(const void* A, const void* B, int M, int N, int K, int lda, int ldb,
                    const int* a_rowmap, const float* bias, int act, const void* aux, int ldaux,
                    const float* res, int ldres, const int* res_rowmap, const int* res_rows,
                    void* C, int ldc, int c_is_bf16, const int* c_rowmap, void* C2, int ldc2, int c2_mode,
                    int beta, float alpha, int variant, const DavNtLn* ln, hipStream_t stream) {
  const int b_kn = (variant >> 12) & 1;   // bit 12: B is [K, N] row-major (dgrad reading W itself)
  variant &= 0xfff;
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (lda & 7) || (ldb & 7)) return DAV_ERR_SHAPE;
  if (b_kn && ((N & 7) || (K & 63))) return DAV_ERR_SHAPE;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return DAV_ERR_ALIGN;
  if (beta && c_is_bf16) return DAV_ERR_DTYPE;
  if (!C && !C2) return DAV_ERR_SHAPE;
  if (C2 && (c2_mode < 1 || c2_mode > 4)) return DAV_ERR_SHAPE;
  if (C2 && c2_mode == 4 && act != 1) return DAV_ERR_SHAPE;
  if ((act == 2 || act == 3) && !aux) return DAV_ERR_SHAPE;
  NTParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
  p.amap = mk(a_rowmap); p.bias = bias; p.act = act; p.aux = (const bf16_t*)aux; p.ldaux = ldaux;
  p.res = res; p.ldres = ldres; p.rmap = mk(res_rowmap); p.res_rows = res_rows;
  p.C = C; p.ldc = ldc; p.c_bf16 = c_is_bf16; p.cmap = mk(c_rowmap); p.C2 = (bf16_t*)C2; p.ldc2 = ldc2; p.c2_mode = C2 ? c2_mode : 0;
  p.beta = beta; p.alpha = alpha; p.b_kn = b_kn;
  static const int nt_debug = getenv("DAV_NT_DEBUG") ? atoi(getenv("DAV_NT_DEBUG")) : 0;
  p.debug = nt_debug;
  p.ln_st = p.ln_st2 = nullptr; p.A2 = nullptr; p.ln_c = nullptr; p.ln_eps = 0.f; p.a_r0 = p.a_r1 = 0;
  p.st_out = nullptr; p.tw_out = nullptr; p.ldtw = 0;
  const bool glds_ok = (K & 63) == 0;
  const bool vec_ok = glds_ok && (N & 3) == 0 && (ldc & 3) == 0 && (!res || (ldres & 3) == 0) && (!C2 || (ldc2 & 3) == 0) &&
                      (!aux || (ldaux & 3) == 0) && !(((uintptr_t)C | (uintptr_t)C2 | (uintptr_t)res) & 15) && !((uintptr_t)aux & 7);
  int cfg = variant >> 4;
  const bool ln_use = ln && (ln->stats || ln->stats_out || ln->twin_out);
  if (ln_use) {
    // LayerNorm folded in: the second-generation bodies with 128-wide (or narrower) wave grids only, tile choice left open
    if (!vec_ok || b_kn || (variant & 15) || !(cfg == 0 || cfg == 3 || cfg == 5 || cfg == 7 || cfg == 8 || cfg == 43 || cfg == 44 || cfg == 45 || cfg == 46)) return DAV_ERR_SHAPE;
    if (ln->stats) {
      if (!ln->ln_c || K > 1024 || alpha != 1.0f || !(ln->eps > 0.f) || (((uintptr_t)ln->stats | (uintptr_t)ln->ln_c) & 15)) return DAV_ERR_SHAPE;
      if (ln->a_r0 < 0 || ln->a_r1 < 0) return DAV_ERR_SHAPE;
      if (ln->a_r0 > 0) {
        if (ln->a_r1 <= 0 || !ln->A2 || !ln->stats2 || a_rowmap || M % (ln->a_r0 + ln->a_r1)) return DAV_ERR_SHAPE;
        if (((uintptr_t)ln->A2 & 15) || ((uintptr_t)ln->stats2 & 7)) return DAV_ERR_ALIGN;
      }
      p.ln_st = ln->stats; p.ln_st2 = ln->stats2; p.A2 = (const bf16_t*)ln->A2; p.ln_c = ln->ln_c; p.ln_eps = ln->eps;
      p.a_r0 = ln->a_r0; p.a_r1 = ln->a_r0 > 0 ? ln->a_r1 : 0;
    } else if (ln->a_r0 || ln->A2) {
      return DAV_ERR_SHAPE;
    }
    if (ln->stats_out || ln->twin_out) {
      if (!C || c_is_bf16 || (N & 63)) return DAV_ERR_SHAPE;                 // statistics of the fp32 result, whole 64-column slots
      if (cfg != 0 && !nt_ln_producer_cfg(cfg)) return DAV_ERR_SHAPE;
      if (ln->twin_out && ((ln->ld_twin & 7) || ln->ld_twin < N)) return DAV_ERR_SHAPE;
      if (((uintptr_t)ln->stats_out & 7) || ((uintptr_t)ln->twin_out & 15)) return DAV_ERR_ALIGN;
      p.st_out = ln->stats_out; p.tw_out = (bf16_t*)ln->twin_out; p.ldtw = ln->ld_twin;
    }
  }
  const bool groupable_cfg = cfg == 0 || cfg == 60 || cfg == 3 || cfg == 5 || cfg == 7 || cfg == 8 || cfg == 50 || cfg == 51 || cfg == 43 || cfg == 44 || cfg == 45 || cfg == 46;
  p.force_cfg = cfg;
  if (vec_ok && groupable_cfg && !(variant & 15) && davb::recording()) {
    // batched: the tile configuration is chosen when the group is issued, from the tile count of the whole group
    davb::push_typed(b_kn ? nt2_issue_auto<true> : nt2_issue_auto<false>, &p, sizeof(p), stream);
    return DAV_OK;
  }
  if (vec_ok && cfg == 0 && !(variant & 15)) {
    // not recording, configuration left open: the same choice (rules, tuned table, issue log) as a recorded group of one
    const void* one = &p;
    if (b_kn) nt2_issue_auto<true>(&one, 1, stream); else nt2_issue_auto<false>(&one, 1, stream);
    return dav_launch_status();
  }
  if (b_kn) {
    if (!vec_ok) return DAV_ERR_SHAPE;
    if (cfg == 0) cfg = nt_auto_config(M, N, K);       // (only reached with the variant bits set; the open choice returned above)
    if (cfg == 60 && !nt256_ok(p)) return DAV_ERR_SHAPE;
    switch (cfg) {
      case 60: launch_nt256<true>(p, stream); break;
      case 3: launch_nt2<128, 128, 4, 2, 2, true>(p, stream); break;
      case 44: launch_nt2<128, 256, 2, 4, 3, true, 32>(p, stream); break;
      case 43: launch_nt2<256, 128, 4, 2, 3, true, 32>(p, stream); break;
      case 45: launch_nt2<256, 128, 4, 2, 2, true, 32>(p, stream); break;
      case 46: launch_nt2<128, 256, 2, 4, 2, true, 32>(p, stream); break;
#ifdef DAV_EXPERIMENTAL      // measured and rejected configurations: kept for the tools that reproduce the measurements
      case 50: launch_nt2<128, 128, 4, 2, 2, true, 64, 2>(p, stream); break;
      case 51: launch_nt2<128, 128, 4, 2, 2, true, 64, 4>(p, stream); break;
      case 52: launch_nt2<192, 128, 4, 2, 2, true>(p, stream); break;      // 192-row tiles, 2 x 80 KB per CU (round 5: -15..30 % alone on the M = 22528 / 3136 shapes, +0.1..0.3 ms in the step)
      case 31: launch_nt2<128, 128, 2, 4, 2, true, 64, 1>(p, stream); break;
      case 32: launch_nt3<true>(p, stream); break;
      case 40: launch_nt2<256, 128, 4, 4, 3, true>(p, stream); break;      // 16 waves, one workgroup per CU, 3 x 48 KB ring
      case 41: launch_nt2<128, 256, 4, 4, 3, true>(p, stream); break;
#endif
      case 8: launch_nt2<128, 64, 2, 2, 2, true>(p, stream); break;
      case 7: launch_nt2<64, 64, 2, 2, 4, true>(p, stream); break;
      case 5: launch_nt2<64, 64, 2, 2, 2, true>(p, stream); break;
      default: return DAV_ERR_SHAPE;          // a configuration this build does not contain (-DDAV_EXPERIMENTAL builds the rejected ones)
    }
    return dav_launch_status();
  }
  if (vec_ok && !(variant & 15)) {
    if (cfg == 0) cfg = nt_auto_config(M, N, K);
    if (cfg == 60 && !nt256_ok(p)) return DAV_ERR_SHAPE;
    switch (cfg) {
      case 60: launch_nt256<false>(p, stream); return dav_launch_status();
      case 3: launch_nt2<128, 128, 4, 2, 2>(p, stream); return dav_launch_status();
      case 5: launch_nt2<64, 64, 2, 2, 2>(p, stream); return dav_launch_status();
      case 7: launch_nt2<64, 64, 2, 2, 4>(p, stream); return dav_launch_status();
      case 8: launch_nt2<128, 64, 2, 2, 2>(p, stream); return dav_launch_status();
      case 43: launch_nt2<256, 128, 4, 2, 3, false, 32>(p, stream); return dav_launch_status();   // 8 waves of 64x64, 3 x 24 KB ring, two workgroups per CU
      case 44: launch_nt2<128, 256, 2, 4, 3, false, 32>(p, stream); return dav_launch_status();
      case 45: launch_nt2<256, 128, 4, 2, 2, false, 32>(p, stream); return dav_launch_status();
      case 46: launch_nt2<128, 256, 2, 4, 2, false, 32>(p, stream); return dav_launch_status();
#ifdef DAV_EXPERIMENTAL      // measured and rejected configurations: kept for the tools that reproduce the measurements
      case 30: {      // phase profile of the dominant configuration (see nt2_body); res_rows carries the int64 output buffer
        if (res) return DAV_ERR_SHAPE;
        const int grid = ((M + 127) / 128) * ((N + 127) / 128);
        DAV_LAUNCH_NOW(gemm_nt2_prof_kernel, dim3(grid), dim3(512), (size_t)65536, stream, p);
        return dav_launch_status();
      }
      case 31: launch_nt2<128, 128, 2, 4, 2, false, 64, 1>(p, stream); return dav_launch_status();
      case 32: launch_nt3<false>(p, stream); return dav_launch_status();
      case 40: launch_nt2<256, 128, 4, 4, 3>(p, stream); return dav_launch_status();   // 16 waves, one workgroup per CU, 3 x 48 KB ring
      case 41: launch_nt2<128, 256, 4, 4, 3>(p, stream); return dav_launch_status();
      case 42: launch_nt2<256, 128, 4, 4, 2>(p, stream); return dav_launch_status();
      case 33: case 34: case 35: case 36: case 37: {      // phase profile of the staggered-halves kernel (+ ablations 1..4); res_rows = int64 output
        if (res) return DAV_ERR_SHAPE;
        const int grid = ((M + 255) / 256) * ((N + 127) / 128);
        const size_t lds = (size_t)6 * 384 * 64;
#define NT3_PROF(A) { (void)hipFuncSetAttribute((const void*)gemm_nt3_prof_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                      DAV_LAUNCH_NOW(gemm_nt3_prof_kernel<A>, dim3(grid), dim3(512), lds, stream, p); }
        if (cfg == 33) NT3_PROF(0) else if (cfg == 34) NT3_PROF(1) else if (cfg == 35) NT3_PROF(2) else if (cfg == 36) NT3_PROF(3) else NT3_PROF(4)
#undef NT3_PROF
        return dav_launch_status();
      }
      case 1: launch_nt2<128, 128, 2, 2, 2>(p, stream); return dav_launch_status();
      case 2: launch_nt2<128, 128, 2, 2, 3>(p, stream); return dav_launch_status();
      case 50: launch_nt2<128, 128, 4, 2, 2, false, 64, 2>(p, stream); return dav_launch_status();
      case 51: launch_nt2<128, 128, 4, 2, 2, false, 64, 4>(p, stream); return dav_launch_status();
      case 52: launch_nt2<192, 128, 4, 2, 2>(p, stream); return dav_launch_status();
      case 4: launch_nt2<128, 128, 2, 4, 3>(p, stream); return dav_launch_status();
      case 48: launch_nt2<128, 128, 4, 2, 4>(p, stream); return dav_launch_status();      // 4 x 32 KB ring, one workgroup per CU (in-flight depth experiment)
      case 6: launch_nt2<64, 64, 2, 2, 3>(p, stream); return dav_launch_status();
      case 9: launch_nt2<128, 64, 2, 2, 3>(p, stream); return dav_launch_status();
      case 10: launch_nt2<128, 64, 4, 1, 3>(p, stream); return dav_launch_status();
      case 11: launch_nt2<64, 128, 2, 2, 3>(p, stream); return dav_launch_status();
      case 12: launch_nt2<128, 128, 2, 2, 4>(p, stream); return dav_launch_status();
      case 13: launch_nt2<256, 128, 4, 2, 2>(p, stream); return dav_launch_status();
      case 14: launch_nt2<256, 128, 4, 2, 3>(p, stream); return dav_launch_status();
      case 15: launch_nt2<128, 128, 2, 2, 2, false, 32>(p, stream); return dav_launch_status();   // 32 KB ring -> 4 WG/CU
      case 16: launch_nt2<128, 128, 2, 2, 3, false, 32>(p, stream); return dav_launch_status();   // 48 KB -> 3 WG/CU
      case 17: launch_nt2<128, 128, 2, 2, 4, false, 32>(p, stream); return dav_launch_status();   // 64 KB -> 2 WG/CU
      case 18: launch_nt2<256, 128, 4, 2, 2, false, 32>(p, stream); return dav_launch_status();   // 48 KB, 8 waves of 64x64
      case 19: launch_nt2<128, 256, 2, 4, 2, false, 32>(p, stream); return dav_launch_status();   // 48 KB, 8 waves of 64x64
      case 20: launch_nt2<128, 128, 2, 4, 2, false, 32>(p, stream); return dav_launch_status();   // 8 waves, 32 KB
      case 21: launch_nt2<128, 128, 2, 4, 3, false, 32>(p, stream); return dav_launch_status();
      case 22: launch_nt2<128, 128, 2, 4, 4, false, 32>(p, stream); return dav_launch_status();   // 64 KB, 1.5 steps of lookahead
#endif
      case 0: break;
      default: return DAV_ERR_SHAPE;          // a configuration this build does not contain (-DDAV_EXPERIMENTAL builds the rejected ones)
    }
  }
  const bool use_glds = glds_ok && !(variant & 1);
  // narrow-N problems (and anything that would leave most CUs idle) use 64-wide tiles
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const bool small = (variant & 2) || N <= 64 || tiles128 < 256;
  if (small) {
    const int grid = ((M + 63) / 64) * ((N + 63) / 64);
    const size_t lds = 2 * 64 * 128 * 2;
    if (use_glds) DAV_LAUNCH((gemm_nt_kernel<64, 64, true>), dim3(grid), dim3(256), lds, stream, p);
    else DAV_LAUNCH((gemm_nt_kernel<64, 64, false>), dim3(grid), dim3(256), lds, stream, p);
  } else {
    const int grid = (int)tiles128;
    const size_t lds = 2 * 128 * 128 * 2;
    if (use_glds) DAV_LAUNCH((gemm_nt_kernel<128, 128, true>), dim3(grid), dim3(256), lds, stream, p);
    else DAV_LAUNCH((gemm_nt_kernel<128, 128, false>), dim3(grid), dim3(256), lds, stream, p);
  }
  return dav_launch_status();
}

extern "C" int dav_nt_tune_set(const int* blob, int n_ints) {
  // blob: entries [cfg, b_kn, n, n x (M, N, K, flags)] as dav_nt_issue_log writes them; n_ints == 0 clears the table
  std::map<std::vector<int>, int> m;
  int i = 0;
  while (i < n_ints) {
    if (i + 3 > n_ints) return DAV_ERR_SHAPE;
    const int cfg = blob[i], bt = blob[i + 1], n = blob[i + 2];
    if (n <= 0 || i + 3 + 4 * n > n_ints) return DAV_ERR_SHAPE;
    if (!(cfg == 60 || cfg == 3 || cfg == 5 || cfg == 8 || cfg == 43 || cfg == 44 || cfg == 45 || cfg == 46)) return DAV_ERR_SHAPE;
    std::vector<std::array<int, 4>> v(n);
    for (int j = 0; j < n; ++j) v[j] = {blob[i + 3 + 4 * j], blob[i + 4 + 4 * j], blob[i + 5 + 4 * j], blob[i + 6 + 4 * j]};
    std::sort(v.begin(), v.end());
    std::vector<int> key;
    key.push_back(bt ? 1 : 0);
    for (auto& q : v) key.insert(key.end(), q.begin(), q.end());
    m[key] = cfg;
    i += 3 + 4 * n;
  }
  std::lock_guard<std::mutex> lk(nt_tuned_mu);
  nt_tuned.swap(m);
  return (int)nt_tuned.size();
}

extern "C" int dav_nt_issue_log(int enable, int* out, int capacity) {
  // enable: 1 start (clears), 0 stop; out != NULL: copy the log (returns its length in ints, or -needed if capacity is short)
  std::lock_guard<std::mutex> lk(nt_log_mu);
  if (out) {
    const int n = (int)nt_log.size();
    if (n > capacity) return -n;
    for (int i = 0; i < n; ++i) out[i] = nt_log[i];
    return n;
  }
  nt_log_on = enable != 0;
  if (enable) nt_log.clear();
  return 0;
}

static int tn_grouped_impl(const DavTnProblem* probs, int count, hipStream_t stream) {
  if (count <= 0 || count > TN_GROUP_MAX) return DAV_ERR_SHAPE;
  static thread_local TNGroup g;   // host staging (one per calling thread); copied by value into the kernel arguments at launch
  long total_tiles = 0;
  for (int i = 0; i < count; ++i) {
    const DavTnProblem& q = probs[i];
    if (q.Mc <= 0 || (q.Mc & 63) || q.N <= 0 || q.K <= 0 || (q.N & 7) || (q.K & 7) || (q.lda & 7) || (q.ldb & 7) || (q.flags & ~1)) return DAV_ERR_SHAPE;
    if (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15 || (q.ldc & 3)) return DAV_ERR_ALIGN;      // 16-byte gradient accesses
    total_tiles += (long)((q.N + 127) / 128) * ((q.K + 127) / 128);
  }
  // aim at ~4 workgroups of 8 waves per CU-slot pair (1024 blocks) but keep >= 8 k-steps of 64 rows per split
  int first = 0;
  for (int i = 0; i < count; ++i) {
    const DavTnProblem& q = probs[i];
    TNParams& p = g.prob[i];
    p.A = (const bf16_t*)q.A; p.B = (const bf16_t*)q.B; p.Mc = q.Mc; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb;
    p.amap = RowMap{q.a_rowmap[0], q.a_rowmap[1], q.a_rowmap[2]};
    p.bmap = RowMap{q.b_rowmap[0], q.b_rowmap[1], q.b_rowmap[2]};
    static const int tn_debug = getenv("DAV_TN_DEBUG") ? atoi(getenv("DAV_TN_DEBUG")) & 6 : 0;
    p.C = q.C; p.ldc = q.ldc; p.beta = (q.flags & 1) ? 0 : 1; p.bias_grad = q.bias_grad; p.debug_plain_store = tn_debug;
    const int tiles = ((q.N + 127) / 128) * ((q.K + 127) / 128), steps = q.Mc >> 6;
    int splits = (int)((1024 + total_tiles - 1) / total_tiles);
    const int max_splits = steps / 8 > 0 ? steps / 8 : 1;
    if (splits > max_splits) splits = max_splits;
    if (q.flags & 1) splits = 1;                         // a written (not accumulated) tile has one owner
    p.splits = splits;
    g.first_block[i] = first;
    first += tiles * splits;
  }
  g.first_block[count] = first;
  // (round 3: -21 % memory-side fetch, -5 % kernel time, -0.2 ms per step on two boxes.  Cutting the WHOLE launch's unit sequence
  // into eight equal-work ranges, one per XCD, took the fetch down 3 x (33.5 -> 11.5 GB per step) and the kernel time UP 6 %:
  // profiles/r03_step_traffic.txt — what bounded this kernel was not the fabric but a compiler-inserted drain of its own DMA, see tn2_tr)
  g.count = count | TN_XCD_RUNS;
  // 4 x 2 waves (32 x 64 wave tiles): +1 % over 2 x 4 in the step; thinner or deeper rings (32-row stages x 3 / 4, 64-row x 3),
  // 4-wave workgroups, three workgroups per CU, 256 x 128 owner-per-tile tiles (round 4) and a 256 x 256 stream-K form with fp32
  // atomics (round 3) were all measured slower (profiles/r02_tn_ring_variants.txt, r04_tn256x128.txt, r03_tn256_group_bench.txt)
  // (by-value copy in an AUTOMATIC: a launch recorded inside dav_batch_begin .. dav_batch_end captures its arguments with [=],
  // which does not copy objects of static storage — two recorded grouped calls would both run with the last table)
  const TNGroup gl = g;
  DAV_LAUNCH((gemm_tn_grouped_kernel<128, 4, 2>), dim3(first), dim3(512), (size_t)2 * 2 * 64 * 128 * 2, stream, gl);
  return dav_launch_status();
}

extern "C" int dav_gemm_tn_grouped_bf16(const DavTnProblem* probs, int count, hipStream_t stream) {
  if (!probs) return DAV_ERR_SHAPE;
  return tn_grouped_impl(probs, count, stream);
}

// ---- gang-scheduled 256 x 256 weight gradients (csrc/gemm_tn_gang.h): host-side plan ------------------------------------------
namespace {
struct TGPlanItem { int prob, r0, c0, nr, nc; long weight; int queue, first; };
// cuts every problem's grid of 256 x 256 tiles into gangs of <= 32 tiles (one XCD's CUs), longest contraction first onto the
// least-loaded of the 8 queues; returns the total tile count
long tn_gang_plan(const DavTnProblem* probs, int count, std::vector<TGPlanItem>& items, int q_start[9]) {
  items.clear();
  for (int i = 0; i < count; ++i) {
    const DavTnProblem& q = probs[i];
    const int tn = (q.N + 255) / 256, tk = (q.K + 255) / 256;
    int nr = tn, nc = tk;
    while (nr * nc > 32) { if (nr >= nc) nr = (nr + 1) / 2; else nc = (nc + 1) / 2; }
    for (int r0 = 0; r0 < tn; r0 += nr)
      for (int c0 = 0; c0 < tk; c0 += nc) {
        const int r = std::min(nr, tn - r0), c = std::min(nc, tk - c0);
        items.push_back(TGPlanItem{i, r0, c0, r, c, (long)r * c * ((q.Mc + 127) / 128), 0, 0});
      }
  }
  std::vector<int> order(items.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return probs[items[a].prob].Mc > probs[items[b].prob].Mc; });
  long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int tiles_q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int oi : order) {
    int best = 0;
    for (int q = 1; q < 8; ++q) if (load[q] < load[best]) best = q;
    items[oi].queue = best; items[oi].first = tiles_q[best];
    load[best] += items[oi].weight; tiles_q[best] += items[oi].nr * items[oi].nc;
  }
  q_start[0] = 0;
  for (int q = 0; q < 8; ++q) q_start[q + 1] = q_start[q] + tiles_q[q];
  return q_start[8];
}
int tn_gang_check(const DavTnProblem* probs, int count) {
  if (count <= 0 || count > 4096) return DAV_ERR_SHAPE;
  for (int i = 0; i < count; ++i) {
    const DavTnProblem& q = probs[i];
    if (q.Mc <= 0 || q.N <= 0 || q.K <= 0 || (q.N & 7) || (q.K & 7) || (q.lda & 7) || (q.ldb & 7)) return DAV_ERR_SHAPE;      // (any contraction length: the ragged last K-tile reads zeros)
    if (q.N > (32767 << 8) || q.K > (32767 << 8) || (q.flags & ~1)) return DAV_ERR_SHAPE;
    if (((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15 || (q.ldc & 3)) return DAV_ERR_ALIGN;
  }
  return DAV_OK;
}
}  // namespace

extern "C" size_t dav_gemm_tn_gang_workspace_bytes(const DavTnProblem* probs, int count) {
  if (!probs || tn_gang_check(probs, count) != DAV_OK) return 0;
  long tiles = 0;
  for (int i = 0; i < count; ++i) tiles += (long)((probs[i].N + 255) / 256) * ((probs[i].K + 255) / 256);
  return sizeof(TGHeader) + (size_t)count * sizeof(TNParams) + (size_t)tiles * sizeof(TGDesc);
}

extern "C" int dav_gemm_tn_gang_bf16(const DavTnProblem* probs, int count, void* workspace, size_t workspace_bytes, hipStream_t stream) {
  if (!probs || !workspace) return DAV_ERR_SHAPE;
  const int chk = tn_gang_check(probs, count);
  if (chk != DAV_OK) return chk;
  if ((uintptr_t)workspace & 15) return DAV_ERR_ALIGN;
  if (workspace_bytes < dav_gemm_tn_gang_workspace_bytes(probs, count)) return DAV_ERR_WORKSPACE;
  std::vector<TGPlanItem> items;
  int q_start[9];
  tn_gang_plan(probs, count, items, q_start);
  // product builds honour the PLACEMENT bits only (8 / 16: same results, other queues); the ablations that make every weight
  // gradient wrong (2 = no epilogue, 4 = no MFMAs) exist in EXPERIMENTAL builds (tools/runs_r05/gang_ablate.sh)
#ifdef DAV_EXPERIMENTAL
  static const int dbg = getenv("DAV_TN_GANG_DEBUG") ? atoi(getenv("DAV_TN_GANG_DEBUG")) & 30 : 0;
#else
  static const int dbg = getenv("DAV_TN_GANG_DEBUG") ? atoi(getenv("DAV_TN_GANG_DEBUG")) & 24 : 0;
#endif
  // the tables go to the workspace through kernel arguments, <= TG_WCH problems / TG_WIT gangs per writer launch
  bool header = true;
  std::vector<std::vector<const TGPlanItem*>> by_prob(count);
  for (const TGPlanItem& it : items) by_prob[it.prob].push_back(&it);
  TGWrite w;
  auto reset = [&](int first) {
    w.ws = (char*)workspace; w.prob_first = first; w.prob_count = 0; w.item_count = 0; w.count_total = count; w.write_header = 0;
    for (int q = 0; q < 9; ++q) w.q_start[q] = q_start[q];
  };
  auto flush = [&]() {
    if (!w.prob_count && !w.item_count && !header) return;
    w.write_header = header ? 1 : 0; header = false;
    const TGWrite wl = w;
    DAV_LAUNCH(gemm_tn_gang_write_kernel, dim3(wl.item_count + 1), dim3(64), 0, stream, wl);
  };
  reset(0);
  for (int i = 0; i < count; ++i) {
    if (w.prob_count == TG_WCH) { flush(); reset(i); }
    const DavTnProblem& q = probs[i];
    TNParams& p = w.prob[w.prob_count++];
    p.A = (const bf16_t*)q.A; p.B = (const bf16_t*)q.B; p.Mc = q.Mc; p.N = q.N; p.K = q.K; p.lda = q.lda; p.ldb = q.ldb;
    p.amap = RowMap{q.a_rowmap[0], q.a_rowmap[1], q.a_rowmap[2]};
    p.bmap = RowMap{q.b_rowmap[0], q.b_rowmap[1], q.b_rowmap[2]};
    p.C = q.C; p.ldc = q.ldc; p.beta = (q.flags & 1) ? 0 : 1; p.bias_grad = q.bias_grad; p.debug_plain_store = 0; p.splits = 1;
    for (const TGPlanItem* it : by_prob[i]) {      // (a very wide weight has more gangs than one writer launch carries: its tickets span several)
      if (w.item_count == TG_WIT) { flush(); reset(i + 1); }
      w.item[w.item_count++] = TGItem{i, q_start[it->queue] + it->first, (it->r0 << 16) | it->c0, (it->nr << 16) | it->nc};
    }
  }
  flush();
  // per device (a process may drive several): the > 64 KB dynamic-LDS opt-in and the CU count the persistent grid is sized from
  static std::mutex dev_mu;
  static int dev_cus[64] = {0};
  int n_wg = 256;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DAV_ERR_HIP;
    std::lock_guard<std::mutex> lk(dev_mu);
    if (!dev_cus[dev]) {
      if (hipFuncSetAttribute((const void*)gemm_tn_gang_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return DAV_ERR_HIP;
      int cus = 256;
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      dev_cus[dev] = cus > 0 ? cus : 256;
    }
    n_wg = dev_cus[dev];
  }
  const long tiles = q_start[8];
  const int grid = (int)std::min<long>(n_wg, tiles);
  DAV_LAUNCH(gemm_tn_gang_kernel, dim3(grid), dim3(512), TNG_LDS, stream, (char*)workspace, count, dbg);
  return dav_launch_status();
}

extern "C" int dav_gemm_tn_bf16(const void* A, const void* B, int Mc, int N, int K, int lda, int ldb,
                                const int* a_rowmap, const int* b_rowmap, float* C, int ldc, int beta,
                                float* bias_grad, int variant, hipStream_t stream) {
  if (Mc <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || (lda & 7) || (ldb & 7)) return DAV_ERR_SHAPE;
  if (((uintptr_t)A | (uintptr_t)B) & 15) return DAV_ERR_ALIGN;
  TNParams p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.Mc = Mc; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb;
  p.amap = mk(a_rowmap); p.bmap = mk(b_rowmap); p.C = C; p.ldc = ldc; p.beta = beta; p.bias_grad = bias_grad;
  p.debug_plain_store = (variant >> 8) & 1;
  variant &= ~256;
  const int steps = (Mc + 63) / 64;
  const int cfg = variant >> 4;           // 0 auto, 1 = 128x128 tiles, 2 = 64x64 tiles (benchmark knob)
  if (!(variant & 15) && (Mc & 63) == 0 && (beta || cfg) && !(ldc & 3) && !((uintptr_t)C & 15)) {      // (16-byte gradient accesses)
    const int t128 = ((N + 127) / 128) * ((K + 127) / 128), t64 = ((N + 63) / 64) * ((K + 63) / 64);
    const bool big = cfg ? cfg == 1 : (Mc >= 20000 && t128 >= 48);   // 64x64 tiles win except on the longest contractions
    const int tiles = big ? t128 : t64;
    int splits = beta ? (768 + tiles - 1) / tiles : 1;      // accumulate mode may split the contraction
    const int max_splits = steps / 6 > 0 ? steps / 6 : 1;      // at least ~6 k-steps of 64 rows per split
    if (splits > max_splits) splits = max_splits;
    p.splits = splits;
    if (big) launch_tn2<128, 2, 4>(p, tiles, stream);
    else launch_tn2<64, 2, 2>(p, tiles, stream);
    return dav_launch_status();
  }
  const int tiles = ((N + 127) / 128) * ((K + 127) / 128);
  int splits = 1;
  if (beta) {   // accumulate mode may split the contraction (atomic adds into the live gradient)
    splits = (1024 + tiles - 1) / tiles;
    if (splits > steps) splits = steps;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  p.splits = splits;
  const size_t lds = 4 * 64 * 256;
  if (variant & 1) DAV_LAUNCH((gemm_tn_kernel<false>), dim3(tiles, splits), dim3(256), lds, stream, p);
  else DAV_LAUNCH((gemm_tn_kernel<true>), dim3(tiles, splits), dim3(256), lds, stream, p);
  return dav_launch_status();
}
