// Softmax attention for gfx950 (MI355X), bf16 in / fp32 softmax / bf16 out.
//
// One workgroup per (batch, head).  All keys/values of that head live in LDS (the path's sequences
// are 8..352 rows: SURVEY.md section 8), each wave owns 16-row tiles and keeps the softmax in
// registers: scores are produced TRANSPOSED (S^T = K.Q^T on the 16x16x32 bf16 MFMA) so that a
// lane's four accumulators are four keys of ONE query row -> the row max/sum are a few lane-local
// ops plus two cross-lane shuffles, and the probabilities feed the second MFMA (O^T = V^T.P^T)
// straight from registers with no LDS round trip.  The V / K / Q / dO tiles are staged ONCE, row-major (16-byte
// coalesced LDS writes); every transposed operand fragment comes from the transposing ds_read_b64_tr_b16.
//
// Covers every attention on the pre-training path through strides only:
//   * timm Attention inside Block (models/vits.py:32-34, models/avmae.py:53-55,83-85) incl. the
//     "fusion tokens are context rows, their query rows are dropped" use at models/deepavfusion.py:104-105
//     (queries start nF rows into the fused qkv buffer, keys/values cover all rows)
//   * CrossAttention                     models/fusion_blocks.py:46-59
//   * the factorised pair attention      models/fusion_blocks.py:250-258 (q/k width 16, v width 64,
//     scale (D/heads)^-0.5 passed by the caller)
#include <algorithm>
#include <vector>

#include <type_traits>
#include "common.h"
#include "dav_kernels.h"

#include <cstdlib>
static int attn_debug() { static const int v = [] { const char* e = getenv("DAV_ATTN_DEBUG"); return e ? atoi(e) : 0; }(); return v; }
int dav_attn_qt = 0;      // dav_tune knob 3: 0 = auto, 1 / 2 = query tiles per wave in the forward kernel

namespace {

struct AttnParams {
  const bf16_t *Q, *K, *V;
  bf16_t* O;          // [B, Nq, H*DV]-style through o_bs / o_rs
  float* LSE;         // [B, H, Nq]  log-sum-exp of the scaled scores
  int B, H, Nq, Nk;
  long q_bs, k_bs, v_bs, o_bs;
  int q_rs, k_rs, v_rs, o_rs;
  float scale;
  // backward only
  const bf16_t* dO; long do_bs; int do_rs;
  const bf16_t* Of;   // forward output (for delta), same strides as O
  float* Delta;       // [B, H, Nq]
  bf16_t *dQ, *dK, *dV;
  long dq_bs, dk_bs, dv_bs;
  int dq_rs, dk_rs, dv_rs;
  int debug;          // DAV_ATTN_DEBUG ablations (timing experiments only): 1 = no tile loop, 2 = no staging
  int pair;           // narrow heads: adjacent heads on the same XCD (pair_heads)
  // additive score bias (window attention, models/swin.py:66-80): bias[b % bias_nb][h][q][0 .. bias_ld) in LOG2 units (already
  // multiplied by log2 e), bias_ld = Nk rounded up to 32 with zero padding; dS (backward, optional): gradient of the biased
  // logits [B][H][Nq][bias_ld] in natural units, what the relative-position table's gradient is reduced from
  const float* bias; int bias_nb, bias_ld;
  float* dS;
  // backward: the dQ buffer holds this many rows BEFORE the first query row of every batch element that belong to keys / values
  // only (the fusion-token context rows of a fused qkv gradient, models/deepavfusion.py:104-105): the dQ kernel zero-fills this
  // head's columns of them, which the qkv weight-gradient / input-gradient GEMMs read (dav_attn_bwd_ctx)
  int dq_ctx;
  // attention dropout (nn.Dropout on the probabilities, models/fusion_blocks.py:14,25 / timm Attention.attn_drop; fine-tuning only):
  // keep[b][h][q][0 .. keep_ld) bytes, non-zero = kept; kept probabilities are scaled by keep_scale = 1 / (1 - p) AFTER the row sum
  // (softmax, then dropout).  keep_ld >= Nk rounded up to 32, a multiple of 4.  Null on the pre-training path.
  const unsigned char* keep; int keep_ld; float keep_scale;
};
__device__ __forceinline__ void keep4(const unsigned char* ptr, float scale, float (&km)[4]) {
  const uint32_t m = *reinterpret_cast<const uint32_t*>(ptr);
#pragma unroll
  for (int r = 0; r < 4; ++r) km[r] = ((m >> (8 * r)) & 0xffu) ? scale : 0.f;
}

// 16-byte-slot XOR swizzle of a row-major LDS tile, chosen so that BOTH access patterns are conflict-free:
//  * ds_read_b128 of 16 consecutive rows at one column slot (its real lane groups are {0-3,12-15,20-27}, ...),
//  * ds_read_b64_tr_b16 of 8 consecutive rows x 32 bytes (two 16-lane groups of a 32-lane half).
// 128-byte rows (8 slots, 2 rows per 256-byte bank row): f = 2 * ((row >> 1) & 3)
//  64-byte rows (4 slots, 4 rows per bank row):           f = (-(row >> 2)) & 3
template <int RB> __device__ __forceinline__ int row_swz(int row) {
  return RB == 128 ? (((row >> 1) & 3) << 1) : ((0 - (row >> 2)) & 3);
}

// stage `nrows` rows of COLS bf16 (global row stride rs) into a swizzled row-major LDS tile with COLSP columns
// and nrows_p rows (zero padded); 16 bytes per lane, consecutive lanes = consecutive chunks of a row
template <int COLS, int COLSP>
__device__ __forceinline__ void stage_tile(char* dst, const bf16_t* src, int nrows, int nrows_p, int rs, int tid, int nthreads) {
  constexpr int CPR = COLSP / 8, RB = COLSP * 2;
  const int total = nrows_p * CPR;
  for (int c = tid; c < total; c += nthreads) {
    const int row = c / CPR, ch = c % CPR;
    uint4 v = uint4{0, 0, 0, 0};
    if (row < nrows && ch * 8 < COLS) v = *reinterpret_cast<const uint4*>(src + (long)row * rs + ch * 8);
    *reinterpret_cast<uint4*>(dst + row * RB + ((ch ^ row_swz<RB>(row)) << 4)) = v;
  }
}

// The same tile through the LDS-DMA path (global_load_lds_dwordx4: no register round trip, every 1 KB piece of a wave in
// flight at once — the register version pays one global-load latency per loop trip, ~10 us per workgroup on the decoder
// sequences, a quarter of the kernel).  A DMA lane's LDS position is fixed (piece base + 16 * lane), so the swizzle moves to
// the SOURCE side: the lane at linear slot s' of row r fetches global slot s' ^ swz(r).  Lanes of padded rows are masked off
// the DMA (it honours EXEC) and store zeros instead.  Only for rows as wide as their LDS tile (head widths 32 / 64).
// The caller waits (s_waitcnt vmcnt(0)) before its barrier.
template <int COLS>
__device__ __forceinline__ void stage_tile_dma(char* dst, const bf16_t* src, int nrows, int nrows_p, int rs, int tid, int nthreads) {
  constexpr int CPR = COLS / 8, RB = COLS * 2;
  const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
  const int npieces = nrows_p * CPR / 64;          // nrows_p is a multiple of 32, CPR of 4
  for (int pc = wave; pc < npieces; pc += nw) {
    const int c = pc * 64 + lane, row = c / CPR, slot = (c % CPR) ^ row_swz<RB>(row);
    if (row < nrows)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, src + (long)row * rs + slot * 8), LDS_PTR(void, dst + pc * 1024), 16, 0, 0);
    else
      *reinterpret_cast<uint4*>(dst + c * 16) = uint4{0, 0, 0, 0};
  }
}
// picks the DMA path when the rows are as wide as the tile
template <int COLS, int COLSP>
__device__ __forceinline__ void stage_rows(char* dst, const bf16_t* src, int nrows, int nrows_p, int rs, int tid, int nthreads) {
  if constexpr (COLS == COLSP) stage_tile_dma<COLS>(dst, src, nrows, nrows_p, rs, tid, nthreads);
  else stage_tile<COLS, COLSP>(dst, src, nrows, nrows_p, rs, tid, nthreads);
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 16-byte fragment (8 consecutive columns of one row) out of a swizzled row-major tile
template <int RB>
__device__ __forceinline__ bf16x8 tile_frag(const char* tile, int row, int kk, int g) {
  const int slot = (kk * 4 + g) ^ row_swz<RB>(row);
  return *reinterpret_cast<const bf16x8*>(tile + row * RB + (slot << 4));
}

// TRANSPOSED fragment out of the same row-major tile through the hardware transposing LDS read: lane (c = lane & 15,
// g = lane >> 4) receives column (colbase + c) at rows {r0 + 4g .. +3} and {r0 + 16 + 4g .. +3} — the contraction
// index order the register-resident P / dS fragments use (see the kernels).
template <int RB>
__device__ __forceinline__ bf16x8 tile_frag_tr(const char* tile, int r0, int colbase, int lane) {
  const int g = lane >> 4, li = lane & 15;
  union { s16x4 h[2]; bf16x8 v; } u;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row = r0 + 16 * h + 4 * g + (li >> 2);
    const int colb = (colbase + 4 * (li & 3)) * 2;
    const int addr = row * RB + ((((colb >> 4) ^ row_swz<RB>(row)) << 4) | (colb & 15));
    u.h[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, tile + addr));
  }
  return u.v;
}

// The same two fragment reads through per-lane LDS byte addresses that the kernels compute ONCE per tile walk and advance by
// a constant per 32-row step (the tile's own rows only move the address by multiples of 32 rows, which neither swizzle sees):
// the key / query loops then carry one integer add per operand column block instead of re-deriving row * RB + swizzle — and
// the (link-time) LDS base of the dynamic array — for every fragment.
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)LDS_PTR(const char, p); }
template <int RB> __device__ __forceinline__ uint32_t frag_off(int fr, int kk, int g) {        // tile_frag(tile, fr, kk, g)
  return (uint32_t)(fr * RB + ((((kk * 4 + g) ^ row_swz<RB>(fr))) << 4));
}
template <int RB> __device__ __forceinline__ uint32_t frag_tr_off(int colbase, int lane) {      // tile_frag_tr(tile, 0, colbase, lane), h = 0
  const int g = lane >> 4, li = lane & 15, row = 4 * g + (li >> 2), colb = (colbase + 4 * (li & 3)) * 2;
  return (uint32_t)(row * RB + ((((colb >> 4) ^ row_swz<RB>(row)) << 4) | (colb & 15)));
}
__device__ __forceinline__ bf16x8 lds_frag(uint32_t a) { return *LDS_PTR(const bf16x8, (uintptr_t)a); }
template <int RB> __device__ __forceinline__ bf16x8 lds_frag_tr(uint32_t a) {                   // rows +0..3 and +16..19 of the lane's column
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, (uintptr_t)a));
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, (uintptr_t)(a + 16 * RB)));
  return u.v;
}
__device__ __forceinline__ f32x4 lds_f4(uint32_t a) { return *LDS_PTR(const f32x4, (uintptr_t)a); }

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  union { uint32_t w[4]; bf16x8 v; } u;
  u.w[0] = pack2bf(a[0], a[1]); u.w[1] = pack2bf(a[2], a[3]);
  u.w[2] = pack2bf(b[0], b[1]); u.w[3] = pack2bf(b[2], b[3]);
  return u.v;
}

// 8 consecutive bf16 of a global row, zero when !ok
__device__ __forceinline__ bf16x8 gfrag(const bf16_t* rowptr, int col, bool ok) {
  union { uint4 q; bf16x8 v; } u;
  u.q = ok ? *reinterpret_cast<const uint4*>(rowptr + col) : uint4{0, 0, 0, 0};
  return u.v;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// CHUNKED = false: all keys of the head resident in LDS, a wave walks over query tiles (the pre-training path's
// sequences).  CHUNKED = true (long sequences, e.g. the 816-row video blocks): blockIdx.y picks a block of
// query tiles, one per wave, and the keys stream through LDS in chunks of ATTN_CHUNK rows with the online softmax
// state carried in registers.
constexpr int ATTN_CHUNK = 256;

// cross-lane max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48) with the gfx950 row/half swaps
// instead of two LDS-pipe ds_bpermute round trips
// The scores come out of MFMAs (never signalling NaNs); fmaxf() would still quiet every operand first (a v_max x, x each, IEEE
// mode) — 21 instructions for the maximum of a lane's 8 scores and its cross-lane part where 10 do.
__device__ __forceinline__ float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float max3_raw(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float rows_max(float x) {
  const unsigned u = __float_as_uint(x);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);     // {x.r0,x.r0,x.r2,x.r2}, {x.r1,x.r1,x.r3,x.r3}
  const float y = max_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const unsigned v = __float_as_uint(y);
  auto c = __builtin_amdgcn_permlane32_swap(v, v, false, false);     // {y.lo,y.lo}, {y.hi,y.hi}
  return max_raw(__uint_as_float(c[0]), __uint_as_float(c[1]));
}

template <int DQK, int DV, bool CHUNKED, int QT, bool DROP = false>
__device__ __forceinline__ void attn_fwd_body(const AttnParams& p, const int bh, const int ychunk) {
  constexpr int DQKP = DQK < 32 ? 32 : DQK, KRB = DQKP * 2, KS = DQKP / 32, VC = DV / 16, DVP = DV < 32 ? 32 : DV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Nkp = (p.Nk + 31) & ~31;
  const int CH = CHUNKED ? ATTN_CHUNK : Nkp;      // key rows resident at a time
  constexpr int VRB = DVP * 2;         // value rows are zero-padded to 32 columns in LDS like narrow q/k rows
  char* Ks = smem;                 // [CH][DQKP]
  char* Vs = smem + CH * KRB;      // [CH][DV]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int b = bh / p.H, h = bh % p.H;
  const int fr = lane & 15, g = lane >> 4;
  const bf16_t* Kg = p.K + b * p.k_bs + h * DQK;
  const bf16_t* Vg = p.V + b * p.v_bs + h * DV;

  if (!CHUNKED && p.debug != 2) {
    stage_rows<DQK, DQKP>(Ks, Kg, p.Nk, Nkp, p.k_rs, tid, blockDim.x);
    stage_rows<DV, DVP>(Vs, Vg, p.Nk, Nkp, p.v_rs, tid, blockDim.x);
    stage_wait();
    __syncthreads();
  }

  const int nqt = p.debug == 1 ? 0 : (p.Nq + 15) >> 4;
  const float sl2 = p.scale * 1.44269504088896341f;
  const float mul = p.bias ? 1.f : sl2;        // with a bias the scores are scaled when it is added
  // MFMA_SUM (measured, off): the row sums by one more MFMA against a block of ones instead of 8 adds per lane and key step — the
  // sums are then those of the bf16-ROUNDED probabilities (LSE off by 1-3e-4) for -5 % on the decoder shapes
  constexpr bool MFMA_SUM = false;
  bf16x8 ones;                         // 8 x bf16 1.0 in four registers the compiler cannot fold back into one (it would rebuild the quad per key step)
  {
    union { uint32_t w[4]; bf16x8 v; } o;
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, 0x3f803f80" : "=v"(o.w[i]));
    ones = o.v;
  }
  uint32_t ka0[KS], va0[VC];           // this lane's fragment addresses at key row 0
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) ka0[kk] = lds_addr(Ks) + frag_off<KRB>(fr, kk, g);
#pragma unroll
  for (int c = 0; c < VC; ++c) va0[c] = lds_addr(Vs) + frag_tr_off<VRB>(c * 16, lane);
  // A wave works on QT query tiles at once (independent softmax chains for the scheduler to interleave; every K / V
  // fragment read from LDS feeds QT MFMAs).  Chunked: exactly one (possibly out-of-range, then fully masked-off)
  // group per wave so that every wave reaches the chunk barriers.
  for (int qt = (CHUNKED ? ychunk * nw + wave : wave) * QT; CHUNKED ? qt >= 0 : qt < nqt; qt = CHUNKED ? -1 : qt + nw * QT) {
    bool qok[QT];
    bf16x8 qf[QT][KS];
    float m[QT], lsum[QT];             // m: reference max in the log2 domain
    f32x4 oacc[QT][VC], lacc[QT];      // lacc (MFMA_SUM): the row sums as one more "value column" of ones through the MFMA pipe
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const int q = (qt + u) * 16 + fr;
      qok[u] = q < p.Nq;
      const bf16_t* qrow = p.Q + b * p.q_bs + (long)(qok[u] ? q : p.Nq - 1) * p.q_rs + h * DQK;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[u][kk] = gfrag(qrow, kk * 32 + 8 * g, kk * 32 + 8 * g < DQK);
      m[u] = -1e30f; lsum[u] = 0.f; lacc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < VC; ++c) oacc[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int c0 = 0; c0 < Nkp; c0 += CH) {
      if (CHUNKED) {
        const int rows = p.Nk - c0 < CH ? p.Nk - c0 : CH, rows_p = Nkp - c0 < CH ? Nkp - c0 : CH;
        __syncthreads();                 // every wave is done with the previous chunk
        stage_rows<DQK, DQKP>(Ks, Kg + (long)c0 * p.k_rs, rows, rows_p, p.k_rs, tid, blockDim.x);
        stage_rows<DV, DVP>(Vs, Vg + (long)c0 * p.v_rs, rows, rows_p, p.v_rs, tid, blockDim.x);
        stage_wait();
        __syncthreads();
      }
      const int cend = c0 + CH < Nkp ? c0 + CH : Nkp;
      uint32_t ka[KS], va[VC];
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) ka[kk] = ka0[kk];
#pragma unroll
      for (int c = 0; c < VC; ++c) va[c] = va0[c];
      for (int k0 = c0; k0 < cend; k0 += 32) {
        bf16x8 kf[2][KS];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < KS; ++kk) kf[t][kk] = lds_frag(ka[kk] + t * 16 * KRB);
        f32x4 st[QT][2];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            st[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
              st[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[u][kk], st[u][t], 0, 0, 0);
          }
        if (p.bias) {                    // logits = scale * q.k + bias, all in log2 units from here on (mul == 1)
#pragma unroll
          for (int u = 0; u < QT; ++u) {
            const int q = (qt + u) * 16 + fr;
            const float* br = p.bias + (((long)(b % p.bias_nb) * p.H + h) * p.Nq + (q < p.Nq ? q : p.Nq - 1)) * p.bias_ld + k0 + 4 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const float4 bv = *reinterpret_cast<const float4*>(br + t * 16);
              st[u][t][0] = __builtin_fmaf(st[u][t][0], sl2, bv.x); st[u][t][1] = __builtin_fmaf(st[u][t][1], sl2, bv.y);
              st[u][t][2] = __builtin_fmaf(st[u][t][2], sl2, bv.z); st[u][t][3] = __builtin_fmaf(st[u][t][3], sl2, bv.w);
            }
          }
        }
        bf16x8 vf[VC];                   // issued here so that the softmax arithmetic covers their LDS latency
#pragma unroll
        for (int c = 0; c < VC; ++c) vf[c] = lds_frag_tr<VRB>(va[c]);
        // only the last key tile can hold padded keys
        if (k0 + 32 > p.Nk) {
#pragma unroll
          for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (k0 + t * 16 + 4 * g + r >= p.Nk) st[u][t][r] = -1e30f;
        }
        bf16x8 pf[QT];
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          // The FIRST maximum is a compiler-visible op reading one register of EACH of the two score MFMAs: hipcc pads nothing
          // inside an asm statement, so an asm v_max3 as the first reader of an MFMA result would rely on whatever happens to
          // sit between the two (advisor finding, round 3); after this op both results are architecturally complete and the asm
          // chain below (which depends on it) may read the rest.  Under -fno-honor-nans it is ONE v_max_f32 (no quieting pair).
          float mx = __builtin_fmaxf(st[u][0][0], st[u][1][0]);
          mx = max3_raw(mx, st[u][0][1], st[u][0][2]);
          mx = max3_raw(mx, st[u][0][3], st[u][1][1]);
          mx = max3_raw(mx, st[u][1][2], st[u][1][3]);
          mx = rows_max(mx) * mul;       // log2 domain (scale * log2(e) > 0 commutes with max)
          // deferred rescale: keep the reference max while no row's tile max exceeds it by more than 2^8 — the
          // probabilities then stay <= 256 (bf16 keeps its relative precision), O and the row sum need no multiply
          if (!__all(mx - m[u] <= 8.f)) {
            const float mn = fmaxf(m[u], mx);
            const float alpha = __builtin_amdgcn_exp2f(m[u] - mn);
            m[u] = mn;
            lsum[u] *= alpha;
            if constexpr (MFMA_SUM) {
#pragma unroll
              for (int r = 0; r < 4; ++r) lacc[u][r] *= alpha;
            }
#pragma unroll
            for (int c = 0; c < VC; ++c)
#pragma unroll
              for (int r = 0; r < 4; ++r) oacc[u][c][r] *= alpha;
          }
          float ps = 0.f;                // fp32 sum of the UNROUNDED probabilities, as torch's softmax has it
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              st[u][t][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t][r], mul, -m[u]));    // v_exp_f32 is 2^x
              if constexpr (!MFMA_SUM) ps += st[u][t][r];
            }
          lsum[u] += ps;
          if constexpr (DROP) {          // dropout on the probabilities (kernels of their own: the pre-training path's are untouched)
            const int q = (qt + u) * 16 + fr;
            const unsigned char* kr = p.keep + (((long)b * p.H + h) * p.Nq + (q < p.Nq ? q : p.Nq - 1)) * p.keep_ld + k0 + 4 * g;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              float km[4];
              keep4(kr + t * 16, p.keep_scale, km);
#pragma unroll
              for (int r = 0; r < 4; ++r) st[u][t][r] *= km[r];
            }
          }
          pf[u] = pack8(st[u][0], st[u][1]);
          if constexpr (MFMA_SUM) lacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[u], lacc[u], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < VC; ++c)
#pragma unroll
          for (int u = 0; u < QT; ++u)
            oacc[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[c], pf[u], oacc[u][c], 0, 0, 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) ka[kk] += 32 * KRB;
#pragma unroll
        for (int c = 0; c < VC; ++c) va[c] += 32 * VRB;
      }
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      float l = lsum[u];
      if constexpr (MFMA_SUM) l = lacc[u][0];
      else { l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64); }
      const float inv = 1.f / l;
      const int q = (qt + u) * 16 + fr;
      if (qok[u]) {
        bf16_t* orow = p.O + b * p.o_bs + (long)q * p.o_rs + h * DV;
#pragma unroll
        for (int c = 0; c < VC; ++c) {
          uint2 w;
          w.x = pack2bf(oacc[u][c][0] * inv, oacc[u][c][1] * inv);
          w.y = pack2bf(oacc[u][c][2] * inv, oacc[u][c][3] * inv);
          *reinterpret_cast<uint2*>(orow + c * 16 + 4 * g) = w;
        }
        if (g == 0 && p.LSE) p.LSE[((long)b * p.H + h) * p.Nq + q] = (m[u] + log2f(l)) * 0.69314718055994531f;   // natural-log LSE
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: dQ (waves own query tiles) + delta
// ------------------------------------------------------------------------------------------------
// QT query tiles per wave: each K / V fragment (and each transposed K fragment for dQ) read from LDS feeds QT MFMAs.
template <int DQK, int DV, bool CHUNKED, int QT = 1, bool DROP = false>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnParams& p, const int bh, const int ychunk) {
  if (p.dq_ctx > 0 && (CHUNKED ? ychunk == 0 : true)) {      // this head's dQ slots of the context-only rows (see AttnParams::dq_ctx)
    const int b_ = bh / p.H, h_ = bh % p.H, cpr = DQK / 8;
    for (int c = threadIdx.x; c < p.dq_ctx * cpr; c += blockDim.x) {
      const int r = c / cpr, cc = c % cpr;
      *reinterpret_cast<uint4*>(p.dQ + b_ * p.dq_bs + (long)(r - p.dq_ctx) * p.dq_rs + h_ * DQK + cc * 8) = uint4{0, 0, 0, 0};
    }
  }
  constexpr int DQKP = DQK < 32 ? 32 : DQK, KRB = DQKP * 2, KS = DQKP / 32, DVP = DV < 32 ? 32 : DV, VRB = DVP * 2, VS = DVP / 32, QC = DQK / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Nkp = (p.Nk + 31) & ~31;
  const int CH = CHUNKED ? ATTN_CHUNK : Nkp;
  char* Ks = smem;                   // [CH][DQKP]
  char* Vs = Ks + CH * KRB;          // [CH][DV]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int b = bh / p.H, h = bh % p.H;
  const int fr = lane & 15, g = lane >> 4;
  const bf16_t* Kg = p.K + b * p.k_bs + h * DQK;
  const bf16_t* Vg = p.V + b * p.v_bs + h * DV;

  if (!CHUNKED && p.debug != 2) {
    stage_rows<DQK, DQKP>(Ks, Kg, p.Nk, Nkp, p.k_rs, tid, blockDim.x);
    stage_rows<DV, DVP>(Vs, Vg, p.Nk, Nkp, p.v_rs, tid, blockDim.x);
    stage_wait();
    __syncthreads();
  }

  const int nqt = p.debug == 1 ? 0 : (p.Nq + 15) >> 4;
  const float sl2 = p.scale * 1.44269504088896341f;
  for (int qt = (CHUNKED ? ychunk * nw + wave : wave) * QT; CHUNKED ? qt >= 0 : qt < nqt; qt = CHUNKED ? -1 : qt + nw * QT) {
    bool qok[QT];
    bf16x8 qf[QT][KS], dof[QT][VS];
    float delta[QT], lse2[QT];
    f32x4 dq[QT][QC];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      const int q = (qt + u) * 16 + fr;
      qok[u] = q < p.Nq;
      const int qc = qok[u] ? q : p.Nq - 1;
      const bf16_t* qrow = p.Q + b * p.q_bs + (long)qc * p.q_rs + h * DQK;
      const bf16_t* dorow = p.dO + b * p.do_bs + (long)qc * p.do_rs + h * DV;
      const bf16_t* orow = p.Of + b * p.o_bs + (long)qc * p.o_rs + h * DV;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qf[u][kk] = gfrag(qrow, kk * 32 + 8 * g, kk * 32 + 8 * g < DQK);
      float d = 0.f;
#pragma unroll
      for (int kk = 0; kk < VS; ++kk) {
        dof[u][kk] = gfrag(dorow, kk * 32 + 8 * g, kk * 32 + 8 * g < DV);
        const bf16x8 of = gfrag(orow, kk * 32 + 8 * g, kk * 32 + 8 * g < DV);
#pragma unroll
        for (int e = 0; e < 8; ++e) d += (float)dof[u][kk][e] * (float)of[e];
      }
      d += __shfl_xor(d, 16, 64);
      d += __shfl_xor(d, 32, 64);
      delta[u] = d;
      const long sidx = ((long)b * p.H + h) * p.Nq + qc;
      lse2[u] = p.LSE[sidx] * 1.44269504088896341f;      // log2 domain
      if (qok[u] && g == 0) p.Delta[sidx] = d;
#pragma unroll
      for (int c = 0; c < QC; ++c) dq[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int c0 = 0; c0 < Nkp; c0 += CH) {
    if (CHUNKED) {
      const int rows = p.Nk - c0 < CH ? p.Nk - c0 : CH, rows_p = Nkp - c0 < CH ? Nkp - c0 : CH;
      __syncthreads();
      stage_rows<DQK, DQKP>(Ks, Kg + (long)c0 * p.k_rs, rows, rows_p, p.k_rs, tid, blockDim.x);
      stage_rows<DV, DVP>(Vs, Vg + (long)c0 * p.v_rs, rows, rows_p, p.v_rs, tid, blockDim.x);
      stage_wait();
      __syncthreads();
    }
    const int cend = c0 + CH < Nkp ? c0 + CH : Nkp;
    uint32_t ka[KS], va[VS], kta[QC];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ka[kk] = lds_addr(Ks) + frag_off<KRB>(fr, kk, g);
#pragma unroll
    for (int kk = 0; kk < VS; ++kk) va[kk] = lds_addr(Vs) + frag_off<VRB>(fr, kk, g);
#pragma unroll
    for (int c = 0; c < QC; ++c) kta[c] = lds_addr(Ks) + frag_tr_off<KRB>(c * 16, lane);
    // no key mask: the padded key rows of the K tile are zero, so whatever dS they get multiplies zeros in dQ = dS.K
    // (two copies of the key loop, picked once per walk: the path's own case — no bias, no dS output — carries no zero bias to
    // subtract the LSE from and no 64-bit score-row addresses; decided inside the loop the compiler kept both alive per step)
    auto key_loop = [&](auto plain_c) {
    constexpr bool PLAIN = decltype(plain_c)::value;
    for (int k0 = c0; k0 < cend; k0 += 32) {
      f32x4 st[QT][2], dp[QT][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 kf[KS], vf[VS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) kf[kk] = lds_frag(ka[kk] + t * 16 * KRB);
#pragma unroll
        for (int kk = 0; kk < VS; ++kk) vf[kk] = lds_frag(va[kk] + t * 16 * VRB);
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          st[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
            st[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kk], qf[u][kk], st[u][t], 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < VS; ++kk)
            dp[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kk], dof[u][kk], dp[u][t], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
          if constexpr (PLAIN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t][r], sl2, -lse2[u]));
              st[u][t][r] = pr * (dp[u][t][r] - delta[u]);     // dS^T
            }
            continue;
          }
          float4 bv = float4{0.f, 0.f, 0.f, 0.f};
          const int q = (qt + u) * 16 + fr;
          const long srow = (((long)b * p.H + h) * p.Nq + (q < p.Nq ? q : p.Nq - 1)) * p.bias_ld + k0 + t * 16 + 4 * g;
          if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + srow - ((long)(b - b % p.bias_nb) * p.H * p.Nq) * p.bias_ld);
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
          float km[4] = {1.f, 1.f, 1.f, 1.f};                // attention dropout: dP reaches the softmax backward through the same mask
          if constexpr (DROP) keep4(p.keep + (((long)b * p.H + h) * p.Nq + (q < p.Nq ? q : p.Nq - 1)) * p.keep_ld + k0 + t * 16 + 4 * g, p.keep_scale, km);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][t][r], sl2, bb[r] - lse2[u]));
            st[u][t][r] = pr * (dp[u][t][r] * km[r] - delta[u]);     // dS^T
          }
          if (p.dS && qok[u]) *reinterpret_cast<float4*>(p.dS + srow) = float4{st[u][t][0], st[u][t][1], st[u][t][2], st[u][t][3]};
        }
      }
      bf16x8 dsf[QT];
#pragma unroll
      for (int u = 0; u < QT; ++u) dsf[u] = pack8(st[u][0], st[u][1]);
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        const bf16x8 kt = lds_frag_tr<KRB>(kta[c]);
#pragma unroll
        for (int u = 0; u < QT; ++u) dq[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[u], dq[u][c], 0, 0, 0);
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) ka[kk] += 32 * KRB;
#pragma unroll
      for (int kk = 0; kk < VS; ++kk) va[kk] += 32 * VRB;
#pragma unroll
      for (int c = 0; c < QC; ++c) kta[c] += 32 * KRB;
    }
    };
    if (!DROP && !p.bias && !p.dS) key_loop(std::true_type{}); else key_loop(std::false_type{});
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
      if (qok[u]) {
        bf16_t* dqrow = p.dQ + b * p.dq_bs + (long)((qt + u) * 16 + fr) * p.dq_rs + h * DQK;
#pragma unroll
        for (int c = 0; c < QC; ++c) {
          uint2 w;
          w.x = pack2bf(dq[u][c][0] * p.scale, dq[u][c][1] * p.scale);
          w.y = pack2bf(dq[u][c][2] * p.scale, dq[u][c][3] * p.scale);
          *reinterpret_cast<uint2*>(dqrow + c * 16 + 4 * g) = w;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: dK, dV (waves own key tiles)
// ------------------------------------------------------------------------------------------------
// KT key tiles per wave: every Q / dO fragment read from LDS (the row-major one for S and dP, the transposed one for dV and
// dK) then feeds KT MFMAs.  With one tile per wave the kernel is bound by the LDS pipe — 16 fragment reads per 8 MFMAs at
// d = 32, 128 of the CU's LDS cycles per wave-step against 70 VALU and 32 MFMA cycles per SIMD.
template <int DQK, int DV, bool CHUNKED, int KT = 1, bool DROP = false>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnParams& p, const int bh, const int ychunk) {
  constexpr int DQKP = DQK < 32 ? 32 : DQK, QRB = DQKP * 2, KS = DQKP / 32, DVP = DV < 32 ? 32 : DV, ORB = DVP * 2, VS = DVP / 32, QC = DQK / 16, VC = DV / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Nqp = (p.Nq + 31) & ~31;
  const int CH = CHUNKED ? ATTN_CHUNK : Nqp;     // query rows resident at a time
  char* Qs = smem;                      // [CH][DQKP]
  char* dOs = Qs + CH * QRB;            // [CH][DV]
  float* lse_s = reinterpret_cast<float*>(dOs + CH * ORB);        // [CH]
  float* del_s = lse_s + CH;                                      // [CH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int b = bh / p.H, h = bh % p.H;
  const int fr = lane & 15, g = lane >> 4;
  const bf16_t* Qg = p.Q + b * p.q_bs + h * DQK;
  const bf16_t* dOg = p.dO + b * p.do_bs + h * DV;

  auto stage_q = [&](int c0) {
    const int rows = p.Nq - c0 < CH ? p.Nq - c0 : CH, rows_p = Nqp - c0 < CH ? Nqp - c0 : CH;
    stage_rows<DQK, DQKP>(Qs, Qg + (long)c0 * p.q_rs, rows, rows_p, p.q_rs, tid, blockDim.x);
    stage_rows<DV, DVP>(dOs, dOg + (long)c0 * p.do_rs, rows, rows_p, p.do_rs, tid, blockDim.x);
    for (int i = tid; i < rows_p; i += blockDim.x) {
      const long sidx = ((long)b * p.H + h) * p.Nq + c0 + i;
      lse_s[i] = i < rows ? p.LSE[sidx] * 1.44269504088896341f : 1e30f;      // log2 domain; 2^(s - 1e30) == 0 for padded query rows
      del_s[i] = i < rows ? p.Delta[sidx] : 0.f;
    }
    stage_wait();
  };
  if (!CHUNKED && p.debug != 2) {
    stage_q(0);
    __syncthreads();
  }

  const float sl2 = p.scale * 1.44269504088896341f;
  const int nkt = p.debug == 1 ? 0 : (p.Nk + 15) >> 4;
  for (int kt = (CHUNKED ? ychunk * nw + wave : wave) * KT; CHUNKED ? kt >= 0 : kt < nkt; kt = CHUNKED ? -1 : kt + nw * KT) {
    bool kok[KT];
    bf16x8 kf[KT][KS], vf[KT][VS];
    f32x4 dk[KT][QC], dv[KT][VC];
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int key = (kt + u) * 16 + fr;
      kok[u] = key < p.Nk;
      const int kc = kok[u] ? key : p.Nk - 1;       // a tile past the end works on a clamped row and stores nothing
      const bf16_t* krow = p.K + b * p.k_bs + (long)kc * p.k_rs + h * DQK;
      const bf16_t* vrow = p.V + b * p.v_bs + (long)kc * p.v_rs + h * DV;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) kf[u][kk] = gfrag(krow, kk * 32 + 8 * g, kk * 32 + 8 * g < DQK);
#pragma unroll
      for (int kk = 0; kk < VS; ++kk) vf[u][kk] = gfrag(vrow, kk * 32 + 8 * g, kk * 32 + 8 * g < DV);
#pragma unroll
      for (int c = 0; c < QC; ++c) dk[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < VC; ++c) dv[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int c0 = 0; c0 < Nqp; c0 += CH) {
    if (CHUNKED) {
      __syncthreads();
      stage_q(c0);
      __syncthreads();
    }
    const int cend = c0 + CH < Nqp ? c0 + CH : Nqp;
    uint32_t qa_[KS], oa[VS], ota[VC], qta[QC];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) qa_[kk] = lds_addr(Qs) + frag_off<QRB>(fr, kk, g);
#pragma unroll
    for (int kk = 0; kk < VS; ++kk) oa[kk] = lds_addr(dOs) + frag_off<ORB>(fr, kk, g);
#pragma unroll
    for (int c = 0; c < VC; ++c) ota[c] = lds_addr(dOs) + frag_tr_off<ORB>(c * 16, lane);
#pragma unroll
    for (int c = 0; c < QC; ++c) qta[c] = lds_addr(Qs) + frag_tr_off<QRB>(c * 16, lane);
    uint32_t la = lds_addr(lse_s) + 16 * g, da = lds_addr(del_s) + 16 * g;     // this lane's four query rows' statistics
    auto query_loop = [&](auto plain_c) {          // (as in the dQ kernel: the no-bias copy of the loop is picked once per walk)
    constexpr bool PLAIN = decltype(plain_c)::value;
    for (int qa = c0; qa < cend; qa += 32) {
      f32x4 s[KT][2], dp[KT][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        bf16x8 qf[KS], dof[VS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[kk] = lds_frag(qa_[kk] + t * 16 * QRB);
#pragma unroll
        for (int kk = 0; kk < VS; ++kk) dof[kk] = lds_frag(oa[kk] + t * 16 * ORB);
        const f32x4 lse4 = lds_f4(la + t * 64), del4 = lds_f4(da + t * 64);
#pragma unroll
        for (int u = 0; u < KT; ++u) {
          s[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          dp[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kk = 0; kk < KS; ++kk)
            s[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[kk], kf[u][kk], s[u][t], 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < VS; ++kk)
            dp[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[kk], vf[u][kk], dp[u][t], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < KT; ++u) {
          if constexpr (PLAIN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][t][r], sl2, -lse4[r]));
              s[u][t][r] = pr;                                 // P[q][key]
              dp[u][t][r] = pr * (dp[u][t][r] - del4[r]);      // dS[q][key]
            }
            continue;
          }
          float bb[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) {                                      // bias[q][this lane's key]: four query rows, one column
            const int key = (kt + u) * 16 + fr, kc = key < p.Nk ? key : p.Nk - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qg = qa + t * 16 + 4 * g + r;        // row in the head (chunk offset included)
              bb[r] = qg < p.Nq ? p.bias[(((long)(b % p.bias_nb) * p.H + h) * p.Nq + qg) * p.bias_ld + kc] : 0.f;
            }
          }
          float km[4] = {1.f, 1.f, 1.f, 1.f};                // attention dropout: keep[q][this lane's key], four query rows
          if constexpr (DROP) {
            const int key = (kt + u) * 16 + fr, kc = key < p.Nk ? key : p.Nk - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qg = qa + t * 16 + 4 * g + r;
              km[r] = (qg < p.Nq && p.keep[(((long)b * p.H + h) * p.Nq + qg) * p.keep_ld + kc]) ? p.keep_scale : 0.f;
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][t][r], sl2, bb[r] - lse4[r]));
            s[u][t][r] = pr * km[r];                                 // (dropped) P[q][key]: what dV contracts with
            dp[u][t][r] = pr * (dp[u][t][r] * km[r] - del4[r]);      // dS[q][key]
          }
        }
      }
      bf16x8 pf[KT], dsf[KT];
#pragma unroll
      for (int u = 0; u < KT; ++u) {
        pf[u] = pack8(s[u][0], s[u][1]);
        dsf[u] = pack8(dp[u][0], dp[u][1]);
      }
#pragma unroll
      for (int c = 0; c < VC; ++c) {
        const bf16x8 dot = lds_frag_tr<ORB>(ota[c]);
#pragma unroll
        for (int u = 0; u < KT; ++u) dv[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[u], dv[u][c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < QC; ++c) {
        const bf16x8 qt = lds_frag_tr<QRB>(qta[c]);
#pragma unroll
        for (int u = 0; u < KT; ++u) dk[u][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[u], dk[u][c], 0, 0, 0);
      }
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) qa_[kk] += 32 * QRB;
#pragma unroll
      for (int kk = 0; kk < VS; ++kk) oa[kk] += 32 * ORB;
#pragma unroll
      for (int c = 0; c < VC; ++c) ota[c] += 32 * ORB;
#pragma unroll
      for (int c = 0; c < QC; ++c) qta[c] += 32 * QRB;
      la += 128; da += 128;
    }
    };
    if (!DROP && !p.bias) query_loop(std::true_type{}); else query_loop(std::false_type{});
    }
#pragma unroll
    for (int u = 0; u < KT; ++u) {
      const int key = (kt + u) * 16 + fr;
      if (kok[u]) {
        bf16_t* dkrow = p.dK + b * p.dk_bs + (long)key * p.dk_rs + h * DQK;
        bf16_t* dvrow = p.dV + b * p.dv_bs + (long)key * p.dv_rs + h * DV;
#pragma unroll
        for (int c = 0; c < QC; ++c) {
          uint2 w;
          w.x = pack2bf(dk[u][c][0] * p.scale, dk[u][c][1] * p.scale);
          w.y = pack2bf(dk[u][c][2] * p.scale, dk[u][c][3] * p.scale);
          *reinterpret_cast<uint2*>(dkrow + c * 16 + 4 * g) = w;
        }
#pragma unroll
        for (int c = 0; c < VC; ++c) {
          uint2 w;
          w.x = pack2bf(dv[u][c][0], dv[u][c][1]);
          w.y = pack2bf(dv[u][c][2], dv[u][c][3]);
          *reinterpret_cast<uint2*>(dvrow + c * 16 + 4 * g) = w;
        }
      }
    }
  }
}

// Head pairing for narrow heads (round 3): with d = 32 a head's K / V / Q rows are 64-byte pieces of the fused qkv rows — every
// 128-byte line is wanted by the workgroups of heads 2j and 2j+1.  Workgroup x runs on XCD x % 8, so neighbours in x never share
// an L2; this permutation of the linear (batch, head) index puts heads 2j and 2j+1 on ids x and x + 8 — the same XCD, dispatched
// back to back — and the second request of a line meets the first in that XCD's L2.  A bijection on [0, n) for n % 16 == 0.
__device__ __forceinline__ int pair_heads(int x, int n) {
  if (n & 15) return x;
  return (((x >> 4) << 3) + (x & 7)) * 2 + ((x >> 3) & 1);
}

// (round 4 tried the backward as ONE kernel from one recomputation of the probabilities — a wave owning three key tiles, walking the
// query tiles once, dS transposed through an LDS patch, the waves' partial dQ tiles reduced per 32-query step: correct and no faster
// (352 x 352: 113 vs 117 us, 228 x 228: 72 vs 60 us, profiles/r04_attn_fused_bwd.txt): halving the exp / fma work moves the bound to the
// LDS pipe at one workgroup per CU.  Removed in round 5; DESIGN_HISTORY section 11.)

// ---- kernels: one grid per problem, or (resident variants) several problems in one grid (batch.h) ------------------
template <int DQK, int DV, bool CHUNKED, int QT, bool DROP = false>
__global__ __launch_bounds__(512) void attn_fwd_kernel(AttnParams p) {
  attn_fwd_body<DQK, DV, CHUNKED, QT, DROP>(p, (DQK <= 32 && p.pair) ? pair_heads(blockIdx.x, gridDim.x) : (int)blockIdx.x, blockIdx.y);
}
// own tiles per wave in the RESIDENT backward kernels.  Two tiles per wave halve the LDS fragment traffic per MFMA; measured
// on the decoders' d = 32, 228 / 352-row problems it changes nothing (73.5 / 131 us vs 73.3 / 126 us): those kernels are bound
// by the softmax recomputation on the VALU (8 v_exp_f32 + 30 plain slots per 32 x 16 scores in EACH of the two kernels), not by
// the LDS pipe, and 11 tile pairs balance worse over the waves than 22 tiles.  Kept at one; the bodies take the tile count.
template <int DQK, int DV> constexpr int bwd_tiles() { return 1; }
#ifndef DKV32_MIN_BLOCKS
#define DKV32_MIN_BLOCKS 4      // HIP's second __launch_bounds__ argument = WAVES PER SIMD: 4 -> 128 registers and two 8-wave workgroups per CU
                                // (the d = 32 dK / dV body wants 130: 2 spilled, 12 bytes of scratch per lane); 3 -> no spill, ONE workgroup per CU
#endif
template <int DQK, int DV, bool CHUNKED, bool DROP = false>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(AttnParams p) {
  attn_bwd_dq_body<DQK, DV, CHUNKED, CHUNKED ? 1 : bwd_tiles<DQK, DV>(), DROP>(p, (DQK <= 32 && p.pair) ? pair_heads(blockIdx.x, gridDim.x) : (int)blockIdx.x, blockIdx.y);
}
// (narrow heads: 4 waves per SIMD = two 8-wave workgroups per CU — at 130 registers only ONE fitted, 2 waves per SIMD under a VALU-bound loop)
template <int DQK, int DV, bool CHUNKED, bool DROP = false>
__global__ __launch_bounds__(512, (DQK <= 32 && DV <= 32 && !DROP) ? DKV32_MIN_BLOCKS : 1) void attn_bwd_dkv_kernel(AttnParams p) {
  attn_bwd_dkv_body<DQK, DV, CHUNKED, CHUNKED ? 1 : bwd_tiles<DQK, DV>(), DROP>(p, (DQK <= 32 && p.pair) ? pair_heads(blockIdx.x, gridDim.x) : (int)blockIdx.x, blockIdx.y);
}

constexpr int ATTN_GROUP_MAX = 8;
struct AttnGroup {
  AttnParams prob[ATTN_GROUP_MAX];
  int first_block[ATTN_GROUP_MAX + 1];
  int count;
};
// WHICH: 0 forward, 1 dQ, 2 dK/dV.  The workgroup size is the largest any problem of the group asks for: the
// surplus waves of a smaller problem stage tiles and then find no tile of their own.
template <int DQK, int DV, int WHICH>
__global__ __launch_bounds__(512, (WHICH == 2 && DQK <= 32 && DV <= 32) ? DKV32_MIN_BLOCKS : 1) void attn_grouped_kernel(const AttnGroup g) {
  int pi = 0;
  while (pi + 1 < g.count && (int)blockIdx.x >= g.first_block[pi + 1]) ++pi;
  int bh = (int)blockIdx.x - g.first_block[pi];
  if (DQK <= 32 && g.prob[pi].pair && !(g.first_block[pi] & 7)) bh = pair_heads(bh, g.first_block[pi + 1] - g.first_block[pi]);
  // (a copy in registers: read through a reference into the by-value table, every field the loops use came back as a scalar load
  // + wait per iteration — the "memory" clobbers of the staging waits forbid hoisting them)
  const AttnParams p = g.prob[pi];
  if (WHICH == 0) attn_fwd_body<DQK, DV, false, 1>(p, bh, 0);
  else if (WHICH == 1) attn_bwd_dq_body<DQK, DV, false, bwd_tiles<DQK, DV>()>(p, bh, 0);
  else attn_bwd_dkv_body<DQK, DV, false, bwd_tiles<DQK, DV>()>(p, bh, 0);
}

template <int DQK> constexpr int padqk() { return DQK < 32 ? 32 : DQK; }      // LDS columns of a q/k (or v/dO) row

// resident-in-LDS variant up to this many bytes (2 workgroups per CU), chunked beyond
constexpr size_t ATTN_RESIDENT_MAX = 80 * 1024;

template <auto KERN>
int raise_lds_cap(size_t lds) {
  static bool done = false;            // once per kernel (160 KiB on gfx950)
  if (lds > 64 * 1024 && !done) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)KERN, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  return 0;
}

inline int waves_for(int rows) { int nw = (rows + 15) / 16; return nw > 8 ? 8 : (nw < 1 ? 1 : nw); }
// waves of a resident workgroup whose waves own `per` 16-row tiles at a time: the fewest rounds 8 waves allow, then the
// fewest waves that still finish in that many rounds (11 tile pairs -> 2 rounds -> 6 waves, not 8 of which 5 idle in round 2)
inline int waves_for_tiles(int rows, int per) {
  const int units = ((rows + 15) / 16 + per - 1) / per, rounds = (units + 7) / 8;
  const int nw = (units + rounds - 1) / rounds;
  return nw < 1 ? 1 : nw;
}

// geometry of the resident (all keys / all queries of a head in LDS) variants
template <int DQK, int DV> constexpr size_t attn_row_bytes() { return (size_t)padqk<DQK>() * 2 + (size_t)padqk<DV>() * 2; }
template <int DQK, int DV, int WHICH> size_t attn_lds(const AttnParams& p) {
  const int Nkp = (p.Nk + 31) & ~31, Nqp = (p.Nq + 31) & ~31;
  return WHICH == 2 ? Nqp * (attn_row_bytes<DQK, DV>() + 8) : Nkp * attn_row_bytes<DQK, DV>();
}
template <int DQK, int DV, int WHICH> int attn_waves(const AttnParams& p) {
  return WHICH == 0 ? waves_for(p.Nq) : waves_for_tiles(WHICH == 2 ? p.Nk : p.Nq, bwd_tiles<DQK, DV>());
}

// issues n >= 1 recorded resident-variant problems of one (head widths, pass) family: davb::GroupFn
template <int DQK, int DV, int WHICH>
void attn_issue(const void* const* params_in, int n, hipStream_t stream) {
  // longest sequences first: a (batch, head) workgroup's run time grows with Nq * Nk and workgroups start in block order
  std::vector<const void*> sorted(params_in, params_in + n);
  std::stable_sort(sorted.begin(), sorted.end(), [](const void* a, const void* b) {
    const AttnParams &x = *(const AttnParams*)a, &y = *(const AttnParams*)b;
    return (long)x.Nq * x.Nk > (long)y.Nq * y.Nk;
  });
  const void* const* params = sorted.data();
  auto single = [&](const AttnParams& p) {
    const size_t lds = attn_lds<DQK, DV, WHICH>(p);
    const int nw = attn_waves<DQK, DV, WHICH>(p);
    if (WHICH == 0) {
      (void)raise_lds_cap<attn_fwd_kernel<DQK, DV, false, 1>>(lds);
      DAV_LAUNCH_NOW((attn_fwd_kernel<DQK, DV, false, 1>), dim3(p.B * p.H), dim3(nw * 64), lds, stream, p);
    } else if (WHICH == 1) {
      (void)raise_lds_cap<attn_bwd_dq_kernel<DQK, DV, false>>(lds);
      DAV_LAUNCH_NOW((attn_bwd_dq_kernel<DQK, DV, false>), dim3(p.B * p.H), dim3(nw * 64), lds, stream, p);
    } else {
      (void)raise_lds_cap<attn_bwd_dkv_kernel<DQK, DV, false>>(lds);
      DAV_LAUNCH_NOW((attn_bwd_dkv_kernel<DQK, DV, false>), dim3(p.B * p.H), dim3(nw * 64), lds, stream, p);
    }
  };
  for (int base = 0; base < n; base += ATTN_GROUP_MAX) {
    const int cnt = n - base < ATTN_GROUP_MAX ? n - base : ATTN_GROUP_MAX;
    if (cnt == 1) {
      single(*(const AttnParams*)params[base]);
      continue;
    }
    AttnGroup g;
    int first = 0, nw = 1;
    size_t lds = 0;
    for (int i = 0; i < cnt; ++i) {
      g.prob[i] = *(const AttnParams*)params[base + i];
      const size_t l = attn_lds<DQK, DV, WHICH>(g.prob[i]);
      lds = l > lds ? l : lds;
      const int w = attn_waves<DQK, DV, WHICH>(g.prob[i]);
      nw = w > nw ? w : nw;
      g.first_block[i] = first;
      first += g.prob[i].B * g.prob[i].H;
    }
    g.first_block[cnt] = first;
    g.count = cnt;
    (void)raise_lds_cap<attn_grouped_kernel<DQK, DV, WHICH>>(lds);
    DAV_LAUNCH_NOW((attn_grouped_kernel<DQK, DV, WHICH>), dim3(first), dim3(nw * 64), lds, stream, g);
  }
}

template <int DQK, int DV, int WHICH>
void attn_resident(const AttnParams& p, hipStream_t stream) {
  if (davb::recording()) {
    davb::push_typed(attn_issue<DQK, DV, WHICH>, &p, sizeof(p), stream);
    return;
  }
  const void* one = &p;
  attn_issue<DQK, DV, WHICH>(&one, 1, stream);
}

template <int DQK, int DV>
int launch_fwd(const AttnParams& p, hipStream_t stream) {
  const size_t row = attn_row_bytes<DQK, DV>();
  const size_t lds = attn_lds<DQK, DV, 0>(p);
  const bool two = dav_attn_qt == 2 || (dav_attn_qt == 0 && p.Nq >= 1024);      // two query tiles per wave
  const int nw = waves_for(two ? (p.Nq + 1) / 2 : p.Nq);
  if (lds <= ATTN_RESIDENT_MAX && !two) {
    attn_resident<DQK, DV, 0>(p, stream);
  } else if (lds <= ATTN_RESIDENT_MAX) {
    if (int rc = raise_lds_cap<attn_fwd_kernel<DQK, DV, false, 2>>(lds)) return rc;
    DAV_LAUNCH((attn_fwd_kernel<DQK, DV, false, 2>), dim3(p.B * p.H), dim3(nw * 64), lds, stream, p);
  } else {
    auto kern = two ? attn_fwd_kernel<DQK, DV, true, 2> : attn_fwd_kernel<DQK, DV, true, 1>;
    const size_t ldc = ATTN_CHUNK * row;
    if (int rc = raise_lds_cap<attn_fwd_kernel<DQK, DV, true, 1>>(ldc)) return rc;
    if (int rc = raise_lds_cap<attn_fwd_kernel<DQK, DV, true, 2>>(ldc)) return rc;
    const int rows_wg = nw * 16 * (two ? 2 : 1);
    DAV_LAUNCH(kern, dim3(p.B * p.H, (p.Nq + rows_wg - 1) / rows_wg), dim3(nw * 64), ldc, stream, p);
  }
  return dav_launch_status();
}

// attention dropout (AttnParams::keep): kernels of their own, one problem per launch (never grouped: fine-tuning only)
template <int DQK, int DV>
int launch_fwd_drop(const AttnParams& p, hipStream_t stream) {
  const size_t lds = attn_lds<DQK, DV, 0>(p);
  const int nw = waves_for(p.Nq);
  if (lds <= ATTN_RESIDENT_MAX) {
    if (int rc = raise_lds_cap<attn_fwd_kernel<DQK, DV, false, 1, true>>(lds)) return rc;
    DAV_LAUNCH((attn_fwd_kernel<DQK, DV, false, 1, true>), dim3(p.B * p.H), dim3(nw * 64), lds, stream, p);
  } else {
    const size_t ldc = ATTN_CHUNK * attn_row_bytes<DQK, DV>();
    if (int rc = raise_lds_cap<attn_fwd_kernel<DQK, DV, true, 1, true>>(ldc)) return rc;
    DAV_LAUNCH((attn_fwd_kernel<DQK, DV, true, 1, true>), dim3(p.B * p.H, (p.Nq + nw * 16 - 1) / (nw * 16)), dim3(nw * 64), ldc, stream, p);
  }
  return dav_launch_status();
}
template <int DQK, int DV>
int launch_bwd_drop(const AttnParams& p, hipStream_t stream, int part) {
  const size_t row = attn_row_bytes<DQK, DV>();
  const size_t lds1 = attn_lds<DQK, DV, 1>(p), lds2 = attn_lds<DQK, DV, 2>(p);
  if (part & 1) {
    if (lds1 <= ATTN_RESIDENT_MAX) {
      const int nw = attn_waves<DQK, DV, 1>(p);
      if (int rc = raise_lds_cap<attn_bwd_dq_kernel<DQK, DV, false, true>>(lds1)) return rc;
      DAV_LAUNCH((attn_bwd_dq_kernel<DQK, DV, false, true>), dim3(p.B * p.H), dim3(nw * 64), lds1, stream, p);
    } else {
      const int nw1 = waves_for(p.Nq);
      const size_t ldc = ATTN_CHUNK * row;
      if (int rc = raise_lds_cap<attn_bwd_dq_kernel<DQK, DV, true, true>>(ldc)) return rc;
      DAV_LAUNCH((attn_bwd_dq_kernel<DQK, DV, true, true>), dim3(p.B * p.H, (p.Nq + nw1 * 16 - 1) / (nw1 * 16)), dim3(nw1 * 64), ldc, stream, p);
    }
  }
  if (part & 2) {
    if (lds2 <= ATTN_RESIDENT_MAX) {
      const int nw = attn_waves<DQK, DV, 2>(p);
      if (int rc = raise_lds_cap<attn_bwd_dkv_kernel<DQK, DV, false, true>>(lds2)) return rc;
      DAV_LAUNCH((attn_bwd_dkv_kernel<DQK, DV, false, true>), dim3(p.B * p.H), dim3(nw * 64), lds2, stream, p);
    } else {
      const int nw2 = waves_for(p.Nk);
      const size_t ldc = ATTN_CHUNK * (row + 8);
      if (int rc = raise_lds_cap<attn_bwd_dkv_kernel<DQK, DV, true, true>>(ldc)) return rc;
      DAV_LAUNCH((attn_bwd_dkv_kernel<DQK, DV, true, true>), dim3(p.B * p.H, (p.Nk + nw2 * 16 - 1) / (nw2 * 16)), dim3(nw2 * 64), ldc, stream, p);
    }
  }
  return dav_launch_status();
}

template <int DQK, int DV>
int launch_bwd(const AttnParams& p, hipStream_t stream, int part = 3) {
  const size_t row = attn_row_bytes<DQK, DV>();
  const size_t lds1 = attn_lds<DQK, DV, 1>(p), lds2 = attn_lds<DQK, DV, 2>(p);
  const int nw1 = waves_for(p.Nq), nw2 = waves_for(p.Nk);
  // dQ first: it also writes Delta, which the dK/dV kernel reads
  if (!(part & 1)) {
  } else if (lds1 <= ATTN_RESIDENT_MAX) {
    attn_resident<DQK, DV, 1>(p, stream);
  } else {
    auto k1 = attn_bwd_dq_kernel<DQK, DV, true>;
    const size_t ldc = ATTN_CHUNK * row;
    if (int rc = raise_lds_cap<attn_bwd_dq_kernel<DQK, DV, true>>(ldc)) return rc;
    DAV_LAUNCH(k1, dim3(p.B * p.H, (p.Nq + nw1 * 16 - 1) / (nw1 * 16)), dim3(nw1 * 64), ldc, stream, p);
  }
  if (!(part & 2)) {
  } else if (lds2 <= ATTN_RESIDENT_MAX) {
    attn_resident<DQK, DV, 2>(p, stream);
  } else {
    auto k2 = attn_bwd_dkv_kernel<DQK, DV, true>;
    const size_t ldc = ATTN_CHUNK * (row + 8);
    if (int rc = raise_lds_cap<attn_bwd_dkv_kernel<DQK, DV, true>>(ldc)) return rc;
    DAV_LAUNCH(k2, dim3(p.B * p.H, (p.Nk + nw2 * 16 - 1) / (nw2 * 16)), dim3(nw2 * 64), ldc, stream, p);
  }
  return dav_launch_status();
}

bool strides_ok(const AttnParams& p, bool bwd) {
  auto a8 = [](long v) { return (v & 7) == 0; };
  bool ok = a8(p.q_bs) && a8(p.k_bs) && a8(p.v_bs) && a8(p.o_bs) && a8(p.q_rs) && a8(p.k_rs) && a8(p.v_rs) && a8(p.o_rs);
  ok = ok && !(((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O) & 15);
  if (bwd) {
    ok = ok && a8(p.do_bs) && a8(p.do_rs) && a8(p.dq_bs) && a8(p.dk_bs) && a8(p.dv_bs) && a8(p.dq_rs) && a8(p.dk_rs) && a8(p.dv_rs);
    ok = ok && !(((uintptr_t)p.dO | (uintptr_t)p.dQ | (uintptr_t)p.dK | (uintptr_t)p.dV) & 15);
  }
  return ok;
}

}  // namespace

static bool bias_ok(const float* bias, int bias_nb, int bias_ld, const float* dS, int B, int Nk) {
  if (!bias) return dS == nullptr;
  const int Nkp = (Nk + 31) & ~31;
  return bias_nb > 0 && B % bias_nb == 0 && bias_ld >= Nkp && !(bias_ld & 3) && !(((uintptr_t)bias | (uintptr_t)dS) & 15);
}

static bool keep_ok(const void* keep, int keep_ld, float keep_scale, int Nk) {
  if (!keep) return true;
  return keep_ld >= ((Nk + 31) & ~31) && !(keep_ld & 3) && !((uintptr_t)keep & 3) && keep_scale > 0.f;
}

static int attn_fwd_any(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq,
                        int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs,
                        long o_bs, int o_rs, float scale, const float* bias, int bias_nb, int bias_ld,
                        const void* keep, int keep_ld, float keep_scale, hipStream_t stream);

extern "C" int dav_attn_bias_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq,
                                 int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs,
                                 long o_bs, int o_rs, float scale, const float* bias, int bias_nb, int bias_ld,
                                 hipStream_t stream) {
  return attn_fwd_any(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale, bias, bias_nb, bias_ld,
                      nullptr, 0, 0.f, stream);
}

extern "C" int dav_attn_drop_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq,
                                 int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs,
                                 long o_bs, int o_rs, float scale, const void* keep, int keep_ld, float keep_scale,
                                 hipStream_t stream) {
  if (!keep) return DAV_ERR_SHAPE;
  return attn_fwd_any(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale, nullptr, 0, 0,
                      keep, keep_ld, keep_scale, stream);
}

static int attn_fwd_any(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq,
                        int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs,
                        long o_bs, int o_rs, float scale, const float* bias, int bias_nb, int bias_ld,
                        const void* keep, int keep_ld, float keep_scale, hipStream_t stream) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0) return DAV_ERR_SHAPE;
  if (!bias_ok(bias, bias_nb, bias_ld, nullptr, B, Nk)) return DAV_ERR_SHAPE;
  if (!keep_ok(keep, keep_ld, keep_scale, Nk)) return ((uintptr_t)keep & 3) ? DAV_ERR_ALIGN : DAV_ERR_SHAPE;
  AttnParams p = {};
  p.keep = (const unsigned char*)keep; p.keep_ld = keep_ld; p.keep_scale = keep_scale;
  p.debug = attn_debug();
  p.pair = 1;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (bf16_t*)O; p.LSE = LSE;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
  p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.o_rs = o_rs; p.scale = scale;
  p.bias = bias; p.bias_nb = bias_nb; p.bias_ld = bias_ld;
  if (!strides_ok(p, false)) return DAV_ERR_ALIGN;
  if (p.keep) {
    if (p.bias) return DAV_ERR_SHAPE;
    if (dqk == 64 && dv == 64) return launch_fwd_drop<64, 64>(p, stream);
    if (dqk == 32 && dv == 32) return launch_fwd_drop<32, 32>(p, stream);
    if (dqk == 16 && dv == 64) return launch_fwd_drop<16, 64>(p, stream);
    if (dqk == 16 && dv == 16) return launch_fwd_drop<16, 16>(p, stream);
    return DAV_ERR_SHAPE;
  }
  if (dqk == 64 && dv == 64) return launch_fwd<64, 64>(p, stream);
  if (dqk == 32 && dv == 32) return launch_fwd<32, 32>(p, stream);
  if (dqk == 16 && dv == 64) return launch_fwd<16, 64>(p, stream);
  if (dqk == 16 && dv == 16) return launch_fwd<16, 16>(p, stream);
  return DAV_ERR_SHAPE;
}

extern "C" int dav_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq,
                            int Nk, int dqk, int dv, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs,
                            long o_bs, int o_rs, float scale, hipStream_t stream) {
  return dav_attn_bias_fwd(Q, K, V, O, LSE, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, scale,
                           nullptr, 0, 0, stream);
}

extern "C" int dav_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                            float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                            long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                            long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                            float scale, hipStream_t stream) {
  return dav_attn_bwd_part(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                           o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, 3, stream);
}

static int attn_bwd_any(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                        float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                        long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                        long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                        float scale, const float* bias, int bias_nb, int bias_ld, float* dS, int dq_ctx_rows, int part,
                        hipStream_t stream, const void* keep = nullptr, int keep_ld = 0, float keep_scale = 0.f) {
  if (B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || part < 1 || part > 3 || dq_ctx_rows < 0) return DAV_ERR_SHAPE;
  if (!bias_ok(bias, bias_nb, bias_ld, dS, B, Nk)) return DAV_ERR_SHAPE;
  if (!keep_ok(keep, keep_ld, keep_scale, Nk)) return ((uintptr_t)keep & 3) ? DAV_ERR_ALIGN : DAV_ERR_SHAPE;
  AttnParams p = {};
  p.keep = (const unsigned char*)keep; p.keep_ld = keep_ld; p.keep_scale = keep_scale;
  p.dq_ctx = dq_ctx_rows;
  p.debug = attn_debug();
  p.pair = 1;
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.Of = (const bf16_t*)O; p.O = nullptr;
  p.dO = (const bf16_t*)dO; p.LSE = const_cast<float*>(LSE); p.Delta = Delta;
  p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk; p.q_bs = q_bs; p.k_bs = k_bs; p.v_bs = v_bs; p.o_bs = o_bs;
  p.q_rs = q_rs; p.k_rs = k_rs; p.v_rs = v_rs; p.o_rs = o_rs; p.do_bs = do_bs; p.do_rs = do_rs;
  p.dq_bs = dq_bs; p.dk_bs = dk_bs; p.dv_bs = dv_bs; p.dq_rs = dq_rs; p.dk_rs = dk_rs; p.dv_rs = dv_rs; p.scale = scale;
  p.bias = bias; p.bias_nb = bias_nb; p.bias_ld = bias_ld; p.dS = dS;
  p.O = (bf16_t*)O;   // only for the alignment check
  if (!strides_ok(p, true)) return DAV_ERR_ALIGN;
  if (p.keep) {
    if (p.bias || p.dS) return DAV_ERR_SHAPE;
    if (dqk == 64 && dv == 64) return launch_bwd_drop<64, 64>(p, stream, part);
    if (dqk == 32 && dv == 32) return launch_bwd_drop<32, 32>(p, stream, part);
    if (dqk == 16 && dv == 64) return launch_bwd_drop<16, 64>(p, stream, part);
    if (dqk == 16 && dv == 16) return launch_bwd_drop<16, 16>(p, stream, part);
    return DAV_ERR_SHAPE;
  }
  if (dqk == 64 && dv == 64) return launch_bwd<64, 64>(p, stream, part);
  if (dqk == 32 && dv == 32) return launch_bwd<32, 32>(p, stream, part);
  if (dqk == 16 && dv == 64) return launch_bwd<16, 64>(p, stream, part);
  if (dqk == 16 && dv == 16) return launch_bwd<16, 16>(p, stream, part);
  return DAV_ERR_SHAPE;
}

extern "C" int dav_attn_bias_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                 float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                 long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                                 long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                                 float scale, const float* bias, int bias_nb, int bias_ld, float* dS, int part,
                                 hipStream_t stream) {
  return attn_bwd_any(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                      do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, bias, bias_nb, bias_ld, dS, 0, part, stream);
}

extern "C" int dav_attn_bwd_ctx(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                                long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                                float scale, int dq_ctx_rows, int part, hipStream_t stream) {
  return attn_bwd_any(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                      do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, nullptr, 0, 0, nullptr, dq_ctx_rows, part, stream);
}

extern "C" int dav_attn_drop_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                 float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                 long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                                 long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                                 float scale, const void* keep, int keep_ld, float keep_scale, int dq_ctx_rows, int part,
                                 hipStream_t stream) {
  if (!keep) return DAV_ERR_SHAPE;
  return attn_bwd_any(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs,
                      do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, nullptr, 0, 0, nullptr, dq_ctx_rows, part, stream,
                      keep, keep_ld, keep_scale);
}

extern "C" int dav_attn_bwd_part(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE,
                                 float* Delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int dqk, int dv,
                                 long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                                 long do_bs, int do_rs, long dq_bs, int dq_rs, long dk_bs, int dk_rs, long dv_bs, int dv_rs,
                                 float scale, int part, hipStream_t stream) {
  return dav_attn_bias_bwd(Q, K, V, O, dO, LSE, Delta, dQ, dK, dV, B, H, Nq, Nk, dqk, dv, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs,
                           o_bs, o_rs, do_bs, do_rs, dq_bs, dq_rs, dk_bs, dk_rs, dv_bs, dv_rs, scale, nullptr, 0, 0, nullptr,
                           part, stream);
}
