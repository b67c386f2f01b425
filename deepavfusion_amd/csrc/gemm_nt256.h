// 256 x 256 x 64 NT GEMM body for gfx950 (tile configuration 60), included by gemm.hip inside its anonymous namespace.
//
// Why this shape: the GEMMs of the step are POWER-bound on MI355X, not issue-bound — the same binary runs 30 % faster on
// zero-filled operands than on random ones (DVFS), and across every schedule tried the run time is the SUM of what the MFMAs,
// the LDS fragment reads and the global -> LDS stream cost alone.  What counts is energy per flop, i.e. bytes moved per flop:
// a 256 x 256 tile streams half the operand bytes of the 128 x 128 one and reads 3/4 of the LDS bytes
// (tools/micro/gemm256.hip, profiles/r03_gemm256_*.txt).
//
// Structure ("ping-pong", one workgroup of 8 waves per CU, 128 KB of LDS):
//   * waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs; per K-tile (64 deep) a wave walks the four
//     64 x 32 quadrants of its 128 x 64 output (rows {h 128 + wr 64 ..}, columns {h 128 + wc 32 ..}, h = 0, 1) in the order
//     (0,0) (0,1) (1,1) (1,0): 8 x v_mfma_f32_32x32x16_bf16 per quadrant, fragments of 12 / 4 / 8 / 0 ds_read_b128 —
//     the B-h0 fragments stay in registers for the fourth quadrant;
//   * group 0 runs  MFMAs(i) ; fragment reads(i+1) ; DMA | barrier,  group 1  fragment reads(i) ; DMA ; MFMAs(i) | barrier:
//     on every SIMD one wave's loads run under the other wave's MFMAs, ONE barrier per quadrant;
//   * LDS = 2 buffers x {A-h0, A-h1, B-h0, B-h1} half-tiles of 16 KB (128 rows x 64 k, 16-byte slots XOR-swizzled by
//     (row >> 1) & 7 on the SOURCE address of the LDS-DMA, which writes lane-linear); every interval restages the half-tile
//     whose last reads retired one barrier earlier, three half-tiles (48 KB) stay in flight (s_waitcnt vmcnt(6), never 0);
//   * b_kn mode (BT): B is given as [K, N] (the input-gradient GEMM reading W itself): its half-tiles are [64 k][128 n] images
//     (256-byte rows, 32-byte granules XOR-swizzled by 2 (k & 3)) read with the transposing ds_read_b64_tr_b16.
//   * accumulators come out transposed (operands swapped): lane l holds C[m = l & 31][n = 8 g + 4 (l >> 5) + 0..3], g = 0..3,
//     so every epilogue access is 16 bytes per lane (bf16 outputs pair two column groups through v_permlane32_swap).
// Requires K % 128 == 0, N % 8 == 0 and 16-byte aligned rows of every output / auxiliary operand (nt256_ok).
#pragma once

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int NT256_HT = 16384;            // bytes per half-tile slot
constexpr int NT256_BREG = 65536;          // B region (A region at 0); slot(d, h) = d * 32768 + h * 16384
constexpr size_t NT256_LDS = 131072;

__host__ __device__ inline bool nt256_ok(const NTParams& p) {
  if ((p.K & 127) || (p.N & 7)) return false;
  if (p.ln_st || p.st_out || p.tw_out || p.a_r0) return false;      // LayerNorm folded in: the 128-wide bodies only
  if (p.C && (p.ldc & (p.c_bf16 ? 7 : 3))) return false;
  if (p.C2 && (p.ldc2 & 7)) return false;
  if (p.res && (p.ldres & 3)) return false;
  if ((p.act == 2 || p.act == 3) && (p.ldaux & 7)) return false;
  if (((uintptr_t)p.C | (uintptr_t)p.C2 | (uintptr_t)p.res | (uintptr_t)p.aux) & 15) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  return true;
}

// Epilogue kinds (compile-time specialisations of the one epilogue below; the generic form is 9000 instructions — more than the
// instruction cache — and cost 15 % on the K = 512 GEMMs):  0 generic (everything nt_epilogue_t does),  1 bf16 C = alpha acc + bias,
// 2 bf16 C = GELU(alpha acc + bias) with the GELU' twin in C2 (c2_mode 4),  3 bf16 C = (alpha acc + bias) * aux (act 3)
__host__ __device__ inline int nt256_kind(const NTParams& p) {
  if (!p.C || !p.c_bf16 || p.res || p.beta) return 0;
  if (p.act == 0 && p.c2_mode == 0) return 1;
  if (p.act == 1 && p.c2_mode == 4) return 2;
  if (p.act == 3 && p.c2_mode == 0) return 3;
  return 0;
}

template <bool BT, int EK>
__device__ __forceinline__ void nt256_body(const NTParams& p, int bid) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + 255) >> 8;
  {
    const int nwg = tiles_m * tiles_n;
    if (bid >= nwg) return;                 // grouped launches pad every problem's block range to a multiple of 8
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // an XCD's contiguous range of tile ids walks panels of <= 4 tile columns row by row: its A rows and B columns stay in that L2
  int bm, bn;
  {
    const int npan = (tiles_n + 3) >> 2;
    const int wn_ = (tiles_n + npan - 1) / npan;
    const int per = tiles_m * wn_;
    const int pnl = bid / per, rem = bid - pnl * per;
    const int wp = (tiles_n - pnl * wn_) < wn_ ? (tiles_n - pnl * wn_) : wn_;
    bm = rem / wp;
    bn = pnl * wn_ + rem % wp;
  }
  const int m0 = bm << 8, n0 = bn << 8;
  const int nk = p.K >> 6;

  // ---- LDS-DMA sources: two 1 KB pieces per wave and half-tile
  const bf16_t* a_src[2][2];
  const bf16_t* b_src[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      {
        const int r = (wave * 2 + e) * 8 + (lane >> 3);                  // row of the [128][64 k] image, 16-byte slot lane & 7
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + h * 128 + r; gm = gm < p.M ? gm : p.M - 1;
        a_src[h][e] = p.A + map_row(gm, p.amap) * p.lda + c * 8;
        if (!BT) {
          int gn = n0 + h * 128 + r; gn = gn < p.N ? gn : p.N - 1;
          b_src[h][e] = p.B + (long)gn * p.ldb + c * 8;
        }
      }
      if (BT) {
        const int kr = (wave * 2 + e) * 4 + (lane >> 4);                 // k row of the [64 k][128 n] image, 16-byte chunk lane & 15
        const int c16 = lane & 15;
        const int g = (c16 >> 1) ^ (2 * (kr & 3));
        int gc = n0 + h * 128 + g * 16 + (c16 & 1) * 8; gc = gc < p.N ? gc : p.N - 8;
        b_src[h][e] = p.B + (long)kr * p.ldb + gc;
      }
    }
  auto issue = [&](bool isA, int h, int d, int kt) {
    kt = kt < nk ? kt : nk - 1;                         // past the end: harmless reload into a free slot (keeps the vmcnt counts uniform)
    const long ko = isA ? (long)kt * 64 : (BT ? (long)kt * 64 * p.ldb : (long)kt * 64);
    char* slot = smem + (isA ? 0 : NT256_BREG) + d * 32768 + h * NT256_HT + wave * 2048;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, (isA ? a_src[h][e] : b_src[h][e]) + ko), LDS_PTR(void, slot + e * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets
  const int sw = (lane >> 1) & 7, hi = lane >> 5;
  uint32_t a_lo[4], b_lo[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_lo[ks] = (wr * 64 + (lane & 31)) * 128 + (((ks * 2 + hi) ^ sw) << 4);
    b_lo[ks] = NT256_BREG + (wc * 32 + (lane & 31)) * 128 + (((ks * 2 + hi) ^ sw) << 4);
  }
  uint32_t bt_lo = 0;
  if (BT) {
    const int gq = lane >> 4, li = lane & 15;
    bt_lo = NT256_BREG + (8 * (gq >> 1) + (li >> 2)) * 256 + ((((wc * 2 + (gq & 1)) ^ (2 * (li >> 2))) << 5) | (8 * (li & 3)));
  }

  f32x16 acc[2][2][2];   // [qm][qn][rf]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][r][e] = 0.f;

  bf16x8 a[2][4], b0[4], b1[4];
  auto read_a = [&](int d, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf)
        a[rf][ks] = *reinterpret_cast<const bf16x8*>(smem + a_lo[ks] + d * 32768 + h * NT256_HT + rf * 4096);
  };
  auto read_b = [&](bf16x8 (&bb)[4], int d, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (!BT) {
        bb[ks] = *reinterpret_cast<const bf16x8*>(smem + b_lo[ks] + d * 32768 + h * NT256_HT);
      } else {
        union { s16x4 h2[2]; bf16x8 v; } u;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          u.h2[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, smem + bt_lo + d * 32768 + h * NT256_HT + ks * 4096 + h2 * 1024));
        bb[ks] = u.v;
      }
    }
  };
  auto mfmas = [&](int qm, int qn) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf)
        acc[qm][qn][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qn ? b1[ks] : b0[ks], a[rf][ks], acc[qm][qn][rf], 0, 0, 0);
  };
  // interval (t, i) restages: i = 0 (t+1).A-h1 -> buffer D^1;  i = 1 (t+2).A-h0 -> D;  i = 2 (t+2).B-h0 -> D;  i = 3 (t+2).B-h1 -> D
  auto issue_for = [&](int i, int d, int t) {
    if (i == 0) issue(true, 1, d ^ 1, t + 1);
    if (i == 1) issue(true, 0, d, t + 2);
    if (i == 2) issue(false, 0, d, t + 2);
    if (i == 3) issue(false, 1, d, t + 2);
  };

  // prologue: K-tile 0 complete, three half-tiles of K-tile 1 (the issue slots of the "intervals" -7 .. -1)
  issue(true, 0, 0, 0); issue(false, 0, 0, 0); issue(false, 1, 0, 0); issue(true, 1, 0, 0);
  issue(true, 0, 1, 1); issue(false, 0, 1, 1); issue(false, 1, 1, 1);
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

#define NT256_SB __builtin_amdgcn_sched_barrier(0)
  if (wr == 0) {
    // ---- group 0: M_i, then the fragment reads of quadrant i+1
    read_a(0, 0); read_b(b0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NT256_SB;
#define NT256_G0(I, D, T)                                                            \
    {                                                                                \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      NT256_SB;                                                                      \
      if ((I) == 0) read_b(b1, D, 1);                                                \
      if ((I) == 1) read_a(D, 1);                                                    \
      if ((I) == 3) { read_a((D) ^ 1, 0); read_b(b0, (D) ^ 1, 0); }                  \
      NT256_SB;                                                                      \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                    \
      NT256_SB;                                                                      \
      __builtin_amdgcn_s_barrier();                                                  \
      NT256_SB;                                                                      \
    }
    for (int t = 0; t < nk; t += 2) {
      NT256_G0(0, 0, t) NT256_G0(1, 0, t) NT256_G0(2, 0, t) NT256_G0(3, 0, t)
      NT256_G0(0, 1, t + 1) NT256_G0(1, 1, t + 1) NT256_G0(2, 1, t + 1) NT256_G0(3, 1, t + 1)
    }
#undef NT256_G0
  } else {
    // ---- group 1: the fragment reads of quadrant i, then M_i
#define NT256_G1(I, D, T)                                                            \
    {                                                                                \
      if ((I) == 0) { read_a(D, 0); read_b(b0, D, 0); }                              \
      if ((I) == 1) read_b(b1, D, 1);                                                \
      if ((I) == 2) read_a(D, 1);                                                    \
      NT256_SB;                                                                      \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
      NT256_SB;                                                                      \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      NT256_SB;                                                                      \
      wait_vmcnt<6>();                                                               \
      NT256_SB;                                                                      \
      __builtin_amdgcn_s_barrier();                                                  \
      NT256_SB;                                                                      \
    }
    for (int t = 0; t < nk; t += 2) {
      NT256_G1(0, 0, t) NT256_G1(1, 0, t) NT256_G1(2, 0, t) NT256_G1(3, 0, t)
      NT256_G1(0, 1, t + 1) NT256_G1(1, 1, t + 1) NT256_G1(2, 1, t + 1) NT256_G1(3, 1, t + 1)
    }
#undef NT256_G1
  }
#undef NT256_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (discarded) reloads have landed: nothing of this workgroup writes LDS any more

  // ---- epilogue, straight from the registers: 16 bytes per lane and access.  Same semantics and order as nt_epilogue_t.
  constexpr bool GEN = EK == 0;
  const int act = GEN ? p.act : (EK == 2 ? 1 : EK == 3 ? 3 : 0);
  const int c2m = GEN ? p.c2_mode : (EK == 2 ? 4 : 0);
  const bool has_aux = (act == 2 || act == 3), has_res = GEN && p.res, c_bf16 = GEN ? (p.c_bf16 != 0) : true, beta = GEN && p.beta;
  const bool has_c = GEN ? (p.C != nullptr) : true;
#pragma unroll
  for (int qn = 0; qn < 2; ++qn) {
    const int nb = n0 + qn * 128 + wc * 32 + 4 * hi;      // + 8 g
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      bv[g] = (p.bias && nb + 8 * g < p.N) ? *reinterpret_cast<const float4*>(p.bias + nb + 8 * g) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const int m = m0 + qm * 128 + wr * 64 + rf * 32 + (lane & 31);
        const bool mok = m < p.M;
        const int mc = mok ? m : p.M - 1;
        const long crow = has_c ? map_row(mc, p.cmap) : 0;
        const long rrow = has_res ? (p.res_rows ? (long)p.res_rows[mc] : map_row(mc, p.rmap)) : 0;
        // all global reads of the fragment row go out before the first use
        float4 rv[4], ov[4];
        uint2 av[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          const bool ok = mok && n < p.N;
          rv[g] = (has_res && ok) ? *reinterpret_cast<const float4*>(p.res + rrow * p.ldres + n) : float4{0.f, 0.f, 0.f, 0.f};
          av[g] = (has_aux && ok) ? *reinterpret_cast<const uint2*>(p.aux + (long)mc * p.ldaux + n) : uint2{0, 0};
          ov[g] = (beta && ok) ? *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + crow * p.ldc + n) : float4{0.f, 0.f, 0.f, 0.f};
        }
        uint2 wc1[4], wc2[4];                                 // packed bf16 of C and of the twin C2, four columns per group
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nb + 8 * g;
          const bool ok = mok && n < p.N;
          const f32x16& q = acc[qm][qn][rf];
          float4 v;
          v.x = q[4 * g + 0] * p.alpha + bv[g].x; v.y = q[4 * g + 1] * p.alpha + bv[g].y;
          v.z = q[4 * g + 2] * p.alpha + bv[g].z; v.w = q[4 * g + 3] * p.alpha + bv[g].w;
          uint2 w2 = uint2{0, 0};
          if (c2m == 1) { w2.x = pack2bf(v.x, v.y); w2.y = pack2bf(v.z, v.w); }
          if (act == 1) {
            float4 d;
            gelu_pair_f(v.x, v.x, d.x); gelu_pair_f(v.y, v.y, d.y); gelu_pair_f(v.z, v.z, d.z); gelu_pair_f(v.w, v.w, d.w);
            if (c2m == 4) { w2.x = pack2bf(d.x, d.y); w2.y = pack2bf(d.z, d.w); }
          } else if (has_aux) {
            float4 gq;
            gq.x = __uint_as_float(av[g].x << 16); gq.y = __uint_as_float(av[g].x & 0xffff0000u);
            gq.z = __uint_as_float(av[g].y << 16); gq.w = __uint_as_float(av[g].y & 0xffff0000u);
            if (act == 2) { gq.x = gelu_grad_f(gq.x); gq.y = gelu_grad_f(gq.y); gq.z = gelu_grad_f(gq.z); gq.w = gelu_grad_f(gq.w); }
            v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w;
          }
          if (c2m == 2) { w2.x = pack2bf(v.x, v.y); w2.y = pack2bf(v.z, v.w); }
          if (has_res) { v.x += rv[g].x; v.y += rv[g].y; v.z += rv[g].z; v.w += rv[g].w; }
          if (has_c && !c_bf16) {
            if (beta) { v.x += ov[g].x; v.y += ov[g].y; v.z += ov[g].z; v.w += ov[g].w; }
            if (ok) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + crow * p.ldc + n) = v;
          }
          if (c2m == 3) { w2.x = pack2bf(v.x, v.y); w2.y = pack2bf(v.z, v.w); }
          wc1[g].x = pack2bf(v.x, v.y); wc1[g].y = pack2bf(v.z, v.w);
          wc2[g] = w2;
        }
        // bf16 outputs: lanes l and l + 32 hold neighbouring 4-column groups of the same row; one half exchange per pair of
        // groups (g, g+1) leaves 8 consecutive columns = 16 bytes in every lane: lanes < 32 own columns 8 g .. 8 g + 7, the upper
        // half 8 (g + 1) .. + 7
        const int ns = n0 + qn * 128 + wc * 32 + 8 * hi;      // + 8 g for the pair starting at g
        if (has_c && c_bf16) {
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            auto r0 = __builtin_amdgcn_permlane32_swap(wc1[g].x, wc1[g + 1].x, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(wc1[g].y, wc1[g + 1].y, false, false);
            const int n = ns + 8 * g;
            if (mok && n < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + n) = uint4{r0[0], r1[0], r0[1], r1[1]};
          }
        }
        if (c2m) {
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            auto r0 = __builtin_amdgcn_permlane32_swap(wc2[g].x, wc2[g + 1].x, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(wc2[g].y, wc2[g + 1].y, false, false);
            const int n = ns + 8 * g;
            if (mok && n < p.N) *reinterpret_cast<uint4*>(p.C2 + (long)m * p.ldc2 + n) = uint4{r0[0], r1[0], r0[1], r1[1]};
          }
        }
      }
  }
}

