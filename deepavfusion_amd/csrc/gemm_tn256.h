// 256 x 256 weight-gradient GEMM for gfx950, included by gemm.hip inside its anonymous namespace (after gemm_nt256.h).
//
//   C_p[N, K] (+)= A_p[Mc, N]^T . B_p[Mc, K]   for the <= 32 problems of a grouped launch (a layer's deferred weight gradients)
//
// Same workgroup body as the NT configuration 60 (csrc/gemm_nt256.h: two groups of four waves half an interval apart, one
// barrier per 64 x 32 quadrant, 2 x 4 half-tile slots of 16 KB, three half-tiles of LDS-DMA in flight), with BOTH operands
// contraction-major: a half-tile is a [64 token rows][128 columns] image (256-byte rows, 32-byte granules XOR-swizzled by
// 2 (row & 3) on the DMA's source side) and every fragment comes from two transposing ds_read_b64_tr_b16.
//
// Work decomposition (what the 128 x 128 kernel could not do without paying 15-35 % for atomics on EVERY tile): the launch is
// a PERSISTENT grid of one workgroup per CU; the unit of work is a pair of K-tiles (128 token rows) of one output tile, all
// units of all problems form one sequence (problem-major, tile-major, contraction-minor), and workgroup w takes the w-th of G
// equal slices of it ("stream-K").  A slice covers the tail of one tile, a few whole tiles and the head of another: whole
// tiles are written / read-modify-written by their only owner, the at most two partial ones meet through fp32 atomics
// (weight gradients are sums: no finaliser protocol is needed).  Every workgroup does the same number of MFMAs whatever the
// mix of contraction lengths (49 .. 352 K-tiles in one launch of the step).
//
// Odd K-tile counts (Mc = 3136: 49) are padded with a K-tile whose A operand is read from a block of zeros.  The bias gradient
// (column sums of A) is taken by the waves that own the first tile column, with v_dot2c_f32_bf16 against ones on the A
// fragments they hold anyway.  DavTnProblem.flags bit 0 (write instead of accumulate) is honoured for tiles that one
// workgroup owns entirely; a problem carrying it is therefore never split (the host rounds its slices to tile boundaries).
#pragma once

__device__ __attribute__((aligned(256))) unsigned short g_tn256_zeros[128];      // 256 bytes of zeros (module-load initialised)

constexpr int TN256_MAX = 32;
struct TN256Group {
  TNParams prob[TN256_MAX];
  int first_unit[TN256_MAX + 1];      // cumulative count of K-tile pairs x tiles
  int count;
};

// row of token m under a row map, without an integer division: q = m / rpb through a multiply-high and one correction
__device__ __forceinline__ long tn256_row(int m, const RowMap& r, uint32_t magic) {
  if (r.rpb <= 0) return (long)m;
  int q = (int)__umulhi((uint32_t)m, magic);
  int rem = m - q * r.rpb;
  if (rem >= r.rpb) { rem -= r.rpb; ++q; }
  return (long)q * r.bs + r.off + rem;
}

// one chunk: K-tile pairs [s, e) of output tile `tile` of problem p
__device__ __forceinline__ void tn256_chunk(const TNParams& p, int tile, int s, int e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_k = (p.K + 255) >> 8;
  const int bn = tile / tiles_k, bk = tile - bn * tiles_k;
  const int n0 = bn << 8, k0 = bk << 8;
  const int nk = p.Mc >> 6, up = (nk + 1) >> 1;
  const int kbeg = 2 * s, kend = 2 * e;                     // K-tiles [kbeg, kend); those >= nk are the zero padding
  const bool whole = (s == 0 && e == up) || (p.debug_plain_store & 1);      // (debug bit 1, timing only: plain stores for partial tiles)
  const uint32_t amagic = p.amap.rpb > 0 ? (uint32_t)(0x100000000ull / (uint32_t)p.amap.rpb) : 0u;
  const uint32_t bmagic = p.bmap.rpb > 0 ? (uint32_t)(0x100000000ull / (uint32_t)p.bmap.rpb) : 0u;

  // ---- LDS-DMA sources.  Piece e of this wave = token rows (2 wave + e) * 4 + (lane >> 4) of the K-tile, 16-byte chunk lane & 15
  // of the 256-byte image row; the image's 32-byte granule g holds source granule g ^ 2 (row & 3).
  int a_col[2], b_col[2], krow[2];
#pragma unroll
  for (int ee = 0; ee < 2; ++ee) krow[ee] = (wave * 2 + ee) * 4 + (lane >> 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c16 = lane & 15;
    // (both pieces of a wave have the same row & 3: (2 wave + e) * 4 is a multiple of 4)
    const int g = (c16 >> 1) ^ (2 * ((lane >> 4) & 3));
    int ac = n0 + h * 128 + g * 16 + (c16 & 1) * 8; ac = ac < p.N ? ac : p.N - 8;
    int bc = k0 + h * 128 + g * 16 + (c16 & 1) * 8; bc = bc < p.K ? bc : p.K - 8;
    a_col[h] = ac; b_col[h] = bc;
  }
  auto issue = [&](bool isA, int h, int d, int kt) {
    char* slot = smem + (isA ? 0 : NT256_BREG) + d * 32768 + h * NT256_HT + wave * 2048;
    // zero A operand for the padding K-tile of an odd count AND for everything past this chunk (group 0 reads one K-tile ahead and
    // sums what it reads into the bias gradient; the trailing reloads are discarded anyway); B: any finite data
    const bool pad = kt >= nk || kt >= kend;
    const int ktc = pad ? nk - 1 : kt;
#pragma unroll
    for (int ee = 0; ee < 2; ++ee) {
      const int m = ktc * 64 + krow[ee];
      const bf16_t* src;
      if (isA) src = pad ? reinterpret_cast<const bf16_t*>(g_tn256_zeros) + (lane & 15) * 8 : p.A + tn256_row(m, p.amap, amagic) * p.lda + a_col[h];
      else src = p.B + tn256_row(m, p.bmap, bmagic) * p.ldb + b_col[h];
      __builtin_amdgcn_global_load_lds(GLB_PTR(void, src), LDS_PTR(void, slot + ee * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (transposing reads): 16-lane group gq reads the [4 k][16 col] block at k = 8 (gq >> 1) (+ 4 h2),
  // columns colbase + 16 (gq & 1); lane li of the group points at row li >> 2, columns 4 (li & 3) .. + 3 of it
  const int gq = lane >> 4, li = lane & 15, hi = lane >> 5;
  uint32_t at_lo[2], bt_lo;
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
    at_lo[rf] = (8 * (gq >> 1) + (li >> 2)) * 256 + ((((wr * 4 + rf * 2 + (gq & 1)) ^ (2 * (li >> 2))) << 5) | (8 * (li & 3)));
  bt_lo = NT256_BREG + (8 * (gq >> 1) + (li >> 2)) * 256 + ((((wc * 2 + (gq & 1)) ^ (2 * (li >> 2))) << 5) | (8 * (li & 3)));

  f32x16 acc[2][2][2];   // [qm][qn][rf]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][r][q] = 0.f;
  const bool do_bias = p.bias_grad != nullptr && bk == 0 && wc == 0;
  float accb[2][2] = {{0.f, 0.f}, {0.f, 0.f}};            // [qm][rf]: this lane's share (its 8 of every 16 k) of the column sums

  bf16x8 a[2][4], b0[4], b1[4];
  auto tr_frag = [&](uint32_t off) {
    union { s16x4 h2[2]; bf16x8 v; } u;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) u.h2[h2] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, smem + off + h2 * 1024));
    return u.v;
  };
  auto read_a = [&](int d, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) a[rf][ks] = tr_frag(at_lo[rf] + d * 32768 + h * NT256_HT + ks * 4096);
  };
  auto read_b = [&](bf16x8 (&bb)[4], int d, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bb[ks] = tr_frag(bt_lo + d * 32768 + h * NT256_HT + ks * 4096);
  };
  auto bias_acc = [&](int qm) {                             // column sums of the A fragments just loaded (rows of dW's bias)
    if (!do_bias) return;
    union { uint32_t u; dav_bf16x2 v; } one; one.u = 0x3f803f80u;
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        union { bf16x8 v; dav_bf16x2 q[4]; } f; f.v = a[rf][ks];
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[qm][rf] = __builtin_amdgcn_fdot2_f32_bf16(f.q[j], one.v, accb[qm][rf], false);
      }
  };
  auto mfmas = [&](int qm, int qn) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf)
        acc[qm][qn][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qn ? b1[ks] : b0[ks], a[rf][ks], acc[qm][qn][rf], 0, 0, 0);
  };
  auto issue_for = [&](int i, int d, int t) {
    if (i == 0) issue(true, 1, d ^ 1, t + 1);
    if (i == 1) issue(true, 0, d, t + 2);
    if (i == 2) issue(false, 0, d, t + 2);
    if (i == 3) issue(false, 1, d, t + 2);
  };

  // prologue (buffers are free: the previous chunk ended with vmcnt(0) + barrier)
  issue(true, 0, 0, kbeg); issue(false, 0, 0, kbeg); issue(false, 1, 0, kbeg); issue(true, 1, 0, kbeg);
  issue(true, 0, 1, kbeg + 1); issue(false, 0, 1, kbeg + 1); issue(false, 1, 1, kbeg + 1);
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

#define TN256_SB __builtin_amdgcn_sched_barrier(0)
  if (wr == 0) {
    read_a(0, 0); read_b(b0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TN256_SB;
    bias_acc(0);
#define TN256_G0(I, D, T)                                                            \
    {                                                                                \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      TN256_SB;                                                                      \
      if ((I) == 0) read_b(b1, D, 1);                                                \
      if ((I) == 1) read_a(D, 1);                                                    \
      if ((I) == 3) { read_a((D) ^ 1, 0); read_b(b0, (D) ^ 1, 0); }                  \
      TN256_SB;                                                                      \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                    \
      TN256_SB;                                                                      \
      if ((I) == 1) bias_acc(1);                                                     \
      if ((I) == 3) bias_acc(0);                                                     \
      TN256_SB;                                                                      \
      __builtin_amdgcn_s_barrier();                                                  \
      TN256_SB;                                                                      \
    }
    for (int t = kbeg; t < kend; t += 2) {
      TN256_G0(0, 0, t) TN256_G0(1, 0, t) TN256_G0(2, 0, t) TN256_G0(3, 0, t)
      TN256_G0(0, 1, t + 1) TN256_G0(1, 1, t + 1) TN256_G0(2, 1, t + 1) TN256_G0(3, 1, t + 1)
    }
#undef TN256_G0
  } else {
#define TN256_G1(I, D, T)                                                            \
    {                                                                                \
      if ((I) == 0) { read_a(D, 0); read_b(b0, D, 0); }                              \
      if ((I) == 1) read_b(b1, D, 1);                                                \
      if ((I) == 2) read_a(D, 1);                                                    \
      TN256_SB;                                                                      \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
      TN256_SB;                                                                      \
      if ((I) == 0) bias_acc(0);                                                     \
      if ((I) == 2) bias_acc(1);                                                     \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      TN256_SB;                                                                      \
      wait_vmcnt<6>();                                                               \
      TN256_SB;                                                                      \
      __builtin_amdgcn_s_barrier();                                                  \
      TN256_SB;                                                                      \
    }
    for (int t = kbeg; t < kend; t += 2) {
      TN256_G1(0, 0, t) TN256_G1(1, 0, t) TN256_G1(2, 0, t) TN256_G1(3, 0, t)
      TN256_G1(0, 1, t + 1) TN256_G1(1, 1, t + 1) TN256_G1(2, 1, t + 1) TN256_G1(3, 1, t + 1)
    }
#undef TN256_G1
  }
#undef TN256_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // every wave is done with LDS: the next chunk's prologue may restage at once

  if (p.debug_plain_store & 2) return;                     // (timing only: no epilogue)
  // ---- epilogue: lane holds C[n = lane & 31 (+ fragment)][k = 8 g + 4 hi + 0..3], four consecutive columns = 16 bytes
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const int n = n0 + qm * 128 + wr * 64 + rf * 32 + (lane & 31);
#pragma unroll
      for (int qn = 0; qn < 2; ++qn) {
        const f32x16& q = acc[qm][qn][rf];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k = k0 + qn * 128 + wc * 32 + 8 * g + 4 * hi;
          if (n >= p.N || k >= p.K) continue;              // N, K are multiples of 8: the four columns are in or out together
          float* c = p.C + (long)n * p.ldc + k;
          if (!whole) {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c + r, q[4 * g + r]);
          } else {
            float4 v = float4{q[4 * g + 0], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]};
            if (p.beta) { const float4 o = *reinterpret_cast<const float4*>(c); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(c) = v;
          }
        }
      }
    }
  if (do_bias) {
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        const float tot = accb[qm][rf] + __shfl_xor(accb[qm][rf], 32, 64);      // the two half-waves hold the two halves of every 16 k
        const int n = n0 + qm * 128 + wr * 64 + rf * 32 + (lane & 31);
        if (hi == 0 && n < p.N) unsafeAtomicAdd(p.bias_grad + n, tot);
      }
  }
}

__global__ __launch_bounds__(512) void gemm_tn256_kernel(const TN256Group g) {
  // logical workgroup id: the dispatcher places block b on XCD b % 8 -> every XCD gets a contiguous eighth of the unit sequence
  // (neighbouring tiles of one problem share the operands' column blocks in that XCD's L2)
  const int G = gridDim.x, b = blockIdx.x;
  const int lb = (G & 7) ? b : (b & 7) * (G >> 3) + (b >> 3);
  const long U = g.first_unit[g.count];
  int u = (int)(U * lb / G);
  const int uend = (int)(U * (lb + 1) / G);
  int pi = 0;
  while (u < uend) {
    while (pi + 1 < g.count && u >= g.first_unit[pi + 1]) ++pi;
    const TNParams& p = g.prob[pi];
    const int up = ((p.Mc >> 6) + 1) >> 1;
    const int local = u - g.first_unit[pi];
    const int tile = local / up, s = local - tile * up;
    // a problem whose tiles must be written by one owner (beta == 0) is never split: the slice is extended to the tile's end
    // (the next workgroup skips what has been taken: see the symmetric rule at the slice start below)
    int e = s + (uend - u) < up ? s + (uend - u) : up;
    if (!p.beta) {
      if (s != 0) { u += up - s; continue; }              // the tile's head belongs to the previous slice, which took all of it
      e = up;
    }
    tn256_chunk(p, tile, s, e);
    u += e - s;
  }
}
