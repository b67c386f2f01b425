// 256 x 256 weight-gradient GEMM with gang scheduling for gfx950, included by gemm.hip inside its anonymous namespace (after
// gemm_nt256.h, whose LDS geometry it shares).
//
//   C_p[N, K] (+)= A_p[Mc, N]^T . B_p[Mc, K]      for ALL weight-gradient problems queued since the last flush (several layers)
//
// Why (profiles/r04_step_traffic.txt, r04_tn_l2_hit.txt): the 128 x 128 owner-per-tile kernel pulls 26.8 GB per step through the
// fabric for 7.8 GB of operands — every tile streams its two 128-column panels over the whole contraction, and the tiles that
// share a panel drift apart in time, so a panel crosses the fabric once per tile (L2 hit rate 53 %).  Two levers, both explicit:
//   * a 256 x 256 tile per workgroup (one workgroup of 8 waves per CU, the ping-pong body of csrc/gemm_nt256.h with BOTH operands
//     contraction-major: [64 rows][128 columns] half-tile images read by the transposing ds_read_b64_tr_b16) halves the bytes per
//     flop on the global -> LDS stream;
//   * tiles that share operand panels run AT THE SAME TIME ON THE SAME XCD: the host cuts every problem's tile grid into "gangs"
//     (sub-grids of <= 32 tiles, r x c tiles sharing r + c panels), deals the gangs out to eight per-XCD queues (longest
//     contraction first, least-loaded queue), and a persistent grid of one workgroup per CU draws tile tickets from the queue of
//     the XCD it actually runs on (s_getreg XCC_ID; placement is a speed matter only).  The 32 workgroups of an XCD start
//     together, draw consecutive tickets = the tiles of one gang, and walk the contraction in step: a panel is fetched once per
//     XCD and k-phase.  An empty queue steals from the next XCD's.
// Tried and dropped (profiles/r05_tn_gang_2_asm_reads_prefetch.txt): group 0 owning the whole LDS-DMA stream while group 1 prefetches
// the panels into L2 a few K-tiles ahead (the gang's tiles wait out the fabric latency of a shared line together) — 2238 vs 1954 us
// on the 12 encoder layers: four DMA pieces per wave and interval cost group 0 more than the warm L2 returns.
// Also tried: issuing an interval's DMA at its top (the slot it restages went free at the barrier just passed) instead of behind the
// fragment reads — 2427 vs 2156 us (decoders), 2212 vs 2084 us (12 encoder layers): the issue slots delay the MFMAs by more than the
// extra flight time returns.
// One owner per tile, whole contraction, fixed order: no atomics on the gradient, results independent of who drew which ticket.
// The merged launch (a dozen layers' problems) is what makes whole tiles per workgroup balance: ~4000 tiles over 256 CUs.
//
// Workspace (caller-owned, dav_gemm_tn_gang_workspace_bytes bytes): [TGHeader][TNParams x count][TGDesc x tiles]; written on the stream
// by gemm_tn_gang_write_kernel launches that carry the tables BY VALUE (no host memory is read at replay time: capture-safe).
#pragma once

__device__ __attribute__((aligned(256))) unsigned short g_tng_zeros[128];      // 256 bytes of zeros (module-load initialised)

struct TGDesc { int prob; int bnbk; };       // one ticket: problem index, (tile row << 16) | tile column
struct TGHeader {
  int head[8];                               // tickets drawn so far, per queue
  int q_start[9];                            // queue q owns desc[q_start[q] .. q_start[q + 1])
  int pad[15];
};
static_assert(sizeof(TGHeader) == 128, "TGHeader layout");

constexpr int TG_WCH = 28;                   // problems per writer launch
constexpr int TG_WIT = 72;                   // gangs per writer launch
struct TGItem { int prob, desc_first, r0c0, nrnc; };
struct TGWrite {
  char* ws;
  int prob_first, prob_count, item_count, count_total, write_header;
  int q_start[9];
  TNParams prob[TG_WCH];
  TGItem item[TG_WIT];
};
static_assert(sizeof(TGWrite) <= 4096, "kernel argument block");

// block b < item_count expands gang b into its tickets; the last block stores the problems (and once the header)
__global__ __launch_bounds__(64) void gemm_tn_gang_write_kernel(const TGWrite w) {
  TGHeader* hd = reinterpret_cast<TGHeader*>(w.ws);
  TNParams* probs = reinterpret_cast<TNParams*>(w.ws + sizeof(TGHeader));
  TGDesc* desc = reinterpret_cast<TGDesc*>(w.ws + sizeof(TGHeader) + (size_t)w.count_total * sizeof(TNParams));
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < w.item_count) {
    const TGItem it = w.item[b];
    const int nr = it.nrnc >> 16, nc = it.nrnc & 0xffff, r0 = it.r0c0 >> 16, c0 = it.r0c0 & 0xffff;
    for (int i = t; i < nr * nc; i += 64) desc[it.desc_first + i] = TGDesc{it.prob, ((r0 + i / nc) << 16) | (c0 + i % nc)};
    return;
  }
  const int* src = reinterpret_cast<const int*>(w.prob);
  int* dst = reinterpret_cast<int*>(probs + w.prob_first);
  for (int i = t; i < w.prob_count * (int)(sizeof(TNParams) / 4); i += 64) dst[i] = src[i];
  if (w.write_header) {
    if (t < 8) hd->head[t] = 0;
    if (t < 9) hd->q_start[t] = w.q_start[t];
  }
}

// row of token m under a row map, without an integer division: q = m / rpb through a multiply-high and one correction
// (magic = floor(2^32 / rpb) estimates q or q - 1 for every m < 2^32; rpb = 1 would need 2^32 itself: 2^32 - 1 gives m - 1, corrected too)
__device__ __forceinline__ uint32_t tng_magic(int rpb) { return rpb > 1 ? (uint32_t)(0x100000000ull / (uint32_t)rpb) : rpb == 1 ? 0xffffffffu : 0u; }
__device__ __forceinline__ long tng_row(int m, const RowMap& r, uint32_t magic) {
  if (r.rpb <= 0) return (long)m;
  int q = (int)__umulhi((uint32_t)m, magic);
  int rem = m - q * r.rpb;
  if (rem >= r.rpb) { rem -= r.rpb; ++q; }
  return (long)q * r.bs + r.off + rem;
}

template <int V> struct TGI { static constexpr int value = V; };
template <int OFF>
__device__ __forceinline__ bf16x8 tng_tr(uint32_t base) {
  static_assert(OFF >= 0 && OFF + 1024 < 65536, "ds offset field");
  union { s16x4 h2[2]; bf16x8 v; } u;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(u.h2[0]), "=&v"(u.h2[1]) : "v"(base), "i"(OFF), "i"(OFF + 1024));
  return u.v;
}

// one whole output tile (bn, bk) of problem p: contraction rows [0, Mc)
__device__ __forceinline__ void tng_tile(const TNParams& p, const int bn, const int bk, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int n0 = bn << 8, k0 = bk << 8;
  const int nk = (p.Mc + 63) >> 6;                          // a ragged last K-tile (Mc % 64 != 0: B x tokens at batch 32, 81 / 95 rows per sample) reads zeros for its missing A rows
  const int kend = (nk + 1) & ~1;                           // K-tiles [0, kend); an odd count is padded with one K-tile of zeros on the A side
  const uint32_t amagic = tng_magic(p.amap.rpb), bmagic = tng_magic(p.bmap.rpb);

  // ---- LDS-DMA sources.  Piece e of this wave = token rows (2 wave + e) * 4 + (lane >> 4) of the K-tile, 16-byte chunk lane & 15
  // of the 256-byte image row; the image's 32-byte granule g holds source granule g ^ 2 (row & 3).
  int a_col[2], b_col[2], krow[2];
#pragma unroll
  for (int ee = 0; ee < 2; ++ee) krow[ee] = (wave * 2 + ee) * 4 + (lane >> 4);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c16 = lane & 15;
    const int g = (c16 >> 1) ^ (2 * ((lane >> 4) & 3));     // (both pieces of a wave have the same row & 3)
    int ac = n0 + h * 128 + g * 16 + (c16 & 1) * 8; ac = ac < p.N ? ac : p.N - 8;
    int bc = k0 + h * 128 + g * 16 + (c16 & 1) * 8; bc = bc < p.K ? bc : p.K - 8;
    a_col[h] = ac; b_col[h] = bc;
  }
  auto issue = [&](bool isA, int h, int d, int kt) {
    char* slot = smem + (isA ? 0 : NT256_BREG) + d * 32768 + h * NT256_HT + wave * 2048;
    // zero A operand for the padding K-tile and for everything past the end (group 0 reads one K-tile ahead and sums what it
    // reads into the bias gradient; the trailing reloads are discarded anyway); B: any finite data
    const bool pad = kt >= nk;
    const int ktc = pad ? nk - 1 : kt;
#pragma unroll
    for (int ee = 0; ee < 2; ++ee) {
      const int m0 = ktc * 64 + krow[ee];
      const int m = m0 < p.Mc ? m0 : p.Mc - 1;              // (rows past the end: A reads zeros, B any valid row)
      const bf16_t* src;
      if (isA) src = (pad || m0 >= p.Mc) ? reinterpret_cast<const bf16_t*>(g_tng_zeros) + (lane & 15) * 8 : p.A + tng_row(m, p.amap, amagic) * p.lda + a_col[h];
      else src = p.B + tng_row(m, p.bmap, bmagic) * p.ldb + b_col[h];
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
  // The two transposing reads of a fragment as ONE asm statement: hipcc puts an s_waitcnt vmcnt(0) in front of every
  // __builtin_amdgcn_ds_read_tr16_b64 issued while an LDS-DMA is in flight (the builtin carries no memory operand its wait-count
  // pass could disambiguate from the DMA's LDS write), which drains the ring at every interval.  An asm read is invisible to that
  // pass — and to its lgkmcnt bookkeeping: every use below sits behind an explicit s_waitcnt lgkmcnt(0) + sched_barrier.
  // (slot offsets as instruction immediates: as register operands the 36 distinct addresses spill)
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(char, smem);
  const uint32_t a_base0 = lds0 + at_lo[0], a_base1 = lds0 + at_lo[1], b_base = lds0 + bt_lo;
  auto read_a = [&](auto dh) {                              // dh = TGI<d * 32768 + h * 16384>
    constexpr int O = decltype(dh)::value;
    a[0][0] = tng_tr<O>(a_base0);         a[1][0] = tng_tr<O>(a_base1);
    a[0][1] = tng_tr<O + 4096>(a_base0);  a[1][1] = tng_tr<O + 4096>(a_base1);
    a[0][2] = tng_tr<O + 8192>(a_base0);  a[1][2] = tng_tr<O + 8192>(a_base1);
    a[0][3] = tng_tr<O + 12288>(a_base0); a[1][3] = tng_tr<O + 12288>(a_base1);
  };
  auto read_b = [&](bf16x8 (&bb)[4], auto dh) {
    constexpr int O = decltype(dh)::value;
    bb[0] = tng_tr<O>(b_base); bb[1] = tng_tr<O + 4096>(b_base); bb[2] = tng_tr<O + 8192>(b_base); bb[3] = tng_tr<O + 12288>(b_base);
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
    if (dbg & 4) return;                                    // (timing only: the operand stream alone)
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

  // prologue (buffers are free: the previous tile ended with vmcnt(0) + barrier)
  issue(true, 0, 0, 0); issue(false, 0, 0, 0); issue(false, 1, 0, 0); issue(true, 1, 0, 0);
  issue(true, 0, 1, 1); issue(false, 0, 1, 1); issue(false, 1, 1, 1);
  wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

#define TNG_SB __builtin_amdgcn_sched_barrier(0)
  if (wr == 0) {
    read_a(TGI<0>{}); read_b(b0, TGI<0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TNG_SB;
    bias_acc(0);
#define TNG_G0(I, D, T)                                                              \
    {                                                                                \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      TNG_SB;                                                                        \
      if ((I) == 0) read_b(b1, TGI<(D) * 32768 + 16384>{});                                               \
      if ((I) == 1) read_a(TGI<(D) * 32768 + 16384>{});                                               \
      if ((I) == 3) { read_a(TGI<((D) ^ 1) * 32768>{}); read_b(b0, TGI<((D) ^ 1) * 32768>{}); }                  \
      TNG_SB;                                                                        \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");                    \
      TNG_SB;                                                                        \
      if ((I) == 1) bias_acc(1);                                                     \
      if ((I) == 3) bias_acc(0);                                                     \
      TNG_SB;                                                                        \
      __builtin_amdgcn_s_barrier();                                                  \
      TNG_SB;                                                                        \
    }
    for (int t = 0; t < kend; t += 2) {
      TNG_G0(0, 0, t) TNG_G0(1, 0, t) TNG_G0(2, 0, t) TNG_G0(3, 0, t)
      TNG_G0(0, 1, t + 1) TNG_G0(1, 1, t + 1) TNG_G0(2, 1, t + 1) TNG_G0(3, 1, t + 1)
    }
#undef TNG_G0
  } else {
#define TNG_G1(I, D, T)                                                              \
    {                                                                                \
      if ((I) == 0) { read_a(TGI<(D) * 32768>{}); read_b(b0, TGI<(D) * 32768>{}); }                              \
      if ((I) == 1) read_b(b1, TGI<(D) * 32768 + 16384>{});                                              \
      if ((I) == 2) read_a(TGI<(D) * 32768 + 16384>{});                                                  \
      TNG_SB;                                                                        \
      issue_for(I, D, T);                                                            \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
      TNG_SB;                                                                        \
      if ((I) == 0) bias_acc(0);                                                     \
      if ((I) == 2) bias_acc(1);                                                     \
      __builtin_amdgcn_s_setprio(1);                                                 \
      mfmas((I) >> 1, ((I) == 1 || (I) == 2) ? 1 : 0);                              \
      __builtin_amdgcn_s_setprio(0);                                                 \
      TNG_SB;                                                                        \
      wait_vmcnt<6>();                                                               \
      TNG_SB;                                                                        \
      __builtin_amdgcn_s_barrier();                                                  \
      TNG_SB;                                                                        \
    }
    for (int t = 0; t < kend; t += 2) {
      TNG_G1(0, 0, t) TNG_G1(1, 0, t) TNG_G1(2, 0, t) TNG_G1(3, 0, t)
      TNG_G1(0, 1, t + 1) TNG_G1(1, 1, t + 1) TNG_G1(2, 1, t + 1) TNG_G1(3, 1, t + 1)
    }
#undef TNG_G1
  }
#undef TNG_SB
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // every wave is done with LDS: the next tile's prologue may restage at once

  if (dbg & 2) return;                                     // (timing only: no epilogue)
  // (the problem table came through memory: without the address space the accesses below are FLAT ones, each followed by a full wait)
  typedef __attribute__((address_space(1))) f32x4 gf4_t;
  __attribute__((address_space(1))) float* const Cg = (__attribute__((address_space(1))) float*)p.C;
  // ---- epilogue: lane holds C[n = lane & 31 (+ fragment)][k = 8 g + 4 hi + 0..3], four consecutive columns = 16 bytes.
  // Accumulating tiles read the old values of a fragment row from CLAMPED addresses (no branch around a load: behind a branch hipcc
  // can no longer count and waits vmcnt(0) — for every store issued so far — in front of each use) and add before the guarded stores.
#pragma unroll
  for (int qm = 0; qm < 2; ++qm)
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      const int n = n0 + qm * 128 + wr * 64 + rf * 32 + (lane & 31);
      const long rowoff = (long)(n < p.N ? n : p.N - 1) * p.ldc;
      f32x4 v[2][4];
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x16& q = acc[qm][qn][rf];
          v[qn][g] = f32x4{q[4 * g + 0], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]};
        }
      if (p.beta) {
        f32x4 old[2][4];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int k = k0 + qn * 128 + wc * 32 + 8 * g + 4 * hi;
            old[qn][g] = *(const gf4_t*)(Cg + rowoff + (k < p.K ? k : p.K - 4));
          }
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int g = 0; g < 4; ++g) v[qn][g] += old[qn][g];
      }
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int k = k0 + qn * 128 + wc * 32 + 8 * g + 4 * hi;
          if (n < p.N && k < p.K) *(gf4_t*)(Cg + rowoff + k) = v[qn][g];      // N, K are multiples of 8: the four columns are in or out together
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

constexpr size_t TNG_LDS = NT256_LDS + 16;                 // + the ticket slot (the ONE shared array: a second __shared__ object de-pipelines the k-loop)

__global__ __launch_bounds__(512) void gemm_tn_gang_kernel(char* __restrict__ ws, const int count, const int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  TGHeader* hd = reinterpret_cast<TGHeader*>(ws);
  const int* probs = reinterpret_cast<const int*>(ws + sizeof(TGHeader));
  const TGDesc* desc = reinterpret_cast<const TGDesc*>(ws + sizeof(TGHeader) + (size_t)count * sizeof(TNParams));
  volatile __attribute__((address_space(3))) int* slot = (volatile __attribute__((address_space(3))) int*)(smem + NT256_LDS);
  int xcd;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
  xcd &= 7;
  if (dbg & 8) xcd = blockIdx.x & 7;                        // (A/B only: the dispatcher's observed round-robin instead of the hardware id)
  if (dbg & 16) xcd = (blockIdx.x >> 5) & 7;                // (A/B only: a placement that is wrong on purpose: gangs spread over all XCDs)
  for (int a = 0; a < 8; ++a) {
    const int q = (xcd + a) & 7;                            // own queue first, then steal round the ring
    const int qs = hd->q_start[q], qn = hd->q_start[q + 1] - qs;
    while (true) {
      if (threadIdx.x == 0) {
        // (a queue seen empty stays empty: no ticket is drawn from it again by this workgroup)
        const int t = __hip_atomic_load(&hd->head[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= qn
                          ? qn : __hip_atomic_fetch_add(&hd->head[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *slot = t;
      }
      __syncthreads();
      const int t = __builtin_amdgcn_readfirstlane(*slot);
      __syncthreads();
      if (t >= qn) break;
      const TGDesc d = desc[qs + t];
      const int pi = __builtin_amdgcn_readfirstlane(d.prob), bnbk = __builtin_amdgcn_readfirstlane(d.bnbk);
      union { TNParams p; int w[sizeof(TNParams) / 4]; } u;
#pragma unroll
      for (int i = 0; i < (int)(sizeof(TNParams) / 4); ++i) u.w[i] = __builtin_amdgcn_readfirstlane(probs[pi * (int)(sizeof(TNParams) / 4) + i]);
      tng_tile(u.p, bnbk >> 16, bnbk & 0xffff, dbg);
    }
  }
}
