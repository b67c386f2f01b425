// Standalone bring-up, step 4: PERSISTENT 256 x 256 x 64 NT GEMM body (bf16 in, fp32 accumulate, bf16 out) for gfx950.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm256p.hip -o build_micro/gemm256p
//
// One workgroup per CU walks its tiles (v = b, b + G, ...).  The LDS-DMA stream never stops at a tile boundary: the half-tile
// sequence [A-h0, B-h0, B-h1, A-h1] x K-tiles simply continues into the next tile, 5 half-tiles (80 KB) in flight.
// Interval = one quadrant (64 x 32 per wave, 8 x v_mfma_f32_32x32x16_bf16) and ONE barrier:
//   group 0 (waves 0-3):  M_i ; W_i | barrier          group 1 (waves 4-7):  W_i ; M_i | barrier
// M_i carries the fragment reads of phase i+1 (issued right behind the last MFMA that uses the register they replace),
// W_i = the interval's LDS-DMA pieces + (last K-tile only) the epilogue of the quadrant that has just been completed.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

struct GP {
  const uint16_t* A; const uint16_t* B; uint16_t* C;
  int M, N, K, lda, ldb, ldc;
};

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2; typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  union { b2 v; uint32_t u; } r; r.v = __builtin_convertvector(f2{lo, hi}, b2); return r.u;
}

// keeps the compiler from hoisting loop-invariant lane arithmetic into long-lived registers (the loop body is at the 256-register
// limit; recomputing a few integer ops next to their use is free beside the MFMAs)
__device__ __forceinline__ int opaque(int x) { asm volatile("" : "+v"(x)); return x; }

constexpr int HT = 16384;          // bytes per half-tile slot: 128 rows x 128 B
constexpr int A_REGION = 0;        // slot(d, h) = d * 32768 + h * 16384
constexpr int B_REGION = 65536;
constexpr int ST = 4;              // VMEM ops of one quadrant's epilogue (bf16 output: 2 row fragments x 2 column pairs)

// younger-VMEM-op budget at the end of an interval: everything issued five or more intervals ago must have landed.
// POS = intervals since the first epilogue interval e0 of the previous / current tile (0..8), NONE = no epilogue nearby
template <int NE> __device__ __forceinline__ void wait_dma() {   // NE = epilogue quadrants inside the 5-interval window
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(10 + NE * ST) : "memory");
}

template <int ABL>
__global__ __launch_bounds__(512) void gemm256p_kernel(GP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = (p.N + 255) >> 8, tiles_m = (p.M + 255) >> 8, ntiles = tiles_m * tiles_n;
  const int G = gridDim.x, b = blockIdx.x;
  const int nk = p.K >> 6;
  const int hi = lane >> 5;

  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7, idx = v >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    m0 = (t / tiles_n) << 8; n0 = (t % tiles_n) << 8;
  };

  // ---- issue context (the tile the DMA stream is currently fetching)
  uint32_t a_go[2][2], b_go[2][2];
  auto set_issue_tile = [&](int v, int parts) {      // parts: 1 = A-h0 and both B halves, 2 = A-h1 (they cross a tile boundary one interval apart)
    int m0, n0; tile_origin(v < ntiles ? v : b, m0, n0);      // past the end: re-fetch a valid tile (harmless, keeps the counts uniform)
    const int lane = opaque(tid & 63);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int r = (wave * 2 + e) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = m0 + h * 128 + r; gm = gm < p.M ? gm : p.M - 1;
        int gn = n0 + h * 128 + r; gn = gn < p.N ? gn : p.N - 1;
        // (mul and add kept apart: fused into v_mad_u64_u32 every offset would occupy a register PAIR)
        if (parts & (h == 0 ? 1 : 2)) a_go[h][e] = (uint32_t)opaque(gm * (p.lda * 2)) + c * 16;
        if (parts & 1) b_go[h][e] = (uint32_t)opaque(gn * (p.ldb * 2)) + c * 16;
      }
  };
  // half-tile s of the per-K-tile order [A-h0, B-h0, B-h1, A-h1] of K-tile kt into buffer d (LDS-DMA through a buffer descriptor:
  // 32-bit per-lane offsets, the K offset rides in the scalar offset)
  const auto ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.A), 0, (int)((size_t)p.M * p.lda * 2), 0x00020000);
  const auto brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.B), 0, (int)((size_t)p.N * p.ldb * 2), 0x00020000);
  auto issue = [&](int s, int d, int kt) {
    if (ABL & 2) return;
    const bool isA = (s == 0 || s == 3);
    const int h = (s >= 2) ? 1 : 0;
    char* slot = smem + (isA ? A_REGION : B_REGION) + d * 32768 + h * HT + wave * 2048;
#pragma unroll
    for (int e = 0; e < 2; ++e)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? ars : brs, LDS_PTR(void, slot + e * 1024), 16, isA ? a_go[h][e] : b_go[h][e], kt * 128, 0, 0);
  };

  const int sw = (lane >> 1) & 7;
  // fragment read addresses: 16-byte slot = (2 ks + hi) ^ sw  ->  offset(ks) = offset(0) ^ (ks << 5)
  const uint32_t a_l0 = A_REGION + (wr * 64 + (lane & 31)) * 128 + ((hi ^ sw) << 4);
  const uint32_t b_l0 = B_REGION + (wc * 32 + (lane & 31)) * 128 + ((hi ^ sw) << 4);

  f32x16 acc[2][2][2];   // [qm][qn][rf]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][r][e] = 0.f;
  bf16x8 a[2][4] = {}, bx[4] = {}, by[4] = {};
#define SB __builtin_amdgcn_sched_barrier(0)
  // Fragment reads go through inline asm with the destination TIED to the register being replaced ("+v"): left to itself the
  // register allocator gives the incoming fragments fresh registers (a second 32-register set of A fragments -> spills).  The
  // compiler does not count these reads; every consumer sits behind the lgkmcnt(0) + barrier that ends the interval.
  auto rd_a = [&](int d, int h, int ks) __attribute__((always_inline)) {
    if (ABL & 4) return;
    const uint32_t o = (uint32_t)opaque((int)a_l0) ^ (ks << 5);
#pragma unroll
    for (int rf = 0; rf < 2; ++rf)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(a[rf][ks]) : "v"(o), "n"(d * 32768 + h * HT + rf * 4096));
  };
  auto rd_b = [&](bf16x8 (&bb)[4], int d, int h, int ks) __attribute__((always_inline)) {
    if (ABL & 4) return;
    const uint32_t o = (uint32_t)opaque((int)b_l0) ^ (ks << 5);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(bb[ks]) : "v"(o), "n"(d * 32768 + h * HT));
  };
  auto mm = [&](int qm, int qn, const bf16x8 (&bb)[4], int ks) __attribute__((always_inline)) {
    if (ABL & 1) { asm volatile("" ::"v"(a[0][ks]), "v"(a[1][ks]), "v"(bb[ks])); return; }
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      acc[qm][qn][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb[ks], a[rf][ks], acc[qm][qn][rf], 0, 0, 0);
    }
  };

  // One interval = TWO quadrants (16 MFMAs per wave).  qm = 0: quadrants (0,0),(0,1) on a = A-h0, while a <- A-h1 of the same K-tile;
  // qm = 1: quadrants (1,1),(1,0) on a = A-h1, while a <- A-h0, b0 <- B-h0, b1 <- B-h1 of the NEXT K-tile (buffer d ^ 1).
  // Each read sits right behind the last MFMA that uses the register it replaces; the interval's LDS-DMA pieces are issued in
  // front (the wave that is busy issuing leaves the matrix pipe to the SIMD's other wave).
  // epilogue chunk c = 0..7 of a quadrant PAIR (qm fixed): quadrant (c >> 2), row fragment (c >> 1) & 1, column-group pair c & 1.
  // transposed accumulators: lane holds C[m = lane & 31][n = 8 g + 4 hi + 0..3]; column groups are paired by permlane32_swap so that
  // every lane stores 16 contiguous bytes; buffer stores: rows past M / columns past N are dropped by the bounds check, so the
  // instruction count (the vmcnt bookkeeping) does not depend on the data.  The fragment's accumulators restart from zero.
  const auto crs = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, (int)((size_t)p.M * p.ldc * 2), 0x00020000);
  auto EC = [&](int qm, int c, int m0, int n0) __attribute__((always_inline)) {
    const int qn = qm == 0 ? (c >> 2) : 1 - (c >> 2), rf = (c >> 1) & 1, g = (c & 1) * 2;
    const int lane = opaque(tid & 63), hi = lane >> 5;
    const int m = m0 + qm * 128 + wr * 64 + rf * 32 + (lane & 31);
    uint32_t ax = pack2bf(acc[qm][qn][rf][4 * g + 0], acc[qm][qn][rf][4 * g + 1]);
    uint32_t ay = pack2bf(acc[qm][qn][rf][4 * g + 2], acc[qm][qn][rf][4 * g + 3]);
    uint32_t cx = pack2bf(acc[qm][qn][rf][4 * g + 4], acc[qm][qn][rf][4 * g + 5]);
    uint32_t cy = pack2bf(acc[qm][qn][rf][4 * g + 6], acc[qm][qn][rf][4 * g + 7]);
    auto r0 = __builtin_amdgcn_permlane32_swap(ax, cx, false, false); ax = r0[0]; cx = r0[1];
    auto r1 = __builtin_amdgcn_permlane32_swap(ay, cy, false, false); ay = r1[0]; cy = r1[1];
    const int n = n0 + qn * 128 + wc * 32 + 8 * g + 8 * hi;
    const uint32_t off = (m < p.M && n < p.N) ? ((uint32_t)m * (uint32_t)p.ldc + n) * 2u : 0xffffffffu;
    if (!(ABL & 8)) __builtin_amdgcn_raw_buffer_store_b128(u32x4{ax, ay, cx, cy}, crs, off, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[qm][qn][rf][4 * g + e] = 0.f;
    SB;
  };

  // One interval = TWO quadrants (16 MFMAs per wave).  qm = 0: quadrants (0,0),(0,1) on a = A-h0, while a <- A-h1 of the same K-tile;
  // qm = 1: quadrants (1,1),(1,0) on a = A-h1, while a <- A-h0, b0 <- B-h0, b1 <- B-h1 of the NEXT K-tile (buffer d ^ 1).
  // Each read sits right behind the last MFMA that uses the register it replaces.  epi: the OTHER quadrant pair (finished one
  // interval ago) leaves in eight chunks, two behind every group of four MFMAs (the matrix pipe stays busy with the MFMAs
  // already queued and with the SIMD's other wave).
  auto M2 = [&](int qm, int d, bool epi, int em0, int en0) __attribute__((always_inline)) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (qm == 0) { mm(0, 0, bx, ks); mm(0, 1, by, ks); SB; rd_a(d, 1, ks); SB; }
      else { mm(1, 1, by, ks); mm(1, 0, bx, ks); SB; rd_a(d ^ 1, 0, ks); rd_b(bx, d ^ 1, 0, ks); rd_b(by, d ^ 1, 1, ks); SB; }
      if (epi) { EC(1 - qm, 2 * ks, em0, en0); EC(1 - qm, 2 * ks + 1, em0, en0); }
    }
    __builtin_amdgcn_s_setprio(0);
    SB;
  };

  // ---- prologue: K-tile 0 (A-h0, B-h0, B-h1, A-h1) and K-tile 1 (A-h0, B-h0, B-h1) of the first tile; pre-read of K-tile 0
  int jt = 0;                                  // tiles done by this workgroup
  set_issue_tile(b, 3);
  issue(0, 0, 0); issue(1, 0, 0); issue(2, 0, 0); issue(3, 0, 0);
  issue(0, 1, 1); issue(1, 1, 1); issue(2, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  SB;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) { rd_a(0, 0, ks); rd_b(bx, 0, 0, ks); rd_b(by, 0, 1, ks); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  SB;

  const int ntw = (ntiles - b + G - 1) / G;    // tiles of this workgroup
  const int nbody = ntw * (nk >> 1);           // loop trips: two K-tiles each
  int t = 0, cm0, cn0, pm0 = 0, pn0 = 0;       // current / previous tile origin
  tile_origin(b, cm0, cn0);
  // end of an interval: everything the NEXT interval reads has landed (younger VMEM ops: NV), then the barrier; the fragment
  // reads issued in this interval are waited for at the start of the next one
#define ENDI(NV) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory"); SB; __builtin_amdgcn_s_barrier(); SB; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SB; }
#pragma clang loop unroll(disable)
  for (int u = 0; u < nbody; ++u) {
    const bool first = (t == 0), last = (t == nk - 2), prev = first && (jt > 0);
    const int nxt = b + (jt + 1) * G;
    // ---- K-tile t (buffer 0)
    // I01: DMA A-h1 of K-tile t+1 (buffer 1); the previous tile's quadrants (1,1),(1,0) leave here
    issue(3, 1, t + 1); SB;
    M2(0, 0, prev, pm0, pn0);
    if (prev) ENDI(2 + 2 * ST + 2 * ST) else ENDI(2)
    // I23: DMA A-h0, B-h0, B-h1 of K-tile t+2 (buffer 0) — the next tile's K-tile 0 when this is the last body
    if (last) set_issue_tile(nxt, 1);
    { const int kt = last ? 0 : t + 2; issue(0, 0, kt); issue(1, 0, kt); issue(2, 0, kt); SB; }
    M2(1, 0, false, 0, 0);
    if (prev) ENDI(6 + 2 * ST) else ENDI(6)
    // ---- K-tile t+1 (buffer 1)
    if (last) set_issue_tile(nxt, 2);
    { const int kt = last ? 0 : t + 2; issue(3, 0, kt); SB; }
    M2(0, 1, false, 0, 0);
    ENDI(2)
    { const int kt = last ? 1 : t + 3; issue(0, 1, kt); issue(1, 1, kt); issue(2, 1, kt); SB; }
    M2(1, 1, last, cm0, cn0);
    if (last) ENDI(6 + 2 * ST) else ENDI(6)
    t += 2;
    if (t == nk) { t = 0; ++jt; pm0 = cm0; pn0 = cn0; tile_origin(b + jt * G < ntiles ? b + jt * G : b, cm0, cn0); }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) EC(1, c, pm0, pn0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// naive reference: one thread per output
__global__ void ref_kernel(GP p, float* out) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)p.M * p.N) return;
  const int m = i / p.N, n = i % p.N;
  float s = 0.f;
  for (int k = 0; k < p.K; ++k) {
    const float a = __uint_as_float((uint32_t)p.A[(size_t)m * p.lda + k] << 16);
    const float b = __uint_as_float((uint32_t)p.B[(size_t)n * p.ldb + k] << 16);
    s += a * b;
  }
  out[i] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint16_t f2bf_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

template <int ABL>
static float run(const GP& p, int iters, int gridcap, hipStream_t st) {
  auto k = gemm256p_kernel<ABL>;
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); set = true; }
  const int ntiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  const int grid = ntiles < gridcap ? ntiles : gridcap;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, st, p);
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, st, p);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main(int argc, char** argv) {
  hipStream_t st; CK(hipStreamCreate(&st));
  const int gridcap = argc > 1 ? atoi(argv[1]) : 256;
  const int shapes[][3] = {{4096, 4096, 4096}, {11264, 2304, 768}, {7168, 3072, 768}, {7168, 768, 3072}, {22528, 2048, 512},
                           {37120, 1536, 512}, {37120, 2048, 512}, {37120, 512, 2048}, {5184, 2304, 768}, {6080, 2304, 768}, {1000, 700, 256}, {8192, 8192, 8192}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    uint32_t seed = 12345u + M + N + K;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf_host(rnd());
    for (auto& v : hB) v = f2bf_host(rnd());
    uint16_t *dA, *dB, *dC; float* dR;
    const int ldc = (N + 7) & ~7;
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dB, hB.size() * 2)); CK(hipMalloc(&dC, (size_t)M * ldc * 2)); CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xff, (size_t)M * ldc * 2));
    GP p{dA, dB, dC, M, N, K, K, K, ldc};
    const float us = run<0>(p, 20, gridcap, st);
    const bool check = (double)M * N * K < 3e11;
    double maxerr = -1, maxref = 0; long bad = 0;
    if (check) {
      hipLaunchKernelGGL(ref_kernel, dim3((unsigned)(((long)M * N + 255) / 256)), dim3(256), 0, st, p, dR);
      CK(hipStreamSynchronize(st));
      std::vector<float> hR((size_t)M * N); std::vector<uint16_t> hC((size_t)M * ldc);
      CK(hipMemcpy(hR.data(), dR, hR.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
      maxerr = 0;
      for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
          uint32_t u = (uint32_t)hC[(size_t)m * ldc + n] << 16; float c; memcpy(&c, &u, 4);
          const float r = hR[(size_t)m * N + n];
          const double e = fabs((double)c - r);
          if (!(e <= 0.02 * fabs(r) + 0.05)) ++bad;
          if (e > maxerr) maxerr = e;
          if (fabs(r) > maxref) maxref = fabs(r);
        }
    }
    const float us1 = run<1>(p, 10, gridcap, st), us2 = run<2>(p, 10, gridcap, st), us4 = run<4>(p, 10, gridcap, st), us8 = run<8>(p, 10, gridcap, st), us15 = run<15>(p, 10, gridcap, st);
    const float us6 = run<6>(p, 10, gridcap, st), us5 = run<5>(p, 10, gridcap, st), us3 = run<3>(p, 10, gridcap, st);
    CK(hipMemset(dA, 0, hA.size() * 2)); CK(hipMemset(dB, 0, hB.size() * 2));
    const float usz = run<0>(p, 20, gridcap, st), usz6 = run<6>(p, 10, gridcap, st);
    printf("      MFMA only %.1f  DMA only %.1f  reads only %.1f | zero-filled operands: full %.1f  MFMA only %.1f us\n", us6, us5, us3, usz, usz6);
    printf("%6d x %5d x %5d : %8.1f us %7.0f TF | maxerr %.3g (ref max %.3g) bad %ld | noMFMA %.1f  noDMA %.1f  noREAD %.1f  noSTORE %.1f  skeleton %.1f us\n", M, N, K, us,
           2.0 * M * N * K / us / 1e6, maxerr, maxref, bad, us1, us2, us4, us8, us15);
    fflush(stdout);
    (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(dC); (void)hipFree(dR);
  }
  return 0;
}
