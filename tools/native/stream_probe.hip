// Measurement aid (NOT part of the product library): what is the ceiling of a CU's global -> LDS operand stream, and does the
// path matter?   hipcc --offload-arch=gfx950 -O3 tools/native/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
//   dma : global_load_lds_dwordx4 (LDS-DMA: the texture path writes LDS directly; what every GEMM loader of the library uses)
//   reg : global_load_dwordx4 into registers, ds_write_b128 into LDS one stage later
// Both: workgroups of 512 lanes, a two-stage ring of 32 KB stages (64 KB LDS -> two workgroups per CU, like the 128 x 128 tiles),
// one barrier per stage, the loads of stage s + 1 in flight while stage s is "consumed" (here: one ds_read per lane, or `work` of them).
// Footprint: every workgroup walks its own window of `win` bytes; small windows stay in L2 (hits), large ones stream from HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))
typedef float f4_t __attribute__((ext_vector_type(4)));

constexpr int NT = 512, STAGE = 32 * 1024, PER = STAGE / 16 / NT;      // 4 x 16-byte chunks per lane and stage

template <int MODE, int WORK>
__global__ __launch_bounds__(NT) void probe(const char* __restrict__ buf, long win, int steps, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const char* base = buf + (long)blockIdx.x * win;
  const long wrap = win / STAGE;
  f4_t acc = {0.f, 0.f, 0.f, 0.f};
  f4_t r[PER];
  auto issue = [&](int s, int stage) {
    const char* src = base + (long)(s % wrap) * STAGE;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PER; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(void, src + (tid + NT * i) * 16), LDS_PTR(void, smem + stage * STAGE + (wave * 64 + NT * i) * 16), 16, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) r[i] = *reinterpret_cast<const f4_t*>(src + (tid + NT * i) * 16);
    }
  };
  auto land = [&](int stage) {                 // reg mode: registers -> LDS
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < PER; ++i) *LDS_PTR(f4_t, smem + stage * STAGE + (tid + NT * i) * 16) = r[i];
    }
  };
  issue(0, 0);
  if (MODE == 1) { land(0); }
  for (int s = 0; s < steps; ++s) {
    const int stage = s & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (s + 1 < steps) issue(s + 1, stage ^ 1);
    // "consume" stage s
#pragma unroll
    for (int w = 0; w < WORK; ++w) {
      const f4_t v = *LDS_PTR(const f4_t, smem + stage * STAGE + ((tid + 64 * w) % 2048) * 16);
      acc += v;
    }
    if (MODE == 1 && s + 1 < steps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); land(stage ^ 1); }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int MODE, int WORK>
static void run(const char* name, const char* buf, long win, int wgs, float* sink) {
  const int steps = 400;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)probe<MODE, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, WORK>), dim3(wgs), dim3(NT), 2 * STAGE, 0, buf, win, steps, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wgs * steps * STAGE;
  const double cus = wgs < 512 ? (wgs + 1) / 2 : 256;
  printf("%-4s work %2d  window %6ld KB  %4d workgroups: %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, %d CUs busy)\n", name, WORK, win >> 10, wgs,
         ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / cus / 2.4e9, (int)cus);
}

int main() {
  char* buf; float* sink;
  const long total = 8l << 30;
  hipMalloc(&buf, total); hipMemset(buf, 0, total); hipMalloc(&sink, 16);
  for (long win : {64l << 10, 1l << 20, 16l << 20}) {       // 64 KB windows: 32 MB in all (L2 / MALL hits); 16 MB windows: 8 GB (HBM)
    for (int wgs : {512, 256}) {
      run<0, 1>("dma", buf, win, wgs, sink);
      run<1, 1>("reg", buf, win, wgs, sink);
      run<0, 24>("dma", buf, win, wgs, sink);
      run<1, 24>("reg", buf, win, wgs, sink);
    }
  }
  return 0;
}
