// Calibration: dense bf16 MFMA rate of this GPU and what one s_memtime tick is worth.
// Every wave issues ITERS x 8 independent v_mfma_f32_16x16x32_bf16 back to back (no memory); 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void k(float* out, long long* ticks, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[7]));
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 20000;
  float* out; long long* ticks;
  hipMalloc(&out, blocks * 512 * 4); hipMalloc(&ticks, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double tk = 0; for (int i = 0; i < blocks; ++i) tk += h[i]; tk /= blocks;
    const double mfma = (double)blocks * 8 * iters * 8;                 // wave-level MFMAs
    const double tf = mfma * 16384 / (ms * 1e-3) / 1e12;
    // per SIMD: 2 waves x iters x 8 MFMAs x 16 cycles
    const double cyc = 2.0 * iters * 8 * 16;
    printf("rep %d: %.3f ms  %.0f TFLOP/s dense bf16 | ticks per wave %.0f, MFMA pipe cycles per SIMD %.0f -> %.3f ticks per MFMA-cycle; tick rate %.3f GHz, implied shader clock %.3f GHz\n",
           rep, ms, tf, tk, cyc, tk / cyc, tk / (ms * 1e-3) / 1e9, cyc / (ms * 1e-3) / 1e9);
  }
  return 0;
}
