// Sustained dense bf16 MFMA rate under the power limit with RANDOM operands that change from instruction to instruction:
// v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 (same flops per instruction; the 32 x 32 shape reads half the
// operand registers per flop).  No memory traffic; 2 waves per SIMD; each variant runs long enough (~0.3 s) to settle the clocks.
// hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ bf16x8 rnd8(unsigned& s, int zero) {
  union { unsigned w[4]; bf16x8 v; } u;
  for (int i = 0; i < 4; ++i) {          // bf16 pairs with exponents around 1.0, random signs and mantissas
    const unsigned r = rnd(s);
    u.w[i] = zero ? 0u : ((r & 0x807f807fu) | 0x3f003f00u);
  }
  return u.v;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int zero) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  bf16x8 a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = rnd8(s, zero); b[i] = rnd8(s, zero); }
  float sum = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + j) & 7], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + j) & 7], acc[i & 3], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) sum += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
  const int zero = argc > 1 && !strcmp(argv[1], "zero");
  const int blocks = 256, iters = 60000;
  float* out; hipMalloc(&out, blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int shape = 0; shape < 2; ++shape) {
      hipEventRecord(e0);
      if (shape == 0) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(512), 0, 0, out, iters, zero);
      else hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(512), 0, 0, out, iters, zero);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 8 * iters * 64 * (shape == 0 ? 16384.0 : 32768.0);      // waves x MFMAs x 2 M N K
      printf("%s operands, %s: %8.2f ms  %7.0f TFLOP/s\n", zero ? "zero" : "random", shape == 0 ? "16x16x32" : "32x32x16", ms, flops / (ms * 1e-3) / 1e12);
    }
  return 0;
}
