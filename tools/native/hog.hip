// Measurement aid (NOT part of the product library; built on the GPU box by tools/step_sensitivity.py):
// a throttled background load on a side stream while the captured step runs.
//   hog_stream(buf, bytes_per_wg_window, total_bytes, wgs, passes, sink, stream)
// `wgs` workgroups of 256 lanes; workgroup w reads the window [w * window, (w + 1) * window) of `buf` `passes` times
// (8 independent 16-byte loads per lane in flight).  A window of many MB streams from HBM (a BANDWIDTH load of ~wgs x 15 GB/s);
// a window of 32 KB stays in the CU's L1 / L2 (the same instruction stream and occupancy with NO memory-side traffic: a CU-time load).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void hog_kernel(const f4_t* __restrict__ buf, long window_f4, int passes, float* sink) {
  const f4_t* base = buf + (long)blockIdx.x * window_f4;
  float acc = 0.f;
  for (int p = 0; p < passes; ++p) {
    for (long i = threadIdx.x; i + 7 * 256 < window_f4; i += 8 * 256) {
      f4_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(base + i + u * 256);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;      // (keeps the loads alive)
}

extern "C" int hog_stream(const void* buf, long window_bytes, int wgs, int passes, float* sink, void* stream) {
  hipLaunchKernelGGL(hog_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (const f4_t*)buf, window_bytes / 16, passes, sink);
  return (int)hipGetLastError();
}
