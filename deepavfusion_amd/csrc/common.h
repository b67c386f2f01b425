// Shared device helpers for the DeepAVFusion gfx950 (CDNA4 / MI355X) kernels.
// wave = 64 lanes everywhere; bf16 storage, fp32 accumulate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "batch.h"

#define DAV_OK 0
#define DAV_ERR_SHAPE (-1)
#define DAV_ERR_DTYPE (-2)
#define DAV_ERR_WORKSPACE (-3)
#define DAV_ERR_HIP (-4)
#define DAV_ERR_ALIGN (-5)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef unsigned short bf16_t;   // raw storage type used in the C ABI

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(T, p) ((const __attribute__((address_space(1))) T*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// fp32 -> bf16, round-to-nearest-even, in hardware: v_cvt_pk_bf16_f32 converts and packs two values per instruction
typedef __attribute__((ext_vector_type(2))) float dav_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 dav_bf16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  union { dav_bf16x2 v; uint32_t u; } r;
  r.v = __builtin_convertvector(dav_f32x2{lo, hi}, dav_bf16x2);
  return r.u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// mean / rstd of one row from its ns = D / 64 partial sums, by the FOUR adjacent lanes q = 0..3 of a row: lane q sums the slots
// q, q + 4, ... in order, then the four sums are combined as (0 + 1) + (2 + 3).  Every user of the statistics (the consuming GEMM,
// the LayerNorm backward) goes through this function, so forward and backward see the same bits.
__device__ __forceinline__ float2 dav_ln_row_stats(const float2* sp, int ns, int q, int D, float eps) {
  float a1 = 0.f, a2 = 0.f;
  float2 t[4];                                   // ns <= 16: all of a lane's slots requested before the first is used
#pragma unroll
  for (int i = 0; i < 4; ++i) t[i] = q + 4 * i < ns ? sp[q + 4 * i] : float2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) { a1 += t[i].x; a2 += t[i].y; }
  a1 += __shfl_xor(a1, 1, 64); a2 += __shfl_xor(a2, 1, 64);
  a1 += __shfl_xor(a1, 2, 64); a2 += __shfl_xor(a2, 2, 64);
  const float inv = 1.0f / (float)D;
  const float mean = a1 * inv;
  return float2{mean, rsqrtf(fmaxf(a2 * inv - mean * mean, 0.f) + eps)};
}

// exact (erf) GELU and its derivative, as nn.GELU() default
// Exact (erf) GELU and its derivative from ONE exponential: erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// far inside the bf16 the results are stored in) — erf(|x|) = 1 - poly(t) e^{-x^2}, t = 1 / (1 + p|x|); with x = z / sqrt(2)
// the same e^{-z^2/2} is the Gaussian density of the derivative.  Phi is formed without cancellation on either side:
// z < 0: Phi = 0.5 poly e,  z >= 0: Phi = 1 - 0.5 poly e.   gelu = z Phi,  gelu' = Phi + z pdf(z).
__device__ __forceinline__ void gelu_pair_f(float z, float& u, float& d) {
  // v_rcp_f32 (1 ulp) instead of an IEEE division (11 instructions), the exponential as one raw v_exp_f32 of
  // z^2 * (-0.5 log2 e), the 0.5 of the tail folded into the polynomial: 18 VALU slots per element instead of 31
  const float e = __builtin_amdgcn_exp2f(z * z * -0.72134752044448170f);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f * 0.70710678118654752f, fabsf(z), 1.0f));
  const float half_tail = e * (t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f)))));   // 0.5 * erfc(|z| / sqrt 2)
  const float cdf = z < 0.f ? half_tail : 1.0f - half_tail;
  u = z * cdf;
  d = __builtin_fmaf(z, 0.39894228040143268f * e, cdf);
}
__device__ __forceinline__ float gelu_f(float x) { float u, d; gelu_pair_f(x, u, d); return u; }
__device__ __forceinline__ float gelu_grad_f(float x) { float u, d; gelu_pair_f(x, u, d); return d; }

#define HIP_CHECK_RET(expr)                    \
  do {                                         \
    hipError_t _e = (expr);                    \
    if (_e != hipSuccess) return DAV_ERR_HIP;  \
  } while (0)

// Launch bookkeeping: hipGetLastError() is sticky process-wide state that other HIP users in the process
// (e.g. torch's event queries -> hipErrorNotReady) also write, so clear it right before each launch and
// latch only errors produced by OUR launches.
inline thread_local int dav_launch_failed = 0;
inline thread_local int dav_last_hip_error = 0;
#define DAV_LAUNCH_NOW(...)                                    \
  do {                                                         \
    (void)hipGetLastError();                                   \
    hipLaunchKernelGGL(__VA_ARGS__);                           \
    const hipError_t _le = hipGetLastError();                  \
    if (_le != hipSuccess) { dav_launch_failed = 1; dav_last_hip_error = (int)_le; } \
  } while (0)
// Inside dav_batch_begin() .. dav_batch_end() (batch.h) a launch is recorded — kernel, geometry and arguments captured by
// value — and replayed by dav_batch_end() in lockstep with the other lanes; families with a grouped kernel record a
// typed parameter block instead (davb::push_typed) before they get here.
#define DAV_LAUNCH(...)                                                          \
  do {                                                                           \
    if (davb::recording()) davb::push_opaque([=]() { DAV_LAUNCH_NOW(__VA_ARGS__); }); \
    else DAV_LAUNCH_NOW(__VA_ARGS__);                                            \
  } while (0)
static inline int dav_launch_status() {
  const int f = dav_launch_failed;
  dav_launch_failed = 0;
  return f ? DAV_ERR_HIP : DAV_OK;
}
