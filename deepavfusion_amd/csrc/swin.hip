// Data movers and the relative-position bias of the Swin decoder blocks (models/swin.py; decoder_arch == 'swin',
// models/avmae.py:37-51, 174-176).  A decoder activation is [B][nF fusion rows | L token rows][C]; a window SEQUENCE is
// [A = win*win window tokens | nF fusion tokens] (models/swin.py:183-185), B * nW of them.
//   window_unfold : per-batch rows -> window sequences (cyclic shift + window_partition + the fusion-token repeat are ONE
//                   gather through `rows`; models/swin.py:172-185).  Also the backward of window_fold (fusion rows scaled 1/nW).
//   window_fold   : window sequences -> per-batch rows (window_reverse + roll back through the inverse map, fusion rows
//                   averaged over the windows, optional residual: models/swin.py:191-201).  Also the backward of
//                   window_unfold (fusion rows SUMMED over the windows).
//   relpos_bias_build / relpos_bias_bwd : bias[w][h][q][k] = table[index[q][k]][h] + mask[w][q][k] on the A x A corner of
//                   the N x N logits, zero elsewhere (models/swin.py:49-53, 66-78), and the table's gradient from dS.
#include "common.h"
#include "dav_kernels.h"

namespace {

template <typename T> __device__ __forceinline__ float ld_as_f(const T* p, long i);
template <> __device__ __forceinline__ float ld_as_f<float>(const float* p, long i) { return p[i]; }
template <> __device__ __forceinline__ float ld_as_f<bf16_t>(const bf16_t* p, long i) { return __uint_as_float((uint32_t)p[i] << 16); }
template <typename T> __device__ __forceinline__ void st_from_f(T* p, long i, float v);
template <> __device__ __forceinline__ void st_from_f<float>(float* p, long i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void st_from_f<bf16_t>(bf16_t* p, long i, float v) { p[i] = f2bf(v); }

// one wave per output row (a slot of a window sequence)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void window_unfold_kernel(const TI* src, const int* rows, int B, int nW, int A, int nF, int L, int C,
                                                            float fusion_scale, TO* out) {
  const int N = A + nF, R = nF + L;
  const long nrows = (long)B * nW * N;
  const int lane = threadIdx.x & 63;
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += ((long)gridDim.x * blockDim.x) >> 6) {
    const int i = (int)(r % N), w = (int)((r / N) % nW), b = (int)(r / ((long)N * nW));
    const bool fus = i >= A;
    const long srow = (long)b * R + (fus ? i - A : nF + rows[w * A + i]);
    const float s = fus ? fusion_scale : 1.f;
    for (int c = lane; c < C; c += 64) st_from_f<TO>(out, r * C + c, s * ld_as_f<TI>(src, srow * C + c));
  }
}

// one wave per output row (a row of the per-batch activation)
__global__ __launch_bounds__(256) void window_fold_kernel(const float* t, const int* inv, const float* res, int B, int nW, int A, int nF,
                                                          int L, int C, float fusion_scale, float* out) {
  const int N = A + nF, R = nF + L;
  const long nrows = (long)B * R;
  const int lane = threadIdx.x & 63;
  for (long r = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < nrows; r += ((long)gridDim.x * blockDim.x) >> 6) {
    const int j = (int)(r % R), b = (int)(r / R);
    if (j >= nF) {                       // token row: exactly one window slot holds it
      const int slot = inv[j - nF];
      const long srow = ((long)b * nW + slot / A) * N + slot % A;
      for (int c = lane; c < C; c += 64) out[r * C + c] = (res ? res[r * C + c] : 0.f) + t[srow * C + c];
    } else {                             // fusion row: one copy per window
      for (int c = lane; c < C; c += 64) {
        float s = 0.f;
        for (int w = 0; w < nW; ++w) s += t[(((long)b * nW + w) * N + A + j) * C + c];
        out[r * C + c] = (res ? res[r * C + c] : 0.f) + fusion_scale * s;
      }
    }
  }
}

__global__ __launch_bounds__(256) void relpos_bias_build_kernel(const float* table, const int* index, const float* mask, int nb, int H,
                                                                int A, int N, int ld, float mul, float* out) {
  const long total = (long)nb * H * N * ld;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % ld), q = (int)((e / ld) % N), h = (int)((e / ((long)ld * N)) % H), w = (int)(e / ((long)ld * N * H));
    float v = 0.f;
    if (q < A && k < A) v = (table[index[q * A + k] * H + h] + (mask ? mask[((long)w * A + q) * A + k] : 0.f)) * mul;
    out[e] = v;
  }
}

// one workgroup per (table entry, head): the (q, k) pairs of that entry (listed once in LDS) over all B * nW sequences
__global__ __launch_bounds__(256) void relpos_bias_bwd_kernel(const float* dS, const int* index, int Bw, int H, int A, int N, int ld,
                                                              float* dtable) {
  __shared__ int pairs[1024];
  __shared__ int npairs;
  __shared__ float red[256];
  const int e = blockIdx.x / H, h = blockIdx.x % H;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int i = 0; i < A * A && n < 1024; ++i)
      if (index[i] == e) pairs[n++] = (i / A) * ld + (i % A);
    npairs = n;
  }
  __syncthreads();
  float s = 0.f;
  const int np = npairs;
  for (long it = threadIdx.x; it < (long)Bw * np; it += blockDim.x) {
    const int bw = (int)(it / np), pi = (int)(it % np);
    s += dS[((long)bw * H + h) * N * ld + pairs[pi]];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dtable[e * H + h] += red[0];
}

int grid_for_rows(long nrows) { long g = (nrows + 3) / 4; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int dav_window_unfold(const void* src, int src_is_bf16, const int* rows, int B, int nW, int A, int nF, int L, int C,
                                 float fusion_scale, void* out, int out_is_bf16, hipStream_t stream) {
  if (B <= 0 || nW <= 0 || A <= 0 || nF < 0 || L != nW * A || C <= 0) return DAV_ERR_SHAPE;
  const int grid = grid_for_rows((long)B * nW * (A + nF));
  if (src_is_bf16 && out_is_bf16)
    DAV_LAUNCH((window_unfold_kernel<bf16_t, bf16_t>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)src, rows, B, nW, A, nF, L, C, fusion_scale, (bf16_t*)out);
  else if (!src_is_bf16 && out_is_bf16)
    DAV_LAUNCH((window_unfold_kernel<float, bf16_t>), dim3(grid), dim3(256), 0, stream, (const float*)src, rows, B, nW, A, nF, L, C, fusion_scale, (bf16_t*)out);
  else if (!src_is_bf16 && !out_is_bf16)
    DAV_LAUNCH((window_unfold_kernel<float, float>), dim3(grid), dim3(256), 0, stream, (const float*)src, rows, B, nW, A, nF, L, C, fusion_scale, (float*)out);
  else
    return DAV_ERR_DTYPE;
  return dav_launch_status();
}

extern "C" int dav_window_fold(const float* t, const int* inv, const float* res, int B, int nW, int A, int nF, int L, int C,
                               float fusion_scale, float* out, hipStream_t stream) {
  if (B <= 0 || nW <= 0 || A <= 0 || nF < 0 || L != nW * A || C <= 0) return DAV_ERR_SHAPE;
  DAV_LAUNCH(window_fold_kernel, dim3(grid_for_rows((long)B * (nF + L))), dim3(256), 0, stream, t, inv, res, B, nW, A, nF, L, C, fusion_scale, out);
  return dav_launch_status();
}

extern "C" int dav_relpos_bias_build(const float* table, const int* index, const float* mask, int nb, int H, int A, int N, int ld,
                                     float mul, float* out, hipStream_t stream) {
  if (nb <= 0 || H <= 0 || A <= 0 || N < A || ld < N) return DAV_ERR_SHAPE;
  const long total = (long)nb * H * N * ld;
  const long g = (total + 255) / 256;
  DAV_LAUNCH(relpos_bias_build_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, stream, table, index, mask, nb, H, A, N, ld, mul, out);
  return dav_launch_status();
}

extern "C" int dav_relpos_bias_bwd(const float* dS, const int* index, int Bw, int H, int A, int N, int ld, int T, float* dtable,
                                   hipStream_t stream) {
  if (Bw <= 0 || H <= 0 || A <= 0 || N < A || ld < N || T <= 0 || A * A > 1024) return DAV_ERR_SHAPE;
  DAV_LAUNCH(relpos_bias_bwd_kernel, dim3(T * H), dim3(256), 0, stream, dS, index, Bw, H, A, N, ld, dtable);
  return dav_launch_status();
}
