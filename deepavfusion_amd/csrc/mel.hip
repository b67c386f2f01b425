// Log-mel front-end on the GPU: replaces the CPU data-loader transform of the reference,
//   aT.MelSpectrogram(sample_rate, n_fft = 0.05 * rate, hop_length = rate / 64, n_mels) -> aT.Log()     (train.py:50-54)
//   followed by [:, :, :-1]                                                                            (datasets.py:242)
// i.e. torchaudio's Spectrogram (torch.stft: centre-padded by reflection, periodic Hann window of n_fft samples, one-sided,
// power 2) -> HTK mel filterbank without normalisation -> log10(x + eps).
//
// One workgroup per (8 consecutive frames, batch element): the windowed samples of the 8 frames go to LDS once, a thread owns
// the bin pair (k, n_fft/2 - k) and runs the length-n_fft DFT sum with cos / sin looked up at (k * n) mod n_fft in an exact
// table — each twiddle read feeds 8 frames x 2 bins (fp32 accumulation, no FFT butterflies: n_fft = 800 = 2^5 * 5^2 is no
// power of two); the power spectra stay in LDS for the mel projection (a thread sums one filter's non-zero band of one frame)
// and the log.  HBM: reads each sample n_fft / hop = 3.2 times (L2-served), writes the output once.
#include "common.h"
#include "dav_kernels.h"

namespace {

constexpr int MEL_FR = 8;      // frames per workgroup: every twiddle read from LDS feeds MEL_FR frames

__global__ __launch_bounds__(256) void logmel_kernel(const float* wave, int S, int n_fft, int hop, int n_freqs, int n_mels,
                                                     const float* window, const float* cos_tab, const float* sin_tab,
                                                     const float* fbank, const int* band_lo, const int* band_hi, float eps,
                                                     int apply_log, int frames, int T_out, float* out) {
  extern __shared__ float sm[];
  float* ct = sm;                         // [n_fft]
  float* st = ct + n_fft;                 // [n_fft]
  float* xs = st + n_fft;                 // [MEL_FR][n_fft] windowed samples of this workgroup's frames
  float* pw = xs + MEL_FR * n_fft;        // [MEL_FR][n_freqs] power spectra
  const int t0 = blockIdx.x * MEL_FR, b = blockIdx.y, tid = threadIdx.x;
  const float* w = wave + (long)b * S;
  for (int n = tid; n < n_fft; n += blockDim.x) { ct[n] = cos_tab[n]; st[n] = sin_tab[n]; }
  for (int e = tid; e < MEL_FR * n_fft; e += blockDim.x) {
    const int f = e / n_fft, n = e - f * n_fft;
    int i = (t0 + f) * hop - n_fft / 2 + n;                 // center = True
    if (i < 0) i = -i;                                      // pad_mode = 'reflect'
    if (i >= S) i = 2 * (S - 1) - i;
    i = i < 0 ? 0 : (i >= S ? S - 1 : i);                   // frames past the end (t0 + f >= frames) only need a valid address
    xs[e] = w[i] * window[n];
  }
  __syncthreads();
  // Bins k and n_fft/2 - k share their twiddles up to the sign (-1)^n: cos(2 pi (N/2 - k) n / N) = (-1)^n cos(2 pi k n / N),
  // sin(...) = -(-1)^n sin(...).  One pass over the samples with separate even-n / odd-n sums therefore yields both bins:
  // X_k = E + O, X_{N/2-k} = (E_re - O_re, -(E_im - O_im)) — same power formula.  (n_fft is even.)
  const int half = n_fft / 2;
  for (int k = tid; 2 * k <= half; k += blockDim.x) {
    float ere[MEL_FR], eim[MEL_FR], ore[MEL_FR], oim[MEL_FR];
#pragma unroll
    for (int f = 0; f < MEL_FR; ++f) ere[f] = eim[f] = ore[f] = oim[f] = 0.f;
    int idx = 0;                                            // (k * n) mod n_fft, updated incrementally: exact
    for (int n = 0; n < n_fft; n += 2) {
      const float c0 = ct[idx], s0 = st[idx];
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
      const float c1 = ct[idx], s1 = st[idx];
      idx += k;
      if (idx >= n_fft) idx -= n_fft;
#pragma unroll
      for (int f = 0; f < MEL_FR; ++f) {
        const float2 x = *reinterpret_cast<const float2*>(xs + f * n_fft + n);      // wave-uniform address: LDS broadcast
        ere[f] = fmaf(x.x, c0, ere[f]);
        eim[f] = fmaf(x.x, s0, eim[f]);
        ore[f] = fmaf(x.y, c1, ore[f]);
        oim[f] = fmaf(x.y, s1, oim[f]);
      }
    }
#pragma unroll
    for (int f = 0; f < MEL_FR; ++f) {
      const float re = ere[f] + ore[f], im = eim[f] + oim[f];
      pw[f * n_freqs + k] = re * re + im * im;
      const float re2 = ere[f] - ore[f], im2 = eim[f] - oim[f];
      pw[f * n_freqs + half - k] = re2 * re2 + im2 * im2;    // (k == half - k: the two expressions agree)
    }
  }
  __syncthreads();
  for (int e = tid; e < MEL_FR * n_mels; e += blockDim.x) {
    const int f = e / n_mels, m = e - f * n_mels, t = t0 + f;
    if (t >= T_out || t >= frames) continue;                 // the reference drops the last frame
    float s = 0.f;
    for (int k = band_lo[m]; k < band_hi[m]; ++k) s = fmaf(pw[f * n_freqs + k], fbank[(long)k * n_mels + m], s);
    out[((long)b * n_mels + m) * T_out + t] = apply_log ? log10f(s + eps) : s;
  }
}

__global__ __launch_bounds__(256) void log10_eps_kernel(const float* x, float eps, long n, float* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = log10f(x[i] + eps);
}

}  // namespace

extern "C" int dav_logmel(const float* wave, int B, int S, int n_fft, int hop, int n_mels, const float* window,
                          const float* cos_tab, const float* sin_tab, const float* fbank, const int* band_lo, const int* band_hi,
                          float eps, int apply_log, int drop_last, float* out, hipStream_t stream) {
  if (B <= 0 || S <= n_fft / 2 || n_fft <= 0 || (n_fft & 1) || hop <= 0 || n_mels <= 0) return DAV_ERR_SHAPE;
  const int n_freqs = n_fft / 2 + 1, frames = S / hop + 1;
  const int T_out = frames - (drop_last ? 1 : 0);
  if (T_out <= 0) return DAV_ERR_SHAPE;
  const size_t lds = (size_t)(2 * n_fft + MEL_FR * (n_fft + n_freqs)) * sizeof(float);
  if (lds > 64 * 1024) return DAV_ERR_SHAPE;
  DAV_LAUNCH(logmel_kernel, dim3((frames + MEL_FR - 1) / MEL_FR, B), dim3(256), lds, stream, wave, S, n_fft, hop, n_freqs, n_mels, window,
             cos_tab, sin_tab, fbank, band_lo, band_hi, eps, apply_log, frames, T_out, out);
  return dav_launch_status();
}

extern "C" int dav_log10_eps(const float* x, float eps, long n, float* y, hipStream_t stream) {
  if (n <= 0) return DAV_ERR_SHAPE;
  long g = (n + 255) / 256;
  g = g > 4096 ? 4096 : g;
  DAV_LAUNCH(log10_eps_kernel, dim3((int)g), dim3(256), 0, stream, x, eps, n, y);
  return dav_launch_status();
}
