// mel.hip -- speech front-end on the GPU: pre-emphasis -> framing -> window -> 512-point
// FFT -> |.|^2 -> Slaney mel filterbank -> dB, batched over utterances.
//
// Restates the reference's offline numpy path (odin/preprocessing/signal.py: pre_emphasis
// :955-967, stft :1442-1562, power_spectrogram :1623-1648, mels_spectrogram :1650-1691,
// power2db :636-680; wrapped by speech.py:655-929).  The reference runs one utterance at
// a time on CPU workers; here a workgroup transforms FPB frames at once entirely in LDS:
//   1. samples are pre-emphasised, windowed (window already carries the 1/sum(w) scale) and
//      written to LDS in bit-reversed order (zero padded to n_fft),
//   2. radix-2 decimation-in-time butterflies, twiddles from an LDS table,
//   3. the power spectrum stays in LDS and is contracted with the [bins, n_mels] filterbank
//      (stored transposed so that lanes read consecutive mel bands),
//   4. 10*log10(max(1e-10, .)) is written; a second launch applies the per-utterance
//      `max - top_db` floor (a global maximum over the utterance, as power2db does).
// HBM-bound by design: reads 4 B/sample, writes 4 B/(frame, band).
#include "odin_device.h"
#include "odin_internal.h"

namespace {

constexpr int FPB = 4;  // frames per workgroup pass

__global__ __launch_bounds__(256) void stft_mel_kernel(const float* __restrict__ y,
                                                       const float* __restrict__ window,
                                                       const float* __restrict__ fbT,
                                                       float* __restrict__ out, int n_samples,
                                                       int frame_length, int step, int n_fft,
                                                       int log2n, int n_frames, int n_mels,
                                                       float preemph, int log_output) {
  ODIN_DYN_SMEM(float, smem);
  const int nb = n_fft / 2 + 1;
  float* re = smem;                       // [FPB][n_fft]
  float* im = re + FPB * n_fft;           // [FPB][n_fft]
  float* twc = im + FPB * n_fft;          // [n_fft/2]
  float* tws = twc + n_fft / 2;           // [n_fft/2]
  float* pw = tws + n_fft / 2;            // [FPB][nb]
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * FPB;
  const float* yb = y + (size_t)b * n_samples;
  for (int k = tid; k < n_fft / 2; k += 256) {
    float s, c;
    sincosf(-6.283185307179586f * (float)k / (float)n_fft, &s, &c);
    twc[k] = c;
    tws[k] = s;
  }
  for (int e = tid; e < FPB * n_fft; e += 256) {
    const int f = e / n_fft, n = e - f * n_fft;
    const int t = t0 + f;
    float v = 0.f;
    if (t < n_frames && n < frame_length) {
      const int i = t * step + n;
      float s = yb[i];
      if (preemph > 0.f && i > 0) s -= preemph * yb[i - 1];
      v = s * window[n];
    }
    unsigned r = __brev((unsigned)n) >> (32 - log2n);
    re[f * n_fft + r] = v;
    im[f * n_fft + r] = 0.f;
  }
  __syncthreads();
  for (int s = 1; s <= log2n; ++s) {
    const int half = 1 << (s - 1);
    for (int e = tid; e < FPB * (n_fft / 2); e += 256) {
      const int f = e / (n_fft / 2), j = e - f * (n_fft / 2);
      const int k = j & (half - 1);
      const int i0 = ((j >> (s - 1)) << s) + k, i1 = i0 + half;
      const int tw = k * (n_fft >> s);
      const float c = twc[tw], sn = tws[tw];
      float* R = re + f * n_fft;
      float* I = im + f * n_fft;
      const float xr = R[i1], xi = I[i1];
      const float tr = c * xr - sn * xi, ti = c * xi + sn * xr;
      const float ur = R[i0], ui = I[i0];
      R[i1] = ur - tr; I[i1] = ui - ti;
      R[i0] = ur + tr; I[i0] = ui + ti;
    }
    __syncthreads();
  }
  for (int e = tid; e < FPB * nb; e += 256) {
    const int f = e / nb, k = e - f * nb;
    const float a = re[f * n_fft + k], c = im[f * n_fft + k];
    pw[e] = a * a + c * c;
  }
  __syncthreads();
  for (int o = tid; o < FPB * n_mels; o += 256) {
    const int f = o / n_mels, m = o - f * n_mels;
    const int t = t0 + f;
    if (t >= n_frames) continue;
    const float* P = pw + f * nb;
    float acc = 0.f;
    for (int k = 0; k < nb; ++k) acc = fmaf(fbT[(size_t)k * n_mels + m], P[k], acc);
    if (log_output) acc = 10.f * log10f(fmaxf(1e-10f, acc));
    out[((size_t)b * n_frames + t) * n_mels + m] = acc;
  }
}

// one workgroup per utterance: global max, then floor at max - top_db (power2db)
__global__ __launch_bounds__(256) void topdb_kernel(float* out, int n, float top_db) {
  __shared__ float red[4];
  float* o = out + (size_t)blockIdx.x * n;
  float m = -3.0e38f;
  for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, o[i]);
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float floor_ = m - top_db;
  for (int i = threadIdx.x; i < n; i += 256) o[i] = fmaxf(o[i], floor_);
}

}  // namespace

extern "C" int odin_stft_mel_db(const float* y, const float* window, const float* melfb_t,
                                float* out, int B, int n_samples, int frame_length,
                                int step_length, int n_fft, int n_mels, float preemph,
                                float top_db, int log_output, void* stream) {
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  if ((1 << log2n) != n_fft || n_fft < frame_length || n_fft > 2048)
    return odin_fail(-2, "stft_mel_db: n_fft must be a power of two in [frame_length, 2048]");
  if (n_samples < frame_length) return odin_fail(-2, "stft_mel_db: utterance shorter than a frame");
  const int n_frames = 1 + (n_samples - frame_length) / step_length;
  const int nb = n_fft / 2 + 1;
  size_t lds = (size_t)(2 * FPB * n_fft + n_fft + FPB * nb) * 4;
  dim3 grid((n_frames + FPB - 1) / FPB, B, 1);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mel_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  ODIN_LAUNCH(stft_mel_kernel, grid, dim3(256), lds, stream, y, window, melfb_t, out, n_samples,
              frame_length, step_length, n_fft, log2n, n_frames, n_mels, preemph, log_output);
  if (log_output && top_db >= 0.f)
    ODIN_LAUNCH(topdb_kernel, dim3(B), dim3(256), 0, stream, out, n_frames * n_mels, top_db);
  return odin_check_launch("stft_mel_db");
}
