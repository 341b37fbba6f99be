// mel.hip -- speech front-end on the GPU: pre-emphasis -> framing -> window -> real FFT ->
// |.|^2 -> Slaney mel filterbank -> dB with the per-utterance top_db floor, batched over
// utterances, ONE launch.
//
// Restates the reference's offline numpy path (odin/preprocessing/signal.py: pre_emphasis
// :955-967, stft :1442-1562, power_spectrogram :1623-1648, mels_spectrogram :1650-1691,
// power2db :636-680; wrapped by speech.py:655-929).  The reference computes in FLOAT64
// throughout (fp32 samples promoted by the float64 window) and its dB output spans 80 dB of
// dynamic range: an fp32 FFT cannot hold the weak bins to 1e-4 (its error floor sits ~1e-7 of
// the strongest bin's amplitude).  So this path computes in float64 too -- the whole front-end
// is ~1.3 GFLOP per 256 utterances, noise beside the training step, and MI355X runs fp64 vector
// code at half its fp32 rate -- and only the stored result is fp32.
//
// One workgroup per utterance (the top_db floor needs the utterance's global maximum):
//   1. FPB frames at a time: samples are pre-emphasised, windowed (the window carries 1/sum(w))
//      and packed two real samples per complex point, z[n] = x[2n] + i x[2n+1], into LDS in base-4
//      digit-reversed order;
//   2. an N/2-point complex FFT, radix-4 decimation in time, twiddles from a float64 table in
//      LDS; the real-input spectrum X[k], k = 0..N/2, follows from Z[k] and conj(Z[N/2-k]);
//   3. the power spectrum stays in LDS and is contracted with the mel filterbank, stored as its
//      non-zero band per filter (Slaney triangles touch 2..40 of the 257 bins);
//   4. 10*log10(max(1e-10, .)) is written and the running maximum kept; after the last frame the
//      workgroup re-reads its own [n_frames, n_mels] block (L2-hot) and applies max - top_db.
// HBM traffic: 4 B/sample read, 4 B/(frame, band) written (+ the same again through L2).
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct cplx {
  double re, im;
};
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }

// base-4 digit reversal of an index of 2*LOG4 bits
__device__ __forceinline__ int rev4(int n, int log4) {
  int r = 0;
  for (int d = 0; d < log4; ++d) {
    r = (r << 2) | (n & 3);
    n >>= 2;
  }
  return r;
}

// tw: e^{-2 pi i k / n_fft}, k = 0 .. n_fft/2 - 1 (float64, host-computed): serves both the
// N/2-point FFT (W_{N/2}^k = tw[2k]) and the real-input split (W_N^k = tw[k]).
__global__ __launch_bounds__(256) void stft_mel_f64_kernel(
    const float* __restrict__ y, const double* __restrict__ window, const double* __restrict__ tw_g,
    const double* __restrict__ fb_vals, const int* __restrict__ fb_band, float* __restrict__ out,
    int n_samples, int frame_length, int step, int n_fft, int log4, int radix2, int fpb,
    int n_frames, int n_mels, double preemph, double top_db, int log_output, int n_out,
    float* __restrict__ bmax /* [B][gridDim.y] block maxima when the frames are split over gridDim.y > 1 */, int fb_cap,
    long long* stamps /* diagnostics: wall-clock stamps of workgroup (0, 0) or null */) {
#ifdef ODIN_SIM
#define MEL_STAMP(i) ((void)0)
#else
  int stamp_n = 0;
#define MEL_STAMP(i)                                                                                              \
  do {                                                                                                            \
    if (stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0 && stamp_n < 60) stamps[stamp_n++] = (long long)wall_clock64(); \
  } while (0)
#endif
  ODIN_DYN_SMEM(double, smem);
  const int H = n_fft / 2, nb = H + 1;
  cplx* tw = reinterpret_cast<cplx*>(smem);                 // [H]
  cplx* Z = tw + H;                                          // [fpb][H]
  double* pw = reinterpret_cast<double*>(Z + (size_t)fpb * H);  // [fpb][nb | pad]
  const int nbp = nb | 1;                                    // odd pitch
  // (Round 6, tried and dropped: the window, the filterbank values and the band table staged in LDS -- 6.7 KB more per
  // workgroup, four instead of five workgroups per CU: 105.6 us against 100.8.  The in-kernel stamps
  // (odin_debug_set_mel_stamps, tools/stamps_mel.py) put a pass of 4 frames at 12 us -- staging 3.0, FFT 4.1 (four
  // barriers, 4- to 16-way bank conflicts of the 16-byte butterfly accesses), real-input split 1.8, mel + log10 2.8 --
  // and a workgroup's four passes at 42 us: the launch is two rounds of those.)
  (void)fb_cap;
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const double ref_value = 1.0;
  const float* yb = y + (size_t)b * n_samples;
  // (only the first n_out frames are stored; the top_db floor is still taken over the whole utterance)
  float* ob = out + (size_t)b * n_out * n_mels;
  for (int k = tid; k < H; k += 256) tw[k] = {tw_g[2 * k], tw_g[2 * k + 1]};
  float vmax = -3.0e38f;
  // (gridDim.y workgroups share an utterance: each takes every gridDim.y-th block of fpb frames; the top_db floor
  // then needs the maximum over all of them and is applied by mel_floor_kernel)
  for (int t0 = blockIdx.y * fpb; t0 < n_frames; t0 += gridDim.y * fpb) {
    __syncthreads();  // previous pass is done with Z / pw (and tw is staged)
    MEL_STAMP(0);
    // ---- 1. stage: pre-emphasis, window, pack, digit-reverse ----
    for (int e = tid; e < fpb * H; e += 256) {
      const int f = e / H, n = e - f * H;
      const int t = t0 + f;
      cplx v = {0.0, 0.0};
      if (t < n_frames) {
        const int j0 = 2 * n, j1 = 2 * n + 1;
        if (j0 < frame_length) {
          const int i = t * step + j0;
          double s = (double)yb[i];
          if (preemph > 0.0 && i > 0) s -= preemph * (double)yb[i - 1];
          v.re = s * window[j0];
        }
        if (j1 < frame_length) {
          const int i = t * step + j1;
          double s = (double)yb[i];
          if (preemph > 0.0) s -= preemph * (double)yb[i - 1];
          v.im = s * window[j1];
        }
      }
      // H = 4^log4, or 2 * 4^log4: then even points go to the first half, odd points to the
      // second (each base-4 digit-reversed) and one radix-2 stage combines the halves at the end
      const int dst = radix2 ? (n & 1) * (H >> 1) + rev4(n >> 1, log4) : rev4(n, log4);
      Z[(size_t)f * H + dst] = v;
    }
    __syncthreads();
    MEL_STAMP(1);
    // ---- 2. radix-4 DIT stages: L = 4, 16, .., H ----
    for (int s = 1; s <= log4; ++s) {
      const int L = 1 << (2 * s), Q = L >> 2;
      const int tstep = 2 * H / L;  // W_L^j = e^{-2 pi i j / L} = tw[j * n_fft / L]
      for (int e = tid; e < fpb * (H / 4); e += 256) {
        const int f = e / (H / 4), q = e - f * (H / 4);
        const int j = q % Q, base = (q / Q) * L;
        cplx* zf = Z + (size_t)f * H + base + j;
        // W_L^{r j}: tw holds e^{-2 pi i k / n_fft}; W_L^j = tw[j * n_fft / L], n_fft / L = 2H / L
        const int ti = j * tstep;
        const cplx a = zf[0];
        cplx bq = zf[Q], c = zf[2 * Q], d = zf[3 * Q];
        if (j != 0) {
          bq = cmul(bq, tw[ti]);
          c = cmul(c, tw[2 * ti]);
          // 3 * ti can reach 3/2 H > H: W^{k + H} = -W^k ... for an n_fft-periodic table of H
          // entries: tw index k >= H means e^{-2 pi i k / n_fft} = -tw[k - H]
          const int t3 = 3 * ti;
          cplx w3 = t3 < H ? tw[t3] : cplx{-tw[t3 - H].re, -tw[t3 - H].im};
          d = cmul(d, w3);
        }
        const cplx apc = cadd(a, c), amc = csub(a, c), bpd = cadd(bq, d), bmd = csub(bq, d);
        // -i * (b - d) = (bmd.im, -bmd.re)
        const cplx mi_bmd = {bmd.im, -bmd.re};
        zf[0] = cadd(apc, bpd);
        zf[Q] = cadd(amc, mi_bmd);
        zf[2 * Q] = csub(apc, bpd);
        zf[3 * Q] = csub(amc, mi_bmd);
      }
      __syncthreads();
    }
    if (radix2) {  // X[k] = A[k] + W_H^k B[k], X[k + H/2] = A[k] - W_H^k B[k]
      const int Hh = H >> 1;
      for (int e = tid; e < fpb * Hh; e += 256) {
        const int f = e / Hh, k = e - f * Hh;
        cplx* zf = Z + (size_t)f * H + k;
        const cplx a = zf[0];
        const cplx bq = k ? cmul(zf[Hh], tw[2 * k]) : zf[Hh];
        zf[0] = cadd(a, bq);
        zf[Hh] = csub(a, bq);
      }
      __syncthreads();
    }
    MEL_STAMP(2);
    // ---- 3. real-input split and power spectrum: X[k] = E[k] + W_N^k O[k] ----
    for (int e = tid; e < fpb * nb; e += 256) {
      const int f = e / nb, k = e - f * nb;
      const cplx* zf = Z + (size_t)f * H;
      const cplx zk = zf[k == H ? 0 : k];
      const cplx zc = zf[(H - k) % H];  // conj taken below
      const cplx E = {0.5 * (zk.re + zc.re), 0.5 * (zk.im - zc.im)};
      // O = -i/2 (Z[k] - conj(Z[H-k])) = ( (zk.im + zc.im)/2 , -(zk.re - zc.re)/2 )
      const cplx O = {0.5 * (zk.im + zc.im), -0.5 * (zk.re - zc.re)};
      const cplx w = k < H ? tw[k] : cplx{-1.0, 0.0};
      const cplx X = cadd(E, cmul(w, O));
      pw[(size_t)f * nbp + k] = X.re * X.re + X.im * X.im;
    }
    __syncthreads();
    MEL_STAMP(3);
    // ---- 4. mel bands, dB ----
    for (int o = tid; o < fpb * n_mels; o += 256) {
      const int f = o / n_mels, m = o - f * n_mels;
      const int t = t0 + f;
      if (t >= n_frames) continue;
      const int k0 = fb_band[3 * m], cnt = fb_band[3 * m + 1], off = fb_band[3 * m + 2];
      const double* P = pw + (size_t)f * nbp + k0;
      double acc = 0.0;
      for (int k = 0; k < cnt; ++k) acc = fma(fb_vals[off + k], P[k], acc);
      float r;
      if (log_output == 3) r = (float)log(acc + 1e-6);  // AudioFeatureLoader(log_mels=True)
      else if (log_output) r = (float)(10.0 * log10(fmax(1e-10, acc) / ref_value));
      else r = (float)acc;
      if (t < n_out) ob[(size_t)t * n_mels + m] = r;
      vmax = fmaxf(vmax, r);
    }
  }
  MEL_STAMP(4);
  if (!log_output || log_output == 3 || top_db < 0.0) return;
  // ---- per-utterance top_db floor (power2db: log_spec.max() - top_db over the whole utterance)
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k));
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  __syncthreads();  // also orders this workgroup's stores before its re-reads below
  if (gridDim.y > 1) {
    if (tid == 0) bmax[(size_t)b * gridDim.y + blockIdx.y] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    return;
  }
  const float floor_ = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - (float)top_db;
  const int n = n_out * n_mels;
  if (log_output == 2) {
    // unit range: the floored dB values span [max - top_db, max] -> (v - max) / top_db + 1 in [0, 1]
    const float mx = floor_ + (float)top_db, inv = 1.f / (float)top_db;
    for (int i = tid; i < n; i += 256) ob[i] = (fmaxf(ob[i], floor_) - mx) * inv + 1.f;
    return;
  }
  for (int i = tid; i < n; i += 256) {
    const float v = ob[i];
    if (v < floor_) ob[i] = floor_;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// n_fft = 512 (the reference's 16 kHz front-end: speech.py:655-929 with frame_length 400 / n_fft 512): the 256-point
// complex FFT as TWO radix-16 passes in registers (round 6).  The kernel above walks four radix-4 stages through LDS with
// a workgroup barrier behind each: a pass of 4 frames took 12 us alone and 19 us with five workgroups on the CU, bound by
// LDS issue (65 KB of LDS traffic per frame, the 16-byte butterfly accesses of the first two stages and the strided
// twiddle reads conflicting 2- to 16-way).  Here 16 lanes own a frame and a lane 16 of its points:
//   pass 1  lane n2 loads z[16 n1 + n2] (n1 = 0..15) straight from the samples (pre-emphasis, window from LDS), takes the
//           16-point DFT over n1 in registers and multiplies by W_256^(n2 k1) -- powers of the lane's own W_256^n2, built by
//           repeated multiplication in float64: no table reads
//   exchange through the frame's LDS block as [k1][n2] with a pitch of 17 points: writes and reads conflict-free
//   pass 2  lane k1 takes the 16-point DFT over n2: Z[k1 + 16 k2]
//   split   Z goes back to LDS in natural order, a lane reads its partners Z[256 - k], forms X[k] = E[k] + W_512^k O[k]
//           and the power spectrum, which overlays the block; the SAME 16 lanes contract it with the mel filterbank
//           (lane u: filters u, u + 16, ..) and write the dB values
// A frame never leaves its 16 lanes, so the loop has no workgroup barrier at all -- wave-private LDS, rendezvous of the
// wave only (odin_wave_sync) -- and 21 KB of conflict-free LDS traffic per frame.  16 frames per pass, 76 KB of LDS, two
// workgroups per CU.  Same arithmetic in float64 as above (the twiddle powers differ from the table by <= 15 ulp).
constexpr int M16_PITCH = 17;                       // complex points per k1 row of the exchange block
constexpr int M16_FRAME = 16 * M16_PITCH;           // complex points of a frame's LDS block (>= 256 natural, >= 257 doubles)

// the value of the lane to the left within the row of 16 lanes (lane 0 of a row: the row's lane 15)
__device__ __forceinline__ float m16_rot1(float v) {
#ifdef ODIN_SIM
  const int lane = threadIdx.x & 63;
  return __shfl(v, (lane & ~15) | ((lane - 1) & 15));
#else
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xF, 0xF, false));   // row_ror:1
#endif
}
__device__ __forceinline__ void m16_r4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
  const cplx a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = csub(x1, x3);
  const cplx mid = {d.im, -d.re};   // -i (x1 - x3)
  x0 = cadd(a, c); x1 = cadd(b, mid); x2 = csub(a, c); x3 = csub(b, mid);
}
// 16-point DFT, natural order in and out: n = 4 b + a, k = c + 4 d;
// X[c + 4 d] = sum_a W4^(a d) W16^(a c) sum_b x[4 b + a] W4^(b c)
__device__ __forceinline__ void m16_dft16(const cplx (&x)[16], cplx (&X)[16]) {
  cplx v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = x[i];
#pragma unroll
  for (int a = 0; a < 4; ++a) m16_r4(v[a], v[4 + a], v[8 + a], v[12 + a]);   // v[4 c + a] = T[a][c]
  const double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, r = 0.70710678118654752440;
  const cplx w1 = {c1, -s1}, w2 = {r, -r}, w3 = {s1, -c1}, w6 = {-r, -r}, w9 = {-c1, s1};
  v[4 + 1] = cmul(v[4 + 1], w1); v[8 + 1] = cmul(v[8 + 1], w2); v[12 + 1] = cmul(v[12 + 1], w3);
  v[4 + 2] = cmul(v[4 + 2], w2); v[8 + 2] = cplx{v[8 + 2].im, -v[8 + 2].re}; v[12 + 2] = cmul(v[12 + 2], w6);
  v[4 + 3] = cmul(v[4 + 3], w3); v[8 + 3] = cmul(v[8 + 3], w6); v[12 + 3] = cmul(v[12 + 3], w9);
#pragma unroll
  for (int c = 0; c < 4; ++c) m16_r4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);   // v[4 c + d] = X[c + 4 d]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) X[c + 4 * d] = v[4 * c + d];
}

__global__ __launch_bounds__(256, 2) void stft_mel512_kernel(
    const float* __restrict__ y, const double* __restrict__ window, const double* __restrict__ tw_g,
    const double* __restrict__ fb_vals, const int* __restrict__ fb_band, float* __restrict__ out,
    int n_samples, int frame_length, int step, int n_frames, int n_mels, double preemph, double top_db,
    int log_output, int n_out, float* __restrict__ bmax) {
  ODIN_DYN_SMEM(double, smem);
  constexpr int H = 256;
  cplx* tw = reinterpret_cast<cplx*>(smem);    // [256]: e^(-2 pi i k / 512)
  double* win = smem + 2 * H;                  // [512]: the window, zero beyond frame_length
  cplx* blk = reinterpret_cast<cplx*>(win + 512);   // [16 frames][M16_FRAME]
  __shared__ float red[4];
  const int tid = threadIdx.x, g = tid >> 4, u = tid & 15;
  const int b = blockIdx.x;
  const float* yb = y + (size_t)b * n_samples;
  float* ob = out + (size_t)b * n_out * n_mels;
  for (int k = tid; k < H; k += 256) tw[k] = {tw_g[2 * k], tw_g[2 * k + 1]};
  for (int k = tid; k < 512; k += 256) win[k] = k < frame_length ? window[k] : 0.0;
  __syncthreads();
  // W_256^u = tw[2 u]
  const cplx wu = tw[2 * u];
  cplx* fb = blk + (size_t)g * M16_FRAME;
  double* pw = reinterpret_cast<double*>(fb);
  float vmax = -3.0e38f;
  // (the clamp of a point beyond the last sample keeps the pair even only when n_samples is)
  const bool al8 = ((((size_t)yb) & 7) == 0) && (step & 1) == 0 && (n_samples & 1) == 0;
  for (int t0 = blockIdx.y * 16; t0 < n_frames; t0 += gridDim.y * 16) {
    const int t = t0 + g;
    const bool live = t < n_frames;
    const int tc = live ? t : n_frames - 1;
    // ---- pass 1: the lane's 16 points z[16 n1 + u] = x[2 n] + i x[2 n + 1] ----
    cplx x[16], A[16];
    {
      // the sample in front of a pair is the left neighbour lane's second one (lane 0 of the frame: lane 15's of the
      // previous n1; the very first one is loaded); pairs that are 8-byte aligned -- even n_samples and step -- are one load
      float y0[16], y1[16];
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        int i0 = tc * step + 2 * (16 * n1 + u);
        if (i0 > n_samples - 2) i0 = n_samples - 2;   // (beyond the frame: the window is zero there)
        if (al8) {
          const float2 v2 = *reinterpret_cast<const float2*>(yb + i0);
          y0[n1] = v2.x; y1[n1] = v2.y;
        } else {
          y0[n1] = yb[i0];
          y1[n1] = yb[i0 + 1];
        }
      }
      const int if0 = tc * step;
      const float pm0 = yb[if0 > 0 ? if0 - 1 : 0];
      const double lv = live ? 1.0 : 0.0;
      float rprev = pm0;
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int j0 = 2 * (16 * n1 + u);
        const int i0 = tc * step + j0;
        const float rcur = m16_rot1(y1[n1]);
        const float prev = u != 0 ? rcur : rprev;
        rprev = rcur;
        double s0 = (double)y0[n1], s1 = (double)y1[n1];
        if (preemph > 0.0) {
          s1 -= preemph * (double)y0[n1];
          if (i0 > 0) s0 -= preemph * (double)prev;
        }
        x[n1] = {s0 * win[j0] * lv, s1 * win[j0 + 1] * lv};
      }
    }
    m16_dft16(x, A);
    {
      cplx wk = wu;
      fb[u] = A[0];
#pragma unroll
      for (int k1 = 1; k1 < 16; ++k1) {
        fb[k1 * M16_PITCH + u] = cmul(A[k1], wk);
        wk = cmul(wk, wu);
      }
    }
    odin_wave_sync();
    // ---- pass 2: lane k1 = u, DFT over n2 ----
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) x[n2] = fb[u * M16_PITCH + n2];
    m16_dft16(x, A);   // A[k2] = Z[u + 16 k2]
    odin_wave_sync();  // every lane has read the exchange block
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) fb[u + 16 * k2] = A[k2];
    odin_wave_sync();
    // ---- real-input split, power spectrum ----
    double P[16], Pn = 0.0;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      const int k = u + 16 * k2;
      const cplx zk = A[k2], zc = fb[(H - k) & (H - 1)];
      const cplx E = {0.5 * (zk.re + zc.re), 0.5 * (zk.im - zc.im)};
      const cplx O = {0.5 * (zk.im + zc.im), -0.5 * (zk.re - zc.re)};
      const cplx X = cadd(E, cmul(tw[k], O));
      P[k2] = X.re * X.re + X.im * X.im;
      if (k == 0) {   // X[256] = E[0] - O[0]
        const double xn = E.re - O.re, xi = E.im - O.im;
        Pn = xn * xn + xi * xi;
      }
    }
    odin_wave_sync();  // every lane has read its partners: the power spectrum overlays the block
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) pw[u + 16 * k2] = P[k2];
    if (u == 0) pw[H] = Pn;
    odin_wave_sync();
    // ---- mel bands, dB: lane u takes filters u, u + 16, .. of its own frame ----
    for (int m = u; m < n_mels; m += 16) {
      const int k0 = fb_band[3 * m], cnt = fb_band[3 * m + 1], off = fb_band[3 * m + 2];
      double acc = 0.0;
      for (int k = 0; k < cnt; ++k) acc = fma(fb_vals[off + k], pw[k0 + k], acc);
      float r;
      if (log_output == 3) r = (float)log(acc + 1e-6);
      else if (log_output) r = (float)(10.0 * log10(fmax(1e-10, acc)));
      else r = (float)acc;
      if (live) {
        if (t < n_out) ob[(size_t)t * n_mels + m] = r;
        vmax = fmaxf(vmax, r);
      }
    }
    odin_wave_sync();  // the block is free for the next pass
  }
  if (!log_output || log_output == 3 || top_db < 0.0) return;
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, k));
  if ((tid & 63) == 0) red[tid >> 6] = vmax;
  __syncthreads();  // also orders this workgroup's stores before its re-reads below
  if (gridDim.y > 1) {
    if (tid == 0) bmax[(size_t)b * gridDim.y + blockIdx.y] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    return;
  }
  const float floor_ = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - (float)top_db;
  const int n = n_out * n_mels;
  if (log_output == 2) {
    const float mx = floor_ + (float)top_db, inv = 1.f / (float)top_db;
    for (int i = tid; i < n; i += 256) ob[i] = (fmaxf(ob[i], floor_) - mx) * inv + 1.f;
    return;
  }
  for (int i = tid; i < n; i += 256) {
    const float v = ob[i];
    if (v < floor_) ob[i] = floor_;
  }
}

// the top_db floor (and the unit-range map) of an utterance whose frames were computed by several workgroups
__global__ __launch_bounds__(256) void mel_floor_kernel(float* __restrict__ out, const float* __restrict__ bmax,
                                                        int nblk, int n, float top_db, int log_output) {
  float* ob = out + (size_t)blockIdx.x * n;
  float mxv = -3.0e38f;
  for (int k = 0; k < nblk; ++k) mxv = fmaxf(mxv, bmax[(size_t)blockIdx.x * nblk + k]);
  const float floor_ = mxv - top_db;
  if (log_output == 2) {
    const float inv = 1.f / top_db;
    for (int i = threadIdx.x; i < n; i += 256) ob[i] = (fmaxf(ob[i], floor_) - mxv) * inv + 1.f;
    return;
  }
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = ob[i];
    if (v < floor_) ob[i] = floor_;
  }
}

}  // namespace

static long long* g_mel_stamps = nullptr;
static bool g_mel_r16 = true;
// tests / A-B runs: 0 = n_fft 512 on the general kernel too; < 0 = only report.  Returns the previous value.
extern "C" int odin_debug_mel_r16(int enable) {
  const int old = g_mel_r16 ? 1 : 0;
  if (enable >= 0) g_mel_r16 = enable != 0;
  return old;
}
// diagnostics: workgroup (0, 0) of the front-end launch records 100 MHz wall-clock stamps (4 per pass + 1) or NULL: off
extern "C" int odin_debug_set_mel_stamps(void* buf) {
  g_mel_stamps = (long long*)buf;
  return 0;
}

extern "C" int odin_stft_mel_db_frames(const float* y, const double* window, const double* twiddles,
                                       const double* fb_vals, const int32_t* fb_band, float* out, int B,
                                       int n_samples, int frame_length, int step_length, int n_fft,
                                       int n_mels, double preemph, double top_db, int log_output,
                                       int n_out_frames, float* workspace, void* stream) {
  const int Hc = n_fft / 2;
  int log4 = 0, radix2 = 0;
  while ((1 << (2 * log4 + 2)) <= Hc) ++log4;
  if ((1 << (2 * log4)) == Hc) radix2 = 0;
  else if (2 * (1 << (2 * log4)) == Hc) radix2 = 1;
  else radix2 = -1;
  if (radix2 < 0 || n_fft < frame_length || n_fft > 2048 || n_fft < 16)
    return odin_fail(-2, "stft_mel_db: n_fft must be a power of two in [16, 2048] and >= frame_length");
  if (log_output == 2 && top_db <= 0.0)
    return odin_fail(-2, "stft_mel_db: the unit-range output (log_output=2) needs top_db > 0");
  if (n_samples < frame_length) return odin_fail(-2, "stft_mel_db: utterance shorter than a frame");
  const int n_frames = 1 + (n_samples - frame_length) / step_length;
  if (n_out_frames < 1 || n_out_frames > n_frames)
    return odin_fail(-2, "stft_mel_db: n_out_frames must be in [1, n_frames]");
  const int H = n_fft / 2, nbp = (H + 1) | 1;
  const bool r16 = g_mel_r16 && n_fft == 512 && n_samples >= 2;   // (the register radix-16 form)
  // frames per pass: 1024 complex points = 28 KB of LDS per workgroup, five workgroups per CU (with 4096 points --
  // 102 KB, one workgroup of 4 waves per CU -- the launch took 213 us instead of 100 us at batch 256)
  int fpb = 1024 / H;
  if (fpb > 16) fpb = 16;
  if (fpb < 1) fpb = 1;
  if (const char* e = ODIN_DIAG_ENV("ODIN_MEL_FPB")) { const int v = atoi(e); if (v >= 1 && v <= fpb) fpb = v; }
  if (r16) fpb = 16;
  const int fb_cap = 0;
  const size_t lds = r16 ? ((size_t)2 * 256 + 512 + (size_t)2 * 16 * M16_FRAME) * 8
                         : ((size_t)2 * H + (size_t)2 * fpb * H + (size_t)fpb * nbp) * 8;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    // (the kernel also has 16 bytes of static LDS: asking for the full 160 KB of DYNAMIC LDS is
    // rejected and leaves a sticky hipErrorInvalidValue behind)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mel_f64_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
      (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_mel512_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  // one workgroup per utterance leaves most of the chip idle at batch 256 (208 us for 0.4 GFLOP of float64): the
  // blocks of fpb frames of an utterance are dealt to up to 8 workgroups, the floor follows in a second launch
  // How many: every workgroup walks ceil(nblocks / gy) passes and the chip holds `slots` workgroups at a time (LDS: five
  // per CU at 28.7 KB), so the launch takes ceil(B gy / slots) rounds of that many passes -- round 5's power-of-two
  // choice (8 at batch 256: 2048 workgroups for 1280 slots, two rounds of 4 passes, 100 us) against 5 (one round of 5).
  const int nblocks = (n_frames + fpb - 1) / fpb;
  int gy = 1;
  if (workspace != nullptr && !ODIN_DIAG_ENV("ODIN_MEL_NOSPLIT")) {
    long per_cu = (long)(160 * 1024) / (long)(lds + 64);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const long slots = per_cu * odin_num_cus();
    long best = -1;
    for (int g = 1; g <= 8 && g <= nblocks; ++g) {
      const long cost = (((long)B * g + slots - 1) / slots) * ((nblocks + g - 1) / g);
      if (best < 0 || cost < best) { best = cost; gy = g; }
    }
    if (const char* e = ODIN_DIAG_ENV("ODIN_MEL_GY")) { const int v = atoi(e); if (v >= 1 && v <= 8) gy = v; }
  }
  if (r16)
    ODIN_LAUNCH(stft_mel512_kernel, dim3(B, gy), dim3(256), lds, stream, y, window, twiddles, fb_vals, (const int*)fb_band,
                out, n_samples, frame_length, step_length, n_frames, n_mels, preemph, top_db, log_output, n_out_frames,
                workspace);
  else
  ODIN_LAUNCH(stft_mel_f64_kernel, dim3(B, gy), dim3(256), lds, stream, y, window, twiddles, fb_vals,
              (const int*)fb_band, out, n_samples, frame_length, step_length, n_fft, log4, radix2,
              fpb, n_frames, n_mels, preemph, top_db, log_output, n_out_frames, workspace, fb_cap, g_mel_stamps);
  if (gy > 1 && log_output != 0 && log_output != 3 && top_db >= 0.0)
    ODIN_LAUNCH(mel_floor_kernel, dim3(B), dim3(256), 0, stream, out, (const float*)workspace, gy,
                n_out_frames * n_mels, (float)top_db, log_output);
  return odin_check_launch("stft_mel_db");
}

extern "C" int odin_stft_mel_db(const float* y, const double* window, const double* twiddles,
                                const double* fb_vals, const int32_t* fb_band, float* out, int B,
                                int n_samples, int frame_length, int step_length, int n_fft,
                                int n_mels, double preemph, double top_db, int log_output,
                                void* stream) {
  const int n_frames = n_samples >= frame_length ? 1 + (n_samples - frame_length) / step_length : 1;
  return odin_stft_mel_db_frames(y, window, twiddles, fb_vals, fb_band, out, B, n_samples, frame_length,
                                 step_length, n_fft, n_mels, preemph, top_db, log_output, n_frames, nullptr, stream);
}
