// pointwise.hip -- the HBM-bound kernels of the VAE step: latent reparameterisation + KL,
// fused observation log-likelihood forward+backward, ELBO finalisation, fused flat
// Keras-Adam, gradient-norm reduction and the counter-based RNG.
//
// Every kernel streams its operands exactly once with 16-byte accesses and reduces with
// wave shuffles; scalars that change from step to step (beta, 1/B, Adam's alpha_t, RNG
// step) are read from device memory so that a captured HIP graph can be replayed.
#include "odin_device.h"
#include "odin_internal.h"
#include "odin_latent_math.h"
#include <cstdlib>
#include <cstdint>

namespace {

constexpr float LOG2PI_F = 1.8378770664093453f;
constexpr float SOFTPLUS_INV1 = 0.5413248546129181f;

// block-wide sum of one float per thread (256 threads); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats LDS */) {
  v = wave_sum64(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) t = (red[0] + red[1]) + (red[2] + red[3]);
  return t;
}

// ------------------------------------------------------------------ latent ----------
// one WAVE per sample (lanes over the latent dimensions, KL summed by a fixed butterfly): with one thread per
// sample the CelebA head (B = 512, D = 45) was two workgroups walking 45 softplus / log evaluations in series
// from strided loads -- 38 us
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* p, const float* eps,
                                                         float* z, float* kl, float* fbmask,
                                                         int B, int D, int analytic,
                                                         float free_bits, const float* cap) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* pb = p + (size_t)b * 2 * D;
  float acc = 0.f;
  for (int d = lane; d < D; d += 64) {
    float loc = pb[d], sc = softplus_f(pb[D + d]), e = eps[(size_t)b * D + d];
    float zz = loc + sc * e;
    z[(size_t)b * D + d] = zz;
    float ls = odin_log(sc);
    if (analytic == 2) acc += ls + 0.5f * (1.f + loc * loc) / (sc * sc) - 0.5f;  // KL(p || q), reverse=False
    else if (analytic) acc += 0.5f * (sc * sc + loc * loc - 1.f) - ls;
    else acc += 0.5f * (zz * zz - e * e) - ls;
  }
  acc = wave_sum64(acc);
  if (lane != 0) return;
  float m = 1.f;
  if (free_bits >= 0.f) {
    float thr = free_bits * (float)D;
    if (!(acc > thr)) { acc = thr; m = 0.f; }
  }
  if (cap != nullptr) {  // BetaCapacityVAE (beta_vae.py:171-177): |kl - C(step)|, gradient sign(kl - C)
    const float d = acc - cap[0];
    m *= d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    acc = fabsf(d);
  }
  kl[b] = acc;
  fbmask[b] = m;
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* p, const float* eps,
                                                         const float* z, const float* dz,
                                                         const float* dz2,
                                                         const float* fbmask, const float* klw,
                                                         const float* dloc_x,
                                                         const float* dscale_x, float* dp, int B,
                                                         int D, int analytic) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  int b = i / D, d = i - b * D;
  const float* pb = p + (size_t)b * 2 * D;
  float loc = pb[d], raw = pb[D + d], sc = softplus_f(raw), e = eps[i], zz = z[i];
  float w = klw[0] * fbmask[b];
  float dloc, dsc;
  if (analytic == 2) {
    const float i2 = 1.f / (sc * sc);
    dloc = w * loc * i2;
    dsc = w * (1.f / sc - (1.f + loc * loc) * i2 / sc);
  } else if (analytic) { dloc = w * loc; dsc = w * (sc - 1.f / sc); }
  else { dloc = w * zz; dsc = w * (zz * e - 1.f / sc); }
  if (dz != nullptr) { float g = dz[i]; dloc += g; dsc += g * e; }
  if (dz2 != nullptr) { float g = dz2[i]; dloc += g; dsc += g * e; }
  if (dloc_x != nullptr) dloc += dloc_x[i];
  if (dscale_x != nullptr) dsc += dscale_x[i];
  dp[(size_t)b * 2 * D + d] = dloc;
  dp[(size_t)b * 2 * D + D + d] = dsc * sigmoid_f(raw);
}

// ------------------------------------------------------------------ ELBO ------------
constexpr int ELBO_CHUNK = 1024;  // elements per workgroup (256 threads x float4)

__global__ __launch_bounds__(256) void elbo_bernoulli_kernel(const float* __restrict__ logits,
                                                             const float* __restrict__ x,
                                                             float* __restrict__ llk_part,
                                                             float* __restrict__ dlogits,
                                                             const float* __restrict__ scale,
                                                             int N, int n_part, int vec) {
  __shared__ float red[4];
  const int b = blockIdx.x / n_part, part = blockIdx.x - b * n_part;
  const size_t base = (size_t)b * N;
  const int i0 = part * ELBO_CHUNK + threadIdx.x * 4;
  const float sc = scale[0];
  float acc = 0.f;
  if (vec && i0 + 3 < N) {
    float4 l = *reinterpret_cast<const float4*>(logits + base + i0);
    float4 t = *reinterpret_cast<const float4*>(x + base + i0);
    float4 g;
    acc += t.x * l.x - softplus_f(l.x); g.x = (sigmoid_f(l.x) - t.x) * sc;
    acc += t.y * l.y - softplus_f(l.y); g.y = (sigmoid_f(l.y) - t.y) * sc;
    acc += t.z * l.z - softplus_f(l.z); g.z = (sigmoid_f(l.z) - t.z) * sc;
    acc += t.w * l.w - softplus_f(l.w); g.w = (sigmoid_f(l.w) - t.w) * sc;
    *reinterpret_cast<float4*>(dlogits + base + i0) = g;
  } else {
    for (int j = 0; j < 4; ++j) {
      int i = i0 + j;
      if (i < N) {
        float l = logits[base + i], t = x[base + i];
        acc += t * l - softplus_f(l);
        dlogits[base + i] = (sigmoid_f(l) - t) * sc;
      }
    }
  }
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) llk_part[blockIdx.x] = s;
}



// ---- QuantizedLogistic(loc, softplus(raw) + e^-7, low=0, high=255, 'sigmoid') -------------
// odin/bay/distributions/quantized.py:50-204 over TFP's QuantizedDistribution (restated in
// oracle/vae_oracle.py: qlogistic_log_prob_elem): log P[Y = y] = log(exp(big) - exp(small)) with
// the survival-function pair right of the median and the cdf pair left of it; floor / ceil are
// taken of y = x * 255 formed in fp32, as the reference does.  Returns the element's log-prob and
// its derivatives wrt (loc, raw).
constexpr float QL_LOW = 0.f, QL_HIGH = 255.f, QL_SUPPORT = 127.5f, QL_MIN_SCALE = 9.1188196555451624e-4f;
__device__ __forceinline__ float softplus_acc(float u) { return fmaxf(u, 0.f) + log1pf(expf(-fabsf(u))); }
__device__ __forceinline__ float sigmoid_acc(float u) {
  const float e = expf(-fabsf(u));
  const float r = 1.f / (1.f + e);
  return u >= 0.f ? r : e * r;
}
__device__ __forceinline__ float qlogistic_elem(float loc, float raw, float t, float& dloc, float& draw) {
  const float m = QL_LOW + QL_SUPPORT * (loc + 1.f);
  const float s = (softplus_acc(raw) + QL_MIN_SCALE) * QL_SUPPORT;
  const float inv_s = 1.f / s;
  const float y = t * QL_HIGH;
  const float NINF = -__builtin_inff();
  // survival side: j = ceil(y), ceil(y - 1); cdf side: j = floor(y), floor(y - 1)
  const float jc = ceilf(y), jc1 = ceilf(y - 1.f), jf = floorf(y), jf1 = floorf(y - 1.f);
  const float u_s = (jc + 0.5f - m) * inv_s, u_s1 = (jc1 + 0.5f - m) * inv_s;
  const float u_c = (jf + 0.5f - m) * inv_s, u_c1 = (jf1 + 0.5f - m) * inv_s;
  const float lsy = jc < QL_HIGH ? (jc < QL_LOW ? 0.f : -softplus_acc(u_s)) : NINF;
  const float lcy = jf < QL_HIGH ? (jf < QL_LOW ? NINF : -softplus_acc(-u_c)) : 0.f;
  const bool use_sf = lsy < lcy;
  float big, small, gb, gs, ub, us;  // g* = d(term)/du (0 on the clamped branches)
  if (use_sf) {
    big = jc1 < QL_HIGH ? (jc1 < QL_LOW ? 0.f : -softplus_acc(u_s1)) : NINF;
    small = lsy;
    gb = (jc1 >= QL_LOW && jc1 < QL_HIGH) ? -sigmoid_acc(u_s1) : 0.f;
    gs = (jc >= QL_LOW && jc < QL_HIGH) ? -sigmoid_acc(u_s) : 0.f;
    ub = u_s1; us = u_s;
  } else {
    big = lcy;
    small = jf1 < QL_HIGH ? (jf1 < QL_LOW ? NINF : -softplus_acc(-u_c1)) : 0.f;
    gb = (jf >= QL_LOW && jf < QL_HIGH) ? sigmoid_acc(-u_c) : 0.f;
    gs = (jf1 >= QL_LOW && jf1 < QL_HIGH) ? sigmoid_acc(-u_c1) : 0.f;
    ub = u_c; us = u_c1;
  }
  float res, wb, ws;
  if (small == NINF) {
    res = big; wb = 1.f; ws = 0.f;
  } else {
    const float d = big - small;                        // >= 0
    const float l1m = d < 0.6931471805599453f ? logf(-expm1f(-d)) : log1pf(-expf(-d));
    res = big + l1m;
    const float r = expf(-d);                           // e^small / e^big
    wb = 1.f / (1.f - r);
    ws = -r * wb;
  }
  const float dm = (wb * gb + ws * gs) * (-inv_s);
  const float ds = (wb * gb * ub + ws * gs * us) * (-inv_s);
  dloc = dm * QL_SUPPORT;
  draw = ds * QL_SUPPORT * sigmoid_acc(raw);
  return res;
}

// ---- streaming form (HBM roofline): one WAVE owns U*256 contiguous elements of one sample.
// All 2*U 16-byte loads of a lane are issued before the first use (U >= 3: >= 6 loads in flight
// per lane, ~100 KB per CU at 16 waves), there is no workgroup barrier and no LDS: the
// per-sample partial is reduced with wave shuffles and stored once per wave; a sample's
// N / (256*U) partials are summed by elbo_finalize.  12 B per element (read logits, x; write
// dlogits).  softplus / sigmoid share one v_exp, one v_log and one v_rcp per element.
__device__ __forceinline__ void bern1(float l, float t, float sc, float& acc, float& g) {
  const float e = odin_exp2(-1.4426950408889634f * fabsf(l));   // exp(-|l|) in (0, 1]
  const float r = odin_rcp(1.f + e);
  acc += t * l - (fmaxf(l, 0.f) + 0.6931471805599453f * odin_log2(1.f + e));
  const float sg = l >= 0.f ? r : e * r;
  g = (sg - t) * sc;
}

template <int U>
__global__ __launch_bounds__(256) void elbo_bernoulli_stream_kernel(
    const float4* __restrict__ logits, const float4* __restrict__ x, float* __restrict__ llk_part,
    float4* __restrict__ dlogits, const float* __restrict__ scale, size_t n_waves) {
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  const int lane = threadIdx.x & 63;
  const size_t base = w * (size_t)(64 * U) + lane;
  float4 l[U], t[U];
#pragma unroll
  for (int u = 0; u < U; ++u) l[u] = logits[base + 64 * u];
#pragma unroll
  for (int u = 0; u < U; ++u) t[u] = x[base + 64 * u];
  const float sc = scale[0];
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float4 g;
    bern1(l[u].x, t[u].x, sc, acc, g.x);
    bern1(l[u].y, t[u].y, sc, acc, g.y);
    bern1(l[u].z, t[u].z, sc, acc, g.z);
    bern1(l[u].w, t[u].w, sc, acc, g.w);
    // non-temporal store: measured on the same traffic (tools/elbo_ceiling.py, odin_debug_stream_probe) a 38 MB cold
    // stream reaches 4.4 TB/s with plain stores and 5.7 TB/s with streaming ones -- the written lines do not wait in
    // the caches behind the reads
    odin_store4_stream(dlogits + base + 64 * u, g);
  }
  acc = wave_sum64(acc);
  if (lane == 0) llk_part[w] = acc;
}

// The same as a persistent grid: 512 workgroups walk the 256-element chunks (64 lanes x float4; a chunk lies inside
// one sample) grid-stride with U chunks of loads in flight per lane and streaming stores; one partial per chunk.
// On the stream probes (tools/elbo_ceiling.py) this launch shape with non-temporal stores is the fastest
// 2-in / 1-out stream of this length on the part (5.6-5.7 TB/s cold against 4.4 for the chunk-per-wave shape).
template <int U>
__global__ __launch_bounds__(256) void elbo_bernoulli_gs_kernel(
    const float4* __restrict__ logits, const float4* __restrict__ x, float* __restrict__ llk_part,
    float4* __restrict__ dlogits, const float* __restrict__ scale, size_t n_chunks) {
  const size_t nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  const float sc = scale[0];
  for (size_t c0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); c0 < n_chunks; c0 += nw * U) {
    float4 l[U], t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = c0 + u * nw;  // (wave-uniform)
      if (c < n_chunks) {
        l[u] = logits[c * 64 + lane];
        t[u] = x[c * 64 + lane];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = c0 + u * nw;
      if (c < n_chunks) {
        float4 g;
        float acc = 0.f;
        bern1(l[u].x, t[u].x, sc, acc, g.x);
        bern1(l[u].y, t[u].y, sc, acc, g.y);
        bern1(l[u].z, t[u].z, sc, acc, g.z);
        bern1(l[u].w, t[u].w, sc, acc, g.w);
        odin_store4_stream(dlogits + c * 64 + lane, g);
        acc = odin_wave_sum64_valu(acc);
        if (lane == 0) llk_part[c] = acc;
      }
    }
  }
}

// The same walk software-pipelined (round 6): the loads of the NEXT U chunks are issued before the current ones are
// evaluated (exp / log / rcp per element) and stored, unconditionally (out-of-range chunks re-read chunk 0: a
// conditional load drains vmcnt); only the stores and the partial are guarded.
template <int U>
__global__ __launch_bounds__(256) void elbo_bernoulli_gsp_kernel(
    const float4* __restrict__ logits, const float4* __restrict__ x, float* __restrict__ llk_part,
    float4* __restrict__ dlogits, const float* __restrict__ scale, size_t n_chunks) {
  const size_t nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  const float sc = scale[0];
  size_t c0 = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c0 >= n_chunks) return;
  f32x4 l[U], t[U], ln[U], tn[U];
  const f32x4* lg = reinterpret_cast<const f32x4*>(logits);
  const f32x4* xg = reinterpret_cast<const f32x4*>(x);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t c = c0 + u * nw < n_chunks ? c0 + u * nw : 0;
    l[u] = lg[c * 64 + lane];
    t[u] = xg[c * 64 + lane];
  }
  for (;;) {
    const size_t c1 = c0 + nw * U;
    const bool more = c1 < n_chunks;   // (wave-uniform)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = (more && c1 + u * nw < n_chunks) ? c1 + u * nw : 0;
      ln[u] = lg[c * 64 + lane];
      tn[u] = xg[c * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t c = c0 + u * nw;
      if (c < n_chunks) {
        float4 g;
        float acc = 0.f;
        bern1(l[u].x, t[u].x, sc, acc, g.x);
        bern1(l[u].y, t[u].y, sc, acc, g.y);
        bern1(l[u].z, t[u].z, sc, acc, g.z);
        bern1(l[u].w, t[u].w, sc, acc, g.w);
        odin_store4_stream(dlogits + c * 64 + lane, g);
        acc = odin_wave_sum64_valu(acc);
        if (lane == 0) llk_part[c] = acc;
      }
    }
    if (!more) break;
#pragma unroll
    for (int u = 0; u < U; ++u) { l[u] = ln[u]; t[u] = tn[u]; }
    c0 = c1;
  }
}

// h [B, n_pix, 2C] (loc | raw scale), x [B, n_pix, C]; N = n_pix*C elements per sample
__global__ __launch_bounds__(256) void elbo_gaussian_kernel(const float* __restrict__ h,
                                                            const float* __restrict__ x,
                                                            float* __restrict__ llk_part,
                                                            float* __restrict__ dh,
                                                            const float* __restrict__ scale,
                                                            int N, int C, int n_part,
                                                            int softplus1) {
  __shared__ float red[4];
  const int b = blockIdx.x / n_part, part = blockIdx.x - b * n_part;
  const float sc = scale[0];
  float acc = 0.f;
  for (int j = 0; j < 4; ++j) {
    int i = part * ELBO_CHUNK + j * 256 + threadIdx.x;
    if (i < N) {
      int pix = i / C, c = i - pix * C;
      size_t hb = ((size_t)b * (N / C) + pix) * 2 * C;
      float loc = h[hb + c], raw = h[hb + C + c], t = x[(size_t)b * N + i];
      if (softplus1 == 2) {  // QuantizedLogistic head
        float gl, gr;
        acc += qlogistic_elem(loc, raw, t, gl, gr);
        dh[hb + c] = -gl * sc;
        dh[hb + C + c] = -gr * sc;
        continue;
      }
      float sd, dsd;
      if (softplus1 == 1) { sd = softplus_f(raw + SOFTPLUS_INV1); dsd = sigmoid_f(raw + SOFTPLUS_INV1); }
      else { sd = raw; dsd = 1.f; }
      float d = (t - loc) / sd;
      acc += -0.5f * d * d - odin_log(sd) - 0.5f * LOG2PI_F;
      dh[hb + c] = -(d / sd) * sc;
      dh[hb + C + c] = -((d * d - 1.f) / sd) * dsd * sc;
    }
  }
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) llk_part[blockIdx.x] = s;
}


// Gaussian head, streaming form.  h is pixel-interleaved ([pixel][loc_0..loc_C-1 | raw_0..raw_C-1]),
// so the lane that owns a pixel needs 2C consecutive floats: loading them per lane (96-byte
// lane stride at C = 3) touches every 128-byte line from several wave-instructions and thrashes
// the 32 KB L1 (measured 4.2 TB/s).  Instead a wave moves its 256*G pixels with lane-linear
// 16-byte loads (every line requested once), transposes through a wave-private LDS image so that
// each lane ends up with whole groups of 4 pixels, and sends the gradients back the same way.
// No workgroup barrier (wave-private LDS, in-order DS pipe); one partial per wave.  20 B/element.
template <int C, int G, int SP1>
__global__ __launch_bounds__(256) void elbo_gaussian_stream_kernel(
    const float4* __restrict__ h, const float4* __restrict__ x, float* __restrict__ llk_part,
    float4* __restrict__ dh, const float* __restrict__ scale, size_t n_waves) {
  constexpr int NH = 2 * C * G, NX = C * G;  // float4 per lane
  __shared__ float4 lds[4][(NH + NX) * 64];
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  const int lane = threadIdx.x & 63;
  float4* hs = lds[threadIdx.x >> 6];
  float4* xs = hs + NH * 64;
  const size_t bh = w * (size_t)(NH * 64), bx = w * (size_t)(NX * 64);
  float4 hv[NH], xv[NX];
#pragma unroll
  for (int j = 0; j < NH; ++j) hv[j] = h[bh + j * 64 + lane];
#pragma unroll
  for (int j = 0; j < NX; ++j) xv[j] = x[bx + j * 64 + lane];
#pragma unroll
  for (int j = 0; j < NH; ++j) hs[j * 64 + lane] = hv[j];
#pragma unroll
  for (int j = 0; j < NX; ++j) xs[j * 64 + lane] = xv[j];
  odin_wave_sync();
  const float sc = scale[0];
  float acc = 0.f;
  float4 o[NH];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    // group q of this lane: 4 pixels = float4 [2C*(q*64+lane), +2C) of the h image
    float hf[8 * C], xf[4 * C];
#pragma unroll
    for (int j = 0; j < 2 * C; ++j) {
      const float4 v = hs[(q * 64 + lane) * 2 * C + j];
      hf[4 * j] = v.x; hf[4 * j + 1] = v.y; hf[4 * j + 2] = v.z; hf[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float4 v = xs[(q * 64 + lane) * C + j];
      xf[4 * j] = v.x; xf[4 * j + 1] = v.y; xf[4 * j + 2] = v.z; xf[4 * j + 3] = v.w;
    }
    float of[8 * C];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float loc = hf[pp * 2 * C + c], raw = hf[pp * 2 * C + C + c];
        const float t = xf[pp * C + c];
        if (SP1 == 2) {  // QuantizedLogistic head
          float gl, gr;
          acc += qlogistic_elem(loc, raw, t, gl, gr);
          of[pp * 2 * C + c] = -gl * sc;
          of[pp * 2 * C + C + c] = -gr * sc;
          continue;
        }
        float sd, dsd;
        if (SP1 == 1) {
          // softplus1(raw) = softplus(raw + softplus^-1(1)); its derivative = sigmoid(same)
          const float a = raw + SOFTPLUS_INV1;
          const float e = odin_exp2(-1.4426950408889634f * fabsf(a));
          const float r = odin_rcp(1.f + e);
          sd = fmaxf(a, 0.f) + 0.6931471805599453f * odin_log2(1.f + e);
          dsd = a >= 0.f ? r : e * r;
        } else {
          sd = raw;
          dsd = 1.f;
        }
        const float inv = 1.f / sd;
        const float d = (t - loc) * inv;
        acc += -0.5f * d * d - 0.6931471805599453f * odin_log2(sd) - 0.5f * LOG2PI_F;
        of[pp * 2 * C + c] = -(d * inv) * sc;
        of[pp * 2 * C + C + c] = -((d * d - 1.f) * inv) * dsd * sc;
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * C; ++j)
      o[q * 2 * C + j] = make_float4(of[4 * j], of[4 * j + 1], of[4 * j + 2], of[4 * j + 3]);
  }
  odin_wave_sync();  // every lane has read its pixels: the image can be overwritten
#pragma unroll
  for (int q = 0; q < G; ++q)
#pragma unroll
    for (int j = 0; j < 2 * C; ++j) hs[(q * 64 + lane) * 2 * C + j] = o[q * 2 * C + j];
  odin_wave_sync();
#pragma unroll
  for (int j = 0; j < NH; ++j) odin_store4_stream(dh + bh + j * 64 + lane, hs[j * 64 + lane]);
  acc = wave_sum64(acc);
  if (lane == 0) llk_part[w] = acc;
}

// sum of n partials with 8 interleaved accumulators and a pairwise finish: many equal-signed
// partials added in one chain round the same way every time (128 equal addends drift 1e-6 relative)
__device__ __forceinline__ float sum_partials8(const float* __restrict__ q, int n) {
  float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int j = 0;
  for (; j + 8 <= n; j += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += q[j + u];
  }
  for (int u = 0; j < n; ++j, ++u) t[u] += q[j];
  return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

__device__ __forceinline__ void elbo_finalize_body(const float* llk_part, int n_part, const float* kl,
                                                    const float* hyper, const float* tcp, float* llk,
                                                    float* out4, int B, float* red /* >= 4 floats LDS */) {
  float sl = 0.f, sk = 0.f;
  // (one thread per sample: 8 independent loads in flight per thread.  A wave per sample with coalesced loads was
  // tried for the Gaussian head's 240 partials per sample: 64 dependent rounds per wave, 18.8 -> 135 us)
  for (int b = threadIdx.x; b < B; b += 256) {
    float t = 0.f;
    t = sum_partials8(llk_part + (size_t)b * n_part, n_part);
    llk[b] = t;
    sl += t;
    sk += kl[b];
  }
  float tl = block_sum_256(sl, red);
  __syncthreads();
  float tk = block_sum_256(sk, red);
  if (threadIdx.x == 0) {
    float beta = hyper[0], tc = (tcp != nullptr) ? hyper[1] * tcp[0] : 0.f;
    float ml = tl / (float)B, mk = beta * tk / (float)B;
    out4[0] = -(ml - mk - tc);
    out4[1] = ml;
    out4[2] = mk;
    out4[3] = tc;
  }
}

__global__ __launch_bounds__(256) void elbo_finalize_kernel(const float* llk_part, int n_part,
                                                            const float* kl, const float* hyper,
                                                            const float* tcp, float* llk,
                                                            float* out4, int B) {
  __shared__ float red[4];
  elbo_finalize_body(llk_part, n_part, kl, hyper, tcp, llk, out4, B, red);
}

// ------------------------------------------------------------------ Adam ------------
__device__ __forceinline__ float adam1(float& th, float g, float& m, float& v, float a, float b1,
                                       float b2, float eps) {
  m = b1 * m + (1.f - b1) * g;
  v = b2 * v + (1.f - b2) * g * g;
  th = th - a * m / (sqrtf(v) + eps);
  return th;
}

// Per-step scalars without a per-step host copy (round 5).  The step's kernels read their hyper-parameters from one
// small device row `cur`; the host used to refresh it with an 80-byte H2D copy before every step graph, which cost
// 7-8 us of idle time per step (copy node + the gaps around it; same-box A/B in profiles/r05_hyper_ring.txt).  Now the
// rows of the coming steps sit in a device ring (filled by the host half a ring at a time, stream-ordered) and the
// step's LAST kernel -- this one -- loads the next step's row into `cur`.  Race-free: the stage-1 launch in front
// of this one copies Adam's own five scalars to `staged` (workgroup 0), so no workgroup of this kernel reads `cur`
// while workgroup 0 rewrites it; every other reader of `cur` runs in an earlier launch of the step.
struct HyperRing {
  const float* ring;   // [rows][row_floats]
  float* cur;          // [row_floats]; word `t_word` holds the step number t as int32
  int rows, row_floats, t_word;
};

struct FinArgs {
  const float* llk_part; const float* kl; const float* hyper; const float* tcp;
  float* llk; float* out4;
  int n_part, B;
};
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ theta,
                                                   const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, const float* __restrict__ hyper,
                                                   const float* __restrict__ gnorm2, float clip,
                                                   int* flag, const float* __restrict__ parts,
                                                   int n_parts, float* __restrict__ gnorm2_out,
                                                   HyperRing hr, FinArgs fin) {
  __shared__ float red[4];
  // (odin_adam_ring_parts: one extra workgroup finalises the step's ELBO -- from the STAGED row: `hyper` itself is
  // being advanced by workgroup 0)
  if (fin.llk_part != nullptr && blockIdx.x + 1 == gridDim.x) {
    elbo_finalize_body(fin.llk_part, fin.n_part, fin.kl, fin.hyper, fin.tcp, fin.llk, fin.out4, fin.B, red);
    return;
  }
  // (hr.ring != null: `hyper` is the staged copy of the five scalars)
  const float a = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
  float gs = hyper[4];
  if (hr.ring != nullptr) {   // (a kernel argument: uniform over the grid)
    int t = 0;
    if (blockIdx.x == 0) t = reinterpret_cast<const int*>(hr.cur)[hr.t_word];
    __syncthreads();           // every thread has read t before its word is rewritten
    if (blockIdx.x == 0 && threadIdx.x < hr.row_floats) {
      const int slot = (t + 1) & (hr.rows - 1);
      hr.cur[threadIdx.x] = hr.ring[(size_t)slot * hr.row_floats + threadIdx.x];
    }
  }
  __shared__ float n2_sh;
  const int nblk = (int)gridDim.x - (fin.llk_part != nullptr ? 1 : 0);   // (workgroups that update parameters)
  if (parts != nullptr) {
    // fused second stage of the gradient-norm reduction: every block sums the stage-1 partials
    // in the same fixed order as sum_stage2 (one launch less on the critical path)
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_parts; i += 256) acc += parts[i];
    const float t = block_sum_256(acc, red);
    if (threadIdx.x == 0) {
      n2_sh = t;
      if (blockIdx.x == 0 && gnorm2_out != nullptr) gnorm2_out[0] = t;
    }
    __syncthreads();
  }
  if (gnorm2 != nullptr || parts != nullptr) {
    float n2 = parts != nullptr ? n2_sh : gnorm2[0];
    // NaN / Inf gradients: skip the update and raise the flag -- unless the caller passed no flag
    // (nan_gradients_policy='ignore', base_networks.py:519-547: the update is applied whatever the gradients hold)
    if ((!(n2 == n2) || n2 > 3.0e38f) && flag != nullptr) {
      if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 1;
      return;
    }
    if (clip > 0.f) gs *= clip / fmaxf(sqrtf(n2), clip);
  }
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)nblk * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<float4*>(theta)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam1(t.x, gg.x * gs, mm.x, vv.x, a, b1, b2, eps);
    adam1(t.y, gg.y * gs, mm.y, vv.y, a, b1, b2, eps);
    adam1(t.z, gg.z * gs, mm.z, vv.z, a, b1, b2, eps);
    adam1(t.w, gg.w * gs, mm.w, vv.w, a, b1, b2, eps);
    reinterpret_cast<float4*>(theta)[i] = t;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    size_t i = (n4 << 2) + threadIdx.x;
    float t = theta[i], mm = m[i], vv = v[i];
    adam1(t, g[i] * gs, mm, vv, a, b1, b2, eps);
    theta[i] = t; m[i] = mm; v[i] = vv;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// odin_adam_step_fold (round 6): an Adam step over a flat parameter buffer whose LAST small gradient pieces are formed
// inside the launch.  FactorVAE's discriminator step ended  ... -> think_wgrad -> slab_reduce -> adam: the first layer's
// [zdim, 1000] weight gradient (0.8 MFLOP) and the reduction of 32 slab rows of 1001 / 7000 floats, each a 5 us launch
// floor in front of an 18 us Adam launch that does not need them until its own first / last few workgroups.  Here
//   workgroups [0, nA)       32 columns of a thin-K Dense layer x 8 row groups: dW[k][n] = sum_b x[b][k] dy[b][n], db[n] =
//                            sum_b dy[b][n] (16 rows of loads in flight per thread, x in LDS), the row groups through LDS in
//                            ascending order, then Adam on the block's (K + 1) x 32 parameters
//   workgroups [nA, nA + nB) thread = element of a slab job: the sum of its rows (ascending), Adam
//   the rest                 the float4 grid-stride walk of adam_kernel over everything else; they also clear `zero`
// The folded gradients are written to g as the separate launches would have left them.  No clip / NaN guard (the
// discriminator's optimiser has none: factor_vae.py:168-176).
struct AdamFold {
  const float* x; const float* dy; int B, K, N; size_t offA; int nA;         // x == null: none
  const float* slab; int rows; size_t stride, nS, offS, endS; int nB;         // slab == null: none; endS: offS + nS rounded up to 4
  unsigned* zero; int zero_n;
};

template <int KT>   // K rounded up to 8 / 16 / 32
__global__ __launch_bounds__(256) void adam_fold_kernel(float* __restrict__ theta, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n, const float* __restrict__ hyper,
                                                        AdamFold f) {
  const float a = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], gs = hyper[4];
  const int wg = (int)blockIdx.x, tid = threadIdx.x;
  if (wg < f.nA) {
    // 32 columns x 8 row groups (rows rg, rg + 8, ..): a thread's chain is B / 8 rows with 16 loads in flight; x staged in
    // LDS when it fits (broadcast reads); the row groups meet in LDS in ascending order
    __shared__ float xs[4096];
    __shared__ float redA[8 * (KT + 1) * 32];
    const int c = tid & 31, rg = tid >> 5;
    const int col = wg * 32 + c;
    const bool xl = f.B * f.K <= 4096;
    if (xl) {
      for (int e = tid; e < f.B * f.K; e += 256) xs[e] = f.x[e];
      __syncthreads();
    }
    float acc[KT], accb = 0.f;
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[k] = 0.f;
    const int colc = col < f.N ? col : f.N - 1;
    for (int b0 = rg; b0 < f.B; b0 += 8 * 16) {
      float dv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {   // (unconditional loads of clamped rows: thin_dense.hip)
        const int b = b0 + 8 * u < f.B ? b0 + 8 * u : f.B - 1;
        dv[u] = f.dy[(size_t)b * f.N + colc];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool ok = b0 + 8 * u < f.B;
        const int b = ok ? b0 + 8 * u : f.B - 1;
        const float d = ok ? dv[u] : 0.f;
        accb += d;
        const float* xr = xl ? xs + b * f.K : f.x + (size_t)b * f.K;   // (the same address in every lane of a row group)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k] = fmaf(k < f.K ? xr[k] : 0.f, d, acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) redA[(rg * (KT + 1) + k) * 32 + c] = acc[k];
    redA[(rg * (KT + 1) + KT) * 32 + c] = accb;
    __syncthreads();
    for (int o = tid; o < (f.K + 1) * 32; o += 256) {
      const int k = o >> 5, cc = o & 31, cg = wg * 32 + cc;
      if (cg >= f.N) continue;
      const int kk = k < f.K ? k : KT;   // (the bias sums sit behind the KT weight rows)
      float gv = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) gv += redA[(r * (KT + 1) + kk) * 32 + cc];
      const size_t i = f.offA + (size_t)k * f.N + cg;
      float t = theta[i], mm = m[i], vv = v[i];
      g[i] = gv;
      adam1(t, gv * gs, mm, vv, a, b1, b2, eps);
      theta[i] = t; m[i] = mm; v[i] = vv;
    }
    return;
  }
  if (wg < f.nA + f.nB) {
    const size_t e = (size_t)(wg - f.nA) * 256 + tid;
    const size_t i = f.offS + e;
    if (i >= f.endS || i >= n) return;
    float gv;
    if (e < f.nS) {
      gv = 0.f;
      for (int r = 0; r < f.rows; ++r) gv += f.slab[(size_t)r * f.stride + e];
      g[i] = gv;
    } else {
      gv = g[i];
    }
    float t = theta[i], mm = m[i], vv = v[i];
    adam1(t, gv * gs, mm, vv, a, b1, b2, eps);
    theta[i] = t; m[i] = mm; v[i] = vv;
    return;
  }
  const int w0 = wg - f.nA - f.nB, nblk = (int)gridDim.x - f.nA - f.nB;
  const size_t stride = (size_t)nblk * 256;
  for (size_t z = (size_t)w0 * 256 + tid; z < (size_t)f.zero_n; z += stride) f.zero[z] = 0u;
  const size_t a0 = f.x != nullptr ? f.offA : 0, a1 = f.x != nullptr ? f.offA + (size_t)(f.K + 1) * f.N : 0;
  const size_t s0 = f.slab != nullptr ? f.offS : 0, s1 = f.slab != nullptr ? f.endS : 0;
  const size_t n4 = n >> 2;
  for (size_t i = (size_t)w0 * 256 + tid; i < n4; i += stride) {
    const size_t e = i << 2;
    if ((e >= a0 && e < a1) || (e >= s0 && e < s1)) continue;
    float4 t = reinterpret_cast<float4*>(theta)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam1(t.x, gg.x * gs, mm.x, vv.x, a, b1, b2, eps);
    adam1(t.y, gg.y * gs, mm.y, vv.y, a, b1, b2, eps);
    adam1(t.z, gg.z * gs, mm.z, vv.z, a, b1, b2, eps);
    adam1(t.w, gg.w * gs, mm.w, vv.w, a, b1, b2, eps);
    reinterpret_cast<float4*>(theta)[i] = t;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (w0 == 0 && tid < (int)(n & 3)) {
    const size_t i = (n4 << 2) + tid;
    if (!((i >= a0 && i < a1) || (i >= s0 && i < s1))) {
      float t = theta[i], mm = m[i], vv = v[i];
      adam1(t, g[i] * gs, mm, vv, a, b1, b2, eps);
      theta[i] = t; m[i] = mm; v[i] = vv;
    }
  }
}

__global__ __launch_bounds__(256) void sumsq_stage1(const float* __restrict__ g, size_t n,
                                                    float* __restrict__ part, const float* __restrict__ adam_hyper,
                                                    float* __restrict__ staged) {
  __shared__ float red[4];
  if (staged != nullptr && blockIdx.x == 0 && threadIdx.x < 5) staged[threadIdx.x] = adam_hyper[threadIdx.x];
  float acc = 0.f;
  const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<const float4*>(g)[i];
    acc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float t = g[(n4 << 2) + threadIdx.x];
    acc += t * t;
  }
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// sumsq_stage1 whose LAST workgroup finalises the step's ELBO instead (elbo_finalize_kernel's work: 4.7 us as a
// launch of its own, nothing in the backward pass depends on it): one launch less per training step.
__global__ __launch_bounds__(256) void sumsq_stage1_fin(const float* __restrict__ g, size_t n,
                                                        float* __restrict__ part, FinArgs f,
                                                        const float* __restrict__ adam_hyper,
                                                        float* __restrict__ staged) {
  __shared__ float red[4];
  if (staged != nullptr && blockIdx.x == 0 && threadIdx.x < 5) staged[threadIdx.x] = adam_hyper[threadIdx.x];
  if (blockIdx.x + 1 == gridDim.x) {
    elbo_finalize_body(f.llk_part, f.n_part, f.kl, f.hyper, f.tcp, f.llk, f.out4, f.B, red);
    return;
  }
  float acc = 0.f;
  const size_t n4 = n >> 2, stride = (size_t)(gridDim.x - 1) * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 t = reinterpret_cast<const float4*>(g)[i];
    acc += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    float t = g[(n4 << 2) + threadIdx.x];
    acc += t * t;
  }
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_stage2(const float* part, int n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = s;
}

// ------------------------------------------------------------------ RNG -------------
__global__ __launch_bounds__(256) void rng_normal_kernel(float* out, size_t n, unsigned k0,
                                                         unsigned k1, const int* step_dev) {
  const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
  const size_t n4 = (n + 3) >> 2, stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float v[4];
    odin_normal4((unsigned)i, (unsigned)(i >> 32), step, k0, k1, v);
    for (int j = 0; j < 4; ++j)
      if (i * 4 + j < n) out[i * 4 + j] = v[j];
  }
}



// ------------------------------------------------------------------ marginal_log_prob --
// VariationalAutoencoder.marginal_log_prob (variational_autoencoder.py:396-513): n samples
// z_k ~ q(z|x) per input, log q(z_k|x), log p(z_k); log-mean-exp over k on device.
__global__ __launch_bounds__(256) void latent_sample_logprob_kernel(
    const float* __restrict__ p, const float* __restrict__ eps, float* __restrict__ z,
    float* __restrict__ logq, float* __restrict__ logp, int n, int B, int D) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // (k, b)
  if (i >= n * B) return;
  const int b = i % B;
  const float* pb = p + (size_t)b * 2 * D;
  float lq = 0.f, lp = 0.f;
  for (int d = 0; d < D; ++d) {
    const float loc = pb[d], sc = softplus_f(pb[D + d]), e = eps[(size_t)i * D + d];
    const float zz = loc + sc * e;
    z[(size_t)i * D + d] = zz;
    lq += -0.5f * e * e - odin_log(sc);
    lp += -0.5f * zz * zz;
  }
  const float c = 0.5f * LOG2PI_F * (float)D;
  logq[i] = lq - c;
  logp[i] = lp - c;
}

// out[b] = sum_j part[b * n_part + j]  (per-sample log-likelihood from the ELBO kernels' partials)
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, int n_part,
                                                        float* __restrict__ out, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  out[b] = sum_partials8(part + (size_t)b * n_part, n_part);
}

// out[b] = log sum_k exp(in[k][b]) - log n   (column-wise, rows of B contiguous floats)
__global__ __launch_bounds__(256) void logmeanexp_rows_kernel(const float* __restrict__ in,
                                                              float* __restrict__ out, int n, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float mx = -3.0e38f;
  for (int k = 0; k < n; ++k) mx = fmaxf(mx, in[(size_t)k * B + b]);
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += expf(in[(size_t)k * B + b] - mx);
  out[b] = mx + logf(s) - logf((float)n);
}

// ------------------------------------------------------------------ gradient policies --
// Networks.optimize between tape.gradient and apply_gradients (odin/networks/base_networks.py:
// 549-596), on the flat gradient buffer, in the reference's order: skip_update_threshold (any
// element >= threshold zeroes EVERY gradient of the step; Adam still runs on the zeros) ->
// per-tensor tf.clip_by_norm -> [tf.clip_by_global_norm: odin_sumsq_adam_flat] ->
// tf.clip_by_value.
__global__ __launch_bounds__(256) void grad_any_ge_kernel(const float* __restrict__ g, size_t n,
                                                          float thr, int* __restrict__ hit) {
  int h = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    h |= (g[i] >= thr) ? 1 : 0;
  // an integer OR: order-independent, so the atomic keeps the step reproducible
  if (h) atomicOr(hit, 1);
}

__global__ __launch_bounds__(256) void grad_zero_if_kernel(float* __restrict__ g, size_t n,
                                                           const int* __restrict__ hit,
                                                           const int* __restrict__ enable,
                                                           int* __restrict__ skipped) {
  if (hit[0] == 0 || (enable != nullptr && enable[0] == 0)) return;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    g[i] = 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0 && skipped != nullptr) skipped[0] += 1;
}

// one workgroup per tensor: ||g||_2 in a fixed order, then g *= clip / max(||g||, clip)
__global__ __launch_bounds__(1024) void clip_by_norm_segments_kernel(float* __restrict__ g,
                                                                     const long long* __restrict__ seg,
                                                                     float clip) {
  __shared__ float red[16];
  __shared__ float sc_sh;
  const long long a = seg[blockIdx.x], b = seg[blockIdx.x + 1];
  float acc = 0.f;
  for (long long i = a + threadIdx.x; i < b; i += 1024) acc += g[i] * g[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    sc_sh = clip / fmaxf(sqrtf(t), clip);
  }
  __syncthreads();
  const float sc = sc_sh;
  if (sc == 1.f) return;
  for (long long i = a + threadIdx.x; i < b; i += 1024) g[i] *= sc;
}

// g = clip_by_value(g * global_scale, -c, c); global_scale = gclip / max(sqrt(gnorm2), gclip) when
// a global-norm clip precedes the value clip (the reference's order), 1 otherwise
__global__ __launch_bounds__(256) void clip_by_value_kernel(float* __restrict__ g, size_t n, float c,
                                                            const float* __restrict__ gnorm2,
                                                            float gclip) {
  float gs = 1.f;
  if (gnorm2 != nullptr && gclip > 0.f) gs = gclip / fmaxf(sqrtf(gnorm2[0]), gclip);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    g[i] = fminf(fmaxf(g[i] * gs, -c), c);
}

inline int grid_for(size_t work_items, int per_block, int cap) {
  size_t g = (work_items + per_block - 1) / per_block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

extern "C" int odin_latent_fwd(const float* p, const float* eps, float* z, float* kl,
                               float* fbmask, int B, int D, int analytic, float free_bits,
                               const float* capacity, void* stream) {
  ODIN_LAUNCH(latent_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, p, eps, z, kl,
              fbmask, B, D, analytic, free_bits, capacity);
  return odin_check_launch("latent_fwd");
}

extern "C" int odin_latent_bwd(const float* p, const float* eps, const float* z, const float* dz,
                               const float* dz_extra, const float* fbmask, const float* klw,
                               const float* dloc_x, const float* dscale_x, float* dp, int B,
                               int D, int analytic, void* stream) {
  ODIN_LAUNCH(latent_bwd_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, p, eps, z, dz,
              dz_extra, fbmask, klw, dloc_x, dscale_x, dp, B, D, analytic);
  return odin_check_launch("latent_bwd");
}

// partial layout: llk_part[b * n_part + j]; n_part is reported to the caller (elbo_finalize sums)
static int elbo_stream_unroll(int n_per_sample) {
  // elements per wave = 256 * U must divide the sample; prefer >= 6 loads in flight per lane
  static const int cand[] = {3, 4, 2, 1};
  if (const char* e = ODIN_DIAG_ENV("ODIN_ELBO_U")) {  // diagnostics: A/B the loads in flight per lane
    const int u = atoi(e);
    return (u >= 1 && u <= 4 && n_per_sample % (256 * u) == 0) ? u : 0;
  }
  for (int u : cand)
    if (n_per_sample % (256 * u) == 0) return u;
  return 0;
}

// launch shape of the persistent Bernoulli ELBO kernel: {workgroups, chunks in flight per wave, software-pipelined}
// Round 6 (profiles/r06_elbo_sweep.txt, cold buffers): the pipelined body with ONE chunk in flight per wave and a grid of
// n_chunks / 6 workgroups (1.5 chunks per wave) reaches the stream probe's rate on both priced shapes -- 6.70 us = 0.704
// of 8 TB/s at batch 256 (probe 0.704; round 5's shape: 0.676), 12.1 us = 0.78 at batch 512 (probe 0.80; 0.736)
static int g_elbo_gs[3] = {0, 1, 1};   // (workgroups 0: from the tensor size)
// diagnostics / sweeps (tools/elbo_sweep6.py): set the launch shape (a negative field keeps its value); returns 0
extern "C" int odin_debug_elbo_shape(int blocks, int U, int pipelined) {
  if (blocks >= 0) g_elbo_gs[0] = blocks;   // (0: from the tensor size)
  if (U > 0) g_elbo_gs[1] = U;
  if (pipelined >= 0) g_elbo_gs[2] = pipelined;
  return 0;
}

extern "C" int odin_elbo_bernoulli_fwd_bwd(const float* logits, const float* x, float* llk_part,
                                           float* dlogits, const float* scale, int B,
                                           int n_per_sample, int* n_part_out, void* stream) {
  // large tensors: the persistent form (one partial per 256-element chunk)
  if (n_per_sample % 256 == 0 && (size_t)B * n_per_sample >= (1u << 20) && !ODIN_DIAG_ENV("ODIN_ELBO_U") &&
      (((uintptr_t)logits | (uintptr_t)x | (uintptr_t)dlogits) & 15) == 0) {
    const int n_part = n_per_sample / 256;
    if (n_part_out) *n_part_out = n_part;
    if (logits == nullptr) return 0;  // dry run: reports the partial count
    const size_t n_chunks = (size_t)B * n_part;
    int blocks = g_elbo_gs[0], U = g_elbo_gs[1];  // (sweeps: profiles/r04_elbo_stream_sweep.txt, r06_elbo_sweep.txt)
    if (blocks <= 0) {
      blocks = (int)((n_chunks / 6 + 511) / 512) * 512;
      blocks = blocks < 512 ? 512 : blocks > 4096 ? 4096 : blocks;
    }
    if (const char* e = ODIN_DIAG_ENV("ODIN_ELBO_GS")) sscanf(e, "%d,%d", &blocks, &U);  // diagnostics sweep
    if (g_elbo_gs[2]) {
#define ODIN_ELBO_GSP_LAUNCH(UU)                                                                             \
  ODIN_LAUNCH((elbo_bernoulli_gsp_kernel<UU>), dim3(blocks), dim3(256), 0, stream, (const float4*)logits,   \
              (const float4*)x, llk_part, (float4*)dlogits, scale, n_chunks)
      if (U == 1) ODIN_ELBO_GSP_LAUNCH(1);
      else if (U == 2) ODIN_ELBO_GSP_LAUNCH(2);
      else if (U == 3) ODIN_ELBO_GSP_LAUNCH(3);
      else ODIN_ELBO_GSP_LAUNCH(4);
#undef ODIN_ELBO_GSP_LAUNCH
      return odin_check_launch("elbo_bernoulli");
    }
#define ODIN_ELBO_GS_LAUNCH(UU)                                                                              \
  ODIN_LAUNCH((elbo_bernoulli_gs_kernel<UU>), dim3(blocks), dim3(256), 0, stream, (const float4*)logits,    \
              (const float4*)x, llk_part, (float4*)dlogits, scale, n_chunks)
    if (U == 2) ODIN_ELBO_GS_LAUNCH(2);
    else if (U == 3) ODIN_ELBO_GS_LAUNCH(3);
    else if (U == 6) ODIN_ELBO_GS_LAUNCH(6);
    else if (U == 8) ODIN_ELBO_GS_LAUNCH(8);
    else ODIN_ELBO_GS_LAUNCH(4);
#undef ODIN_ELBO_GS_LAUNCH
    return odin_check_launch("elbo_bernoulli");
  }
  const int U = elbo_stream_unroll(n_per_sample);
  if (U > 0 && (((uintptr_t)logits | (uintptr_t)x | (uintptr_t)dlogits) & 15) == 0) {
    const int n_part = n_per_sample / (256 * U);
    if (n_part_out) *n_part_out = n_part;
    if (logits == nullptr) return 0;  // dry run: reports the partial count
    const size_t n_waves = (size_t)B * n_part;
    const dim3 grid((unsigned)((n_waves + 3) / 4));
#define ODIN_ELBO_STREAM(UU)                                                                    \
  ODIN_LAUNCH((elbo_bernoulli_stream_kernel<UU>), grid, dim3(256), 0, stream,                   \
              (const float4*)logits, (const float4*)x, llk_part, (float4*)dlogits, scale, n_waves)
    if (U == 3) ODIN_ELBO_STREAM(3);
    else if (U == 4) ODIN_ELBO_STREAM(4);
    else if (U == 2) ODIN_ELBO_STREAM(2);
    else ODIN_ELBO_STREAM(1);
#undef ODIN_ELBO_STREAM
    return odin_check_launch("elbo_bernoulli");
  }
  int n_part = (n_per_sample + ELBO_CHUNK - 1) / ELBO_CHUNK;
  if (n_part_out) *n_part_out = n_part;
  if (logits == nullptr) return 0;
  int vec = (n_per_sample % 4 == 0) ? 1 : 0;
  ODIN_LAUNCH(elbo_bernoulli_kernel, dim3(B * n_part), dim3(256), 0, stream, logits, x, llk_part,
              dlogits, scale, n_per_sample, n_part, vec);
  return odin_check_launch("elbo_bernoulli");
}

// ---- Gaussian head: Conv2D 1x1 (Cin -> 2C maps, linear) -> Independent(Normal(loc, scale)).log_prob(target) forward +
// backward in ONE pass over the [pixels, Cin] activation (image_networks.py:505-511 + :95-102; the audio VAE's
// decoder, examples/vae/vae_audio.py:84-110).  Unfused this is four launches -- pw1x1 forward, the ELBO kernel, pw1x1
// weight gradient, pw1x1 data gradient -- and three passes over the activation (252 MB at [256, 96, 80, 32]); here
// every 16 bytes of it are read once and the gradient wrt it is written once.
// Lane <-> (pixel, channel quad) as in pw1x1.hip: the Q = Cin/4 lanes of a pixel butterfly their partial dots into
// the 2C logits (every lane of the pixel then holds them), evaluate the C elements of the pixel, and turn
// d(-llk)/d(logits) straight into this lane's four channels of dh = (dl w1^T) * act'(h), its share of dW1 / db1 and of
// the column sums of dh (the bias gradient of a Conv2DTranspose below).  A workgroup walks units of 256 pixels of
// one sample grid-stride (unit u = sample * n_units_per_sample + j; its 8 waves write llk_part[8 u + wave]: per sample
// n_part = 8 * units contiguous partials, the layout odin_elbo_finalize sums).
// Fixed-order reductions throughout: bit-reproducible.
constexpr int GH_NT = 512;   // threads per workgroup
constexpr int GH_U = 4;      // pixel groups in flight per lane (8: 116 -> 187 us, register pressure)
// SP1 = 0 / 1: Normal(loc, raw | softplus1(raw)) over 2C maps; SP1 = 3: Bernoulli(logits) over C maps
// (image_networks.py:87-93)
template <int C, int SP1>
__global__ __launch_bounds__(GH_NT) void gauss_head_kernel(
    const float4* __restrict__ h, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ target, float* __restrict__ logits, float* __restrict__ dlogits,
    float4* __restrict__ dh, float* __restrict__ llk_part, float* __restrict__ wslab, float* __restrict__ colsum,
    const float* __restrict__ scale, unsigned* dh_amax, int n_units, int n_part, int n_pix, int Q, int h_act,
    int upw) {
  constexpr int CO = SP1 == 3 ? C : 2 * C, NWV = GH_NT / 64;
  __shared__ float wl[64 * CO];
  __shared__ float red[NWV * 8 * (5 * CO + 4) + 16];
  const int CI = 4 * Q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < CI * CO; e += GH_NT) wl[e] = w1[e];
  __syncthreads();
  const int q = tid % Q;
  float wr[4][CO], br[CO];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < CO; ++o) wr[k][o] = wl[(4 * q + k) * CO + o];
#pragma unroll
  for (int o = 0; o < CO; ++o) br[o] = b1[o];
  const float sc = scale[0];
  const int ppg = GH_NT / Q;           // pixels per group
  const int ppu = ppg * GH_U;          // pixels per unit
  float acc[4][CO], db[CO];
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  float amx = 0.f;
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    db[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k][o] = 0.f;
  }
  // upw > 0: the n_part / upw workgroups of a sample deal its units round-robin (upw each: the launcher picks a divisor
  // of the units per sample), so neighbouring workgroups stream neighbouring 32 KB chunks at the same time, and a wave
  // keeps ONE log-likelihood partial for all of its units; upw == 0: units grid-stride, one partial per (unit, wave)
  float llk_run = 0.f;
  const int npw = upw > 0 ? n_part / upw : 1;
  const int u_first = upw > 0 ? ((int)blockIdx.x / npw) * n_part + (int)blockIdx.x % npw : (int)blockIdx.x;
  const int u_step = upw > 0 ? npw : (int)gridDim.x;
  const int u_end = upw > 0 ? ((int)blockIdx.x / npw + 1) * n_part : n_units;
  // the loads of the NEXT unit are issued before the current one is evaluated (C = 1: +20 registers, still four waves
  // per SIMD); with more channels the registers are not there and a unit is loaded when it is due
  constexpr bool PREF = (C == 1);
  float4 vn[GH_U];
  float tn[GH_U][C];
  auto load_unit = [&](int uu, float4 (&vv)[GH_U], float (&tt)[GH_U][C]) {
    const int b = uu / n_part, part = uu - b * n_part;
    const int pin = part * ppu + tid / Q;
    const size_t pbase = (size_t)b * n_pix;
#pragma unroll
    for (int u = 0; u < GH_U; ++u) {
      const int pi = pin + u * ppg;
      const bool ok = pi < n_pix && uu < u_end;
      const size_t p = pbase + pi;
      vv[u] = ok ? h[p * Q + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < C; ++c) tt[u][c] = ok ? target[p * C + c] : 0.f;
    }
  };
  if (PREF && u_first < u_end) load_unit(u_first, vn, tn);
  for (int u0 = u_first; u0 < u_end; u0 += u_step) {
    const int b = u0 / n_part, part = u0 - b * n_part;
    const int pin = part * ppu + tid / Q;          // pixel inside the sample (of group 0)
    const size_t pbase = (size_t)b * n_pix;
    float4 v[GH_U];
    float t[GH_U][C];
    if (PREF) {
#pragma unroll
      for (int u = 0; u < GH_U; ++u) {
        v[u] = vn[u];
#pragma unroll
        for (int c = 0; c < C; ++c) t[u][c] = tn[u][c];
      }
      load_unit(u0 + u_step, vn, tn);
    } else {
      load_unit(u0, v, t);
    }
    float llk = 0.f;
#pragma unroll
    for (int u = 0; u < GH_U; ++u) {
      const int pi = pin + u * ppg;
      const bool ok = pi < n_pix;            // (uniform over the Q lanes of the pixel)
      const size_t p = pbase + pi;
      float lg[CO];
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        float a = v[u].x * wr[0][o];
        a = fmaf(v[u].y, wr[1][o], a);
        a = fmaf(v[u].z, wr[2][o], a);
        a = fmaf(v[u].w, wr[3][o], a);
        for (int m = 1; m < Q; m <<= 1) a += __shfl_xor(a, m);  // fixed-order butterfly (wave-uniform trip count)
        lg[o] = a + br[o];
      }
      float dl[CO];
      float l1 = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        if constexpr (SP1 == 3) {
          bern1(lg[c], t[u][c], sc, l1, dl[c]);
          continue;
        }
        const float loc = lg[c], raw = lg[SP1 == 3 ? c : C + c];
        float sd, dsd;
        if (SP1 == 1) {  // softplus1(raw) = softplus(raw + softplus^-1(1)); its derivative = sigmoid(same)
          const float a = raw + SOFTPLUS_INV1;
          const float e = odin_exp2(-1.4426950408889634f * fabsf(a));
          const float r = odin_rcp(1.f + e);
          sd = fmaxf(a, 0.f) + 0.6931471805599453f * odin_log2(1.f + e);
          dsd = a >= 0.f ? r : e * r;
        } else {
          sd = raw;
          dsd = 1.f;
        }
        const float inv = 1.f / sd;
        const float d = (t[u][c] - loc) * inv;
        l1 += -0.5f * d * d - 0.6931471805599453f * odin_log2(sd) - 0.5f * LOG2PI_F;
        dl[c] = -(d * inv) * sc;
        dl[SP1 == 3 ? c : C + c] = -((d * d - 1.f) * inv) * dsd * sc;
      }
      if (!ok) {
#pragma unroll
        for (int o = 0; o < CO; ++o) dl[o] = 0.f;
        l1 = 0.f;
      }
      if (q == 0) llk += l1;
      if (ok && q == 0) {
#pragma unroll
        for (int o = 0; o < CO; ++o) logits[p * CO + o] = lg[o];
        if (dlogits != nullptr) {
#pragma unroll
          for (int o = 0; o < CO; ++o) dlogits[p * CO + o] = dl[o];
        }
      }
      float r[4];
      const float hv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          s = fmaf(dl[o], wr[k][o], s);
          acc[k][o] = fmaf(hv[k], dl[o], acc[k][o]);
        }
        r[k] = s * odin_act_grad(h_act, hv[k]);
        amx = fmaxf(amx, fabsf(r[k]));
      }
#pragma unroll
      for (int o = 0; o < CO; ++o) db[o] += dl[o];
      cs.x += r[0]; cs.y += r[1]; cs.z += r[2]; cs.w += r[3];
      if (ok) odin_store4_stream(dh + p * Q + q, make_float4(r[0], r[1], r[2], r[3]));  // (252 MB: read next from HBM anyway)
    }
    // one log-likelihood partial per (unit, wave): no workgroup barrier inside the loop, the waves run free and the
    // next unit's loads overlap this one's arithmetic
    if (upw > 0) {
      llk_run += llk;
    } else {
      llk = wave_sum64(llk);
      if (lane == 0) llk_part[(size_t)u0 * NWV + wave] = llk;
    }
  }
  if (upw > 0) {
    llk_run = wave_sum64(llk_run);
    if (lane == 0) llk_part[(size_t)blockIdx.x * NWV + wave] = llk_run;
  }
  // ---- this workgroup's slab rows: lanes of equal q inside a wave (xor butterfly over the masks >= Q), then the
  // waves in order ----
  constexpr int RW = 5 * CO + 4;
  float mine[RW];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < CO; ++o) mine[k * CO + o] = acc[k][o];
#pragma unroll
  for (int o = 0; o < CO; ++o) mine[4 * CO + o] = db[o];
  mine[5 * CO] = cs.x; mine[5 * CO + 1] = cs.y; mine[5 * CO + 2] = cs.z; mine[5 * CO + 3] = cs.w;
#pragma unroll
  for (int e = 0; e < RW; ++e) {
    float a = mine[e];
    for (int m = Q; m < 64; m <<= 1) a += __shfl_xor(a, m);
    mine[e] = a;
  }
  __syncthreads();
  if (lane < Q) {
#pragma unroll
    for (int e = 0; e < RW; ++e) red[(wave * 8 + lane) * RW + e] = mine[e];
  }
  __syncthreads();
  float* row = wslab + (size_t)blockIdx.x * (CI * CO + CO);
  for (int e = tid; e < CI * CO + CO; e += GH_NT) {
    float s = 0.f;
    if (e < CI * CO) {
      const int c = e / CO, o = e - c * CO;
      for (int w = 0; w < NWV; ++w) s += red[(w * 8 + (c >> 2)) * RW + (c & 3) * CO + o];
    } else {
      for (int w = 0; w < NWV; ++w) s += red[(w * 8) * RW + 4 * CO + (e - CI * CO)];
    }
    row[e] = s;
  }
  if (colsum != nullptr && tid < CI) {
    float s = 0.f;
    for (int w = 0; w < NWV; ++w) s += red[(w * 8 + (tid >> 2)) * RW + 5 * CO + (tid & 3)];
    colsum[(size_t)blockIdx.x * CI + tid] = s;
  }
  if (dh_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(dh_amax, amx, tid, GH_NT, red, blockIdx.x);
  }
}

// h [B * n_pix, Cin] = the decoder's activation below the 1x1 head (act h_act already applied), w1 [Cin, 2C], b1 [2C],
// target [B, n_pix, C].  Writes logits [B, n_pix, 2C], dlogits (optional) = -scale * d llk / d logits, dh [B * n_pix,
// Cin], llk_part [B][n_part], one row (dW1 | db1) per workgroup of wslab and (optional) one row of Cin column sums of
// dh per workgroup of colsum_slab.  NULL h: dry run that reports n_part / rows.  -2: shapes outside this kernel.
extern "C" int odin_gaussian_head_fwd_bwd(const float* h, const float* w1, const float* b1, const float* target,
                                          float* logits, float* dlogits, float* dh, float* llk_part,
                                          int* n_part_out, float* wslab, int* rows_out, float* colsum_slab,
                                          const float* scale, int B, int n_pix, int Cin, int C, int softplus1,
                                          int h_act, uint32_t* dh_amax, void* stream) {
  if (!(Cin == 8 || Cin == 16 || Cin == 32) || (C != 1 && C != 3) ||
      (softplus1 != 0 && softplus1 != 1 && softplus1 != 3) || B < 1 ||
      n_pix < 1 || (size_t)B * n_pix * Cin * 4 >= (1ull << 40))
    return odin_fail(-2, "gaussian_head: shapes outside the fused kernel");
  const int Q = Cin / 4;
  const int ppu = GH_NT / Q * GH_U;
  const int n_part = (n_pix + ppu - 1) / ppu;
  const long n_units = (long)B * n_part;
  if (n_units > (1L << 30)) return odin_fail(-2, "gaussian_head: too many units");
  int grid = (int)(n_units < ODIN_MAX_COLSUM_BLOCKS ? n_units : ODIN_MAX_COLSUM_BLOCKS);
  int parts = n_part * (GH_NT / 64);  // one partial per unit and wave ...
  // ... unless a divisor d of the units per sample gives a grid of B * n_part / d <= 512 workgroups with d consecutive
  // units each: then a wave keeps one partial for its d units (96 x 80 pixels: 30 units, d = 15, 16 partials per
  // sample instead of 240 -- odin_elbo_finalize sums them with one thread per sample)
  int upw = 0;
  for (int d = 1; d <= n_part; ++d)
    if (n_part % d == 0 && n_units / d <= ODIN_MAX_COLSUM_BLOCKS) { upw = d; break; }
  if (upw > 0) {
    grid = (int)(n_units / upw);
    parts = (n_part / upw) * (GH_NT / 64);
  }
  if (n_part_out) *n_part_out = parts;
  if (rows_out) *rows_out = grid;
  if (h == nullptr) return 0;  // dry run
  if ((((uintptr_t)h | (uintptr_t)dh) & 15) != 0) return odin_fail(-2, "gaussian_head: unaligned activation");
#define ODIN_GH(CC, SP)                                                                                          \
  ODIN_LAUNCH((gauss_head_kernel<CC, SP>), dim3(grid), dim3(GH_NT), 0, stream, (const float4*)h, w1, b1, target, \
              logits, dlogits, (float4*)dh, llk_part, wslab, colsum_slab, scale, (unsigned*)dh_amax,             \
              (int)n_units, n_part, n_pix, Q, h_act, upw)
  if (softplus1 == 3) { if (C == 1) ODIN_GH(1, 3); else ODIN_GH(3, 3); }
  else if (C == 1 && softplus1 == 1) ODIN_GH(1, 1);
  else if (C == 1) ODIN_GH(1, 0);
  else if (softplus1 == 1) ODIN_GH(3, 1);
  else ODIN_GH(3, 0);
#undef ODIN_GH
  return odin_check_launch("gaussian_head");
}

extern "C" int odin_elbo_gaussian_fwd_bwd(const float* h, const float* x, float* llk_part,
                                          float* dh, const float* scale, int B, int n_pix, int C,
                                          int softplus1, int* n_part_out, void* stream) {
  int N = n_pix * C;
  // streaming form: C in {1, 3}, whole waves of 256*G pixels per sample
  const int G = (C == 1) ? 2 : 1;
  if ((C == 1 || C == 3) && n_pix % (256 * G) == 0 &&
      (((uintptr_t)h | (uintptr_t)x | (uintptr_t)dh) & 15) == 0) {
    const int n_part = n_pix / (256 * G);
    if (n_part_out) *n_part_out = n_part;
    if (h == nullptr) return 0;  // dry run
    const size_t n_waves = (size_t)B * n_part;
    const dim3 grid((unsigned)((n_waves + 3) / 4));
#define ODIN_GAUSS_STREAM(CC, GG, SP)                                                           \
  ODIN_LAUNCH((elbo_gaussian_stream_kernel<CC, GG, SP>), grid, dim3(256), 0, stream,            \
              (const float4*)h, (const float4*)x, llk_part, (float4*)dh, scale, n_waves)
    if (C == 3 && softplus1 == 2) ODIN_GAUSS_STREAM(3, 1, 2);
    else if (C == 3 && softplus1 == 1) ODIN_GAUSS_STREAM(3, 1, 1);
    else if (C == 3) ODIN_GAUSS_STREAM(3, 1, 0);
    else if (softplus1 == 2) ODIN_GAUSS_STREAM(1, 2, 2);
    else if (softplus1 == 1) ODIN_GAUSS_STREAM(1, 2, 1);
    else ODIN_GAUSS_STREAM(1, 2, 0);
#undef ODIN_GAUSS_STREAM
    return odin_check_launch("elbo_gaussian");
  }
  int n_part = (N + ELBO_CHUNK - 1) / ELBO_CHUNK;
  if (n_part_out) *n_part_out = n_part;
  if (h == nullptr) return 0;  // dry run
  ODIN_LAUNCH(elbo_gaussian_kernel, dim3(B * n_part), dim3(256), 0, stream, h, x, llk_part, dh,
              scale, N, C, n_part, softplus1);
  return odin_check_launch("elbo_gaussian");
}

// ---- MixtureQuantizedLogistic(params, n_components = K, n_channels = C, low = 0, high = 255, 'sigmoid')
// (odin/bay/distributions/quantized.py:206-349; image_networks.py:72-85): h [B, n_pix, K * n_out],
// n_out = 1 + C + C + C (C - 1) / 2 = (mixture logit | loc | raw scale | channel coefficients) per
// component.  One lane owns one pixel: log p = logsumexp_k(log_softmax(logit)_k + sum_c log QL_kc(x_c))
// with loc_ki += sum_{j<i} coef * (2 x_j - 1); the gradient is written through the responsibilities.
// HBM-bound: 8 B per parameter (read h, write dh) + 4 B per target value.
template <int C, int K>
__global__ __launch_bounds__(256) void elbo_mixql_kernel(const float* __restrict__ h,
                                                        const float* __restrict__ x,
                                                        float* __restrict__ llk_part,
                                                        float* __restrict__ dh,
                                                        const float* __restrict__ scale, int n_pix,
                                                        int n_part) {
  constexpr int NCO = C * (C - 1) / 2, NO = 2 * C + NCO + 1;
  __shared__ float red[4];
  const int b = blockIdx.x / n_part, part = blockIdx.x - b * n_part;
  const int pix = part * 256 + threadIdx.x;
  const float sc = scale[0];
  float acc = 0.f;
  if (pix < n_pix) {
    const size_t pb = (size_t)b * n_pix + pix;
    const float* hp = h + pb * (K * NO);
    float* dp = dh + pb * (K * NO);
    float t[C], xt[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { t[c] = x[pb * C + c]; xt[c] = 2.f * t[c] - 1.f; }
    float lg[K], comp[K], gl[K][C], gr[K][C];
    float lmax = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < K; ++k) { lg[k] = hp[k * NO]; lmax = fmaxf(lmax, lg[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) se += expf(lg[k] - lmax);
    const float lse = lmax + logf(se);
    float cmax = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float loc = hp[k * NO + 1 + c];
        const float raw = hp[k * NO + 1 + C + c];
        // coefficient index of (channel c, earlier channel j): c (c - 1) / 2 + j (the reference's loop order)
#pragma unroll
        for (int j = 0; j < c; ++j) loc = fmaf(xt[j], hp[k * NO + 1 + 2 * C + c * (c - 1) / 2 + j], loc);
        q += qlogistic_elem(loc, raw, t[c], gl[k][c], gr[k][c]);
      }
      comp[k] = lg[k] - lse + q;
      cmax = fmaxf(cmax, comp[k]);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) s += expf(comp[k] - cmax);
    acc = cmax + logf(s);
    const float inv_s = 1.f / s;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float r = expf(comp[k] - cmax) * inv_s;  // responsibility of component k
      dp[k * NO] = -(r - expf(lg[k] - lse)) * sc;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        dp[k * NO + 1 + c] = -r * gl[k][c] * sc;
        dp[k * NO + 1 + C + c] = -r * gr[k][c] * sc;
#pragma unroll
        for (int j = 0; j < c; ++j) dp[k * NO + 1 + 2 * C + c * (c - 1) / 2 + j] = -r * gl[k][c] * xt[j] * sc;
      }
    }
  }
  const float tot = block_sum_256(acc, red);
  if (threadIdx.x == 0) llk_part[blockIdx.x] = tot;
}

extern "C" int odin_elbo_mixqlogistic_fwd_bwd(const float* h, const float* x, float* llk_part,
                                              float* dh, const float* scale, int B, int n_pix, int C,
                                              int K, int* n_part_out, void* stream) {
  if (!((C == 1 || C == 3) && K == 10))
    return odin_fail(-2, "odin_elbo_mixqlogistic_fwd_bwd: 1 or 3 channels, 10 components");
  const int n_part = (n_pix + 255) / 256;
  if (n_part_out) *n_part_out = n_part;
  if (h == nullptr) return 0;  // dry run
  if (C == 1)
    ODIN_LAUNCH((elbo_mixql_kernel<1, 10>), dim3(B * n_part), dim3(256), 0, stream, h, x, llk_part, dh, scale,
                n_pix, n_part);
  else
    ODIN_LAUNCH((elbo_mixql_kernel<3, 10>), dim3(B * n_part), dim3(256), 0, stream, h, x, llk_part, dh, scale,
                n_pix, n_part);
  return odin_check_launch("elbo_mixqlogistic");
}

extern "C" int odin_elbo_finalize(const float* llk_part, int n_part, const float* kl,
                                  const float* hyper, const float* tc, float* llk, float* out4,
                                  int B, void* stream) {
  ODIN_LAUNCH(elbo_finalize_kernel, dim3(1), dim3(256), 0, stream, llk_part, n_part, kl, hyper,
              tc, llk, out4, B);
  return odin_check_launch("elbo_finalize");
}

extern "C" int odin_adam_step_flat(float* theta, const float* g, float* m, float* v, size_t n,
                                   const float* hyper, const float* gnorm2, float clip,
                                   int32_t* flag, void* stream) {
  int grid = grid_for(n / 4 + 1, 256, 2048);
  ODIN_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper, gnorm2,
              clip, (int*)flag, (const float*)nullptr, 0, (float*)nullptr, HyperRing{}, FinArgs{});
  return odin_check_launch("adam");
}



// see adam_fold_kernel.  The folded ranges start at multiples of 4 floats, the Dense one also ends at one ((K + 1) N % 4
// == 0); the slab range may end ragged (its last float4 is finished with the ordinary gradient values).
extern "C" int odin_adam_step_fold(float* theta, float* g, float* m, float* v, size_t n, const float* hyper,
                                   const odin_adam_fold* fo, void* stream) {
  if (fo == nullptr) return odin_fail(-2, "adam_step_fold: no fold description");
  AdamFold f;
  memset(&f, 0, sizeof(f));
  int KT = 8;
  if (fo->x != nullptr) {
    if (fo->dy == nullptr || fo->B < 1 || fo->K < 1 || fo->K > 32 || fo->N < 1 || (fo->w_off & 3) != 0 ||
        (((size_t)(fo->K + 1) * fo->N) & 3) != 0 || fo->w_off + (size_t)(fo->K + 1) * fo->N > n)
      return odin_fail(-2, "adam_step_fold: Dense piece outside the folded regime (K <= 32, 4-float aligned range)");
    f.x = fo->x; f.dy = fo->dy; f.B = fo->B; f.K = fo->K; f.N = fo->N; f.offA = fo->w_off;
    f.nA = (fo->N + 31) / 32;
    KT = fo->K <= 8 ? 8 : fo->K <= 16 ? 16 : 32;
  }
  if (fo->slab != nullptr) {
    if (fo->slab_rows < 1 || fo->slab_n < 1 || (fo->slab_off & 3) != 0 || fo->slab_off + fo->slab_n > n)
      return odin_fail(-2, "adam_step_fold: slab piece outside the folded regime");
    f.slab = fo->slab; f.rows = fo->slab_rows; f.stride = fo->slab_stride; f.nS = fo->slab_n; f.offS = fo->slab_off;
    f.endS = (fo->slab_off + fo->slab_n + 3) & ~(size_t)3;
    if (f.endS > n) f.endS = n;
    f.nB = (int)((f.endS - f.offS + 255) / 256);
    if (f.x != nullptr && f.offS < f.offA + (size_t)(f.K + 1) * f.N && f.offA < f.endS)
      return odin_fail(-2, "adam_step_fold: the folded ranges overlap");
  }
  f.zero = reinterpret_cast<unsigned*>(fo->zero); f.zero_n = fo->zero != nullptr ? fo->zero_n : 0;
  const int grid = grid_for(n / 4 + 1, 256, 2048) + f.nA + f.nB;
  if (KT == 8) ODIN_LAUNCH((adam_fold_kernel<8>), dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper, f);
  else if (KT == 16) ODIN_LAUNCH((adam_fold_kernel<16>), dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper, f);
  else ODIN_LAUNCH((adam_fold_kernel<32>), dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper, f);
  return odin_check_launch("adam_fold");
}

extern "C" int odin_latent_sample_logprob(const float* p, const float* eps, float* z, float* logq,
                                          float* logp, int n, int B, int D, void* stream) {
  ODIN_LAUNCH(latent_sample_logprob_kernel, dim3((n * B + 255) / 256), dim3(256), 0, stream, p, eps,
              z, logq, logp, n, B, D);
  return odin_check_launch("latent_sample_logprob");
}

extern "C" int odin_sum_parts(const float* part, int n_part, float* out, int B, void* stream) {
  ODIN_LAUNCH(sum_parts_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, part, n_part, out, B);
  return odin_check_launch("sum_parts");
}

extern "C" int odin_logmeanexp_rows(const float* in, float* out, int n, int B, void* stream) {
  ODIN_LAUNCH(logmeanexp_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, in, out, n, B);
  return odin_check_launch("logmeanexp_rows");
}

extern "C" int odin_grad_skip_threshold(float* g, size_t n, float threshold, const int32_t* enable,
                                        int32_t* hit, int32_t* skipped_count, void* stream) {
  if (int rc = odin_zero_u32((uint32_t*)hit, 1, stream)) return rc;  // (a kernel, not a memset node: runtime.hip)
  int grid = grid_for(n, 256 * 4, 2048);
  ODIN_LAUNCH(grad_any_ge_kernel, dim3(grid), dim3(256), 0, stream, (const float*)g, n, threshold,
              (int*)hit);
  ODIN_LAUNCH(grad_zero_if_kernel, dim3(grid), dim3(256), 0, stream, g, n, (const int*)hit,
              (const int*)enable, (int*)skipped_count);
  return odin_check_launch("grad_skip_threshold");
}

extern "C" int odin_clip_by_norm_segments(float* g, const int64_t* seg_offsets, int n_segments,
                                          float clipnorm, void* stream) {
  if (n_segments <= 0) return 0;
  ODIN_LAUNCH(clip_by_norm_segments_kernel, dim3(n_segments), dim3(1024), 0, stream, g,
              (const long long*)seg_offsets, clipnorm);
  return odin_check_launch("clip_by_norm_segments");
}

extern "C" int odin_clip_by_value(float* g, size_t n, float clipvalue, const float* gnorm2,
                                  float global_clipnorm, void* stream) {
  int grid = grid_for(n, 256 * 4, 2048);
  ODIN_LAUNCH(clip_by_value_kernel, dim3(grid), dim3(256), 0, stream, g, n, clipvalue, gnorm2,
              global_clipnorm);
  return odin_check_launch("clip_by_value");
}

namespace {
// ring == null: no ring.  Otherwise `cur` = the device row the step's kernels read (the row that holds `hyper`),
// `staged` >= 8 floats of scratch, `rows` a power of two.
bool hyper_ring_args(const float* ring, float* cur, float* staged, int rows, int row_floats, int t_word, HyperRing& hr) {
  memset(&hr, 0, sizeof(hr));
  if (ring == nullptr) return true;
  if (cur == nullptr || staged == nullptr || rows < 2 || (rows & (rows - 1)) != 0 || row_floats < 1 ||
      row_floats > 64 || t_word < 0 || t_word >= row_floats)
    return false;
  hr.ring = ring; hr.cur = cur; hr.rows = rows; hr.row_floats = row_floats; hr.t_word = t_word;
  return true;
}
}  // namespace

extern "C" int odin_sumsq_adam_flat(float* theta, const float* g, float* m, float* v, size_t n,
                                    const float* hyper, float* workspace, float* gnorm2_out,
                                    float clip, int32_t* flag, void* stream) {
  int g1 = grid_for(n / 4 + 1, 256, 1024);
  ODIN_LAUNCH(sumsq_stage1, dim3(g1), dim3(256), 0, stream, g, n, workspace, (const float*)nullptr, (float*)nullptr);
  int grid = grid_for(n / 4 + 1, 256, 2048);
  ODIN_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper,
              (const float*)nullptr, clip, (int*)flag, (const float*)workspace, g1, gnorm2_out, HyperRing{}, FinArgs{});
  return odin_check_launch("sumsq_adam");
}

// odin_sumsq_adam_flat / odin_sumsq_adam_finalize_flat (llk_part != NULL) as the LAST launches of a step whose
// per-step scalars live in a device ring: the stage-1 launch stages Adam's five scalars (hyper[0..4]), the Adam
// launch loads ring row (t + 1) mod rows into `cur`, t = the int32 at cur[t_word].
extern "C" int odin_sumsq_adam_ring(float* theta, const float* g, float* m, float* v, size_t n, const float* hyper,
                                    float* workspace, float* gnorm2_out, float clip, int32_t* flag,
                                    const float* llk_part, int n_part, const float* kl, const float* elbo_hyper,
                                    const float* tc, float* llk, float* out4, int B, const float* ring, float* cur,
                                    float* staged, int rows, int row_floats, int t_word, void* stream) {
  HyperRing hr;
  if (ring == nullptr || !hyper_ring_args(ring, cur, staged, rows, row_floats, t_word, hr))
    return odin_fail(-2, "odin_sumsq_adam_ring: bad ring arguments (rows must be a power of two)");
  int g1 = grid_for(n / 4 + 1, 256, 1024);
  if (llk_part != nullptr) {
    FinArgs f;
    f.llk_part = llk_part; f.kl = kl; f.hyper = elbo_hyper; f.tcp = tc; f.llk = llk; f.out4 = out4;
    f.n_part = n_part; f.B = B;
    ODIN_LAUNCH(sumsq_stage1_fin, dim3(g1 + 1), dim3(256), 0, stream, g, n, workspace, f, hyper, staged);
  } else {
    ODIN_LAUNCH(sumsq_stage1, dim3(g1), dim3(256), 0, stream, g, n, workspace, hyper, staged);
  }
  int grid = grid_for(n / 4 + 1, 256, 2048);
  ODIN_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, (const float*)staged,
              (const float*)nullptr, clip, (int*)flag, (const float*)workspace, g1, gnorm2_out, hr, FinArgs{});
  return odin_check_launch("sumsq_adam_ring");
}

// The update of a step whose gradient-norm partials were left by odin_slab_reduce_sumsq (`parts`, n_parts) together
// with the staged hyper-parameter row (`staged`: the whole row, alpha_off / elbo_off = the offsets of Adam's five
// scalars and of the ELBO weights in it): ONE launch -- norm, clip scale, NaN guard, Adam, the ring advance of
// odin_sumsq_adam_ring, and (llk_part != NULL) the step's ELBO finalisation in an extra workgroup.
extern "C" int odin_adam_ring_parts(float* theta, const float* g, float* m, float* v, size_t n, const float* staged,
                                    int alpha_off, int elbo_off, const float* parts, int n_parts, float* gnorm2_out,
                                    float clip, int32_t* flag, const float* llk_part, int n_part, const float* kl,
                                    const float* tc, float* llk, float* out4, int B, const float* ring, float* cur,
                                    int rows, int row_floats, int t_word, void* stream) {
  // (ring == NULL: no ring to advance -- a caller that writes `hyper` from the host every step, FactorVAE's iteration)
  HyperRing hr;
  if (staged == nullptr || parts == nullptr || n_parts < 1 ||
      !hyper_ring_args(ring, cur, const_cast<float*>(staged), rows, row_floats, t_word, hr))
    return odin_fail(-2, "odin_adam_ring_parts: bad arguments");
  FinArgs f;
  memset(&f, 0, sizeof(f));
  if (llk_part != nullptr) {
    f.llk_part = llk_part; f.kl = kl; f.hyper = staged + elbo_off; f.tcp = tc; f.llk = llk; f.out4 = out4;
    f.n_part = n_part; f.B = B;
  }
  int grid = grid_for(n / 4 + 1, 256, 2048) + (llk_part != nullptr ? 1 : 0);
  ODIN_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, staged + alpha_off,
              (const float*)nullptr, clip, (int*)flag, parts, n_parts, gnorm2_out, hr, f);
  return odin_check_launch("adam_ring_parts");
}

extern "C" int odin_sumsq_adam_finalize_flat(float* theta, const float* g, float* m, float* v, size_t n,
                                             const float* hyper, float* workspace, float* gnorm2_out,
                                             float clip, int32_t* flag, const float* llk_part, int n_part,
                                             const float* kl, const float* elbo_hyper, const float* tc,
                                             float* llk, float* out4, int B, void* stream) {
  // (same stage-1 partition as odin_sumsq_adam_flat: the gradient norm is bit-identical)
  int g1 = grid_for(n / 4 + 1, 256, 1024);
  FinArgs f;
  f.llk_part = llk_part; f.kl = kl; f.hyper = elbo_hyper; f.tcp = tc; f.llk = llk; f.out4 = out4;
  f.n_part = n_part; f.B = B;
  ODIN_LAUNCH(sumsq_stage1_fin, dim3(g1 + 1), dim3(256), 0, stream, g, n, workspace, f, (const float*)nullptr,
              (float*)nullptr);
  int grid = grid_for(n / 4 + 1, 256, 2048);
  ODIN_LAUNCH(adam_kernel, dim3(grid), dim3(256), 0, stream, theta, g, m, v, n, hyper,
              (const float*)nullptr, clip, (int*)flag, (const float*)workspace, g1, gnorm2_out, HyperRing{}, FinArgs{});
  return odin_check_launch("sumsq_adam_finalize");
}

extern "C" int odin_sumsq_flat(const float* g, size_t n, float* workspace, float* out,
                               void* stream) {
  int grid = grid_for(n / 4 + 1, 256, 1024);
  ODIN_LAUNCH(sumsq_stage1, dim3(grid), dim3(256), 0, stream, g, n, workspace, (const float*)nullptr, (float*)nullptr);
  ODIN_LAUNCH(sum_stage2, dim3(1), dim3(256), 0, stream, (const float*)workspace, grid, out);
  return odin_check_launch("sumsq");
}

// ------------------------------------------------------------------ tiny Dense layers ---
// The bottleneck projections (latent 128 -> 2D, first decoder layer D -> 128 and their
// data-gradients: K*N <= 4096 weights, batch 256) are ~0.1 MFLOP each.  On the MFMA path they
// are two workgroups staging tiles for 15 us; here the whole weight matrix sits in LDS, a thread
// owns one output element, and the launch costs its floor.
//   forward:        y[b][n]  = act(sum_k x[b][k] w[k][n] + bias[n])
//   data-gradient:  dx[b][k] = (sum_n dy[b][n] w[k][n]) * act'(aux[b][k]), plus per-block column
//                   sums of dx (the bias gradient of the producing layer) into slab[blockIdx.x][K]
template <bool DGRAD>
__global__ __launch_bounds__(256) void tiny_dense_kernel(const float* __restrict__ in,
                                                         const float* __restrict__ w,
                                                         const float* __restrict__ bias_or_aux,
                                                         float* __restrict__ out,
                                                         float* __restrict__ colsum, int B, int K,
                                                         int N, int act, int nop, int sb) {
  ODIN_DYN_SMEM(float, sm);  // w [K][N + 1] | in [sb][NR] | red [sb][nop]
  const int NR = DGRAD ? N : K, NO = DGRAD ? K : N;
  const int WS = N + 1;
  float* wl = sm;
  float* xl = sm + K * WS;
  float* red = xl + sb * NR;
  for (int e = threadIdx.x; e < K * N; e += 256) wl[(e / N) * WS + (e % N)] = w[e];
  const int b0 = blockIdx.x * sb;
  for (int e = threadIdx.x; e < sb * NR; e += 256) {
    const int s = e / NR, r = e - s * NR;
    xl[e] = (b0 + s < B) ? in[(size_t)(b0 + s) * NR + r] : 0.f;
  }
  __syncthreads();
  const int s = threadIdx.x / nop, o = threadIdx.x - s * nop;
  float v = 0.f;
  const bool live = s < sb && o < NO && b0 + s < B;
  if (live) {
    const float* xr = xl + s * NR;
    float acc = 0.f;
    if (DGRAD) {
      const float* wr = wl + o * WS;
      for (int r = 0; r < NR; ++r) acc = fmaf(xr[r], wr[r], acc);
      if (bias_or_aux != nullptr) acc *= odin_act_grad(act, bias_or_aux[(size_t)(b0 + s) * NO + o]);
    } else {
      for (int r = 0; r < NR; ++r) acc = fmaf(xr[r], wl[r * WS + o], acc);
      acc = odin_act(act, acc + (bias_or_aux != nullptr ? bias_or_aux[o] : 0.f));
    }
    out[(size_t)(b0 + s) * NO + o] = acc;
    v = acc;
  }
  if (DGRAD && colsum != nullptr) {
    if (threadIdx.x < sb * nop) red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < NO) {
      float t = 0.f;
      for (int q = 0; q < sb; ++q) t += red[q * nop + threadIdx.x];
      colsum[(size_t)blockIdx.x * NO + threadIdx.x] = t;
    }
  }
}

// geometry shared by the launcher and the dry run: outputs padded to a power of two <= 256;
// samples per block capped so that the staged input rows (sb x NR floats) stay within 32 KB:
// with the <= 17 KB weight image the launch fits the default 64 KB dynamic-LDS limit for every
// shape odin_tiny_dense_ok admits (narrow heads such as Dense(128 -> 1) or Dense(256 -> 4))
static void tiny_dense_geom(int NO, int NR, int B, int* nop, int* sb, int* blocks) {
  int p2 = 1;
  while (p2 < NO) p2 <<= 1;
  *nop = p2;
  int s = 256 / p2;
  const int cap = 8192 / (NR > 0 ? NR : 1);
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  *sb = s;
  *blocks = (B + *sb - 1) / *sb;
}

bool odin_tiny_dense_ok(int B, int K, int N) {
  return (long)K * N <= 4096 && K <= 256 && N <= 256 && B >= 1 && B <= 1024 && !ODIN_DIAG_ENV("ODIN_NOTINYDENSE");
}

int odin_tiny_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K,
                        int N, int act, void* stream) {
  int nop, sb, blocks;
  tiny_dense_geom(N, K, B, &nop, &sb, &blocks);
  const size_t lds = (size_t)(K * (N + 1) + sb * K + 256) * 4;
  ODIN_LAUNCH((tiny_dense_kernel<false>), dim3(blocks), dim3(256), lds, stream, x, w, bias, y,
              (float*)nullptr, B, K, N, act, nop, sb);
  return odin_check_launch("tiny_dense_fwd");
}

int odin_tiny_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          float* colsum_slab, int* slab_rows_out, int B, int K, int N, void* stream) {
  int nop, sb, blocks;
  tiny_dense_geom(K, N, B, &nop, &sb, &blocks);
  if (slab_rows_out) *slab_rows_out = blocks;
  if (dx == nullptr) return 0;  // dry run
  const size_t lds = (size_t)(K * (N + 1) + sb * N + 256) * 4;
  ODIN_LAUNCH((tiny_dense_kernel<true>), dim3(blocks), dim3(256), lds, stream, dy, w,
              (aux != nullptr && aux_act != 0) ? aux : (const float*)nullptr, dx, colsum_slab, B, K, N,
              aux_act, nop, sb);
  return odin_check_launch("tiny_dense_dgrad");
}

// ------------------------------------------------------------------ input pipeline ----
// batch gather + ImageDataset.normalize from an HBM-resident uint8 dataset
// (odin/fuel/image_data/_base.py:130-147; dSprites pre-multiplies its 0/1 pixels by 255,
// fuel/image_data/shapes.py:69-72,80).  16 pixels (one 16-byte load) per thread.
__global__ __launch_bounds__(256) void gather_normalize_u8_kernel(const unsigned char* __restrict__ data,
                                                                  const int* __restrict__ idx,
                                                                  float* __restrict__ out, int B,
                                                                  int n_per, float premul, int mode) {
  const int per16 = n_per >> 4;
  const long total = (long)B * per16;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int b = (int)(t / per16), q = (int)(t - (long)b * per16);
    const uint4 raw = *reinterpret_cast<const uint4*>(data + (size_t)idx[b] * n_per + (size_t)q * 16);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
    float* o = out + (size_t)b * n_per + (size_t)q * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = (float)((w[k] >> (8 * j)) & 0xFFu) * premul;
        if (mode == 3) {
          v[j] = x;  // binarised data passes through
        } else {
          x = fminf(fmaxf(x, 0.f), 255.f);
          if (mode == 0) x = fminf(fmaxf(x / 255.f, 1e-6f), 1.f - 1e-6f);                  // 'probs'
          else if (mode == 1) x = fminf(fmaxf(x / 255.f * 2.f - 1.f, -1.f + 1e-6f), 1.f - 1e-6f);  // 'tanh'
          v[j] = x;                                                                           // 'raster'
        }
      }
      *reinterpret_cast<float4*>(o + 4 * k) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

extern "C" int odin_gather_normalize_u8(const uint8_t* data, const int32_t* idx, float* out, int B,
                                        int n_per, float premul, int mode, void* stream) {
  if (n_per % 16 != 0) return odin_fail(-2, "gather_normalize_u8: pixels per image must be a multiple of 16");
  if (mode < 0 || mode > 3) return odin_fail(-2, "gather_normalize_u8: mode must be 0..3");
  int grid = grid_for((size_t)B * (n_per / 16), 256, 4096);
  ODIN_LAUNCH(gather_normalize_u8_kernel, dim3(grid), dim3(256), 0, stream, data, (const int*)idx, out,
              B, n_per, premul, mode);
  return odin_check_launch("gather_normalize_u8");
}

// ---- diagnostics: hand-written streams with the traffic of the fused Bernoulli ELBO kernel (two arrays read, one
// written, 12 bytes per element) and NO arithmetic to speak of -- the ceiling a launch of that length can reach on
// this part (bench.py: elbo_kernel.frac_of_own_stream).  variant 0: the ELBO kernel's own shape (one wave = 64 x U
// float4 per array, all loads first); 1: persistent grid, grid-stride, 4 float4 pairs in flight per lane; 2: the same
// with non-temporal loads and stores; 3: 8 pairs in flight ----
// NT bits: 1 = non-temporal loads of a, 2 = of b, 4 = non-temporal stores
template <int U, int NT>
__global__ __launch_bounds__(256) void stream_probe_kernel(const f32x4* __restrict__ a, const f32x4* __restrict__ b,
                                                           f32x4* __restrict__ out, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
    f32x4 x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        x[u] = (NT & 1) ? __builtin_nontemporal_load(a + i) : a[i];
        y[u] = (NT & 2) ? __builtin_nontemporal_load(b + i) : b[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + u * stride;
      if (i < n4) {
        const f32x4 r = x[u] + y[u];
        if (NT & 4) __builtin_nontemporal_store(r, out + i); else out[i] = r;
      }
    }
  }
}
template <int U>
__global__ __launch_bounds__(256) void stream_probe_wave_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                                float4* __restrict__ out, size_t n_waves) {
  const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_waves) return;
  const size_t base = w * (size_t)(64 * U) + (threadIdx.x & 63);
  float4 x[U], y[U];
#pragma unroll
  for (int u = 0; u < U; ++u) x[u] = a[base + 64 * u];
#pragma unroll
  for (int u = 0; u < U; ++u) y[u] = b[base + 64 * u];
#pragma unroll
  for (int u = 0; u < U; ++u)
    out[base + 64 * u] = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w);
}

extern "C" int odin_debug_stream_probe(const float* a, const float* b, float* out, size_t n, int variant, int blocks,
                                       void* stream) {
  if ((n & 3) != 0 || n == 0) return odin_fail(-2, "stream_probe: n must be a positive multiple of 4");
  const size_t n4 = n >> 2;
  const float4 *a4 = (const float4*)a, *b4 = (const float4*)b;
  float4* o4 = (float4*)out;
  const f32x4 *av = (const f32x4*)a, *bv = (const f32x4*)b;
  f32x4* ov = (f32x4*)out;
  if (blocks <= 0) blocks = odin_num_cus() * 8;
  if (variant == 0) {
    if (n4 % (64 * 3) != 0) return odin_fail(-2, "stream_probe 0: n must be a multiple of 768");
    const size_t nw = n4 / (64 * 3);
    ODIN_LAUNCH((stream_probe_wave_kernel<3>), dim3((unsigned)((nw + 3) / 4)), dim3(256), 0, stream, a4, b4, o4, nw);
  } else if (variant == 1) {
    ODIN_LAUNCH((stream_probe_kernel<4, 0>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 2) {
    ODIN_LAUNCH((stream_probe_kernel<4, 7>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 3) {
    ODIN_LAUNCH((stream_probe_kernel<8, 0>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 4) {
    ODIN_LAUNCH((stream_probe_kernel<8, 7>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 5) {
    ODIN_LAUNCH((stream_probe_kernel<2, 0>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 6) {
    ODIN_LAUNCH((stream_probe_kernel<4, 2>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 7) {
    ODIN_LAUNCH((stream_probe_kernel<4, 3>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 8) {
    ODIN_LAUNCH((stream_probe_kernel<4, 4>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else if (variant == 9) {
    ODIN_LAUNCH((stream_probe_kernel<4, 6>), dim3(blocks), dim3(256), 0, stream, av, bv, ov, n4);
  } else {
    return odin_fail(-2, "stream_probe: variant 0..9");
  }
  return odin_check_launch("stream_probe");
}

// batch gather from an HBM-resident float32 dataset that is already normalised (fit() on a tensor:
// odin/networks/base_networks.py:642-812 feeds `train` batch by batch): out[b] = data[idx[b]], 16 bytes per thread
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ data,
                                                              const int* __restrict__ idx, float* __restrict__ out,
                                                              int B, int n_per) {
  const int per4 = n_per >> 2;
  const long total = (long)B * per4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int b = (int)(t / per4), q = (int)(t - (long)b * per4);
    reinterpret_cast<float4*>(out + (size_t)b * n_per)[q] =
        reinterpret_cast<const float4*>(data + (size_t)idx[b] * n_per)[q];
  }
}

extern "C" int odin_gather_rows_f32(const float* data, const int32_t* idx, float* out, int B, int n_per,
                                    void* stream) {
  if (n_per % 4 != 0) return odin_fail(-2, "gather_rows_f32: floats per row must be a multiple of 4");
  int grid = grid_for((size_t)B * (n_per / 4), 256, 4096);
  ODIN_LAUNCH(gather_rows_f32_kernel, dim3(grid), dim3(256), 0, stream, data, (const int*)idx, out, B, n_per);
  return odin_check_launch("gather_rows_f32");
}

extern "C" int odin_rng_normal(float* out, size_t n, uint64_t seed, const int32_t* step_dev,
                               void* stream) {
  int grid = grid_for((n + 3) / 4, 256, 2048);
  ODIN_LAUNCH(rng_normal_kernel, dim3(grid), dim3(256), 0, stream, out, n, (unsigned)seed,
              (unsigned)(seed >> 32), (const int*)step_dev);
  return odin_check_launch("rng_normal");
}
