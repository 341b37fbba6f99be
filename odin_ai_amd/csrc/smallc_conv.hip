// smallc_conv.hip -- first-layer convolutions (Cin <= 4: grey / RGB images) forward and
// weight-gradient.  These layers are HBM-bound (dSprites conv1: 0.27 GFLOP against 37 MB
// of traffic at batch 256) with a reduction depth of only 16-48, so nothing is staged in LDS:
// the matrix-core kernels below gather their operands straight from HBM/L2 (one tap per lane
// per MFMA) and store 16-byte NHWC pieces; vector-ALU kernels cover the shapes the row-block
// mapping does not (output widths that are not a multiple of 32 / odd).
// (Reference: first Conv2D of every get_networks encoder,
// odin/networks/image_networks.py:244,463,679; `CenterAt0` :121-126 folded into the load.)
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct SCParams {
  const float* x;
  const float* w;     // [KH][KW][CI][CO]
  const float* bias;
  const float* dy;    // wgrad: [B,OH,OW,CO]
  float* y;           // fwd out / wgrad slab
  int B, H, W, CI, OH, OW, CO, KH, KW, S, pt, pl, act, center;
  int pix_per_block, slab_stride;
  unsigned* y_amax;   // fwd: optional range word of y (odin_conv_desc.y_amax)
  const unsigned* dy_amax;   // wgrad on planes: range word of dy
};

// t / d for 0 <= t < 64, 1 <= d <= 8 (tap decoding: a runtime integer division is ~30 instructions, and the
// matrix-core kernels decode 24 taps per lane before their first load)
__device__ __forceinline__ int sc_smalldiv(int t, int d) { return (t * (256 / d + 1)) >> 8; }

// forward: thread = (output pixel, 4 consecutive output channels)
__global__ __launch_bounds__(256) void smallc_fwd_kernel(SCParams p) {
  ODIN_DYN_SMEM(float, wl);  // [KH*KW*CI][CO] + bias[CO]
  const int K = p.KH * p.KW * p.CI;
  for (int e = threadIdx.x; e < K * p.CO; e += 256) wl[e] = p.w[e];
  for (int e = threadIdx.x; e < p.CO; e += 256) wl[K * p.CO + e] = p.bias ? p.bias[e] : 0.f;
  __syncthreads();
  const int cg = p.CO >> 2;  // channel groups per pixel
  const int total = p.B * p.OH * p.OW * cg;  // < 2^31 (checked on the host)
  const int ohw = p.OH * p.OW;
  float amx = 0.f;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int pix = t / cg, g4 = t - pix * cg;
    const int b = pix / ohw, rem = pix - b * ohw;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    const float4 bb = *reinterpret_cast<const float4*>(wl + K * p.CO + 4 * g4);
    float4 acc = bb;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = oh * p.S - p.pt + kh;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = ow * p.S - p.pl + kw;
        if (iw < 0 || iw >= p.W) continue;
        const float* xp = p.x + (((size_t)b * p.H + ih) * p.W + iw) * p.CI;
        const float* wp = wl + ((kh * p.KW + kw) * p.CI) * p.CO + 4 * g4;
        for (int c = 0; c < p.CI; ++c) {
          float xv = xp[c];
          if (p.center) xv = 2.f * xv - 1.f;
          const float4 wv = *reinterpret_cast<const float4*>(wp + c * p.CO);
          acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
          acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
        }
      }
    }
    acc.x = odin_act(p.act, acc.x); acc.y = odin_act(p.act, acc.y);
    acc.z = odin_act(p.act, acc.z); acc.w = odin_act(p.act, acc.w);
    amx = odin_amax3(odin_amax3(amx, acc.x, acc.y), acc.z, acc.w);
    *reinterpret_cast<float4*>(p.y + (size_t)pix * p.CO + 4 * g4) = acc;
  }
  __shared__ float ared[16];
  odin_amax_commit_wg(p.y_amax, amx, threadIdx.x, 256, ared, blockIdx.x);   // (the range word of y, if asked for)
}

// forward, compile-time taps: all KH*KW*CI input taps of a thread are requested up front as
// range-checked loads over the whole input tensor (padding taps carry an out-of-range offset
// and read zeros) -- no branch between the loads, so they overlap instead of paying one memory
// latency each.
template <int KH, int KW, int CI>
__global__ __launch_bounds__(256) void smallc_fwd_kernel_t(SCParams p) {
  ODIN_DYN_SMEM(float, wl);  // [KH*KW*CI][CO] + bias[CO]
  constexpr int K = KH * KW * CI;
  for (int e = threadIdx.x; e < K * p.CO; e += 256) wl[e] = p.w[e];
  for (int e = threadIdx.x; e < p.CO; e += 256) wl[K * p.CO + e] = p.bias ? p.bias[e] : 0.f;
  __syncthreads();
  const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * p.W * CI * 4));
  const int cg = p.CO >> 2;
  const int total = p.B * p.OH * p.OW * cg;
  const int ohw = p.OH * p.OW;
  float amx = 0.f;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    const int pix = t / cg, g4 = t - pix * cg;
    const int b = pix / ohw, rem = pix - b * ohw;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    const int ih0 = oh * p.S - p.pt, iw0 = ow * p.S - p.pl;
    float xv[K];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
        const unsigned off = ok ? (unsigned)((((b * p.H + ih) * p.W + iw) * CI) * 4) : ODIN_OOB;
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          float v = odin_run_load1(XR, ok ? off + 4 * c : ODIN_OOB);
          if (p.center) v = ok ? 2.f * v - 1.f : 0.f;
          xv[(kh * KW + kw) * CI + c] = v;
        }
      }
    }
    float4 acc = *reinterpret_cast<const float4*>(wl + K * p.CO + 4 * g4);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float4 wv = *reinterpret_cast<const float4*>(wl + k * p.CO + 4 * g4);
      acc.x = fmaf(xv[k], wv.x, acc.x); acc.y = fmaf(xv[k], wv.y, acc.y);
      acc.z = fmaf(xv[k], wv.z, acc.z); acc.w = fmaf(xv[k], wv.w, acc.w);
    }
    acc.x = odin_act(p.act, acc.x); acc.y = odin_act(p.act, acc.y);
    acc.z = odin_act(p.act, acc.z); acc.w = odin_act(p.act, acc.w);
    amx = odin_amax3(odin_amax3(amx, acc.x, acc.y), acc.z, acc.w);
    *reinterpret_cast<float4*>(p.y + (size_t)pix * p.CO + 4 * g4) = acc;
  }
  __shared__ float ared[16];
  odin_amax_commit_wg(p.y_amax, amx, threadIdx.x, 256, ared, blockIdx.x);   // (the range word of y, if asked for)
}

// weight gradient.  Workgroup = 16 waves; a wave owns 64/CW pixels per iteration (CW = 32
// or 64 lanes per pixel = output channels), every lane keeps all TK = KH*KW*CI rows of dW for
// its channel in registers; the x taps of a pixel are wave-broadcast loads.  Partial sums are
// combined across the lane halves by a shuffle and across waves through LDS; one slab row
// [TK*CO | CO] per workgroup.
template <int TK>
__global__ __launch_bounds__(1024) void smallc_wgrad_kernel(SCParams p) {
  ODIN_DYN_SMEM(float, red);  // [16 waves][(TK + 1) * CO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cw = p.CO <= 32 ? 32 : 64;
  const int ppw = 64 / cw;                   // pixels per wave per iteration
  const int co = lane % cw, sub = lane / cw;
  const bool live = co < p.CO;
  float acc[TK];
#pragma unroll
  for (int k = 0; k < TK; ++k) acc[k] = 0.f;
  // cooperative tap loads: lane j of a pixel's lane group fetches tap j (+cw, ...) of that
  // pixel with ONE vector load; the taps are then broadcast inside the group by shuffles
  // (16 broadcast loads per pixel would saturate the vector-memory issue rate).
  constexpr int NL = (TK + 31) / 32;  // loads per lane (cw >= 32)
  int lkh[NL], lkw[NL], lc[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int k = co + j * cw;
    const int tap = k / p.CI;
    lc[j] = k - tap * p.CI;
    lkh[j] = (k < TK) ? tap / p.KW : -100000;
    lkw[j] = tap % p.KW;
  }
  float bacc = 0.f;
  const int total = p.B * p.OH * p.OW;
  const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun DR = odin_run(p.dy, (unsigned)((size_t)total * p.CO * 4));
  const int p0 = blockIdx.x * p.pix_per_block;
  int p1 = p0 + p.pix_per_block;
  if (p1 > total) p1 = total;
  const int ohw = p.OH * p.OW;
  const int lbase = lane - co;  // first lane of this pixel's group
  const int n_it = (p1 - p0 + 16 * ppw - 1) / (16 * ppw);
  for (int it = 0; it < n_it; ++it) {  // uniform trip count: shuffles need every lane
    const int pix = p0 + (it * 16 + wave) * ppw + sub;
    const bool pv = pix < p1;
    const int pc = pv ? pix : p0;
    const int b = pc / ohw, rem = pc - b * ohw;
    const int oh = rem / p.OW, ow = rem - oh * p.OW;
    // branch-free range-checked loads (masked lanes / padding taps read zeros)
    const float g = odin_run_load1(DR, (live && pv) ? (unsigned)((pc * p.CO + co) * 4) : ODIN_OOB);
    bacc += g;
    const int ih0 = oh * p.S - p.pt, iw0 = ow * p.S - p.pl;
    float xt[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int ih = ih0 + lkh[j], iw = iw0 + lkw[j];
      const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      float xv = odin_run_load1(XR, ok ? (unsigned)((((b * p.H + ih) * p.W + iw) * p.CI + lc[j]) * 4) : ODIN_OOB);
      if (p.center) xv = ok ? 2.f * xv - 1.f : 0.f;
      xt[j] = xv;
    }
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      // tap k lives in load k / cw of lane lbase + k % cw; cw is 32 or 64
      const float xv = (cw == 32) ? __shfl(xt[k >> 5 < NL ? k >> 5 : 0], lbase + (k & 31))
                                  : __shfl(xt[0], lbase + (k & 63));
      acc[k] = fmaf(xv, g, acc[k]);
    }
  }
  // combine the pixel halves of a wave (CW = 32), then the 16 waves
  if (cw == 32) {
#pragma unroll
    for (int k = 0; k < TK; ++k) acc[k] += __shfl_xor(acc[k], 32);
    bacc += __shfl_xor(bacc, 32);
  }
  const int n_out = (TK + 1) * p.CO;
  if (live && sub == 0) {
#pragma unroll
    for (int k = 0; k < TK; ++k) red[wave * n_out + k * p.CO + co] = acc[k];
    red[wave * n_out + TK * p.CO + co] = bacc;
  }
  __syncthreads();
  float* row = p.y + (size_t)blockIdx.x * p.slab_stride;
  for (int e = tid; e < n_out; e += 1024) {
    float t = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) t += red[w2 * n_out + e];
    row[e] = t;
  }
}

// Forward on the matrix cores: Y[co][pixel] = W^T[co][k] * X[k][pixel] with k = (kh, kw, c)
// padded to an even count.  A = weights (NK2 registers per 32 output channels, loaded once),
// B = one gathered x tap per lane per k-pair (tap 2u + h of pixel l31), straight from HBM/L2.
// A wave owns 32 consecutive pixels of one output row per iteration (OW % 32 == 0).
template <int NK2, int RB>
__global__ __launch_bounds__(256) void smallc_fwd_mfma_kernel(SCParams p) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int l31 = lane & 31, h = lane >> 5;
  const int K = p.KH * p.KW * p.CI;
  const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun WR = odin_run(p.w, (unsigned)((size_t)K * p.CO * 4));
  float a[NK2][RB];
  int tkh[NK2], tkw[NK2], tc[NK2];
  bool tok[NK2];
#pragma unroll
  for (int u = 0; u < NK2; ++u) {
    const int k = 2 * u + h;
    const int tap = sc_smalldiv(k, p.CI);
    tc[u] = k - tap * p.CI;
    tkh[u] = sc_smalldiv(tap, p.KW);
    tkw[u] = tap - tkh[u] * p.KW;
    tok[u] = k < K;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int co = rb * 32 + l31;
      a[u][rb] = odin_run_load1(WR, (k < K && co < p.CO) ? (unsigned)((k * p.CO + co) * 4) : ODIN_OOB);
    }
  }
  float bias_r[RB][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
      bias_r[rb][i] = (p.bias != nullptr && n < p.CO) ? p.bias[n] : 0.f;
    }
  const int cpr = p.OW >> 5;  // 32-pixel column blocks per output row
  const int n_it = p.B * p.OH * cpr;
  const int gw = (blockIdx.x * 256 + tid) >> 6, nw = (gridDim.x * 256) >> 6;
  // Persistent waves, the NEXT block's taps requested before the current block's MFMAs: one wave per block
  // (round 2) paid the weight fetch, the tap decoding and a full L2 round trip per 24 MFMAs -- 37 us for the
  // 64x64x3 first layer whose matrix time is 6 us.
  float amx = 0.f;   // running max |y| of this lane: the range word of the activation
  auto gather = [&](int it, float (&b)[NK2]) {
    const int itc = it < n_it ? it : n_it - 1;          // (beyond the end: a valid block, never used)
    const int r = itc / cpr, q0 = (itc - r * cpr) << 5;
    const int bb = r / p.OH, oh = r - bb * p.OH;
    const int ow = q0 + l31;
    const int ih0 = oh * p.S - p.pt, iw0 = ow * p.S - p.pl;
#pragma unroll
    for (int u = 0; u < NK2; ++u) {
      const int ih = ih0 + tkh[u], iw = iw0 + tkw[u];
      const bool ok = tok[u] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
      float v = odin_run_load1(XR, ok ? (unsigned)((((bb * p.H + ih) * p.W + iw) * p.CI + tc[u]) * 4) : ODIN_OOB);
      if (p.center) v = ok ? 2.f * v - 1.f : 0.f;
      b[u] = v;
    }
  };
  auto block = [&](int it, const float (&b)[NK2]) {
    const int r = it / cpr, q0 = (it - r * cpr) << 5;
    const int ow = q0 + l31;
    float* outp = p.y + ((size_t)(r * p.OW + ow)) * p.CO + 4 * h;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      f32x16 acc = f32x16_zero();
#pragma unroll
      for (int u = 0; u < NK2; ++u) acc = mfma32(a[u][rb], b[u], acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = rb * 32 + 8 * q + 4 * h;
        if (n + 3 < p.CO) {
          const float4 o = make_float4(odin_act(p.act, acc[4 * q] + bias_r[rb][4 * q]),
                                       odin_act(p.act, acc[4 * q + 1] + bias_r[rb][4 * q + 1]),
                                       odin_act(p.act, acc[4 * q + 2] + bias_r[rb][4 * q + 2]),
                                       odin_act(p.act, acc[4 * q + 3] + bias_r[rb][4 * q + 3]));
          amx = odin_amax3(odin_amax3(amx, o.x, o.y), o.z, o.w);
          *reinterpret_cast<float4*>(outp + rb * 32 + 8 * q) = o;
        }
      }
    }
  };
  float b0[NK2], b1[NK2];
  int it = gw;  // wave-uniform
  if (it < n_it) {
    gather(it, b0);
    for (;;) {
      gather(it + nw, b1);
      ODIN_SCHED_FENCE();
      block(it, b0);
      ODIN_SCHED_FENCE();
      it += 2 * nw;
      if (it - nw >= n_it) break;
      gather(it, b0);
      ODIN_SCHED_FENCE();
      block(it - nw, b1);
      ODIN_SCHED_FENCE();
      if (it >= n_it) break;
    }
  }
  __shared__ float ared[16];
  odin_amax_commit_wg(p.y_amax, amx, tid, 256, ared, blockIdx.x);   // (the range word of y, if asked for)
}

// The same forward with the input rows staged in LDS: for RGB images (K = 48) the direct gather is bound by the
// texture-address path -- 24 four-byte gathers per lane per 32 pixels, each touching ~12 cache lines: 38 us for
// the 64x64x3 first layer at batch 256 whatever the occupancy or prefetch depth -- while every input element is
// needed by 4 taps x 2 rows.  A workgroup owns NR consecutive output rows of one image: its S (NR - 1) + KH input
// rows arrive with coalesced 16-byte loads (CenterAt0 applied here, SAME-padding rows zeroed), the taps are
// gathered from LDS (4-byte ds_reads, <= 2-way bank conflicts), wave w multiplies rows w, w + 4, ...
// the activation as a compile-time constant where it is ELU (every conv layer of the reference's stacks): the run-time
// switch of odin_act compiles to scalar branches per ELEMENT (132 in this kernel's epilogue; round 6, cf. blk_planes.hip)
template <bool ELU>
__device__ __forceinline__ float sc_act(int rt, float v) {
  if (ELU) {
    const float em1 = odin_exp2(v * 1.44269504088896341f) - 1.f;
    return v > 0.f ? v : em1;
  }
  return odin_act(rt, v);
}

template <int NK2, int RB, bool ELU>
__global__ __launch_bounds__(256) void smallc_fwd_lds_kernel(SCParams p, int NR) {
  ODIN_DYN_SMEM(float, xs);  // [S (NR - 1) + KH rows][W * CI]
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  const int K = p.KH * p.KW * p.CI;
  const int rowf = p.W * p.CI;                 // floats per input row
  const int nrows = p.S * (NR - 1) + p.KH;
  const int gpi = p.OH / NR;                   // row groups per image
  const int bb = blockIdx.x / gpi, oh0 = (blockIdx.x - bb * gpi) * NR;
  const int ih_first = oh0 * p.S - p.pt;
  // ---- stage the rows: all loads of a thread before its stores ----
  {
    const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * rowf * 4));
    const int n4 = (nrows * rowf) >> 2;        // (rowf % 4 == 0: checked on the host)
    const int r4 = rowf >> 2;
    for (int e0 = 0; e0 < n4; e0 += 256 * 4) {
      float4 v[4];
      bool real[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256 + tid;
        const int r = e / r4, c4 = e - r * r4;
        const int ih = ih_first + r;
        real[u] = e < n4 && ih >= 0 && ih < p.H;
        v[u] = odin_run_load4(XR, real[u] ? (unsigned)((((bb * p.H + ih) * rowf) + 4 * c4) * 4) : ODIN_OOB);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < n4) {
          float4 t = v[u];
          if (p.center && real[u]) { t.x = 2.f * t.x - 1.f; t.y = 2.f * t.y - 1.f; t.z = 2.f * t.z - 1.f; t.w = 2.f * t.w - 1.f; }
          reinterpret_cast<float4*>(xs)[e] = t;
        }
      }
    }
  }
  // ---- weights / bias / tap tables (in flight while the rows land) ----
  const OdinRun WR = odin_run(p.w, (unsigned)((size_t)K * p.CO * 4));
  float a[NK2][RB];
  int toff[NK2], tkw[NK2];
  bool tok[NK2];
#pragma unroll
  for (int u = 0; u < NK2; ++u) {
    const int k = 2 * u + h;
    const int tap = sc_smalldiv(k, p.CI);
    const int c = k - tap * p.CI;
    const int kh = sc_smalldiv(tap, p.KW);
    tkw[u] = tap - kh * p.KW;
    toff[u] = (kh * p.W + tkw[u]) * p.CI + c;   // relative to the block's first (row, column) tap
    tok[u] = k < K;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int co = rb * 32 + l31;
      a[u][rb] = odin_run_load1(WR, (k < K && co < p.CO) ? (unsigned)((k * p.CO + co) * 4) : ODIN_OOB);
    }
  }
  float bias_r[RB][16];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = rb * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
      bias_r[rb][i] = (p.bias != nullptr && n < p.CO) ? p.bias[n] : 0.f;
    }
  __syncthreads();
  const int cpr = (p.OW + 31) >> 5;   // (the last 32-pixel block of a row may be ragged: the audio VAE's 40-pixel rows)
  float amx = 0.f;   // running max |y| of this lane: the range word of the activation (the plane layer above reads it)
  for (int j = wave; j < NR * cpr; j += 4) {  // wave-uniform: (row of the group, 32-pixel column block)
    const int rr = j / cpr, q0 = (j - rr * cpr) << 5;
    const int ow = q0 + l31;
    const int iw0 = ow * p.S - p.pl;
    const int base = (rr * p.S * p.W + iw0) * p.CI;
    float b[NK2];
#pragma unroll
    for (int u = 0; u < NK2; ++u) {
      const int iw = iw0 + tkw[u];
      const bool ok = tok[u] && iw >= 0 && iw < p.W;
      b[u] = ok ? xs[base + toff[u]] : 0.f;
    }
    float* outp = p.y + ((size_t)((bb * p.OH + oh0 + rr) * p.OW + ow)) * p.CO + 4 * h;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      f32x16 acc = f32x16_zero();
#pragma unroll
      for (int u = 0; u < NK2; ++u) acc = mfma32(a[u][rb], b[u], acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = rb * 32 + 8 * q + 4 * h;
        if (n + 3 < p.CO && ow < p.OW) {
          const float4 o = make_float4(sc_act<ELU>(p.act, acc[4 * q] + bias_r[rb][4 * q]),
                                       sc_act<ELU>(p.act, acc[4 * q + 1] + bias_r[rb][4 * q + 1]),
                                       sc_act<ELU>(p.act, acc[4 * q + 2] + bias_r[rb][4 * q + 2]),
                                       sc_act<ELU>(p.act, acc[4 * q + 3] + bias_r[rb][4 * q + 3]));
          amx = odin_amax3(odin_amax3(amx, o.x, o.y), o.z, o.w);
          *reinterpret_cast<float4*>(outp + rb * 32 + 8 * q) = o;
        }
      }
    }
  }
  __shared__ float ared[16];
  odin_amax_commit_wg(p.y_amax, amx, tid, 256, ared, blockIdx.x);
}

// The RGB first layer (K = 4 * 4 * 3 = 48, 32 output channels) of the kernel above on the f16 matrix pipe (round 6): its 24
// v_mfma_f32_32x32x2f32 per 32-pixel block are 1536 matrix-pipe cycles -- at batch 512 (CelebA) 10 us of fp32 MFMA time
// under a launch whose 67 MB of output need 13 us; measured 38 us.  With the operands as two f16 planes (odin_device.h:
// x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per 16 k-values) the block is 9 MFMAs = 288 cycles: the weights' planes
// live in 24 registers, a lane gathers its pixel's 24 k-values from the staged rows as before and splits them (6 x
// odin_split_h4).  Same staging, epilogue and range word; image values and weights are inside the f16 window unscaled.
template <bool ELU>
__global__ __launch_bounds__(256) void smallc_fwd_lds_h_kernel(SCParams p, int NR) {
  ODIN_DYN_SMEM(float, xs);  // [S (NR - 1) + KH rows][W * CI]
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  const int rowf = p.W * p.CI;                 // floats per input row
  const int nrows = p.S * (NR - 1) + p.KH;
  const int gpi = p.OH / NR;                   // row groups per image
  const int bb = blockIdx.x / gpi, oh0 = (blockIdx.x - bb * gpi) * NR;
  const int ih_first = oh0 * p.S - p.pt;
  // ---- stage the rows: all loads of a thread before its stores ----
  {
    const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * rowf * 4));
    const int n4 = (nrows * rowf) >> 2;        // (rowf % 4 == 0: checked on the host)
    const int r4 = rowf >> 2;
    for (int e0 = 0; e0 < n4; e0 += 256 * 4) {
      float4 v[4];
      bool real[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256 + tid;
        const int r = e / r4, c4 = e - r * r4;
        const int ih = ih_first + r;
        real[u] = e < n4 && ih >= 0 && ih < p.H;
        v[u] = odin_run_load4(XR, real[u] ? (unsigned)((((bb * p.H + ih) * rowf) + 4 * c4) * 4) : ODIN_OOB);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * 256 + tid;
        if (e < n4) {
          float4 t = v[u];
          if (p.center && real[u]) { t.x = 2.f * t.x - 1.f; t.y = 2.f * t.y - 1.f; t.z = 2.f * t.z - 1.f; t.w = 2.f * t.w - 1.f; }
          reinterpret_cast<float4*>(xs)[e] = t;
        }
      }
    }
  }
  // ---- weight planes (A operand: row = output channel l31, k = 16 s + 8 h + j), tap tables of this lane's 24 k ----
  const OdinRun WR = odin_run(p.w, (unsigned)((size_t)48 * p.CO * 4));
  u32x4 ah[3], al[3];
  int toff[24], tkw[24];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * s + 8 * h + j;
      const int tap = sc_smalldiv(k, 3);
      const int c = k - tap * 3;
      const int kh = tap >> 2;
      tkw[8 * s + j] = tap & 3;
      toff[8 * s + j] = (kh * p.W + (tap & 3)) * 3 + c;
      wv[j] = odin_run_load1(WR, l31 < p.CO ? (unsigned)((k * p.CO + l31) * 4) : ODIN_OOB);
    }
    u32x2 h0, l0, h1, l1;
    odin_split_h4<false>(make_float4(wv[0], wv[1], wv[2], wv[3]), 1.f, ODIN_LO_SCALE, h0, l0);
    odin_split_h4<false>(make_float4(wv[4], wv[5], wv[6], wv[7]), 1.f, ODIN_LO_SCALE, h1, l1);
    ah[s][0] = h0.x; ah[s][1] = h0.y; ah[s][2] = h1.x; ah[s][3] = h1.y;
    al[s][0] = l0.x; al[s][1] = l0.y; al[s][2] = l1.x; al[s][3] = l1.y;
  }
  float bias_r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = 8 * (i >> 2) + 4 * h + (i & 3);
    bias_r[i] = (p.bias != nullptr && n < p.CO) ? p.bias[n] : 0.f;
  }
  __syncthreads();
  const int cpr = (p.OW + 31) >> 5;
  float amx = 0.f;
  for (int j = wave; j < NR * cpr; j += 4) {  // wave-uniform: (row of the group, 32-pixel column block)
    const int rr = j / cpr, q0 = (j - rr * cpr) << 5;
    const int ow = q0 + l31;
    const int iw0 = ow * p.S - p.pl;
    const int base = (rr * p.S * p.W + iw0) * 3;
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      float b[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int iw = iw0 + tkw[8 * s + e];
        b[e] = (iw >= 0 && iw < p.W) ? xs[base + toff[8 * s + e]] : 0.f;
      }
      u32x2 h0, l0, h1, l1;
      odin_split_h4<false>(make_float4(b[0], b[1], b[2], b[3]), 1.f, ODIN_LO_SCALE, h0, l0);
      odin_split_h4<false>(make_float4(b[4], b[5], b[6], b[7]), 1.f, ODIN_LO_SCALE, h1, l1);
      u32x4 bh, bl;
      bh[0] = h0.x; bh[1] = h0.y; bh[2] = h1.x; bh[3] = h1.y;
      bl[0] = l0.x; bl[1] = l0.y; bl[2] = l1.x; bl[3] = l1.y;
      acx = mfma32_f16(ah[s], bl, acx);
      acc = mfma32_f16(ah[s], bh, acc);
      acx = mfma32_f16(al[s], bh, acx);
    }
    float* outp = p.y + ((size_t)((bb * p.OH + oh0 + rr) * p.OW + ow)) * p.CO + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = 8 * q + 4 * h;
      if (n + 3 < p.CO && ow < p.OW) {
        float o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o4[i] = sc_act<ELU>(p.act, fmaf(acx[4 * q + i], ODIN_LO_UNSCALE, acc[4 * q + i]) + bias_r[4 * q + i]);
        const float4 o = make_float4(o4[0], o4[1], o4[2], o4[3]);
        amx = odin_amax3(odin_amax3(amx, o.x, o.y), o.z, o.w);
        *reinterpret_cast<float4*>(outp + 8 * q) = o;
      }
    }
  }
  __shared__ float ared[16];
  odin_amax_commit_wg(p.y_amax, amx, tid, 256, ared, blockIdx.x);
}

// Weight gradient on the matrix cores, operands straight from HBM/L2 (no LDS staging): with
// K = KH*KW*Cin <= 63 the whole dW is RB x CB accumulator tiles (rows = taps (+ one bias row whose
// A operand is the constant 1), columns = output channels) and the MFMA reduction index is the
// pixel: per pixel pair a lane loads ONE x tap per row block (a gather: tap l31 of pixel p + h) and
// ONE dy value per column block (coalesced).  U pixel pairs are requested before their MFMAs.
// 262144 pixels of the dSprites first layer = 128 MFMAs per SIMD.
// NW waves per workgroup: the kernel is a stream over DY (33.5 MB for the dSprites first layer) with
// 4-byte loads, 8 of them in flight per wave -- latency-bound at one wave per SIMD (4 waves: 22 us =
// 1.7 TB/s); the slab-row cap fixes the number of workgroups, so the occupancy comes from 16-wave
// workgroups whose partial tiles meet in LDS.
template <int RB, int CB, int NW>
__global__ __launch_bounds__(NW * 64) void smallc_wgrad_mfma_kernel(SCParams p) {
  ODIN_DYN_SMEM(float, red);  // [NW waves][RB*CB][16][64]
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  const int K = p.KH * p.KW * p.CI;
  const int total = p.B * p.OH * p.OW;
  const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun DR = odin_run(p.dy, (unsigned)((size_t)total * p.CO * 4));
  // row (tap) geometry of this lane for each row block
  int tkh[RB], tkw[RB], tc[RB];
  float aone[RB];
  bool arow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int k = rb * 32 + l31;
    const int tap = k / p.CI;
    tc[rb] = k - tap * p.CI;
    tkh[rb] = tap / p.KW;
    tkw[rb] = tap - tkh[rb] * p.KW;
    arow[rb] = k < K;
    aone[rb] = (k == K) ? 1.f : 0.f;  // bias row
  }
  f32x16 acc[RB][CB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = f32x16_zero();
  // a block owns whole output rows (pix_per_block is a multiple of OW, OW even): the row decode
  // is wave-uniform scalar work, per pixel pair only the column changes
  const int r0 = blockIdx.x * (p.pix_per_block / p.OW);
  int r1 = r0 + p.pix_per_block / p.OW;
  if (r1 > p.B * p.OH) r1 = p.B * p.OH;
  constexpr int U = 4;
  for (int r = r0 + wave; r < r1; r += NW) {  // wave-uniform
    const int bb = r / p.OH, oh = r - bb * p.OH;
    const int ih0 = oh * p.S - p.pt;
    unsigned rowoff[RB];
    bool rowok[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int ih = ih0 + tkh[rb];
      rowok[rb] = arow[rb] && ih >= 0 && ih < p.H;
      rowoff[rb] = (unsigned)((((bb * p.H + ih) * p.W) * p.CI + tc[rb]) * 4);
    }
    const unsigned dyoff = (unsigned)((r * p.OW) * p.CO * 4);
    for (int q0 = 0; q0 < p.OW; q0 += 2 * U) {
      float a[U][RB], b[U][CB];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ow = q0 + 2 * u + h;
        const bool pv = ow < p.OW;
        const int iw0 = ow * p.S - p.pl;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const int iw = iw0 + tkw[rb];
          const bool ok = pv && rowok[rb] && iw >= 0 && iw < p.W;
          float v = odin_run_load1(XR, ok ? rowoff[rb] + (unsigned)(iw * p.CI * 4) : ODIN_OOB);
          if (p.center) v = ok ? 2.f * v - 1.f : 0.f;
          a[u][rb] = v + (pv ? aone[rb] : 0.f);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
          const int co = cb * 32 + l31;
          b[u][cb] = odin_run_load1(DR, (pv && co < p.CO) ? dyoff + (unsigned)((ow * p.CO + co) * 4) : ODIN_OOB);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int cb = 0; cb < CB; ++cb) acc[rb][cb] = mfma32(a[u][rb], b[u][cb], acc[rb][cb]);
    }
  }
  // combine the NW waves in a fixed order, then write the slab row [K*CO | CO]
  float* mine = red + wave * (RB * CB * 16 * 64);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[((rb * CB + cb) * 16 + r) * 64 + lane] = acc[rb][cb][r];
  __syncthreads();
  float* row = p.y + (size_t)blockIdx.x * p.slab_stride;
  for (int e = tid; e < RB * CB * 16 * 64; e += NW * 64) {
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; wv += 4)
      t += (red[e + wv * RB * CB * 1024] + red[e + (wv + 1) * RB * CB * 1024]) +
           (red[e + (wv + 2) * RB * CB * 1024] + red[e + (wv + 3) * RB * CB * 1024]);
    const int ln = e & 63, r = (e >> 6) & 15, blk = e >> 10;
    const int rb = blk / CB, cb = blk - rb * CB;
    const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int co = cb * 32 + (ln & 31);
    if (co < p.CO && k <= K) row[(size_t)k * p.CO + co] = t;  // k == K: the bias row
  }
}

}  // namespace

bool odin_smallc_applicable(const odin_conv_desc* d) {
  const int K = d->KH * d->KW * d->Cin;
  if ((long)d->B * d->OH * d->OW * d->Cout >= (1L << 29)) return false;  // byte offsets fit 31 bits
  if ((long)d->B * d->H * d->W * d->Cin >= (1L << 29)) return false;
  return d->Cin <= 4 && (d->Cout % 4) == 0 && d->Cout <= 64 && (K == 16 || K == 25 || K == 48) &&
         (size_t)16 * (K + 1) * d->Cout * 4 <= 150 * 1024;
}

static bool g_sc_planes = true;
// tests / A-B runs: 0 = the RGB first layer's forward on the fp32 matrix instructions (round 5); < 0 = only report
extern "C" int odin_debug_smallc_planes(int enable) {
  const int old = g_sc_planes ? 1 : 0;
  if (enable >= 0) g_sc_planes = enable != 0;
  return old;
}
static void sc_fill(SCParams& p, const odin_conv_desc* d) {
  memset(&p, 0, sizeof(p));
  p.B = d->B; p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.S = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.act = d->act; p.center = d->center;
}

static int smallc_fwd_launch(const float* x, const float* w, const float* bias, float* y,
                             const odin_conv_desc* d, void* stream, bool* tracked);

// (the range word of y, d->y_amax: kept by every variant's epilogue)
int odin_smallc_fwd(const float* x, const float* w, const float* bias, float* y,
                    const odin_conv_desc* d, void* stream) {
  bool tracked = false;
  const int rc = smallc_fwd_launch(x, w, bias, y, d, stream, &tracked);
  (void)tracked;   // (every variant folds max|y| into d->y_amax from its own epilogue)
  return rc;
}

static int smallc_fwd_launch(const float* x, const float* w, const float* bias, float* y,
                             const odin_conv_desc* d, void* stream, bool* tracked) {
  SCParams p;
  sc_fill(p, d);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.y_amax = d->y_amax;
  const long total = (long)d->B * d->OH * d->OW * (d->Cout / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  size_t lds = (size_t)(d->KH * d->KW * d->Cin + 1) * d->Cout * 4;
  const bool small_x = (size_t)d->B * d->H * d->W * d->Cin * 4 < 0x7FFFFFF0u;
  // (rows of whole 32-pixel blocks; the LDS form also takes a ragged last block -- the audio VAE's 40-pixel rows, round 6)
  if (small_x && ((d->OW % 32) == 0 || d->OW > 32) && !ODIN_DIAG_ENV("ODIN_SMALLC_VALU")) {
    const int K = d->KH * d->KW * d->Cin;
    const int nk2 = (K + 1) / 2, rb = (d->Cout + 31) / 32;
    const long n_it = (long)d->B * d->OH * (d->OW / 32);
    // rows staged in LDS (RGB first layers; also the grey ones when the row groups divide evenly)
    if (!ODIN_DIAG_ENV("ODIN_SMALLC_NOLDS") && ((d->W * d->Cin) % 4) == 0 && (K == 16 || K == 48)) {
      // (rows per workgroup: 16 for RGB, 8 for one channel -- 14.2 vs 14.9 us on 64x64x1, profiles/r03_enc0bench.txt)
      int NR = ODIN_DIAG_ENV("ODIN_SMALLC_NR") ? atoi(ODIN_DIAG_ENV("ODIN_SMALLC_NR")) : (d->Cin == 1 ? 8 : 16);
      while (NR > 1 && ((d->OH % NR) != 0 ||
                        (size_t)(d->stride * (NR - 1) + d->KH) * d->W * d->Cin * 4 > 48 * 1024))
        NR >>= 1;
      const size_t l3 = (size_t)(d->stride * (NR - 1) + d->KH) * d->W * d->Cin * 4;
      const long gb = (long)d->B * (d->OH / NR);
      if (l3 <= 48 * 1024 && gb < (1L << 30)) {
        if (nk2 == 8 && rb == 1) {
          if (d->act == ODIN_ACT_ELU) ODIN_LAUNCH((smallc_fwd_lds_kernel<8, 1, true>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          else ODIN_LAUNCH((smallc_fwd_lds_kernel<8, 1, false>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          *tracked = true;
          return odin_check_launch("smallc_fwd_lds");
        }
        if (nk2 == 8 && rb == 2) {
          if (d->act == ODIN_ACT_ELU) ODIN_LAUNCH((smallc_fwd_lds_kernel<8, 2, true>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          else ODIN_LAUNCH((smallc_fwd_lds_kernel<8, 2, false>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          *tracked = true;
          return odin_check_launch("smallc_fwd_lds");
        }
        if (nk2 == 24 && rb == 1 && d->Cin == 3 && d->KW == 4 && d->Cout == 32 && g_sc_planes && !odin_exact_fp32()) {
          // the RGB first layer on two f16 planes
          if (d->act == ODIN_ACT_ELU) ODIN_LAUNCH((smallc_fwd_lds_h_kernel<true>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          else ODIN_LAUNCH((smallc_fwd_lds_h_kernel<false>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          *tracked = true;
          return odin_check_launch("smallc_fwd_lds(f16x2)");
        }
        if (nk2 == 24 && rb == 1) {
          if (d->act == ODIN_ACT_ELU) ODIN_LAUNCH((smallc_fwd_lds_kernel<24, 1, true>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          else ODIN_LAUNCH((smallc_fwd_lds_kernel<24, 1, false>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          *tracked = true;
          return odin_check_launch("smallc_fwd_lds");
        }
        if (nk2 == 24 && rb == 2) {
          if (d->act == ODIN_ACT_ELU) ODIN_LAUNCH((smallc_fwd_lds_kernel<24, 2, true>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          else ODIN_LAUNCH((smallc_fwd_lds_kernel<24, 2, false>), dim3((unsigned)gb), dim3(256), l3, stream, p, NR);
          *tracked = true;
          return odin_check_launch("smallc_fwd_lds");
        }
      }
    }
    // persistent waves (2 x 4-wave workgroups per CU-pair ... 4 blocks per wave at batch 256)
    if ((d->OW % 32) == 0) {
    long bl = (n_it + 3) / 4;
    const long cap = ODIN_DIAG_ENV("ODIN_SMALLC_BL") ? atol(ODIN_DIAG_ENV("ODIN_SMALLC_BL")) : 4L * odin_num_cus();
    if (bl > cap) bl = cap;
    if (nk2 == 8 && rb == 1) { ODIN_LAUNCH((smallc_fwd_mfma_kernel<8, 1>), dim3((unsigned)bl), dim3(256), 0, stream, p); return odin_check_launch("smallc_fwd_mfma"); }
    if (nk2 == 8 && rb == 2) { ODIN_LAUNCH((smallc_fwd_mfma_kernel<8, 2>), dim3((unsigned)bl), dim3(256), 0, stream, p); return odin_check_launch("smallc_fwd_mfma"); }
    if (nk2 == 24 && rb == 1) { ODIN_LAUNCH((smallc_fwd_mfma_kernel<24, 1>), dim3((unsigned)bl), dim3(256), 0, stream, p); return odin_check_launch("smallc_fwd_mfma"); }
    if (nk2 == 24 && rb == 2) { ODIN_LAUNCH((smallc_fwd_mfma_kernel<24, 2>), dim3((unsigned)bl), dim3(256), 0, stream, p); return odin_check_launch("smallc_fwd_mfma"); }
    }
  }
  if (small_x && d->KH == 4 && d->KW == 4 && d->Cin == 1)
    ODIN_LAUNCH((smallc_fwd_kernel_t<4, 4, 1>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  else if (small_x && d->KH == 4 && d->KW == 4 && d->Cin == 3)
    ODIN_LAUNCH((smallc_fwd_kernel_t<4, 4, 3>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  else if (small_x && d->KH == 5 && d->KW == 5 && d->Cin == 1)
    ODIN_LAUNCH((smallc_fwd_kernel_t<5, 5, 1>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  else
    ODIN_LAUNCH(smallc_fwd_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, p);
  return odin_check_launch("smallc_fwd");
}

// The same weight gradient for the first layers of the image stacks (4x4 / stride 2, pads (1, 1), 64-pixel input rows
// of CI = 1 or 3 channels, <= 32 output channels): the four input rows of an output row are STAGED in LDS by the wave
// that owns the row -- 1 (CI = 1) or 3 (CI = 3) coalesced 16-byte loads per lane instead of one 4-byte gather per tap
// row and pixel pair -- with four zero floats either side of each row, so a tap read needs no bounds test: its address
// is a per-lane constant plus the pixel step.  A row of ones feeds the bias row of the tile, a row of zeros the unused
// tile rows.  fp32 MFMAs do not run beside VALU work: the ~6 VALU per gather of the kernel above were
// half of its time.  The next row's loads are in flight while the current row is multiplied.
// PL (round 6, the RGB layer): the same walk on the f16 matrix pipe -- x (image values, the ones row) unscaled, dy times the
// power of two of its range word, both as two planes (odin_device.h), 16 pixels per k-step: 6 v_mfma_f32_32x32x16_f16 per
// 16 pixels and row block instead of 8 v_mfma_f32_32x32x2f32 (1024 -> 384 matrix-pipe cycles per output row); 8 waves per
// workgroup (two workgroups per CU) so that a wave may hold the main + cross accumulators beside its operands.
// W_: input row width (64; 80 = the audio VAE's 96 x 80 patches, round 6); output rows of W_ / 2 pixels, a multiple of 8.
template <int RB, int CI, int NW_ = 16, bool PL = false, int W_ = 64>
__global__ __launch_bounds__(NW_ * 64) void smallc_wgrad_lds_kernel(SCParams p) {
  constexpr int NW = NW_, U = 4, W = W_, OW = W_ / 2;
  constexpr int K = 16 * CI;
  constexpr int NCH = W * CI / 4;                 // 16-byte chunks per input row
  constexpr int NI = (4 * NCH + 63) / 64;         // chunks per lane and output row
  constexpr int RS = W * CI + 8;                  // floats per staged row
  constexpr int WS = 6 * RS;                      // per wave: 4 data rows, the ones row, the zero row
  ODIN_DYN_SMEM(float, smem);  // staging [NW][WS] inside the loop, partial tiles [NW][RB][16][64] behind it
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  float* st = smem + wave * WS;
  const OdinRun XR = odin_run(p.x, (unsigned)((size_t)p.B * p.H * W * CI * 4));
  const OdinRun DR = odin_run(p.dy, (unsigned)((size_t)p.B * p.OH * OW * p.CO * 4));
  // guards, ones row, zero row (the data rows' interiors are rewritten for every output row)
  for (int e = lane; e < WS; e += 64) st[e] = (e >= 4 * RS && e < 5 * RS) ? 1.f : 0.f;
  // this lane's tile rows: k = tap * CI + c -> staged row kh, float 4 + (2 ow - 1 + kw) CI + c for pixel ow
  int abase[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int k = rb * 32 + l31;
    const int tap = k / CI, c = k - tap * CI, kh = tap >> 2, kw = tap & 3;
    int idx = kh * RS + 4 + (kw - 1) * CI + c;
    if (k == K) idx = 4 * RS + 4;   // bias row: ones
    if (k > K) idx = 5 * RS + 4;    // unused row of the tile: zeros
    abase[rb] = (wave * WS + idx + h * 2 * CI) * 4;  // bytes; pixel ow = q0 + 2 u + h: + (q0 + 2 u) 2 CI floats
  }
  unsigned boff[U];
#pragma unroll
  for (int u = 0; u < U; ++u) boff[u] = l31 < p.CO ? (unsigned)(((2 * u + h) * p.CO + l31) * 4) : ODIN_OOB_V;
  // staging chunks of this lane: chunk e = lane + 64 i of the 4 NCH chunks of rows ih0 .. ih0 + 3
  int cdst[NI];
  unsigned csrc[NI];
  bool cj0[NI], cj3[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = lane + 64 * i, j = e / NCH, ci = e - j * NCH;
    const bool in = e < 4 * NCH;
    cdst[i] = in ? (wave * WS + j * RS + 4 + 4 * ci) * 4 : -1;
    csrc[i] = in ? (unsigned)(e * 16) : ODIN_OOB;
    cj0[i] = j == 0;
    cj3[i] = j == 3;
  }
  f32x16 acc[RB], acx[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) { acc[rb] = f32x16_zero(); acx[rb] = f32x16_zero(); }
  // PL: dy is carried times 2^gk (its range word: the data gradient of the layer above kept it)
  int gk = 0;
  if (PL) gk = odin_range_shift(odin_range_finish(odin_range_issue(p.dy_amax, lane)));
  const float g_s = odin_pow2(gk), g_s2k = odin_pow2(gk + 11);
  unsigned boffp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) boffp[j] = l31 < p.CO ? (unsigned)(((8 * h + j) * p.CO + l31) * 4) : ODIN_OOB_V;
  const int r0 = blockIdx.x * (p.pix_per_block / OW);
  int r1 = r0 + p.pix_per_block / OW;
  if (r1 > p.B * p.OH) r1 = p.B * p.OH;
  float4 nx[NI];
  auto fetch = [&](int r) {  // rows 2 oh - 1 .. 2 oh + 2 of image bb: contiguous in memory
    const int bb = r / p.OH, oh = r - bb * p.OH;
    const unsigned base = (unsigned)(((bb * p.H + 2 * oh - 1) * W * CI) * 4);
    const bool top = oh == 0, bot = oh == p.OH - 1, live = r < r1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool ok = live && !(cj0[i] && top) && !(cj3[i] && bot);
      nx[i] = odin_run_load4(XR, ok ? base + csrc[i] : ODIN_OOB);
      if (p.center && ok) {
        nx[i].x = 2.f * nx[i].x - 1.f; nx[i].y = 2.f * nx[i].y - 1.f;
        nx[i].z = 2.f * nx[i].z - 1.f; nx[i].w = 2.f * nx[i].w - 1.f;
      }
    }
  };
  fetch(r0 + wave);
  for (int r = r0 + wave; r < r1; r += NW) {  // wave-uniform
    odin_wave_sync();  // (the wave's reads of the previous row are issued; LDS runs a wave's accesses in order)
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (cdst[i] >= 0) *reinterpret_cast<float4*>(reinterpret_cast<char*>(smem) + cdst[i]) = nx[i];
    odin_wave_sync();
    fetch(r + NW);
    const unsigned dyrow = (unsigned)(r * OW * p.CO * 4);
    if (PL) {
#pragma unroll
      for (int q0 = 0; q0 < OW; q0 += 16) {
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = odin_run_load1s(DR, boffp[j], dyrow + (unsigned)(q0 * p.CO * 4));
        u32x2 h0, l0, h1, l1;
        odin_split_h4<true>(make_float4(bv[0], bv[1], bv[2], bv[3]), g_s, g_s2k, h0, l0);
        odin_split_h4<true>(make_float4(bv[4], bv[5], bv[6], bv[7]), g_s, g_s2k, h1, l1);
        u32x4 bh, bl;
        bh[0] = h0.x; bh[1] = h0.y; bh[2] = h1.x; bh[3] = h1.y;
        bl[0] = l0.x; bl[1] = l0.y; bl[2] = l1.x; bl[3] = l1.y;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          float av[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)   // pixel q0 + 8 h + j (abase carries h * 2 CI floats: 14 h CI more here)
            av[j] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + abase[rb] +
                                                    ((q0 + j) * 2 * CI + 14 * h * CI) * 4);
          odin_split_h4<false>(make_float4(av[0], av[1], av[2], av[3]), 1.f, ODIN_LO_SCALE, h0, l0);
          odin_split_h4<false>(make_float4(av[4], av[5], av[6], av[7]), 1.f, ODIN_LO_SCALE, h1, l1);
          u32x4 ah, al;
          ah[0] = h0.x; ah[1] = h0.y; ah[2] = h1.x; ah[3] = h1.y;
          al[0] = l0.x; al[1] = l0.y; al[2] = l1.x; al[3] = l1.y;
          acx[rb] = mfma32_f16(ah, bl, acx[rb]);
          acc[rb] = mfma32_f16(ah, bh, acc[rb]);
          acx[rb] = mfma32_f16(al, bh, acx[rb]);
        }
      }
      continue;
    }
#pragma unroll 1
    for (int q0 = 0; q0 < OW; q0 += 2 * U) {
      float a[U][RB], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          a[u][rb] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + abase[rb] +
                                                     (q0 + 2 * u) * 2 * CI * 4);
        b[u] = odin_run_load1s(DR, boff[u], dyrow + (unsigned)(q0 * p.CO * 4));
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = mfma32(a[u][rb], b[u], acc[rb]);
    }
  }
  // combine the NW waves in a fixed order, then write the slab row [K*CO | CO]
  __syncthreads();  // (the partial tiles overlay the staging areas)
  float* mine = smem + wave * (RB * 16 * 64);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      mine[(rb * 16 + r) * 64 + lane] = PL ? fmaf(acx[rb][r], ODIN_LO_UNSCALE, acc[rb][r]) * odin_pow2(-gk) : acc[rb][r];
  __syncthreads();
  float* row = p.y + (size_t)blockIdx.x * p.slab_stride;
  for (int e = tid; e < RB * 16 * 64; e += NW * 64) {
    float t = 0.f;
#pragma unroll
    for (int wv = 0; wv < NW; wv += 4)
      t += (smem[e + wv * RB * 1024] + smem[e + (wv + 1) * RB * 1024]) +
           (smem[e + (wv + 2) * RB * 1024] + smem[e + (wv + 3) * RB * 1024]);
    const int ln = e & 63, r = (e >> 6) & 15, rb = e >> 10;
    const int k = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
    const int co = ln & 31;
    if (co < p.CO && k <= K) row[(size_t)k * p.CO + co] = t;  // k == K: the bias row
  }
}

int odin_smallc_wgrad(const float* x, const float* dy, float* slab, int* rows_out,
                      const odin_conv_desc* d, void* stream) {
  SCParams p;
  sc_fill(p, d);
  p.x = x; p.dy = dy; p.y = slab;
  const long total = (long)d->B * d->OH * d->OW;
  int rows = ODIN_MAX_SLAB_BLOCKS;
  if (total < rows * 64L) rows = (int)((total + 63) / 64);
  p.pix_per_block = (int)((total + rows - 1) / rows);
  const bool use_mfma = !ODIN_DIAG_ENV("ODIN_SMALLC_VALU") && (d->OW % 2) == 0;
  if (use_mfma)  // whole output rows per block
    p.pix_per_block = (p.pix_per_block + d->OW - 1) / d->OW * d->OW;
  rows = (int)((total + p.pix_per_block - 1) / p.pix_per_block);
  p.slab_stride = d->KH * d->KW * d->Cin * d->Cout + d->Cout;
  if (rows_out) *rows_out = rows;
  if (slab == nullptr) return 0;
  const int K = d->KH * d->KW * d->Cin;
  if (use_mfma) {
    // matrix-core kernel: RB row blocks of 32 (taps + the bias row), CB column blocks of 32
    const int RB = (K + 1 + 31) / 32, CB = (d->Cout + 31) / 32;
    const int NW = (RB * CB == 4) ? 8 : 16;   // NW x RB x CB x 4 KB of LDS for the partial tiles
    const size_t l2 = (size_t)NW * RB * CB * 1024 * 4;
    // rows staged in LDS: the first layers of the image stacks
    if ((d->Cin == 1 || d->Cin == 3) && ((d->W == 64 && d->OW == 32) || (d->Cin == 1 && d->W == 80 && d->OW == 40)) &&
        d->H == 2 * d->OH && d->KH == 4 && d->KW == 4 &&
        d->stride == 2 && d->pad_t == 1 && d->pad_l == 1 && d->Cout <= 32 && !ODIN_DIAG_ENV("ODIN_SMALLC_NOLDS") &&
        (size_t)d->B * d->OH * d->OW * d->Cout * 4 < (1ull << 31)) {
      const size_t stage = (size_t)16 * 6 * (d->W * d->Cin + 8) * 4;
      const size_t l3 = stage > (size_t)16 * RB * 4096 ? stage : (size_t)16 * RB * 4096;
#ifndef ODIN_SIM
      static bool attr3 = false;
      if (!attr3) {
        const void* fns[4] = {reinterpret_cast<const void*>(&smallc_wgrad_lds_kernel<1, 1>),
                              reinterpret_cast<const void*>(&smallc_wgrad_lds_kernel<2, 3>),
                              reinterpret_cast<const void*>(&smallc_wgrad_lds_kernel<2, 3, 8, true>),
                              reinterpret_cast<const void*>(&smallc_wgrad_lds_kernel<1, 1, 16, false, 80>)};
        for (const void* f : fns)
          if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr3 = true;
      }
#endif
      if (d->Cin == 3 && g_sc_planes && !odin_exact_fp32()) {
        // the RGB layer on two f16 planes: 8 waves per workgroup, the range word of dy (one counted pass if the caller
        // brought none)
        p.dy_amax = odin_range_word_of(dy, (size_t)d->B * d->OH * d->OW * d->Cout, d->dy_amax, stream);
        if (p.dy_amax == nullptr) return odin_fail(-3, "smallc_wgrad: no range word for dy");
        const size_t stage8 = (size_t)8 * 6 * (64 * 3 + 8) * 4, tiles8 = (size_t)8 * RB * 4096;
        ODIN_LAUNCH((smallc_wgrad_lds_kernel<2, 3, 8, true>), dim3(rows), dim3(512), stage8 > tiles8 ? stage8 : tiles8, stream, p);
        return odin_check_launch("smallc_wgrad_lds(f16x2)");
      }
      if (d->Cin == 1 && d->W == 80) ODIN_LAUNCH((smallc_wgrad_lds_kernel<1, 1, 16, false, 80>), dim3(rows), dim3(1024), l3, stream, p);
      else if (d->Cin == 1) ODIN_LAUNCH((smallc_wgrad_lds_kernel<1, 1>), dim3(rows), dim3(1024), l3, stream, p);
      else ODIN_LAUNCH((smallc_wgrad_lds_kernel<2, 3>), dim3(rows), dim3(1024), l3, stream, p);
      return odin_check_launch("smallc_wgrad_lds");
    }
#ifndef ODIN_SIM
    static bool attr2 = false;
    if (!attr2) {
      const void* fns[4] = {reinterpret_cast<const void*>(&smallc_wgrad_mfma_kernel<1, 1, 16>),
                            reinterpret_cast<const void*>(&smallc_wgrad_mfma_kernel<1, 2, 16>),
                            reinterpret_cast<const void*>(&smallc_wgrad_mfma_kernel<2, 1, 16>),
                            reinterpret_cast<const void*>(&smallc_wgrad_mfma_kernel<2, 2, 8>)};
      for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          (void)hipGetLastError();
      attr2 = true;
    }
#endif
    if (RB == 1 && CB == 1) ODIN_LAUNCH((smallc_wgrad_mfma_kernel<1, 1, 16>), dim3(rows), dim3(1024), l2, stream, p);
    else if (RB == 1 && CB == 2) ODIN_LAUNCH((smallc_wgrad_mfma_kernel<1, 2, 16>), dim3(rows), dim3(1024), l2, stream, p);
    else if (RB == 2 && CB == 1) ODIN_LAUNCH((smallc_wgrad_mfma_kernel<2, 1, 16>), dim3(rows), dim3(1024), l2, stream, p);
    else ODIN_LAUNCH((smallc_wgrad_mfma_kernel<2, 2, 8>), dim3(rows), dim3(512), l2, stream, p);
    return odin_check_launch("smallc_wgrad_mfma");
  }
  size_t lds = (size_t)16 * (K + 1) * d->Cout * 4;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_wgrad_kernel<16>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_wgrad_kernel<25>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_wgrad_kernel<48>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  if (K == 16) ODIN_LAUNCH(smallc_wgrad_kernel<16>, dim3(rows), dim3(1024), lds, stream, p);
  else if (K == 25) ODIN_LAUNCH(smallc_wgrad_kernel<25>, dim3(rows), dim3(1024), lds, stream, p);
  else ODIN_LAUNCH(smallc_wgrad_kernel<48>, dim3(rows), dim3(1024), lds, stream, p);
  return odin_check_launch("smallc_wgrad");
}
