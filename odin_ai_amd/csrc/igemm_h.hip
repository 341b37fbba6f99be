// igemm_h.hip -- convolutions of ANY spatial size as implicit GEMMs on the f16 matrix pipe with fp32 operands carried
// as two planes (odin_device.h: x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per 16 k-values into a main and a
// cross accumulator, <= 3 * 2^-22 per product), both operands straight from L2 -- the plane arithmetic of
// fconv_planes / tconv_planes / wgrad_planes without their whole-row tiles of 8 / 16 / 32 pixels.  Those kernels stage
// row windows through LDS and need row widths that are powers of two; the audio VAE's spectrogram stack
// (examples/vae/vae_audio.py:84-110 on the image stack of image_networks.py:460-513: 96 x 80 -> 48 x 40 -> 24 x 20
// -> 12 x 10 -> 6 x 5) fits none of them and ran on the fp32 gather kernels (0.31 of the fp32 MFMA peak).  Layers
// from ~1.5 GFLOP per launch come here (odin_igemm_h_applicable); the small-spatial layers of the image stacks
// (encoder3 / decoder2: 8 x 8 and 4 x 4 images, 0.5 GFLOP) stay on igemm.hip, whose workgroups split a tile's
// reduction over their waves and pair a layer's weight and data gradient in one launch (same-box A/B: the dSprites
// step 0.544 ms there, 0.595 ms here).
//
//   forward / data gradient (igemm_h_kernel):  C[m][j] = sum_k A(m, k) * Wt(k, j)
//     rows m = output pixels (for the transposed gathers ordered by stride class, so that a tile's rows share their
//     valid taps: 4 of the 16 taps of a 4x4/s2 kernel); k = (tap, channel), 16 consecutive channels of one tap per
//     MFMA step: lane (row l31, half h) loads the 8 channels 8 h .. 8 h + 7 of its gathered pixel (32 bytes) and of
//     its weight column, splits both into planes and issues the three MFMAs.
//     ONE WAVE OWNS ONE 32 x 32 TILE and walks tiles grid-stride: no partial tiles, no workgroup barrier in the loop
//     (the row table of a tile is wave-private LDS); the column sums of a data gradient (the bias gradient of a
//     Conv2DTranspose below) are kept per lane across the tiles of a wave and meet once at the end.  Where the weight
//     planes of one stride class and 32 output columns fit in 64 KB of LDS (igemm_hw_*: all the audio stack's layers
//     but the 64-channel 16-tap ones) a workgroup splits them once and B comes from LDS.
//   weight gradient (igemm_h_wgrad_kernel):    dW[(tap, cu)][cv] = sum_m U(pix(m, tap), cu) * V(m, cv)
//     the reduction over the pixels m runs 16 per MFMA step; split over gridDim.z workgroups (one slab row each) and
//     their waves, partial tiles meet in LDS in wave order.
// A gradient operand is scaled by the power of two of its range word on its way into the planes (odin_range_shift),
// the data gradient keeps the range word of its output.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>
#include <type_traits>

namespace {

#ifdef ODIN_SIM
#define IH_UNIFORM(x) (x)
#else
#define IH_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// four waves per SIMD: a tile is 8-32 steps behind a row decode and a first batch of gathers whose latency only other
// waves can cover (two waves per SIMD left ~3 k cycles exposed per tile)
#ifdef ODIN_SIM
#define IH_WAVES_ATTR
#else
#define IH_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
constexpr int IH_NW = 4;  // waves (= tiles in flight) per workgroup

struct IHParams {
  const float* in;    // gathered tensor [B, H, W, CI]
  const float* w;
  float* out;         // [B, OH, OW, CO]
  const float* bias;  // forward
  const float* aux;   // data gradient: multiply by act'(aux), aux shaped like out
  float* colsum;      // data gradient: slab [gridDim.y][CO] of column sums (may be null)
  const unsigned* in_amax;  // SC: range word of `in` (a gradient tensor, or an activation: in_cond)
  int in_cond;              // `in` is an ACTIVATION: scaled only when its bound leaves [2^-8, 2^15)
  unsigned* out_amax;       // data gradient: range word of `out` (may be null)
  int B, H, W, CI, OH, OW, CO, KH, KW, S, pt, pl;
  int gpt;            // 16-channel groups per tap = CI / 16
  int act, aux_act;
  int Mc;             // rows per stride class
  int tpc;            // 32-row tiles per stride class
  int ntile;          // row tiles in all = classes * tpc
  unsigned dbg;       // diagnostics build (ODIN_IH_DBG): bit 0 = gathers read nothing, bit 1 = nothing is stored
};

template <bool SC>
__device__ __forceinline__ void ih_split8(const float (&v)[8], float s, float s2k, u32x4& hi, u32x4& lo) {
  u32x2 h0, l0, h1, l1;
  odin_split_h4<SC>(make_float4(v[0], v[1], v[2], v[3]), s, s2k, h0, l0);
  odin_split_h4<SC>(make_float4(v[4], v[5], v[6], v[7]), s, s2k, h1, l1);
  hi[0] = h0.x; hi[1] = h0.y; hi[2] = h1.x; hi[3] = h1.y;
  lo[0] = l0.x; lo[1] = l0.y; lo[2] = l1.x; lo[3] = l1.y;
}

// position of a wave inside the reduction of its tile: tap (a, c) of the stride class, channel group cg
struct IHCursor {
  int t, cg, a, c;
};

template <bool TMODE, bool BKC, bool SC>
__global__ __launch_bounds__(IH_NW * 64) void igemm_h_kernel(IHParams p) {
  __shared__ int rowoff[IH_NW][32];
  __shared__ float cred[IH_NW * 32 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = IH_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int j = blockIdx.x * 32 + l31;
  const bool b_ok = j < p.CO;
  const float bj = (p.bias != nullptr && b_ok) ? p.bias[j] : 0.f;
  const OdinRun RA = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun RB = odin_run(p.w, (unsigned)((size_t)p.KH * p.KW * p.CI * p.CO * 4));
  const OdinRun RO = odin_run(p.out, (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4));
  const bool has_aux = p.aux != nullptr;
  const OdinRun RX = odin_run(has_aux ? p.aux : p.in, has_aux ? (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4) : 0u);
  // a gradient input is carried times 2^gk (its maximum lands in [2^14, 2^15)), the sums are scaled back; an ACTIVATION
  // input (in_cond) only when its bound leaves the safe window (a wave-uniform flag around the split)
  const unsigned in_mb = SC ? odin_range_load(p.in_amax) : 0u;
  const bool sc_on = SC && (!p.in_cond || odin_act_needs_scale(in_mb));
  const int gk = sc_on ? odin_range_shift(in_mb) : 0;
  const float in_s = odin_pow2(gk), in_s2k = odin_pow2(gk + 11);
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11);
  const int SS = TMODE ? p.S : 1;     // 1 or 2
  const int ssh = SS >> 1;            // x / SS = x >> ssh,  x % SS = x & (SS - 1)
  const int ohs = p.OH >> ssh, ows = p.OW >> ssh;
  const unsigned img = (unsigned)(ohs * ows);
  const int sg = TMODE ? -1 : 1;
  const int gpt = p.gpt;
  const unsigned bl = b_ok ? (unsigned)((BKC ? j * p.CI + 8 * h : 8 * h * p.CO + j) * 4) : ODIN_OOB_V;
  float csum = 0.f, amx = 0.f;
  for (int ty = IH_UNIFORM((int)blockIdx.y * IH_NW + wave); ty < p.ntile; ty += (int)gridDim.y * IH_NW) {
    // ---- this tile's stride class and its taps (wave-uniform) ----
    const int cls = TMODE ? ty / p.tpc : 0;
    const int tile = TMODE ? ty - cls * p.tpc : ty;
    const int cy = cls >> ssh, cx = cls - (cy << ssh);
    const int kh0 = TMODE ? (cy + p.pt) & (SS - 1) : 0, kw0 = TMODE ? (cx + p.pl) & (SS - 1) : 0;
    const int nkh = (p.KH - kh0 + SS - 1) >> ssh, nkw = (p.KW - kw0 + SS - 1) >> ssh;
    const int ntap = nkh * nkw;
    // ---- this lane's A row: one output pixel ----
    const unsigned q = (unsigned)(tile * 32 + l31);
    const bool a_ok = q < (unsigned)p.Mc;
    const unsigned b = q / img, r = q - b * img;
    const unsigned ys = r / (unsigned)ows, xs = r - ys * (unsigned)ows;
    const int oy = (int)ys * SS + cy, ox = (int)xs * SS + cx;
    // gathered pixel of tap (a, c): (Y0 + sg a, X0 + sg c)
    const int Y0 = TMODE ? (oy + p.pt - kh0) >> ssh : oy * p.S - p.pt;
    const int X0 = TMODE ? (ox + p.pl - kw0) >> ssh : ox * p.S - p.pl;
    unsigned mask = 0;
    {
      int t = 0;
      for (int a = 0; a < nkh; ++a)
        for (int c = 0; c < nkw; ++c, ++t) {
          const unsigned in = ((unsigned)(Y0 + sg * a) < (unsigned)p.H) & ((unsigned)(X0 + sg * c) < (unsigned)p.W);
          mask |= in << t;
        }
    }
    mask = a_ok ? mask : 0u;
    const int lanebase = (((int)b * p.H + Y0) * p.W + X0) * p.CI + 8 * h;  // floats
    odin_wave_sync();  // (the previous tile's epilogue has read the table)
    if (h == 0) rowoff[wave][l31] = a_ok ? (((int)b * p.OH + oy) * p.OW + ox) : -1;
    odin_wave_sync();
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
    float a0[8], b0[8], a1[8], b1[8];
    IHCursor cur = {0, 0, 0, 0};
    auto load_group = [&](float (&av)[8], float (&bv)[8]) {
      // branch-free: a group beyond the reduction reads out of range (zeros, no traffic)
      const unsigned live = (unsigned)(cur.t - ntap) >> 31;   // 1 / 0 (wave-uniform)
      const unsigned dead = live - 1u;
      const int tapoff = sg * (cur.a * p.W + cur.c) * p.CI;
      const unsigned va = live & (mask >> (cur.t & 31)) & 1u;
      const unsigned ao = (unsigned)((lanebase + tapoff + 16 * cur.cg) * 4);
      const float4 x0 = odin_run_load4(RA, ao | (va - 1u)), x1 = odin_run_load4(RA, (ao + 16u) | (va - 1u));
      av[0] = x0.x; av[1] = x0.y; av[2] = x0.z; av[3] = x0.w; av[4] = x1.x; av[5] = x1.y; av[6] = x1.z; av[7] = x1.w;
      const int wt = (kh0 + cur.a * SS) * p.KW + kw0 + cur.c * SS;  // weight tap
      if constexpr (BKC) {
        const unsigned so = (unsigned)(wt * p.CO * p.CI + 16 * cur.cg) * 4u;
        const float4 y0 = odin_run_load4s(RB, bl | dead, so), y1 = odin_run_load4s(RB, bl | dead, so + 16u);
        bv[0] = y0.x; bv[1] = y0.y; bv[2] = y0.z; bv[3] = y0.w; bv[4] = y1.x; bv[5] = y1.y; bv[6] = y1.z; bv[7] = y1.w;
      } else {
        const unsigned so = (unsigned)((wt * p.CI + 16 * cur.cg) * p.CO) * 4u;
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = odin_run_load1s(RB, bl | dead, so + (unsigned)(e * p.CO * 4));
      }
      // advance (wave-uniform)
      cur.cg += 1;
      if (cur.cg == gpt) {
        cur.cg = 0; cur.t += 1; cur.c += 1;
        if (cur.c == nkw) { cur.c = 0; cur.a += 1; }
      }
    };
    auto mul = [&](const float (&av)[8], const float (&bv)[8]) {
      u32x4 ah, al, bh, bl2;
      if (SC && sc_on) ih_split8<true>(av, in_s, in_s2k, ah, al); else ih_split8<false>(av, 1.f, ODIN_LO_SCALE, ah, al);
      ih_split8<false>(bv, 1.f, ODIN_LO_SCALE, bh, bl2);
      acx = mfma32_f16(ah, bl2, acx);
      acc = mfma32_f16(ah, bh, acc);
      acx = mfma32_f16(al, bh, acx);
    };
    const int ngroups = ntap * gpt;
    load_group(a0, b0);
    for (int g = 0;;) {
      load_group(a1, b1);
      ODIN_SCHED_FENCE();
      mul(a0, b0);
      ODIN_SCHED_FENCE();
      g += 2;
      if (g - 1 >= ngroups) break;
      load_group(a0, b0);
      ODIN_SCHED_FENCE();
      mul(a1, b1);
      ODIN_SCHED_FENCE();
      if (g >= ngroups) break;
    }
    // the epilogue's store offsets and act'(aux) factors: fetched here, when the operand batches are dead (32 registers
    // less through the main loop; the round trip is covered by the other waves of the SIMD)
    unsigned ooff[16];
    float auxv[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int po = rowoff[wave][(rr & 3) + 8 * (rr >> 2) + 4 * h];
      const unsigned ok = (unsigned)b_ok & (((unsigned)po >> 31) ^ 1u);
      ooff[rr] = (unsigned)((po * p.CO + j) * 4) | (ok - 1u);
    }
    if (has_aux) {   // (one uniform branch around the 16 loads, not one per load)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) auxv[rr] = odin_run_load1(RX, ooff[rr]);
    }
    // ---- epilogue: lane holds column j, rows (rr & 3) + 8 (rr >> 2) + 4 h ----
    // (the activation codes are wave-uniform: one switch around the 16-element loop instead of branches inside it)
    auto epilogue = [&](auto ACT_, auto AUX_) {
      constexpr int A_ = decltype(ACT_)::value, X_ = decltype(AUX_)::value;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        float v = odin_act(A_, fmaf(acx[rr], o_sx, acc[rr] * o_s) + bj);
        if (X_ != 0) v *= odin_act_grad(X_, auxv[rr]);
        odin_run_store1(RO, ooff[rr], v);           // range-checked: nothing is written for masked rows
        const float vv = ((int)ooff[rr] >= 0) ? v : 0.f;
        csum += vv;
        amx = fmaxf(amx, fabsf(vv));
      }
    };
    using IH_I0 = std::integral_constant<int, 0>;
    using IH_I1 = std::integral_constant<int, ODIN_ACT_ELU>;
    using IH_I2 = std::integral_constant<int, ODIN_ACT_RELU>;
    if (has_aux) {   // data gradient: linear output, act'(aux)
      if (p.aux_act == ODIN_ACT_ELU) epilogue(IH_I0{}, IH_I1{});
      else epilogue(IH_I0{}, IH_I2{});
    } else if (p.act == ODIN_ACT_ELU) epilogue(IH_I1{}, IH_I0{});
    else if (p.act == ODIN_ACT_RELU) epilogue(IH_I2{}, IH_I0{});
    else epilogue(IH_I0{}, IH_I0{});
  }
  if (p.colsum != nullptr) {
    csum += __shfl_xor(csum, 32);
    if (h == 0) cred[wave * 32 + l31] = csum;
    __syncthreads();
    if (wave == 0 && h == 0 && b_ok) {
      float t = 0.f;
      for (int w = 0; w < IH_NW; ++w) t += cred[w * 32 + l31];
      p.colsum[(size_t)blockIdx.y * p.CO + j] = t;
    }
  }
  if (p.out_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, IH_NW * 64, cred, blockIdx.x + gridDim.x * blockIdx.y);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same with the WEIGHT PLANES IN LDS.  A workgroup serves one stride class and one 32-column block: it splits that
// slice of the weights ONCE (<= 32 steps of 16 k-values: 2 KB per step, both planes, already in the MFMA B layout:
// [step][plane][half][column] x 16 bytes) and its waves then walk the tiles of the class reading B with two
// ds_read_b128 per step -- on the global-memory variant every tile re-fetched and re-split its 4-32 KB of weights
// through the vector memory pipe, which the A gathers need (audio decoder4: 270 -> see DESIGN 3.8c).
// ALLC (transposed gathers whose four stride classes fit in LDS together): a wave runs ALL classes of its 32 coarse
// pixels back to back -- their 16 gathers fall on the same 3 x 3 input neighbourhood, so the input is read from HBM
// once; with one class per workgroup the classes of a pixel ran on different XCDs at different times and the audio
// decoder4's 63 MB input crossed the fabric ~16 times (196 us = 1 GB at ~5 TB/s).
template <bool TMODE, bool BKC, bool SC, bool ALLC, int IH_U, int NW>
__device__ __forceinline__ void igemm_hw_body(const IHParams& p) {
  ODIN_DYN_SMEM(char, wlds);   // [nsteps][2][2][32] x 16 B
  __shared__ int rowoff[NW][32];
  __shared__ float cred[NW * 32 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = IH_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int j = blockIdx.x * 32 + l31;
  const bool b_ok = j < p.CO;
  // (the input's range word: requested here, finished behind the weight staging -- odin_device.h: odin_range_issue)
  const OdinRangeReq in_rq = odin_range_issue(SC ? p.in_amax : nullptr, lane);
  const float bj = (p.bias != nullptr && b_ok) ? p.bias[j] : 0.f;
  const OdinRun RA = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun RB = odin_run(p.w, (unsigned)((size_t)p.KH * p.KW * p.CI * p.CO * 4));
  const OdinRun RO = odin_run(p.out, (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4));
  const bool has_aux = p.aux != nullptr;
  const OdinRun RX = odin_run(has_aux ? p.aux : p.in, has_aux ? (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4) : 0u);
  const int SS = TMODE ? p.S : 1;     // 1 or 2
  const int ssh = SS >> 1;
  const int ncls = SS * SS;
  const int ohs = p.OH >> ssh, ows = p.OW >> ssh;
  const unsigned img = (unsigned)(ohs * ows);
  const int sg = TMODE ? -1 : 1;
  const int gpt = p.gpt;
  // ---- this workgroup's stride class(es) and their taps ----
  const int cls_lo = ALLC ? 0 : (int)blockIdx.y % ncls, cls_hi = ALLC ? ncls : cls_lo + 1;
  const int slot = ALLC ? (int)blockIdx.y : (int)blockIdx.y / ncls;
  const int nslot = ALLC ? (int)gridDim.y : (int)gridDim.y / ncls;
  auto class_taps = [&](int cls, int& kh0, int& kw0, int& nkh, int& nkw) {
    const int cy = cls >> ssh, cx = cls - (cy << ssh);
    kh0 = TMODE ? (cy + p.pt) & (SS - 1) : 0;
    kw0 = TMODE ? (cx + p.pl) & (SS - 1) : 0;
    nkh = (p.KH - kh0 + SS - 1) >> ssh;
    nkw = (p.KW - kw0 + SS - 1) >> ssh;
  };
  // first step of class c in LDS (prefix sums of taps x channel groups)
  int sbase[5];
  sbase[cls_lo] = 0;
  for (int c = cls_lo; c < cls_hi; ++c) {
    int kh0, kw0, nkh, nkw;
    class_taps(c, kh0, kw0, nkh, nkw);
    sbase[c + 1] = sbase[c] + nkh * nkw * gpt;
  }
  // ---- the weight planes of (class, column block): entry (step s, half hh, column jl) <- 8 weights ----
  for (int e = tid; e < sbase[cls_hi] * 64; e += NW * 64) {
    const int s = e >> 6, hh = (e >> 5) & 1, jl = e & 31;
    int wc = cls_lo;
    while (wc + 1 < cls_hi && s >= sbase[wc + 1]) ++wc;
    int kh0, kw0, nkh, nkw;
    class_taps(wc, kh0, kw0, nkh, nkw);
    const int sl = s - sbase[wc];
    const int t = sl / gpt, cg = sl - t * gpt;
    const int a = t / nkw, c = t - a * nkw;
    const int wt = (kh0 + a * SS) * p.KW + kw0 + c * SS;
    const int jj = blockIdx.x * 32 + jl;
    float wv[8];
    if (BKC) {
      const unsigned off = jj < p.CO ? (unsigned)(((wt * p.CO + jj) * p.CI + 16 * cg + 8 * hh) * 4) : ODIN_OOB;
      const float4 y0 = odin_run_load4(RB, off), y1 = odin_run_load4(RB, off == ODIN_OOB ? ODIN_OOB : off + 16u);
      wv[0] = y0.x; wv[1] = y0.y; wv[2] = y0.z; wv[3] = y0.w; wv[4] = y1.x; wv[5] = y1.y; wv[6] = y1.z; wv[7] = y1.w;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        wv[q] = odin_run_load1(RB, jj < p.CO ? (unsigned)((((wt * p.CI + 16 * cg + 8 * hh + q) * p.CO) + jj) * 4) : ODIN_OOB);
    }
    u32x4 wh, wl;
    ih_split8<false>(wv, 1.f, ODIN_LO_SCALE, wh, wl);
    *reinterpret_cast<u32x4*>(wlds + ((((s * 2 + 0) * 2 + hh) * 32 + jl) << 4)) = wh;
    *reinterpret_cast<u32x4*>(wlds + ((((s * 2 + 1) * 2 + hh) * 32 + jl) << 4)) = wl;
  }
  __syncthreads();
  const unsigned in_mb = SC ? odin_range_finish(in_rq) : 0u;
  const bool sc_on = SC && (!p.in_cond || odin_act_needs_scale(in_mb));
  const int gk = sc_on ? odin_range_shift(in_mb) : 0;
  const float in_s = odin_pow2(gk), in_s2k = odin_pow2(gk + 11);
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11);
  const unsigned dbg_a = (p.dbg & 1u) ? 0xFFFFFFFFu : 0u, dbg_s = (p.dbg & 2u) ? 0xFFFFFFFFu : 0u;
  const char* wme0 = wlds + ((h * 32 + l31) << 4);   // this lane's entry of step 0, plane 0; + 2048 per step, + 1024 lo plane
  float csum = 0.f, amx = 0.f;
  for (int tile = IH_UNIFORM(slot * NW + wave); tile < p.tpc; tile += nslot * NW) {
    // ---- this lane's A row: one (coarse) output pixel ----
    const unsigned q = (unsigned)(tile * 32 + l31);
    const bool a_ok = q < (unsigned)p.Mc;
    const unsigned b = q / img, r = q - b * img;
    const unsigned ys = r / (unsigned)ows, xs = r - ys * (unsigned)ows;
   for (int cls = cls_lo; cls < cls_hi; ++cls) {
    const int cy = cls >> ssh, cx = cls - (cy << ssh);
    int kh0, kw0, nkh, nkw;
    class_taps(cls, kh0, kw0, nkh, nkw);
    const int ntap = nkh * nkw;
    const int ngroups = ntap * gpt;
    const char* wme = wme0 + (sbase[cls] << 11);
    const int oy = (int)ys * SS + cy, ox = (int)xs * SS + cx;
    const int Y0 = TMODE ? (oy + p.pt - kh0) >> ssh : oy * p.S - p.pt;
    const int X0 = TMODE ? (ox + p.pl - kw0) >> ssh : ox * p.S - p.pl;
    unsigned mask = 0;
    {
      int t = 0;
      for (int a = 0; a < nkh; ++a)
        for (int c = 0; c < nkw; ++c, ++t) {
          const unsigned in = ((unsigned)(Y0 + sg * a) < (unsigned)p.H) & ((unsigned)(X0 + sg * c) < (unsigned)p.W);
          mask |= in << t;
        }
    }
    mask = a_ok ? mask : 0u;
    const int lanebase = (((int)b * p.H + Y0) * p.W + X0) * p.CI + 8 * h;  // floats
    odin_wave_sync();  // (the previous tile's epilogue has read the table)
    if (h == 0) rowoff[wave][l31] = a_ok ? (((int)b * p.OH + oy) * p.OW + ox) : -1;
    odin_wave_sync();
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
    // batches of IH_U steps, double-buffered: the gathers of the next four steps are in flight while the current four
    // multiply (one step ahead left an L2 round trip exposed at every step: two waves per SIMD cannot cover it)
    float a0[IH_U][8], a1[IH_U][8];
    IHCursor cur = {0, 0, 0, 0};
    auto load_a = [&](float (&av)[8]) {
      const unsigned live = (unsigned)(cur.t - ntap) >> 31;   // 1 / 0 (wave-uniform)
      const int tapoff = sg * (cur.a * p.W + cur.c) * p.CI;
      const unsigned va = live & (mask >> (cur.t & 31)) & 1u;
      const unsigned ao = (unsigned)((lanebase + tapoff + 16 * cur.cg) * 4);
      const float4 x0 = odin_run_load4(RA, ao | (va - 1u) | dbg_a), x1 = odin_run_load4(RA, (ao + 16u) | (va - 1u) | dbg_a);
      av[0] = x0.x; av[1] = x0.y; av[2] = x0.z; av[3] = x0.w; av[4] = x1.x; av[5] = x1.y; av[6] = x1.z; av[7] = x1.w;
      cur.cg += 1;
      if (cur.cg == gpt) {
        cur.cg = 0; cur.t += 1; cur.c += 1;
        if (cur.c == nkw) { cur.c = 0; cur.a += 1; }
      }
    };
    auto mul = [&](const float (&av)[8], int s) {
      // (a step beyond the reduction carries zeros in A: any valid weight step will do)
      const int sc = s < ngroups ? s : ngroups - 1;
      u32x4 ah, al;
      if (SC && sc_on) ih_split8<true>(av, in_s, in_s2k, ah, al); else ih_split8<false>(av, 1.f, ODIN_LO_SCALE, ah, al);
      const u32x4 bh = *reinterpret_cast<const u32x4*>(wme + (sc << 11));
      const u32x4 bl2 = *reinterpret_cast<const u32x4*>(wme + (sc << 11) + 1024);
      acx = mfma32_f16(ah, bl2, acx);
      acc = mfma32_f16(ah, bh, acc);
      acx = mfma32_f16(al, bh, acx);
    };
    auto mul_load = [&](const float (&av)[IH_U][8], int g, float (&nav)[IH_U][8]) {
#pragma unroll
      for (int u = 0; u < IH_U; ++u) {
        mul(av[u], g + u);
        ODIN_SCHED_FENCE();
        load_a(nav[u]);
        ODIN_SCHED_FENCE();
      }
    };
#pragma unroll
    for (int u = 0; u < IH_U; ++u) load_a(a0[u]);
    for (int g = 0;;) {
      mul_load(a0, g, a1);
      g += IH_U;
      if (g >= ngroups) break;
      mul_load(a1, g, a0);
      g += IH_U;
      if (g >= ngroups) break;
    }
    // the epilogue's store offsets and act'(aux) factors: fetched HERE, when the gather batches are dead (their registers
    // are reused) -- before the main loop they cost 32 live registers through it, and the round trip they hid is
    // covered by the other waves of the SIMD
    unsigned ooff[16];
    float auxv[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int po = rowoff[wave][(rr & 3) + 8 * (rr >> 2) + 4 * h];
      const unsigned ok = (unsigned)b_ok & (((unsigned)po >> 31) ^ 1u);
      ooff[rr] = (unsigned)((po * p.CO + j) * 4) | (ok - 1u);
    }
    if (has_aux) {   // (one uniform branch around the 16 loads, not one per load)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) auxv[rr] = odin_run_load1(RX, ooff[rr]);
    }
    // (the activation codes are wave-uniform: one switch around the 16-element loop instead of branches inside it)
    auto epilogue = [&](auto ACT_, auto AUX_) {
      constexpr int A_ = decltype(ACT_)::value, X_ = decltype(AUX_)::value;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        float v = odin_act(A_, fmaf(acx[rr], o_sx, acc[rr] * o_s) + bj);
        if (X_ != 0) v *= odin_act_grad(X_, auxv[rr]);
        odin_run_store1(RO, ooff[rr] | dbg_s, v);
        const float vv = ((int)ooff[rr] >= 0) ? v : 0.f;
        csum += vv;
        amx = fmaxf(amx, fabsf(vv));
      }
    };
    using IH_I0 = std::integral_constant<int, 0>;
    using IH_I1 = std::integral_constant<int, ODIN_ACT_ELU>;
    using IH_I2 = std::integral_constant<int, ODIN_ACT_RELU>;
    if (has_aux) {   // data gradient: linear output, act'(aux)
      if (p.aux_act == ODIN_ACT_ELU) epilogue(IH_I0{}, IH_I1{});
      else epilogue(IH_I0{}, IH_I2{});
    } else if (p.act == ODIN_ACT_ELU) epilogue(IH_I1{}, IH_I0{});
    else if (p.act == ODIN_ACT_RELU) epilogue(IH_I2{}, IH_I0{});
    else epilogue(IH_I0{}, IH_I0{});
   }  // classes
  }
  if (p.colsum != nullptr) {
    csum += __shfl_xor(csum, 32);
    if (h == 0) cred[wave * 32 + l31] = csum;
    __syncthreads();
    if (wave == 0 && h == 0 && b_ok) {
      float t = 0.f;
      for (int w = 0; w < NW; ++w) t += cred[w * 32 + l31];
      p.colsum[(size_t)blockIdx.y * p.CO + j] = t;
    }
  }
  if (p.out_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, NW * 64, cred, blockIdx.x + gridDim.x * blockIdx.y);
  }
}

// The transposed gathers keep 16-32 KB of weight planes per workgroup: four waves per SIMD fit, and a tile there is only
// 8-16 steps behind a row decode and a first batch of gathers whose latency only other waves can cover (forced to 128
// registers: audio decoder4 forward 168 -> 152 us, decoder3 79 -> 74).  The strided gathers (16 taps: 64 KB of planes,
// two workgroups per CU whatever the register count) keep their 4-step batches and two waves per SIMD (forced to 128
// registers they spill: decoder4 data gradient 152 -> 197 us).
template <bool ALLC>
__global__ __launch_bounds__(IH_NW * 64) IH_WAVES_ATTR void igemm_hw_t_kernel(IHParams p) {
  igemm_hw_body<true, true, false, ALLC, 2, IH_NW>(p);
}
// (a gradient input adds the scale multiplies: at 128 registers it spills -- conv data gradients 58 -> 69 us)
template <bool ALLC>
__global__ __launch_bounds__(IH_NW * 64) void igemm_hw_tg_kernel(IHParams p) {
  igemm_hw_body<true, true, true, ALLC, 4, IH_NW>(p);
}
// (forward: eight waves share the 64 KB of planes of the strided gathers -- two workgroups per CU = four waves per
// SIMD: audio encoder1 / 2 forward 42.5 / 22.8 -> 37.7 / 21.7 us; the data gradients of the transposed layers were
// slower that way, 155 -> 172 us, and keep four waves per workgroup)
constexpr int IH_NWF = 8;
__global__ __launch_bounds__(IH_NWF * 64) void igemm_hw_f_kernel(IHParams p) {
  igemm_hw_body<false, false, false, false, 4, IH_NWF>(p);
}
__global__ __launch_bounds__(IH_NW * 64) void igemm_hw_fg_kernel(IHParams p) {
  igemm_hw_body<false, false, true, false, 4, IH_NW>(p);
}

template <typename K>
int ih_set_lds(K kern, size_t bytes) {
#ifndef ODIN_SIM
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess)
    return odin_fail(-4, "igemm_h: cannot raise the dynamic LDS limit");
#else
  (void)kern; (void)bytes;
#endif
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
struct IHWParams {
  const float* u;   // fine tensor   [B, FH, FW, CU]
  const float* v;   // coarse tensor [B, h, w, CV]
  float* slab;      // [gridDim.z][slab_stride]: (dW [KH*KW*CU][CV] | column sums of V [CV])
  const unsigned* g_amax;  // range word of the gradient operand (SCU: u, SCV: v)
  const unsigned* a_amax;  // AS instances: range word of the other (activation) operand
  int slab_stride;
  int B, FH, FW, CU, h, w, CV, KH, KW, S, pt, pl;
  int M;            // B * h * w
  int chunk;        // pixels per workgroup (multiple of 16, <= IHW_CHUNK)
  int want_bias;
};

constexpr int IHW_CHUNK = 2048;

// NW waves split the 16-pixel steps of a workgroup's chunk; SCU / SCV: that operand is a gradient
// AS: the activation operand comes with its range word and is carried times its own power of two
template <int NW, bool SCU, bool SCV, bool AS = false>
__global__ __launch_bounds__(NW * 64) void igemm_h_wgrad_kernel(IHWParams p) {
  __shared__ float red[NW > 1 ? (NW - 1) * 16 * 64 : 64];
  // per coarse pixel m of this workgroup: byte offset of the fine pixel (y S - pt, x S - pl), and the bit mask of the
  // taps (kh KW + kw) whose fine pixel lies inside the image (0 beyond the workgroup's pixels) -- a lane's validity
  // test is one bit extract, eight consecutive entries arrive with two ds_read_b128
  constexpr int TBN = IHW_CHUNK + 16 * (2 * NW + 1);
  __shared__ __attribute__((aligned(16))) int tb_off[TBN];
  __shared__ __attribute__((aligned(16))) unsigned tb_msk[TBN];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = IH_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int I = p.KH * p.KW * p.CU;
  const int i = blockIdx.y * 32 + l31, j = blockIdx.x * 32 + l31;
  const bool i_ok = i < I, j_ok = j < p.CV;
  const int tap = i / p.CU, cu = i - tap * p.CU;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int rowc4 = ((kh * p.FW + kw) * p.CU + cu) * 4;
  const unsigned tapbit = i_ok ? (unsigned)tap : 31u;   // (bit 31 of a mask is never set)
  const int mlo = blockIdx.z * p.chunk;
  const int mhi = (mlo + p.chunk < p.M) ? mlo + p.chunk : p.M;
  // (the two range words: requested here, finished behind the table build -- odin_device.h: odin_range_issue)
  const OdinRangeReq g_rq = odin_range_issue((SCU || SCV) ? p.g_amax : nullptr, lane);
  const OdinRangeReq a_rq = odin_range_issue(AS ? p.a_amax : nullptr, lane);
  for (int e = tid; e < TBN; e += NW * 64) {
    const int m = mlo + e;
    int off = 0;
    unsigned msk = 0u;
    if (m < mhi && e < p.chunk) {
      const int b = m / (p.h * p.w), r = m - b * (p.h * p.w), y = r / p.w, x = r - y * p.w;
      const int fy = y * p.S - p.pt, fx = x * p.S - p.pl;
      off = ((b * p.FH + fy) * p.FW + fx) * p.CU * 4;
      // taps kw with 0 <= fx + kw < FW as a bit run, repeated for the rows kh with 0 <= fy + kh < FH
      const int wlo = fx < 0 ? -fx : 0, whi = (p.FW - fx < p.KW) ? p.FW - fx : p.KW;
      const unsigned run = whi > wlo ? ((1u << whi) - 1u) & ~((1u << wlo) - 1u) : 0u;
      for (int a = 0; a < p.KH; ++a)
        if ((unsigned)(fy + a) < (unsigned)p.FH) msk |= run << (a * p.KW);
    }
    tb_off[e] = off;
    tb_msk[e] = msk;
  }
  __syncthreads();
  const OdinRun RU = odin_run(p.u, (unsigned)((size_t)p.B * p.FH * p.FW * p.CU * 4));
  const OdinRun RV = odin_run(p.v, (unsigned)((size_t)mhi * p.CV * 4));   // (rows beyond this workgroup's pixels read zeros)
  const int gk = (SCU || SCV) ? odin_range_shift(odin_range_finish(g_rq)) : 0;
  const float g_s = (SCU || SCV) ? odin_pow2(gk) : 1.f, g_s2k = (SCU || SCV) ? odin_pow2(gk + 11) : ODIN_LO_SCALE;
  const unsigned a_mb = AS ? odin_range_finish(a_rq) : 0u;
  const bool a_on = AS && odin_act_needs_scale(a_mb);   // (an activation is scaled only outside the safe window)
  const int ak = a_on ? odin_range_shift(a_mb) : 0;
  const float a_s = odin_pow2(ak), a_s2k = odin_pow2(ak + 11);
  const int nsteps = (mhi - mlo + 15) >> 4;
  f32x16 acc = f32x16_zero(), acx = f32x16_zero();
  float csum = 0.f;
  float a0[8], b0[8], a1[8], b1[8];
  const unsigned vb0 = j_ok ? (unsigned)(((mlo + 8 * h) * p.CV + j) * 4) : ODIN_OOB;
  const unsigned vstep = (unsigned)(16 * p.CV * 4), vrow = (unsigned)(p.CV * 4);
  const bool do_bias = p.want_bias && blockIdx.y == 0;
  auto load_step = [&](int s, float (&av)[8], float (&bv)[8]) {
    // the 8 pixels 16 s + 8 h .. + 7 of this lane's half (steps beyond the chunk: mask 0 / rows beyond mhi)
    const int4 o0 = *reinterpret_cast<const int4*>(&tb_off[16 * s + 8 * h]);
    const int4 o1 = *reinterpret_cast<const int4*>(&tb_off[16 * s + 8 * h + 4]);
    const uint4 m0 = *reinterpret_cast<const uint4*>(&tb_msk[16 * s + 8 * h]);
    const uint4 m1 = *reinterpret_cast<const uint4*>(&tb_msk[16 * s + 8 * h + 4]);
    const int o[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
    const unsigned mk[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
    const unsigned vb = vb0 + (unsigned)s * vstep;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      av[e] = odin_run_load1(RU, ((mk[e] >> tapbit) & 1u) ? (unsigned)(o[e] + rowc4) : ODIN_OOB);
      bv[e] = odin_run_load1(RV, vb + (unsigned)e * vrow);
    }
  };
  auto mul = [&](const float (&av)[8], const float (&bv)[8]) {
    u32x4 ah, al, bh, bl;
    if (SCU) ih_split8<true>(av, g_s, g_s2k, ah, al);
    else if (AS && a_on) ih_split8<true>(av, a_s, a_s2k, ah, al);
    else ih_split8<false>(av, 1.f, ODIN_LO_SCALE, ah, al);
    if (SCV) ih_split8<true>(bv, g_s, g_s2k, bh, bl);
    else if (AS && a_on) ih_split8<true>(bv, a_s, a_s2k, bh, bl);
    else ih_split8<false>(bv, 1.f, ODIN_LO_SCALE, bh, bl);
    acx = mfma32_f16(ah, bl, acx);
    acc = mfma32_f16(ah, bh, acc);
    acx = mfma32_f16(al, bh, acx);
    if (do_bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) csum += bv[e];
    }
  };
  int s = wave;
  if (s < nsteps) {
    load_step(s, a0, b0);
    for (;;) {
      load_step(s + NW, a1, b1);   // (beyond the chunk: mask 0 / rows beyond mhi: zeros, no traffic)
      ODIN_SCHED_FENCE();
      mul(a0, b0);
      ODIN_SCHED_FENCE();
      s += 2 * NW;
      if (s - NW >= nsteps) break;
      load_step(s, a0, b0);
      ODIN_SCHED_FENCE();
      mul(a1, b1);
      ODIN_SCHED_FENCE();
      if (s >= nsteps) break;
    }
  }
  // ---- combine the NW partial tiles in wave order (main + 2^-11 cross, scaled back) ----
  const float o_s = (SCU || SCV) ? odin_pow2(-gk) : 1.f, o_sx = (SCU || SCV) ? odin_pow2(-gk - 11) : ODIN_LO_UNSCALE;
  f32x16 tot;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) tot[rr] = fmaf(acx[rr], o_sx, acc[rr] * o_s);
  if (AS) {   // (the two scales are taken back one after the other: their sum may leave one factor's exponent range)
    const float a_o = odin_pow2(-ak);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) tot[rr] *= a_o;
  }
  if (NW > 1) {
    if (wave > 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) red[((wave - 1) * 16 + rr) * 64 + lane] = tot[rr];
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < NW; ++w) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) tot[rr] += red[((w - 1) * 16 + rr) * 64 + lane];
      }
    }
  }
  float* row = p.slab + (size_t)blockIdx.z * p.slab_stride;
  if (p.want_bias && blockIdx.y == 0) {
    // column sums of V over this workgroup's pixels: halves h = 0 / 1 and the NW waves (raw fp32 values)
    __syncthreads();
    const float t = csum + __shfl_xor(csum, 32);
    if (h == 0) red[wave * 32 + l31] = t;
    __syncthreads();
    if (wave == 0 && h == 0 && j_ok) {
      float sm = 0.f;
      for (int w = 0; w < NW; ++w) sm += red[w * 32 + l31];
      row[(size_t)I * p.CV + j] = sm;
    }
  }
  if (wave != 0 || !j_ok) return;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int ii = blockIdx.y * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
    if (ii < I) row[(size_t)ii * p.CV + j] = tot[rr];
  }
}

bool ih_enabled() { return !odin_exact_fp32() && ODIN_DIAG_ENV("ODIN_NOIGEMMH") == nullptr; }

// Below ~1.2 GFLOP per launch the fp32 implicit GEMM (igemm.hip) wins: its workgroups split the reduction of a tile over
// their waves and the weight and data gradient of a layer share one launch (same-box A/B over the six workloads: the
// dSprites step 0.544 ms with the small layers (0.5 GFLOP) there, 0.595 ms with them here; the audio stack's layers of
// 2-16 GFLOP are 10-60 % faster here).  1.2 GFLOP is also where igemm.hip stops taking the transposed gathers: between
// the two limits the 5x5/s2 layers of the MNIST stack (1.29 GFLOP at batch 128) fell to the generic fp32 kernels
// (encoder3's data gradient: 226 us).
double g_ih_min_flop = 1.2e9;

}  // namespace

static int g_ih_ldsw_steps = 32;
// diagnostics: the largest weight slice (16-value steps of one stride class and 32 columns: 2 KB each) kept in LDS;
// returns the previous value (negative: only read)
extern "C" int odin_debug_igemm_h_ldsw_steps(int steps) {
  const int old = g_ih_ldsw_steps;
  if (steps >= 0) g_ih_ldsw_steps = steps;
  return old;
}

// tests: the launch size from which convolutions come here (0: every applicable shape); returns the previous value
extern "C" double odin_debug_igemm_h_min_flop(double flop) {
  const double old = g_ih_min_flop;
  if (flop >= 0.0) g_ih_min_flop = flop;
  return old;
}

// forward / data gradient: 16-channel steps, at least 128 tiles (fewer: the fp32 kernel splits the reduction of a tile
// over the waves of a workgroup, igemm.hip)
bool odin_igemm_h_applicable(int tmode, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                             int center) {
  if (!ih_enabled() || center) return false;
  if (CI < 16 || (CI & 15) != 0 || CI > 8192 || KH * KW > 25 || KH < 1 || KW < 1 || KH > 8 || KW > 8 || S < 1 || S > 2)
    return false;
  if (tmode && (OH % S || OW % S || KH < S || KW < S)) return false;
  const int SS = tmode ? S : 1;
  const long Mc = (long)B * (OH / SS) * (OW / SS);
  const long tiles = (long)SS * SS * ((Mc + 31) / 32) * ((CO + 31) / 32);
  if (tiles < 128 && g_ih_min_flop > 0.0) return false;   // (tests force small shapes here: min flop 0)
  const double flop = 2.0 * B * (tmode ? (double)H * W : (double)OH * OW) * KH * KW * CI * CO;
  if (flop < g_ih_min_flop) return false;
  // strided gathers whose weight planes do not fit in LDS (> 32 steps: 5x5 kernels, 64 channels x 16 taps) run with both
  // operands from L2: below 2 GFLOP the fp32 kernel, which takes strided gathers up to 5 GFLOP, is faster (MNIST
  // encoder3 forward, 1.29 GFLOP: 22.6 us there, 34.7 us here)
  if (!tmode && KH * KW * (CI / 16) > 32 && flop < 2.0e9 && g_ih_min_flop > 0.0) return false;
  if ((long)B * H * W * CI >= (1L << 29) || (long)B * OH * OW * CO >= (1L << 29) || (long)KH * KW * CI * CO >= (1L << 29))
    return false;
  return true;
}

// rows of column sums the data-gradient launch writes (= its gridDim.y)
int odin_igemm_h_rows(int tmode, int B, int OH, int OW, int S) {
  const int SS = tmode ? S : 1;
  const long Mc = (long)B * (OH / SS) * (OW / SS);
  const long ntile = (long)SS * SS * ((Mc + 31) / 32);
  long gy = (ntile + IH_NW - 1) / IH_NW;
  if (gy > ODIN_MAX_COLSUM_BLOCKS) gy = ODIN_MAX_COLSUM_BLOCKS;
  gy = (gy + SS * SS - 1) / (SS * SS) * (SS * SS);   // (whole stride classes: igemm_hw_kernel)
  if (gy > ODIN_MAX_COLSUM_BLOCKS) gy -= SS * SS;
  return (int)gy;
}

int odin_igemm_h_launch(int tmode, const float* in, const float* w, const float* bias, const float* aux, int aux_act,
                        float* out, float* colsum, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                        int S, int pt, int pl, int act, const uint32_t* in_amax, int in_is_grad, uint32_t* out_amax,
                        void* stream) {
  IHParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.out = out; p.bias = bias; p.out_amax = out_amax;
  p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act; p.colsum = colsum;
  p.B = B; p.H = H; p.W = W; p.CI = CI; p.OH = OH; p.OW = OW; p.CO = CO;
  p.KH = KH; p.KW = KW; p.S = S; p.pt = pt; p.pl = pl; p.gpt = CI / 16; p.act = act;
  const int SS = tmode ? S : 1;
  p.Mc = B * (OH / SS) * (OW / SS);
  p.tpc = (p.Mc + 31) / 32;
  p.ntile = SS * SS * p.tpc;
  if (const char* e = ODIN_DIAG_ENV("ODIN_IH_DBG")) p.dbg = (unsigned)atoi(e);
  if (in_is_grad) {
    p.in_amax = odin_range_word_of(in, (size_t)B * H * W * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "igemm_h: no range word for the gradient input");
  } else if (in_amax != nullptr) {
    // an ACTIVATION that comes with its range word (odin_conv_desc.x_amax) takes the scaled instances too: any
    // magnitude keeps its 22 bits; without a word the input is carried unscaled (|x| <= 65504)
    p.in_amax = in_amax;
    p.in_cond = 1;
    in_is_grad = 1;
  }
  // every wave walks tiles grid-stride: the whole chip in one wave of workgroups for the big layers (and the column
  // sums stay within ODIN_MAX_COLSUM_BLOCKS rows)
  int gy = (p.ntile + IH_NW - 1) / IH_NW;
  const int gx = (CO + 31) / 32;
  int cap = ODIN_MAX_COLSUM_BLOCKS;
  if (colsum == nullptr) {
    cap = (8 * odin_num_cus() + gx - 1) / gx;   // ~8 workgroups (32 waves) per CU
    if (cap < 1) cap = 1;
  }
  if (gy > cap) gy = cap;
  const int ncls = SS * SS;
  gy = (gy + ncls - 1) / ncls * ncls;             // whole stride classes
  if (gy > cap && gy > ncls) gy -= ncls;
  dim3 grid(gx, gy, 1);
  // weight planes of one (class, column block) in LDS when they fit in 64 KB (2 KB per 16-value step)
  const int max_steps = ((KH + SS - 1) / SS) * ((KW + SS - 1) / SS) * p.gpt;
  // (round 6, measured and dropped: up to 64 steps -- 128 KB, one workgroup per CU -- so that the 5 x 5 layers over 32
  // channels of the MNIST stack (50 steps) keep their weight planes in LDS too: MNIST conv 0.990 -> 1.005 ms, CelebA
  // 1.165 -> 1.184 (tools/r06_ldsw_ab.py, odin_debug_igemm_h_ldsw_steps): the occupancy lost costs more than the L2
  // reads of the weights)
  if (max_steps <= g_ih_ldsw_steps && !ODIN_DIAG_ENV("ODIN_IH_NOLDSW")) {
    const size_t lds = (size_t)max_steps * 2048;
    // (eight tiles per workgroup there: half the rows when no column sums are asked for; with column sums the row
    // count odin_igemm_h_rows promised stays -- surplus workgroups write zero rows)
    dim3 gridf = grid;
    if (colsum == nullptr && gridf.y > 1) gridf.y = (gridf.y + 1) / 2;
#define ODIN_IHW_T(S_, A_, LDS_)                                                                  \
  do {                                                                                            \
    if (S_) {                                                                                     \
      if (int rc = ih_set_lds(&igemm_hw_tg_kernel<A_>, LDS_)) return rc;                          \
      ODIN_LAUNCH((igemm_hw_tg_kernel<A_>), grid, dim3(IH_NW * 64), LDS_, stream, p);             \
    } else {                                                                                      \
      if (int rc = ih_set_lds(&igemm_hw_t_kernel<A_>, LDS_)) return rc;                           \
      ODIN_LAUNCH((igemm_hw_t_kernel<A_>), grid, dim3(IH_NW * 64), LDS_, stream, p);              \
    }                                                                                             \
  } while (0)
#define ODIN_IHW_F(S_, LDS_)                                                                      \
  do {                                                                                            \
    if (S_) {                                                                                     \
      if (int rc = ih_set_lds(&igemm_hw_fg_kernel, LDS_)) return rc;                              \
      ODIN_LAUNCH(igemm_hw_fg_kernel, grid, dim3(IH_NW * 64), LDS_, stream, p);                   \
    } else {                                                                                      \
      if (int rc = ih_set_lds(&igemm_hw_f_kernel, LDS_)) return rc;                               \
      ODIN_LAUNCH(igemm_hw_f_kernel, gridf, dim3(IH_NWF * 64), LDS_, stream, p);                  \
    }                                                                                             \
  } while (0)
    // (measured on the audio stack: decoder4 forward 196 -> 212 us, encoder2 data gradient 70 -> 86 -- the kernel is
    // bound by instruction issue, not by the re-read of its input (no gathers and no stores at all: 139 us); the
    // instances exist in the diagnostics build only)
#ifdef ODIN_DIAG
    const int all_steps = KH * KW * p.gpt;
    if (tmode && SS > 1 && all_steps <= 32 && ODIN_DIAG_ENV("ODIN_IH_ALLC")) {
      const size_t lds_all = (size_t)all_steps * 2048;
      if (in_is_grad) ODIN_IHW_T(true, true, lds_all); else ODIN_IHW_T(false, true, lds_all);
    } else
#endif
    if (tmode) {
      if (in_is_grad) ODIN_IHW_T(true, false, lds); else ODIN_IHW_T(false, false, lds);
    } else {
      if (in_is_grad) ODIN_IHW_F(true, lds); else ODIN_IHW_F(false, lds);
    }
#undef ODIN_IHW_T
#undef ODIN_IHW_F
    return odin_check_launch("igemm_h(f16x2)");
  }
#define ODIN_IH(T_, K_, S_) ODIN_LAUNCH((igemm_h_kernel<T_, K_, S_>), grid, dim3(IH_NW * 64), 0, stream, p)
  // Conv2D forward / Conv2DTranspose data gradient: weights [tap][k][j]; the transposed gathers: [tap][j][k]
  if (tmode) { if (in_is_grad) ODIN_IH(true, true, true); else ODIN_IH(true, true, false); }
  else { if (in_is_grad) ODIN_IH(false, false, true); else ODIN_IH(false, false, false); }
#undef ODIN_IH
  return odin_check_launch("igemm_h(f16x2)");
}

// weight gradient: fine tensor (FH, FW, CU) gathered around the pixels of the coarse one (h, w, CV)
bool odin_igemm_h_wgrad_applicable(int B, int FH, int FW, int CU, int h, int w, int CV, int KH, int KW, int S,
                                   int center) {
  if (!ih_enabled() || center) return false;
  if (CU < 8 || (CU & 7) != 0 || CU > 8192 || KH * KW > 30 || KH < 1 || KW < 1 || KW > 8 || S < 1 || S > 4) return false;
  if (FH > 8192 || FW > 8192) return false;
  const long M = (long)B * h * w;
  if ((M < 512 && g_ih_min_flop > 0.0) || M < 16 || (M + ODIN_MAX_SLAB_BLOCKS - 1) / ODIN_MAX_SLAB_BLOCKS > IHW_CHUNK - 16)
    return false;
  if (2.0 * M * KH * KW * CU * CV < g_ih_min_flop) return false;
  if ((long)B * FH * FW * CU >= (1L << 29) || M * CV >= (1L << 29)) return false;
  return true;
}

// reduction splits (= slab rows) of the weight-gradient launch
int odin_igemm_h_wgrad_rows(int B, int h, int w, int KH, int KW, int CU, int CV) {
  const int M = B * h * w;
  const long tiles = (long)((KH * KW * CU + 31) / 32) * ((CV + 31) / 32);
  int R = (M + IHW_CHUNK - 17) / (IHW_CHUNK - 16);
  // enough workgroups to fill the chip twice, at least 128 pixels each
  while (tiles * R < 2048 && M / (R * 2) >= 128 && R * 2 <= ODIN_MAX_SLAB_BLOCKS) R *= 2;
  return R;
}

int odin_igemm_h_wgrad_launch(const float* u, const float* v, float* slab, int slab_stride, int B, int FH, int FW,
                              int CU, int h, int w, int CV, int KH, int KW, int S, int pt, int pl, int want_bias,
                              int grad_u, const uint32_t* g_amax, const uint32_t* a_amax, void* stream) {
  IHWParams p;
  memset(&p, 0, sizeof(p));
  p.u = u; p.v = v; p.slab = slab; p.slab_stride = slab_stride; p.a_amax = a_amax;
  p.B = B; p.FH = FH; p.FW = FW; p.CU = CU; p.h = h; p.w = w; p.CV = CV;
  p.KH = KH; p.KW = KW; p.S = S; p.pt = pt; p.pl = pl; p.want_bias = want_bias;
  p.M = B * h * w;
  const int R = odin_igemm_h_wgrad_rows(B, h, w, KH, KW, CU, CV);
  p.chunk = (((p.M + R - 1) / R) + 15) & ~15;
  if (p.chunk > IHW_CHUNK) return odin_fail(-2, "igemm_h wgrad: chunk beyond the pixel table");
  const float* g = grad_u ? u : v;
  p.g_amax = odin_range_word_of(g, grad_u ? (size_t)B * FH * FW * CU : (size_t)p.M * CV, g_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "igemm_h wgrad: no range word for the gradient operand");
  dim3 grid((CV + 31) / 32, (KH * KW * CU + 31) / 32, R);
  const int nsteps = (p.chunk + 15) / 16;
  // four waves share a workgroup's pixel table (its fill is ~100 VALU per entry) and split its steps: measured on the
  // audio stack's five weight gradients 210 / 105 / 57 / 33 / 68 us with one wave, 162 / 81 / 42 / 28 / 50 with four
  int nw = 1;
  while (nw < 4 && nsteps / (nw * 2) >= 4) nw *= 2;
  if (const char* e = ODIN_DIAG_ENV("ODIN_IHW_NW")) nw = atoi(e);
#define ODIN_IHW(N_)                                                                                        \
  do {                                                                                                      \
    if (a_amax != nullptr) {                                                                                \
      if (grad_u) ODIN_LAUNCH((igemm_h_wgrad_kernel<N_, true, false, true>), grid, dim3(N_ * 64), 0, stream, p);  \
      else ODIN_LAUNCH((igemm_h_wgrad_kernel<N_, false, true, true>), grid, dim3(N_ * 64), 0, stream, p);   \
    } else if (grad_u) ODIN_LAUNCH((igemm_h_wgrad_kernel<N_, true, false>), grid, dim3(N_ * 64), 0, stream, p);   \
    else ODIN_LAUNCH((igemm_h_wgrad_kernel<N_, false, true>), grid, dim3(N_ * 64), 0, stream, p);          \
  } while (0)
  if (nw >= 4) ODIN_IHW(4);
  else if (nw == 2) ODIN_IHW(2);
  else ODIN_IHW(1);
#undef ODIN_IHW
  return odin_check_launch("igemm_h_wgrad(f16x2)");
}
