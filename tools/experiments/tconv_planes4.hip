// tconv_planes4.hip -- the transposed 4x4 / stride-2 gathers of tconv_planes.hip (Conv2DTranspose forward,
// Conv2D data gradient, the fused decoder tail; same arithmetic: fp32 operands as three exact bf16
// planes, six plane products per 16 k-values, fp32 accumulation) restructured around what the round-3
// counters showed (profiles/r03_kpmc_planes.txt): with two waves per SIMD the 8-wave kernel spent 75 % of
// the SIMD's time ISSUING (336 VALU + 146 SALU + 70 LDS instructions per wave and tile at one quad-cycle
// each), 34 % of every wave's life stalled behind its partner's MFMA and 29 % at the per-tile barrier.
//
//   * 4 waves, one per SIMD, up to 512 registers: a wave owns one (column parity, row parity) class and
//     ALL 64 input positions of the tile (two 32-pixel groups, two accumulator chains).  Its 4 taps x
//     2 k-halves x 3 planes of WEIGHT FRAGMENTS live in 96 registers for the whole kernel: no weight image
//     in LDS (96 KB), half the LDS reads per MFMA, no 6 k-cycle weight staging per workgroup.
//   * the freed LDS holds a deeper row ring (3 RP + 4 rows): the rows of tile T + 2 are stored during tile
//     T, so the first fragment reads of tile T + 1 are issued BEFORE the barrier that ends tile T.
//   * per-tile addressing is scalar: every global access is a buffer instruction with a per-lane offset
//     computed once and a wave-uniform `soffset` per tile; the row walk is branch-free.
// LDS row layout, swizzle and fragment order are those of tconv_planes.hip.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>
#include <utility>

namespace {

struct T4Params {
  const float* in;     // [B, H, W, CS]
  const float* w;      // [16 taps][CO][CS]
  const float* bias;   // EPI 1 / 3: [CO]
  const float* aux;    // EPI 2: [B, 2H, 2W, CO], out *= ELU'(aux)
  float* out;          // [B, 2H, 2W, CO]  (EPI 3: dL/d pre-activation of this layer)
  float* colsum;       // EPI 2: [gridDim.x][CO]
  const float* w1;     // EPI 3: [CO][C1]
  const float* b1;     // [C1]
  const float* target; // [B, 2H, 2W, C1]
  float* logits;       // optional [B, 2H, 2W, C1]
  float* llk_part;     // [n_tiles]
  float* slab;         // [gridDim.x][CO * C1 + C1 + CO]
  const float* scale;  // device scalar 1/B
  int B, H, CO;
  int CS, ci_off;      // channels per input pixel in memory (32 or 64) and the first of this pass's 32
  int tiles_per_img, n_tiles, tiles_per_wg;
};

template <int... Is, class F>
__device__ __forceinline__ void t4_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void t4_static_for(F&& f) {
  t4_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct T4Yes { static constexpr bool value = true; };
struct T4No { static constexpr bool value = false; };

// x + the value of lane ^ 32
__device__ __forceinline__ float t4_pairsum32(float x) {
#ifdef ODIN_SIM
  return x + __shfl_xor(x, 32);
#else
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

// four consecutive fp32 values -> their three bf16 planes (4 bf16 = 8 bytes each), exact
__device__ __forceinline__ void t4_split4(const float4& v, u32x2& h, u32x2& m, u32x2& l) {
  h = odin_u2(odin_pack_bf16(v.x, v.y), odin_pack_bf16(v.z, v.w));
  const float r0 = odin_bf16_rest(v.x), r1 = odin_bf16_rest(v.y), r2 = odin_bf16_rest(v.z),
              r3 = odin_bf16_rest(v.w);
  m = odin_u2(odin_pack_bf16(r0, r1), odin_pack_bf16(r2, r3));
  l = odin_u2(odin_pack_bf16(odin_bf16_rest(r0), odin_bf16_rest(r1)),
              odin_pack_bf16(odin_bf16_rest(r2), odin_bf16_rest(r3)));
}

struct T4Item {
  float4 v;
  int dst;  // byte offset of the hi-plane store inside the ring (no item: inside the spare slot)
};

// EPI 0: raw partial sums (first of two reduction passes over 64 input channels); 1: bias + ELU;
// 2: x ELU'(aux) + column sums; 3: fused Bernoulli tail with C1 logit maps.  ACC: add the partial sums
// the previous pass left in `out` before the epilogue.
template <int EPI, int C1, int W, bool ACC>
__global__ __launch_bounds__(256) void tconv_planes4_kernel(T4Params p) {
  constexpr int RP = 64 / W;              // input rows per tile
  constexpr int NSLOT = 3 * RP + 4;       // rows of tile T (RP + 2), T + 1 (RP), T + 2 (RP) + two image seams
  constexpr int PB = (W + 2) * 64;        // one plane of a row: W + 2 pixels x 32 bf16
  constexpr int RB = 3 * PB;
  constexpr int CPR = W / 8;              // 1 KB load items (8 pixels x 32 channels fp32) per row
  constexpr int RSTEP = 4 / CPR;          // rows between a wave's consecutive items
  constexpr int NI0 = (W == 32) ? 4 : 3;  // items per wave for the RP + 2 rows of the first tile
  constexpr int NI = 3;                   // ... for the <= RP + 1 new rows of every further tile
  ODIN_DYN_SMEM(char, ring);
  constexpr int NRED = 32 * (1 + (EPI == 3 ? C1 : 0)) + 4;  // per-wave reduction row
  __shared__ float cred[4 * NRED];
  __shared__ float llk_red[2][4];  // per-wave log-likelihood partials of a sample, by tile parity
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const int HP = p.H + 1;
  const int OH = 2 * p.H, OW = 2 * W;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  // ---- SAME-padding pixels (pc = 0 and pc = W + 1) of every ring row and plane: zero for ever ----
  for (int e = tid; e < NSLOT * 24; e += 256) {
    const int sl = e / 24, rem = e - sl * 24;
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(ring + sl * RB + pl * PB + (side ? (W + 1) * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- row fills: wave-uniform walk over the padded rows (global padded row g = HP * b + gi, gi = 0 the
  // zero row between images); wave `wave` moves item e = wave + 4 j of a tile's new rows: row e / CPR,
  // 1 KB chunk e % CPR = wave % CPR.  Branch-free: selects on wave-uniform integers. ----
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * p.H * W * p.CS * 4));
  int f_gi, f_b, f_slot, f_g, need_g0, ft_t;
  {
    const int b0 = T0 / p.tiles_per_img, t0 = T0 - b0 * p.tiles_per_img;
    f_g = HP * b0 + RP * t0;
    f_gi = RP * t0;
    f_b = b0;
    f_slot = f_g % NSLOT;
    need_g0 = f_g;
    ft_t = t0;
  }
  const int f_c = wave & (CPR - 1), f_r0 = wave / CPR;
  const int f_px = 8 * f_c + (lane >> 3), f_ch4 = lane & 7, f_pc = f_px + 1;
  const int f_lds_lane = f_pc * 64 + ((((f_ch4 >> 1) ^ ((f_pc >> 2) & 3))) << 4) + (f_ch4 & 1) * 8;
  const unsigned f_g_lane = (unsigned)((f_px * p.CS + p.ci_off + 4 * f_ch4) * 4);
  const unsigned f_rowbytes = (unsigned)(W * p.CS * 4);
  auto load_fill = [&](auto n_items, T4Item (&it)[4], bool live) __attribute__((always_inline)) {
    constexpr int N = decltype(n_items)::value;
    const int nrows = live ? need_g0 + RP + 2 - f_g : 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int r = f_r0 + RSTEP * j;
      const int valid = r < nrows;
      int gi = f_gi + r;
      const int wrap = gi >= HP;
      gi -= wrap ? HP : 0;
      const int b = f_b + wrap;
      int slot = f_slot + r;
      slot -= slot >= NSLOT ? NSLOT : 0;
      it[j].dst = (valid ? slot : NSLOT) * RB + f_lds_lane;  // (no item: the spare slot behind the ring)
      const int real = valid & (gi != 0) & (b < p.B);
      const unsigned row_off = real ? (unsigned)(b * p.H + gi - 1) * f_rowbytes : 0u;
      it[j].v = odin_run_load4s(IN, real ? f_g_lane : ODIN_OOB_V, row_off);
    }
    f_g += nrows;
    f_gi += nrows;
    const int w2 = f_gi >= HP;
    f_gi -= w2 ? HP : 0;
    f_b += w2;
    f_slot += nrows;
    f_slot -= f_slot >= NSLOT ? NSLOT : 0;
    const int seam = live & (ft_t + 1 == p.tiles_per_img);
    need_g0 += live ? RP + seam : 0;
    ft_t = live ? (seam ? 0 : ft_t + 1) : ft_t;
  };
  auto store_fill1 = [&](const T4Item& it) __attribute__((always_inline)) {
    u32x2 h, m, l;
    t4_split4(it.v, h, m, l);
    char* d = ring + it.dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PB) = m;
    *reinterpret_cast<u32x2*>(d + 2 * PB) = l;
  };

  // ---- this wave's pixel class: column parity x row parity; two groups of 32 input positions ----
  const int cpw = wave & 1, rpar = wave >> 1;
  // input row of the tile per group (per lane when a 32-pixel group spans 2 or 4 rows) and input column
  int rp[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
    rp[g] = (W == 32) ? g : (W == 16) ? 2 * g + (l31 >> 4) : 4 * g + (l31 >> 3);
  const int i_in = (W == 32) ? l31 : (W == 16) ? (l31 & 15) : (l31 & 7);
  // column taps: parity 0 -> kw = 1 (padded column pc = i + 1), kw = 3 (pc = i); parity 1 -> kw = 0
  // (pc = i + 2), kw = 2 (pc = i + 1); row taps alike with kh / padded rows
  const int kw_a = cpw ? 0 : 1, kw_b = kw_a + 2;
  const int kh_a = rpar ? 0 : 1, kh_b = kh_a + 2;
  const int d_a = cpw ? 2 : 1;
  int offA[2], offB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int pa_ = i_in + d_a, pb_ = pa_ - 1;
    offA[kk] = pa_ * 64 + (((2 * kk + half) ^ ((pa_ >> 2) & 3)) << 4);
    offB[kk] = pb_ * 64 + (((2 * kk + half) ^ ((pb_ >> 2) & 3)) << 4);
  }

  // ---- prologue: the first tile's rows in flight while the weight fragments are fetched and split ----
  T4Item itA[4], itB[4];
  load_fill(std::integral_constant<int, NI0>{}, itA, true);
  // weight fragments: tap t (0: a/a, 1: a/b, 2: b/a, 3: b/b in (row tap, column tap)), k-half kk, plane:
  // lane (co = l31, half) supplies channels 16 kk + 8 half .. + 7 of W[tap][n0 + co][.]
  u32x4 wf[4][2][3];
  {
    float4 wv[4][2][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int tap = ((t < 2) ? kh_a : kh_b) * 4 + ((t & 1) ? kw_b : kw_a);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          wv[t][kk][q] = *reinterpret_cast<const float4*>(
              p.w + ((size_t)(tap * p.CO + n0 + l31) * p.CS + p.ci_off + 16 * kk + 8 * half + 4 * q));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x2 h0, m0, l0, h1, m1, l1;
        t4_split4(wv[t][kk][0], h0, m0, l0);
        t4_split4(wv[t][kk][1], h1, m1, l1);
        wf[t][kk][0] = u32x4{h0.x, h0.y, h1.x, h1.y};
        wf[t][kk][1] = u32x4{m0.x, m0.y, m1.x, m1.y};
        wf[t][kk][2] = u32x4{l0.x, l0.y, l1.x, l1.y};
      }
  }

  // ---- epilogue constants: accumulator register r holds channel n0 + (r & 3) + 8 (r >> 2) + 4 half.
  // Plain (unpacked) VALU arithmetic only: v_pk_*_f32 serialises with the bf16 matrix pipe.
  auto ch_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * half; };
  constexpr float LOG2E = 1.44269504088896341f;
  float bias_r[(EPI == 1 || EPI == 3) ? 16 : 1];
  if (EPI == 1 || EPI == 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = p.bias[n0 + ch_of(r)];
  }
  float w1r[(EPI == 3) ? 16 : 1][(EPI == 3) ? C1 : 1];
  float dw1[(EPI == 3) ? 16 : 1][(EPI == 3) ? C1 : 1];
  float b1r[(EPI == 3) ? C1 : 1];
  float db1[(EPI == 3) ? C1 : 1];
  if (EPI == 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int oc = 0; oc < C1; ++oc) {
        w1r[r][oc] = p.w1[(n0 + ch_of(r)) * C1 + oc];
        dw1[r][oc] = 0.f;
      }
#pragma unroll
    for (int oc = 0; oc < C1; ++oc) { b1r[oc] = p.b1[oc]; db1[oc] = 0.f; }
  }
  float csum[(EPI >= 2) ? 16 : 1];
#pragma unroll
  for (int r = 0; r < ((EPI >= 2) ? 16 : 1); ++r) csum[r] = 0.f;
  const float sc = (EPI == 3) ? p.scale[0] : 0.f;
  float llk_lane = 0.f;
  // global tensors behind buffer descriptors: per-lane offsets once, one scalar per tile
  const unsigned out_bytes = (unsigned)((size_t)p.B * OH * OW * p.CO * 4);
  const OdinRun OUT = odin_run(p.out, out_bytes);
  const OdinRun AUX = odin_run(EPI == 2 ? p.aux : nullptr, EPI == 2 ? out_bytes : 0u);
  const unsigned tgt_bytes = (EPI == 3) ? (unsigned)((size_t)p.B * OH * OW * C1 * 4) : 0u;
  const OdinRun TG = odin_run(EPI == 3 ? p.target : nullptr, tgt_bytes);
  const OdinRun LG = odin_run(EPI == 3 ? p.logits : nullptr, (EPI == 3 && p.logits != nullptr) ? tgt_bytes : 0u);
  unsigned out_lane[2], tgt_lane[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const unsigned pix = (unsigned)((2 * rp[g] + rpar) * OW + 2 * i_in + cpw);  // inside the tile's 2 RP output rows
    out_lane[g] = (pix * p.CO + n0 + 4 * half) * 4;
    tgt_lane[g] = pix * C1 * 4;
  }
  const unsigned lg_lane_mask = (half == 0) ? 0u : ODIN_OOB_V;  // the half == 0 lane of a pixel stores its logits

  store_fill1(itA[0]);
  store_fill1(itA[1]);
  store_fill1(itA[2]);
  if (NI0 == 4) store_fill1(itA[3]);
  load_fill(std::integral_constant<int, NI>{}, itA, T0 + 1 < T1);
  store_fill1(itA[0]);
  store_fill1(itA[1]);
  store_fill1(itA[2]);
  load_fill(std::integral_constant<int, NI>{}, itA, T0 + 2 < T1);  // stored during the first tile
  __syncthreads();  // pads and the rows of the first two tiles are in LDS

  int b_cur = T0 / p.tiles_per_img, t_cur = T0 - b_cur * p.tiles_per_img;
  int sl0 = (HP * b_cur + RP * t_cur) % NSLOT;  // ring slot of the tile's first padded row
  // state of the PREVIOUS tile, whose epilogue rides in the current tile's MFMA stream
  f32x16 pa[2] = {f32x16_zero(), f32x16_zero()};
  unsigned tileP_out = 0, tileP_tgt = 0;  // scalar byte offsets of the previous tile in out / target
  float4 axP[2][4], pvP[2][4];
  float tgtP[2][(EPI == 3) ? C1 : 1] = {};
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int q = 0; q < 4; ++q) axP[g][q] = pvP[g][q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dl[(EPI == 3) ? C1 : 1] = {};

  // The epilogue of one 32-pixel group of a tile (lane = pixel x 16 channels) as micro-ops of a few VALU
  // instructions; op k of group G is issued right behind MFMA slot 48 G + (k < 8 ? 2 k : k + 8).
  auto elu_r = [&](int G, int r) __attribute__((always_inline)) {
    const float t = pa[G][r] + bias_r[(EPI == 1 || EPI == 3) ? r : 0];
    const float em1 = odin_exp2(t * LOG2E) - 1.f;
    pa[G][r] = t > 0.f ? t : em1;
  };
  auto store_q = [&](int G, int q) __attribute__((always_inline)) {
    odin_run_store4s(OUT, out_lane[G] + 32 * q, tileP_out,
                     make_float4(pa[G][4 * q], pa[G][4 * q + 1], pa[G][4 * q + 2], pa[G][4 * q + 3]));
  };
  float t_dot = 0.f, lgt = 0.f, eabs = 0.f;  // dot product, logit, exp(-|logit|) of the logit map in flight
  constexpr int N_EPI_OPS = (EPI == 3) ? 12 + 4 * C1 + 8 : 12;
  static_assert(N_EPI_OPS + 8 <= 48, "a group's epilogue micro-ops must fit its 48 MFMA slots");
  auto epi_op = [&](int G, int k) __attribute__((always_inline)) {
    if (k < 8) {
      const int r0 = 2 * k, r1 = r0 + 1;
      if (ACC) {
        const int q = k >> 1;
        pa[G][r0] += (k & 1) ? pvP[G][q].z : pvP[G][q].x;
        pa[G][r1] += (k & 1) ? pvP[G][q].w : pvP[G][q].y;
      }
      if (EPI == 1 || EPI == 3) { elu_r(G, r0); elu_r(G, r1); }
      if (EPI == 2) {
        const int q = k >> 1;
        const float a0 = (k & 1) ? axP[G][q].z : axP[G][q].x, a1 = (k & 1) ? axP[G][q].w : axP[G][q].y;
        pa[G][r0] = fmaf(pa[G][r0], fminf(a0, 0.f), pa[G][r0]);  // x ELU'(aux) = 1 + min(aux, 0)
        pa[G][r1] = fmaf(pa[G][r1], fminf(a1, 0.f), pa[G][r1]);
        csum[r0] += pa[G][r0];
        csum[r1] += pa[G][r1];
      }
    }
    if (EPI <= 2) {
      if (k >= 8 && k < 12) store_q(G, k - 8);
    }
    if (EPI == 3) {
      // per logit map oc: 4 ops (dot, logistic terms, likelihood, its gradient)
      if (k >= 8 && k < 8 + 4 * C1) {
        const int oc = (k - 8) >> 2, ph = (k - 8) & 3;
        if (ph == 0) {
          // two chains of eight: a dependent FMA waits for its predecessor
          float ta = pa[G][0] * w1r[0][oc], tb = pa[G][1] * w1r[1][oc];
#pragma unroll
          for (int r = 2; r < 16; r += 2) {
            ta = fmaf(pa[G][r], w1r[r][oc], ta);
            tb = fmaf(pa[G][r + 1], w1r[r + 1][oc], tb);
          }
          t_dot = ta + tb;
        }
        if (ph == 1) {
          lgt = t4_pairsum32(t_dot) + b1r[oc];  // the other 16 channels live in lane ^ 32
          eabs = odin_exp2(-LOG2E * fabsf(lgt));
          odin_run_store1s(LG, (tgt_lane[G] + 4 * oc) | lg_lane_mask, tileP_tgt, lgt);
        }
        if (ph == 2) {
          // log p(x | logit) = x l - softplus(l); the half == 0 lane of a pixel owns the scalar results
          const float sp = fmaxf(lgt, 0.f) + 0.6931471805599453f * odin_log2(1.f + eabs);
          llk_lane += half == 0 ? tgtP[G][oc] * lgt - sp : 0.f;
        }
        if (ph == 3) {
          const float rr = odin_rcp(1.f + eabs);
          const float sg = lgt >= 0.f ? rr : eabs * rr;
          const float dsig = (sg - tgtP[G][oc]) * sc;
          db1[oc] += half == 0 ? dsig : 0.f;
          dl[oc] = dsig;
        }
      }
      constexpr int G0 = 8 + 4 * C1;
      if (k >= G0 && k < G0 + 8) {
#pragma unroll
        for (int r = 2 * (k - G0); r < 2 * (k - G0) + 2; ++r) {
          float gs = w1r[r][0] * dl[0];
          dw1[r][0] = fmaf(pa[G][r], dl[0], dw1[r][0]);
#pragma unroll
          for (int oc = 1; oc < C1; ++oc) {
            gs = fmaf(w1r[r][oc], dl[oc], gs);
            dw1[r][oc] = fmaf(pa[G][r], dl[oc], dw1[r][oc]);
          }
          pa[G][r] = fmaf(gs, fminf(pa[G][r], 0.f), gs);  // x ELU'(y) = 1 + min(y, 0)
          csum[r] += pa[G][r];
        }
      }
      if (k >= G0 + 8 && k < G0 + 12) store_q(G, k - G0 - 8);
    }
  };

  // the log-likelihood partial of a sample is flushed once, behind its last tile
  auto flush_llk = [&](int T) __attribute__((always_inline)) {
    if (EPI == 3) {
      const float tt = odin_wave_sum64_valu(llk_lane);
      llk_red[T & 1][wave] = tt;  // (every lane holds the sum) summed over the 4 waves behind the next barrier
      llk_lane = 0.f;
    }
  };

  // fragment reads of one step (tap tp, k-half kk) for both pixel groups
  u32x4 fb[2][2][3];  // [buffer][group][plane]
  auto loads = [&](int s, int slot0, u32x4 (&Bf)[2][3]) __attribute__((always_inline)) {
    const int tp = s >> 1, kk = s & 1;
    const bool ra = tp < 2, ca = (tp & 1) == 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      // padded row of the tap relative to the tile's first: row tap a -> rp + (rpar ? 2 : 1), b -> one less
      int sl = slot0 + rp[g] + (rpar ? 2 : 1) - (ra ? 0 : 1);
      sl -= sl >= NSLOT ? NSLOT : 0;
      const char* bp = ring + sl * RB + (ca ? offA[kk] : offB[kk]);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) Bf[g][pl] = *reinterpret_cast<const u32x4*>(bp + pl * PB);
    }
  };
  loads(0, sl0, fb[0]);

  // one tile = 96 MFMA slots: step s = m / 12 (4 taps x 2 k-halves), inside a step the two groups'
  // chains alternate; the fragment reads of step s + 1 (of the NEXT tile's step 0 behind step 7) go out
  // with the first MFMA of step s
  auto run_tile = [&](auto with_epi, int T) __attribute__((always_inline)) {
    constexpr bool WE = decltype(with_epi)::value;
    f32x16 acc[2] = {f32x16_zero(), f32x16_zero()};
    // next tile's ring slot (for the read-ahead behind step 7) and this tile's global offsets
    int t_nxt = t_cur + 1, sl_nxt = sl0 + RP;
    const int seam = t_nxt == p.tiles_per_img;
    sl_nxt += seam;
    sl_nxt -= sl_nxt >= NSLOT ? NSLOT : 0;
    const unsigned tile_pix = (unsigned)((b_cur * OH + 2 * RP * t_cur) * OW);
    const unsigned tile_out = tile_pix * (unsigned)p.CO * 4u, tile_tgt = tile_pix * (unsigned)C1 * 4u;
    float4 axN[2][4], pvN[2][4];
    float tgtN[2][(EPI == 3) ? C1 : 1] = {};
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int q = 0; q < 4; ++q) axN[g][q] = pvN[g][q] = make_float4(0.f, 0.f, 0.f, 0.f);
    ODIN_SCHED_FENCE();
    t4_static_for<96>([&](auto M) __attribute__((always_inline)) {
      constexpr int m = decltype(M)::value;
      constexpr int s = m / 12, u = m % 12, g = u & 1, pr = u >> 1, cur = s & 1, nxt = cur ^ 1;
      constexpr int tp = s >> 1, kk = s & 1;
      if constexpr (u == 0) {
        if constexpr (s < 7) loads(s + 1, sl0, fb[nxt]);
        else if (T + 1 < T1) loads(0, sl_nxt, fb[nxt]);
      }
      // plane products, smallest first: 0*2, 2*0, 1*1, 0*1, 1*0, 0*0 (weights x pixels)
      constexpr int pw = (pr == 0) ? 0 : (pr == 1) ? 2 : (pr == 2) ? 1 : (pr == 3) ? 0 : (pr == 4) ? 1 : 0;
      constexpr int px = (pr == 0) ? 2 : (pr == 1) ? 0 : (pr == 2) ? 1 : (pr == 3) ? 1 : (pr == 4) ? 0 : 0;
      acc[g] = mfma32_bf16(wf[tp][kk][pw], fb[cur][g][px], acc[g]);
      // group G = m / 48 of the previous tile: the 8 activation ops behind every other MFMA of its first
      // 16 slots, the remaining ops one per MFMA
      constexpr int G = m / 48, me = m % 48;
      constexpr int k = me < 16 ? ((me & 1) ? -1 : me / 2) : me - 8;
      if constexpr (WE && k >= 0 && k < N_EPI_OPS) epi_op(G, k);
      if constexpr (m == 44) load_fill(std::integral_constant<int, NI>{}, itB, T + 3 < T1);  // rows of tile T + 3
      if constexpr (m == 45 || m == 93) {  // this tile's epilogue operands (used one tile later), group m / 48
        constexpr int GG = m / 48;
        if (EPI == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q) axN[GG][q] = odin_run_load4s(AUX, out_lane[GG] + 32 * q, tile_out);
        }
        if (EPI == 3) {
#pragma unroll
          for (int oc = 0; oc < C1; ++oc) tgtN[GG][oc] = odin_run_load1s(TG, tgt_lane[GG] + 4 * oc, tile_tgt);
        }
        if (ACC) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pvN[GG][q] = odin_run_load4s(OUT, out_lane[GG] + 32 * q, tile_out);
        }
      }
      if constexpr (WE && m == 94) {
        if (t_cur == 0) flush_llk(T - 1);
      }
      // rows of tile T + 2 (loaded during tile T - 1): split + store
      if constexpr (m == 46) store_fill1(itA[0]);
      if constexpr (m == 47) store_fill1(itA[1]);
      if constexpr (m == 95) store_fill1(itA[2]);
      ODIN_SCHED_FENCE();
    });
    pa[0] = acc[0];
    pa[1] = acc[1];
    tileP_out = tile_out;
    tileP_tgt = tile_tgt;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { axP[g][q] = axN[g][q]; pvP[g][q] = pvN[g][q]; }
#pragma unroll
      for (int oc = 0; oc < ((EPI == 3) ? C1 : 1); ++oc) tgtP[g][oc] = tgtN[g][oc];
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) itA[j] = itB[j];
    // the walk to the next tile
    sl0 = sl_nxt;
    t_cur = seam ? 0 : t_nxt;
    b_cur += seam;
  };

  run_tile(T4No{}, T0);
  __syncthreads();
#pragma unroll 1
  for (int T = T0 + 1; T < T1; ++T) {
    run_tile(T4Yes{}, T);
    __syncthreads();  // every wave is past tile T's rows; tile T + 2's rows are stored
    if (EPI == 3 && tid == 0) {
      // tile T - 1 closed a sample (t_cur now names tile T + 1; tile T opened one when it is 1 ... or the
      // image has one tile): its slot carries the sample's (this workgroup's share of the) sum, the other
      // tiles' slots a zero
      const int t_of_T = (t_cur == 0 ? p.tiles_per_img : t_cur) - 1;
      const float* q = llk_red[(T - 1) & 1];
      p.llk_part[T - 1] = t_of_T == 0 ? (q[0] + q[1]) + (q[2] + q[3]) : 0.f;
    }
  }
#pragma unroll
  for (int G = 0; G < 2; ++G)
#pragma unroll
    for (int k = 0; k < N_EPI_OPS; ++k) epi_op(G, k);
  flush_llk(T1 - 1);

  // ---- per-workgroup partial sums: 32 pixel lanes by shuffles, then the 4 waves through LDS ----
  if (EPI >= 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ch = (r & 3) + 8 * (r >> 2) + 4 * half;
      float vv = csum[r];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
      if (l31 == 0) cred[wave * NRED + ch] = vv;
      if (EPI == 3) {
#pragma unroll
        for (int oc = 0; oc < C1; ++oc) {
          float ww = dw1[r][oc];
#pragma unroll
          for (int m = 16; m >= 1; m >>= 1) ww += __shfl_xor(ww, m);
          if (l31 == 0) cred[wave * NRED + 32 + ch * C1 + oc] = ww;
        }
      }
    }
  }
  if (EPI == 3) {
#pragma unroll
    for (int oc = 0; oc < C1; ++oc) {
      float vv = db1[oc];  // non-zero in the half == 0 lanes only
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
      if (lane == 0) cred[wave * NRED + 32 + 32 * C1 + oc] = vv;
    }
  }
  if (EPI == 2 && p.colsum != nullptr) {
    __syncthreads();
    if (tid < 32) {
      float tt = 0.f;
      for (int wv = 0; wv < 4; ++wv) tt += cred[wv * NRED + tid];
      p.colsum[(size_t)blockIdx.x * p.CO + n0 + tid] = tt;
    }
  }
  if (EPI == 3) {
    __syncthreads();
    if (tid == 0) {
      const float* q = llk_red[(T1 - 1) & 1];
      p.llk_part[T1 - 1] = (q[0] + q[1]) + (q[2] + q[3]);
    }
    // slab row: [dW1 (CO, C1) | db1 (C1) | column sums of out (CO)]
    float* row = p.slab + (size_t)blockIdx.x * (p.CO * C1 + C1 + p.CO);
    for (int e = tid; e < 32 + 32 * C1 + C1; e += 256) {
      float tt = 0.f;
      for (int wv = 0; wv < 4; ++wv) tt += cred[wv * NRED + e];
      if (e < 32) {
        row[p.CO * C1 + C1 + e] = tt;
      } else if (e < 32 + 32 * C1) {
        const int ch = (e - 32) / C1, oc = (e - 32) - ch * C1;
        row[ch * C1 + oc] = tt;
      } else {
        row[p.CO * C1 + (e - 32 - 32 * C1)] = tt;
      }
    }
  }
}

template <int EPI, int C1, bool ACC>
int t4_launch_w(const T4Params& p, int W, dim3 grid, void* stream) {
  const int RP = 64 / W;
  const size_t lds = (size_t)(3 * RP + 4 + 1) * 3 * (W + 2) * 64;  // ring + spare slot
  constexpr int W3 = (EPI == 3 ? 16 : 8);  // (the fused tail has no 8-pixel geometry)
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[3] = {reinterpret_cast<const void*>(&tconv_planes4_kernel<EPI, C1, 32, ACC>),
                          reinterpret_cast<const void*>(&tconv_planes4_kernel<EPI, C1, 16, ACC>),
                          reinterpret_cast<const void*>(&tconv_planes4_kernel<EPI, C1, W3, ACC>)};
    for (int i = 0; i < 3; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  if (W == 32) ODIN_LAUNCH((tconv_planes4_kernel<EPI, C1, 32, ACC>), grid, dim3(256), lds, stream, p);
  else if (W == 16) ODIN_LAUNCH((tconv_planes4_kernel<EPI, C1, 16, ACC>), grid, dim3(256), lds, stream, p);
  else ODIN_LAUNCH((tconv_planes4_kernel<EPI, C1, W3, ACC>), grid, dim3(256), lds, stream, p);
  return odin_check_launch("tconv_planes4(bf16x3)");
}

}  // namespace

// Same contract as odin_tconv_planes_launch (tconv_planes.hip), which dispatches here unless ODIN_TP8 is set.
int odin_tconv_planes4_launch(const float* in, const float* w, const float* bias, const float* aux,
                              float* out, float* colsum, int* rows_out, const float* w1, const float* b1,
                              const float* target, float* logits, float* llk_part, int* n_part_out,
                              float* slab, const float* scale, int C1, int B, int H, int W, int CI,
                              int CO, int epi, void* stream) {
  T4Params p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.w1 = w1; p.b1 = b1; p.target = target; p.logits = logits; p.llk_part = llk_part; p.slab = slab;
  p.scale = scale;
  p.B = B; p.H = H; p.CO = CO;
  p.CS = CI; p.ci_off = 0;
  const int RP = 64 / W;
  p.tiles_per_img = H / RP;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  p.tiles_per_wg = (p.n_tiles + cap - 1) / cap;
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (n_part_out) *n_part_out = p.tiles_per_img;
  if (out == nullptr) return 0;  // dry run
  dim3 grid(gx, gy, 1);
  if (CI == 64) {
    if (epi == 3) return odin_fail(-2, "tconv_planes4 tail: 32 input channels only");
    T4Params q = p;
    q.colsum = nullptr;
    const int rc = t4_launch_w<0, 1, false>(q, W, grid, stream);
    if (rc != 0) return rc;
    p.ci_off = 32;
    return epi == 1 ? t4_launch_w<1, 1, true>(p, W, grid, stream) : t4_launch_w<2, 1, true>(p, W, grid, stream);
  }
  if (epi == 1) return t4_launch_w<1, 1, false>(p, W, grid, stream);
  if (epi == 2) return t4_launch_w<2, 1, false>(p, W, grid, stream);
  if (C1 == 1) return t4_launch_w<3, 1, false>(p, W, grid, stream);
  if (C1 == 3) return t4_launch_w<3, 3, false>(p, W, grid, stream);
  return odin_fail(-2, "tconv_planes4 tail: one or three logit maps only");
}
