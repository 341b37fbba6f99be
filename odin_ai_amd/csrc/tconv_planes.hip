// tconv_planes.hip -- TRANSPOSED gather convolution 4x4 / stride 2 over 32 reduction channels with
// fp32 operands carried through the f16 matrix pipe as two f16 planes (odin_device.h: x = h + 2^-11 l;
// the three plane products h*h, h*l, l*h are accumulated in fp32 by v_mfma_f32_32x32x16_f16 into a main
// and a cross accumulator -- error <= 3 * 2^-22 per product, below the rounding noise of an fp32 dot
// product of these reduction lengths; round 2-3 carried three bf16 planes and six products).
//
// Serves (TF `SAME`, pads (1, 1)):
//   Conv2DTranspose(k4, s2) forward from 32 channels          (image_networks.py:503-506, decoder4)
//   Conv2D(k4, s2) DATA GRADIENT into 32-channel inputs       (tape.gradient of encoder1)
//   the fused decoder tail: decoder4 -> Conv2D 1x1 (decoder6) -> Independent(Bernoulli) log-prob AND
//   its backward in the epilogue (image_networks.py:505-511, 87-93; variational_autoencoder.py:528-530)
// i.e.  out[b, oh, ow, n] = sum over (kh, kw) with (oh + 1 - kh), (ow + 1 - kw) even, c < 32 of
//       in[b, (oh + 1 - kh) / 2, (ow + 1 - kw) / 2, c] * W[kh, kw, n, c]
//
// Why planes, and why split ONCE: measured on gfx950 (tools/micro/mfma_fillers.hip,
// mfma_bf16_fillers.hip) v_mfma_f32_*_f32 shares the vector ALU's issue -- every VALU instruction of
// an epilogue adds its full time to an fp32 MFMA stream, from the same wave or from a partner wave --
// while bf16 MFMAs run beside the VALU (5 VALU instructions per 32-cycle MFMA are free) at 16x the
// fp32 MFMA rate; f16 MFMAs behave the same (tools/micro/mfma_f16_split.hip).  Three f16 plane products
// per 16 k-values take 5.3x less matrix time than the fp32 MFMAs of the same block, and the epilogue hides
// behind them.  Splitting costs ~3 VALU instructions per element, so every input element is split
// exactly once per workgroup, on its way from HBM into LDS: global_load -> registers -> two
// ds_write_b64 (no LDS-DMA: DMA cannot convert).  The weights are split once per workgroup too.
//
// Structure: 8 waves, all alike.  LDS = weight planes [tap][plane][k-piece][co][8 f16] (64 KB) +
// a rolling window of input rows, slot = global padded row mod NSLOT, each [plane][pixel][32 f16]
// with the 16-byte k-pieces XOR-swizzled by (pixel >> 2) so that the 16-lane groups of ds_read_b128
// (MI355X_MICROARCH.md, LDS) hit 16 distinct slots of the 256-byte bank row.  A tile = RP input
// rows (RP * W = 64) = 2 RP output rows; a wave owns one (column parity, row parity) class and one
// 32-pixel group of the tile: 4 taps x 2 k-halves x 3 plane products = 24 MFMAs per tile, with the
// previous tile's epilogue, the split + store of the next tile's rows and the loads of the one after
// scheduled into the same MFMA stream.  One workgroup barrier per tile.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>
#include <utility>

namespace {

struct TPParams {
  const float* in;     // [B, H, W, 32]
  const float* w;      // [16 taps][CO][32]
  const float* bias;   // EPI 1 / 3: [CO]
  const float* aux;    // EPI 2: [B, 2H, 2W, CO], out *= ELU'(aux)
  float* out;          // [B, 2H, 2W, CO]  (EPI 3: dL/d pre-activation of this layer)
  float* colsum;       // EPI 2: [gridDim.x][CO]
  // fused tail (EPI 3)
  const float* w1;     // [CO][C1]
  const float* b1;     // [C1]
  const float* target; // [B, 2H, 2W, C1]
  float* logits;       // optional [B, 2H, 2W, C1]
  float* llk_part;     // [n_tiles]
  float* slab;         // [gridDim.x][CO * C1 + C1 + CO]
  const float* scale;  // device scalar 1/B
  int B, H, CO;
  int CS, ci_off;      // channels per input pixel in memory (32 or 64) and the first of this pass's 32
  int tiles_per_img, n_tiles, tiles_per_wg;
  const unsigned* in_amax;  // SC instances: the input is a gradient tensor; its range word (odin_device.h)
  unsigned* out_amax;       // EPI 2 / 3: range word of `out`, a gradient tensor (may be null)
  long long* stamps;   // diagnostics: s_memtime stamps of workgroup 0 (wave 0: [0,32), wave 4: [32,64))
  int part_off;        // PL instances: byte offset (dynamic LDS) of the partial-sum buffer [tile][q][thread] x 16 bytes
};

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define TP_STAMP(k) ((void)0)
#else
#define TP_STAMP(k)                                                                               \
  do {                                                                                            \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (wave & 3) == 0 && \
        stamp_i < 31)                                                                             \
      p.stamps[8 * wave + stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

// index of MFMA slot m among the slots >= g0 that run_tile does not reserve for its own row fills,
// operand loads and log-likelihood flush (32-34, 36, 42); -1 for a reserved or earlier slot
__host__ __device__ constexpr int tp_late_index(int g0, int m) {  // (g0 <= 32; loop-free so that it folds early)
  return (m < g0 || m == 32 || m == 33 || m == 34 || m == 36 || m == 42)
             ? -1
             : (m - g0) - ((m > 32) + (m > 33) + (m > 34) + (m > 36) + (m > 42));
}
static_assert(tp_late_index(25, 47) >= 11, "three logit maps: 8 gradient + 4 store slots must fit");

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the slot index of the tile's MFMA
// stream is a compile-time constant inside f (no dead branches for the optimiser to clear away)
template <int... Is, class F>
__device__ __forceinline__ void tp_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void tp_static_for(F&& f) {
  tp_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

struct TpYes { static constexpr bool value = true; };
struct TpNo { static constexpr bool value = false; };

constexpr int TP_NPL = 2;                        // f16 planes per operand
constexpr int TP_TAPB = TP_NPL * 4 * 512;        // one tap of the weight planes: [plane][k-piece][co][8 f16]
constexpr int TP_WBYTES = 16 * TP_TAPB;          // weight planes: 64 KB

// x + the value of lane ^ 32
__device__ __forceinline__ float tp_pairsum32(float x) {
#ifdef ODIN_SIM
  return x + __shfl_xor(x, 32);
#else
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

struct alignas(8) TpEnt {
  int x, y;
};

struct TpItem {
  float4 v;
  int dst;  // byte offset of the hi-plane store inside the ring
  int ok;   // wave-uniform: this wave has an item (no item: nothing is split or stored)
};

// EPI 1: bias + ELU; EPI 2: x ELU'(aux) + column sums; EPI 3: fused Bernoulli tail with C1 logit maps
// DBG (diagnostics only, ODIN_TP_DBG): 1 = no global stores of `out`, 2 = no LDS reads in the MFMA loop,
// 4 = no epilogue micro-ops
// EPI 0: raw partial sums (first of two reduction passes over 64 input channels); ACC: add the partial
// sums the previous pass left in `out` before the epilogue.
// SCM 1: the input tensor is a gradient (scaled by 2^gk on its way into the planes, the result scaled back); SCM 2: an
// ACTIVATION that comes with its range word -- scaled the same way, but only when its bound leaves [2^-8, 2^15)
// (odin_device.h: odin_act_needs_scale): a wave-uniform flag, one scalar branch around each split and around the
// accumulator combine.  (Round 5's first form held two whole bodies behind ONE branch at the top of the kernel: every
// launch then waited for the 32 scalar loads of the word before its first weight load -- +1.5 us per launch.)
// PL (round 6): the raw partial sums of the first of two reduction passes stay in LDS ([tile][q][thread] x 16 bytes, the
// SAME thread reads them back in the second pass) instead of travelling through `out`: decoder3's forward moved 120 MB
// for 50 MB of algorithmic traffic that way (PMC, profiles/r05_pmc_traffic.json).
template <int EPI, int C1, int W, int DBG = 0, bool ACC = false, int SCM = 0, bool PL = false>
__device__ __forceinline__ void tp_body(const TPParams& p) {
  constexpr int NPL = TP_NPL;
  constexpr int RP = 64 / W;              // input rows per tile
  constexpr int NSLOT = 2 * RP + 3;       // live rows of a tile (RP + 2) + the next tile's (RP + 1 at an image seam)
  constexpr int PB = (W + 2) * 64;        // one plane of a row: W + 2 pixels x 32 f16
  constexpr int RB = NPL * PB;
  constexpr int CPR = W / 8;              // 1 KB load items (8 pixels x 32 channels fp32) per row
  constexpr int CSHIFT = (W == 32) ? 2 : (W == 16) ? 1 : 0;
  ODIN_DYN_SMEM(char, smem);
  char* wl = smem;
  char* ring = smem + TP_WBYTES;
  constexpr int NRED = 32 * (1 + (EPI == 3 ? C1 : 0)) + 4;  // per-wave reduction row
  __shared__ float cred[8 * NRED];
  __shared__ float llk_red[2][8];  // per-wave log-likelihood partials of a tile, by tile parity
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5;
  int stamp_i = 0;
  (void)stamp_i;
  TP_STAMP(1);
#if defined(ODIN_DIAG) && !defined(ODIN_SIM)
  if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    p.stamps[64] = clock64();
    p.stamps[65] = wall_clock64();
  }
#endif
  const int n0 = blockIdx.y * 32;
  const int HP = p.H + 1;
  const int OH = 2 * p.H, OW = 2 * W;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  const OdinRangeReq in_rq = odin_range_issue(SCM != 0 ? p.in_amax : nullptr, lane);   // (finished behind the prologue's loads)
  // ---- the loads of the prologue go out FIRST: the 8 weight loads of a thread (fp32 [tap][co][32]) and the wave's
  // items of fill 0 (all rows of tile T0, offsets computed directly) are in flight while the zero fills and the table
  // arithmetic run -- the layers with 8- and 16-pixel rows run only 1-4 tiles per workgroup ----
  float4 wv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int e = tid + 512 * j;
    const int ci4 = e & 7, co = (e >> 3) & 31, tap = e >> 8;
    wv[j] = *reinterpret_cast<const float4*>(p.w + ((size_t)(tap * p.CO + n0 + co) * p.CS + p.ci_off + 4 * ci4));
  }
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * p.H * W * p.CS * 4));
  float amx = 0.f;  // running max |out| of this lane (the range word of `out`)
  const int f_c = wave & (CPR - 1), f_r0 = wave >> CSHIFT;
  constexpr int F_RJ = 8 >> CSHIFT;  // rows between a wave's two items
  const int f_px = 8 * f_c + (lane >> 3), f_ch4 = lane & 7, f_pc = f_px + 1;
  const int f_lds_lane = f_pc * 64 + ((((f_ch4 >> 1) ^ ((f_pc >> 2) & 3))) << 4) + (f_ch4 & 1) * 8;
  const unsigned f_g_lane = (unsigned)((f_px * p.CS + p.ci_off + 4 * f_ch4) * 4);
  const unsigned f_rowbytes = (unsigned)(W * p.CS * 4);
  TpItem itA[2], itB[2];
  {
    const int tpi = p.tiles_per_img;
    const int b0 = odin_div_small(T0, tpi), t0 = T0 - b0 * tpi;
    const int start = HP * b0 + RP * t0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = f_r0 + F_RJ * j, G = start + r;
      const bool valid = r < RP + 2;
      const int b = b0 + (RP * t0 + r >= HP ? 1 : 0), gi = G - b * HP;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      itA[j].dst = (G % NSLOT) * RB + f_lds_lane;
      itA[j].ok = valid;
      itA[j].v = odin_run_load4(IN, (real ? (unsigned)(G - b - 1) * f_rowbytes : 0xFFFF0000u) + f_g_lane);
    }
  }
  ODIN_SCHED_FENCE();
  // (the range word of the input: requested at the top, finished in front of the first split below.  A scaled input
  // is carried times 2^gk -- its maximum lands in [2^14, 2^15) -- and the two accumulators are scaled back)
  bool scl = SCM == 1;
  float in_s = 1.f, in_s2k = ODIN_LO_SCALE, out_s = 1.f, out_sx = ODIN_LO_UNSCALE;   // (set where the word is finished)

  // ---- SAME-padding pixels (pc = 0 and pc = W + 1) of every ring row and plane: zero for ever ----
  for (int e = tid; e < NSLOT * 8 * NPL; e += 512) {
    const int sl = e / (8 * NPL), rem = e - sl * (8 * NPL);
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(ring + sl * RB + pl * PB + (side ? (W + 1) * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- row fills: wave-uniform walk over the padded rows; a wave moves up to two 1 KB items ----
  // Which rows a fill moves and where they land (image seams, ring wrap-arounds) is index arithmetic: ~150 dependent
  // scalar instructions per tile when done inside the MFMA stream, where both waves of a SIMD executed them at the
  // same time and the matrix pipe idled behind them (fconv_planes.hip: 14 of 71 us).  It is done once here, by all
  // threads, into an LDS table; the tile loop reads its entries (wave-uniform addresses) and adds lane offsets.
  // Fill f >= 1 brings the rows tile T0 + f needs beyond those of tile T0 + f - 1, fill 0 all of tile T0's.
  constexpr int RPF = 2 * F_RJ;  // rows a fill can carry (row r = f_r0 + F_RJ j of item j)
  const int NF = p.tiles_per_wg + 3;
  TpEnt* tr = reinterpret_cast<TpEnt*>(ring + NSLOT * RB);  // [NF][RPF] row: (ring byte offset or -1, global byte offset)
  {
    const int tpi = p.tiles_per_img;
    for (int e = tid; e < NF * RPF; e += 512) {
      const int f = e / RPF, r = e - f * RPF;
      const int T = T0 + f, b1 = odin_div_small(T, tpi), t1 = T - b1 * tpi;
      const int end = HP * b1 + RP * t1 + RP + 2;
      int start = end - (RP + 2);
      if (f > 0) {
        const int b0 = t1 > 0 ? b1 : b1 - 1, t0 = t1 > 0 ? t1 - 1 : tpi - 1;  // tile T - 1: this image or the one before
        start = HP * b0 + RP * t0 + RP + 2;
      }
      const int G = start + r;  // global padded row HP * b + gi; gi == 0: the zero row between images
      const bool valid = T < T1 && G < end;
      const int b = odin_div_small(G, HP), gi = G - b * HP;
      const bool real = valid && gi != 0 && b < p.B;
      tr[e] = TpEnt{valid ? (G % NSLOT) * RB : -1, real ? (int)((unsigned)(G - b - 1) * f_rowbytes) : (int)0xFFFF0000u};
    }
  }
  auto fill_entries = [&](TpEnt (&en)[2], int f) {
#pragma unroll
    for (int j = 0; j < 2; ++j) en[j] = tr[f * RPF + f_r0 + F_RJ * j];
  };
  auto fill_loads = [&](TpItem (&it)[2], const TpEnt (&en)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      it[j].dst = en[j].x + f_lds_lane;
#ifdef ODIN_SIM
      it[j].ok = en[j].x >= 0;
#else
      it[j].ok = __builtin_amdgcn_readfirstlane(en[j].x) >= 0;  // (wave-uniform: a scalar branch around the stores)
#endif
      it[j].v = odin_run_load4(IN, (unsigned)en[j].y + f_g_lane);
    }
  };
  // (a wave-uniform branch: ~30 VALU + 3 LDS stores per item, and half of the 16 item slots of a tile are
  // empty -- the kernel is bound by instruction issue, profiles/r03_kpmc_planes_8wave.txt)
  auto store_fill1 = [&](const TpItem& it) {
    if (it.ok) {
      u32x2 h, l;
      if (SCM == 1 || (SCM == 2 && scl)) odin_split_h4<true>(it.v, in_s, in_s2k, h, l);
      else odin_split_h4<false>(it.v, 1.f, ODIN_LO_SCALE, h, l);
      char* d = ring + it.dst;
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + PB) = l;
    }
  };
  auto store_fill = [&](const TpItem (&it)[2]) {
    store_fill1(it[0]);
    store_fill1(it[1]);
  };

  // ---- this wave's pixel class: column parity x row parity, 32 pixels of the tile ----
  const int role = wave & 3, grp = wave >> 2;
  const int cpw = role & 1, rpar = role >> 1;
  // input row of the tile (per lane when a 32-pixel group spans 2 or 4 rows) and input column: ow = 2 i + cpw
  const int rp = (W == 32) ? grp : (W == 16) ? 2 * grp + (l31 >> 4) : 4 * grp + (l31 >> 3);
  const int i_in = (W == 32) ? l31 : (W == 16) ? (l31 & 15) : (l31 & 7);
  // column taps: parity 0 -> kw = 1 (padded column pc = i + 1), kw = 3 (pc = i); parity 1 -> kw = 0
  // (pc = i + 2), kw = 2 (pc = i + 1); row taps alike with kh / padded rows
  const int kw_a = cpw ? 0 : 1, kw_b = kw_a + 2;
  const int kh_a = rpar ? 0 : 1, kh_b = kh_a + 2;
  const int d_a = cpw ? 2 : 1;
  int offA[2], offB[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int pa = i_in + d_a, pb = pa - 1;
    offA[kk] = pa * 64 + (((2 * kk + half) ^ ((pa >> 2) & 3)) << 4);
    offB[kk] = pb * 64 + (((2 * kk + half) ^ ((pb >> 2) & 3)) << 4);
  }
  const int roff_a = rp + (rpar ? 2 : 1);  // padded row of row tap a relative to the tile's first
  const char* wlane = wl + half * 512 + l31 * 16;
  const int tap_aa = (kh_a * 4 + kw_a) * TP_TAPB, tap_ab = (kh_a * 4 + kw_b) * TP_TAPB;
  const int tap_ba = (kh_b * 4 + kw_a) * TP_TAPB, tap_bb = (kh_b * 4 + kw_b) * TP_TAPB;

  // ---- epilogue constants: accumulator register r holds channel n0 + (r & 3) + 8 (r >> 2) + 4 half.
  // Plain (unpacked) VALU arithmetic only: v_pk_*_f32 serialises with the bf16 matrix pipe, plain VALU
  // instructions run beside it, ~5.5 per 32-cycle MFMA per SIMD (tools/micro/mfma_bf16_mix.hip).
  auto ch_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * half; };
  constexpr float LOG2E = 1.44269504088896341f;
  // EPI 3 with several logit maps: the per-channel constants (bias | w1[oc][ch]) live in LDS and are
  // fetched four channels at a time right before their use (48 + 16 registers would not fit beside the
  // 16 C1 dW1 accumulators under 256 registers)
  constexpr bool CL = (EPI == 3 && C1 > 1);
  __shared__ float cst[CL ? 32 * (1 + C1) : 4];
  if (CL) {
    for (int e = tid; e < 32 * (1 + C1); e += 512) {
      const int ch = e & 31, oc = (e >> 5) - 1;
      cst[e] = e < 32 ? p.bias[n0 + ch] : p.w1[(n0 + ch) * C1 + oc];
    }
  }
  const float* cl = cst + 4 * half;  // this lane's channels of group q: cl[8 q .. 8 q + 3]
  float bias_r[((EPI == 1 || EPI == 3) && !CL) ? 16 : 1];
  if ((EPI == 1 || EPI == 3) && !CL) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = p.bias[n0 + ch_of(r)];
  }
  float w1r[(EPI == 3 && !CL) ? 16 : 1][(EPI == 3 && !CL) ? C1 : 1];
  float dw1[(EPI == 3) ? 16 : 1][(EPI == 3) ? C1 : 1];
  float b1r[(EPI == 3) ? C1 : 1];
  float db1[(EPI == 3) ? C1 : 1];
  if (EPI == 3) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int oc = 0; oc < C1; ++oc) {
        if (!CL) w1r[r][oc] = p.w1[(n0 + ch_of(r)) * C1 + oc];
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
  // global tensors behind buffer descriptors: the per-lane part of an address once per kernel, one scalar
  // (`soffset`) per tile
  const unsigned out_bytes = (unsigned)((size_t)p.B * OH * OW * p.CO * 4);
  const OdinRun OUT = odin_run(p.out, out_bytes);
  const OdinRun AUX = odin_run(EPI == 2 ? p.aux : nullptr, EPI == 2 ? out_bytes : 0u);
  const unsigned tgt_bytes = (EPI == 3) ? (unsigned)((size_t)p.B * OH * OW * C1 * 4) : 0u;
  const OdinRun TG = odin_run(EPI == 3 ? p.target : nullptr, tgt_bytes);
  const OdinRun LG = odin_run(EPI == 3 ? p.logits : nullptr, (EPI == 3 && p.logits != nullptr) ? tgt_bytes : 0u);

  // ---- prologue, second half: the weights -> planes [tap][plane][k-piece][co][8 f16], the first tile's rows ->
  // ring (both loaded at the top of the kernel), then ONE barrier publishes them together with the pads and the table
  if (SCM != 0) {
    // the input's range word (requested first thing in the kernel): the bound, the scale flag, the powers of two
    const unsigned in_mb = odin_range_finish(in_rq);
    scl = SCM == 1 || odin_act_needs_scale(in_mb);
    const int gk = scl ? odin_range_shift(in_mb) : 0;
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11);
    out_s = odin_pow2(-gk); out_sx = odin_pow2(-gk - 11);
  }
  // the bias as the C operand of a tile's first main-accumulator MFMA (times the input's power of two: the
  // accumulator holds scaled sums until the combine)
  f32x16 biasv = f32x16_zero();
  if ((EPI == 1 || EPI == 3) && !CL) {
#pragma unroll
    for (int r = 0; r < 16; ++r) biasv[r] = bias_r[r] * in_s;
  }
  {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = tid + 512 * j;
      const int ci4 = e & 7, co = (e >> 3) & 31, tap = e >> 8;
      u32x2 h, l;
      odin_split_h4<false>(wv[j], 1.f, ODIN_LO_SCALE, h, l);
      char* d = wl + tap * TP_TAPB + (ci4 >> 1) * 512 + co * 16 + (ci4 & 1) * 8;
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + 2048) = l;
    }
  }
  store_fill(itA);
  TP_STAMP(2);
  __syncthreads();  // weights, pads, the table and the first tile's rows are in LDS
  TP_STAMP(10);
  TpEnt en[2];
  fill_entries(en, 1);
  fill_loads(itA, en);

  int b_cur = T0 / p.tiles_per_img, t_cur = T0 - b_cur * p.tiles_per_img;
  int sl0 = (HP * b_cur + RP * t_cur) % NSLOT;  // ring slot of the tile's first padded row
  // this lane's pixel inside the tile's 2 RP output rows: (2 rp + rpar, 2 i + cpw)
  const unsigned pix_lane = (unsigned)((2 * rp + rpar) * OW + 2 * i_in + cpw);
  const unsigned out_lane = (pix_lane * p.CO + n0 + 4 * half) * 4, tgt_lane = pix_lane * C1 * 4;
  const unsigned lg_lane = (half == 0) ? tgt_lane : ODIN_OOB_V;  // the half == 0 lane of a pixel stores its logits
  unsigned tileP_out = 0, tileP_tgt = 0;  // scalar byte offsets of the previous tile in out / target
  // state of the PREVIOUS tile, whose epilogue rides in the current tile's MFMA stream
  f32x16 pa = f32x16_zero();
  float4 axP[4], pvP[4];
  float tgtP[(EPI == 3) ? C1 : 1] = {};
#pragma unroll
  for (int q = 0; q < 4; ++q) axP[q] = pvP[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  float dl[(EPI == 3) ? C1 : 1] = {};

  // The epilogue of one tile (lane = pixel (oh, 2 i + cpw) x 16 channels) as a list of micro-ops of a
  // few VALU instructions each: op k is issued right behind MFMA k of the next tile's stream (the
  // compiler's own interleaving bunched the MFMAs and left the VALU work exposed; sched_group_barrier
  // pipelines of this size are not honoured, so the order is fixed in the source, fence by fence).
  // It works in place on `pa`, two accumulator registers per op.
  auto elu_b = [&](int r, float b) __attribute__((always_inline)) {
    const float t = pa[r] + b;
    const float em1 = odin_exp2(t * LOG2E) - 1.f;
    pa[r] = t > 0.f ? t : em1;
  };
  // (the bias sits in the main accumulator from the tile's first MFMA on -- biasv below -- so the epilogue adds none:
  // 16 VALU instructions less per tile and wave)
  auto elu_r = [&](int r) { elu_b(r, 0.f); };
  char* plds = smem + p.part_off + tid * 16;   // (PL) this thread's 16-byte slot of a (tile, q) block of 512
  int tileP_loc = 0;                            // (PL) the previous tile's index inside this workgroup
  auto store_q = [&](int q) __attribute__((always_inline)) {
    if (DBG & 1) return;
    if (EPI >= 1) amx = odin_amax3(odin_amax3(amx, pa[4 * q], pa[4 * q + 1]), pa[4 * q + 2], pa[4 * q + 3]);
    if (PL && EPI == 0) {
      *reinterpret_cast<float4*>(plds + (tileP_loc * 4 + q) * 8192) =
          make_float4(pa[4 * q], pa[4 * q + 1], pa[4 * q + 2], pa[4 * q + 3]);
      return;
    }
    odin_run_store4s(OUT, out_lane + 32 * q, tileP_out,
                     make_float4(pa[4 * q], pa[4 * q + 1], pa[4 * q + 2], pa[4 * q + 3]));
  };
  float t_dot = 0.f, lgt = 0.f, eabs = 0.f;  // dot product, logit, exp(-|logit|) of the logit map in flight
  constexpr int N_EPI_OPS = (EPI == 3) ? 12 + 4 * C1 + 8 : 12;
  auto epi_op = [&](int k) {
    if (k < 8) {
      const int r0 = 2 * k, r1 = r0 + 1;
      if (ACC) {
        const int q = k >> 1;
        pa[r0] += (k & 1) ? pvP[q].z : pvP[q].x;
        pa[r1] += (k & 1) ? pvP[q].w : pvP[q].y;
      }
      if (EPI == 1 || EPI == 3) { elu_r(r0); elu_r(r1); }
      if (EPI == 2) {
        const int q = k >> 1;
        const float a0 = (k & 1) ? axP[q].z : axP[q].x, a1 = (k & 1) ? axP[q].w : axP[q].y;
        pa[r0] = fmaf(pa[r0], fminf(a0, 0.f), pa[r0]);  // x ELU'(aux) = 1 + min(aux, 0)
        pa[r1] = fmaf(pa[r1], fminf(a1, 0.f), pa[r1]);
        csum[r0] += pa[r0];
        csum[r1] += pa[r1];
      }
    }
    if (EPI <= 2) {
      if (k >= 8 && k < 12) store_q(k - 8);
    }
    if (EPI == 3) {
      // per logit map oc: 4 ops (dot, logistic terms, likelihood, its gradient)
      if (k >= 8 && k < 8 + 4 * C1) {
        const int oc = (k - 8) >> 2, ph = (k - 8) & 3;
        if (ph == 0) {
          t_dot = pa[0] * w1r[0][oc];
#pragma unroll
          for (int r = 1; r < 16; ++r) t_dot = fmaf(pa[r], w1r[r][oc], t_dot);
        }
        if (ph == 1) {
          lgt = tp_pairsum32(t_dot) + b1r[oc];  // the other 16 channels live in lane ^ 32
          eabs = odin_exp2(-LOG2E * fabsf(lgt));
          odin_run_store1s(LG, lg_lane + 4 * oc, tileP_tgt, lgt);
        }
        if (ph == 2) {
          // log p(x | logit) = x l - softplus(l); the half == 0 lane of a pixel owns the scalar results
          const float sp = fmaxf(lgt, 0.f) + 0.6931471805599453f * odin_log2(1.f + eabs);
          llk_lane += half == 0 ? tgtP[oc] * lgt - sp : 0.f;
        }
        if (ph == 3) {
          const float rr = odin_rcp(1.f + eabs);
          const float sg = lgt >= 0.f ? rr : eabs * rr;
          const float dsig = (sg - tgtP[oc]) * sc;
          db1[oc] += half == 0 ? dsig : 0.f;
          dl[oc] = dsig;
        }
      }
      constexpr int G0 = 8 + 4 * C1;
      if (k >= G0 && k < G0 + 8) {
#pragma unroll
        for (int r = 2 * (k - G0); r < 2 * (k - G0) + 2; ++r) {
          float gs = w1r[r][0] * dl[0];
          dw1[r][0] = fmaf(pa[r], dl[0], dw1[r][0]);
#pragma unroll
          for (int oc = 1; oc < C1; ++oc) {
            gs = fmaf(w1r[r][oc], dl[oc], gs);
            dw1[r][oc] = fmaf(pa[r], dl[oc], dw1[r][oc]);
          }
          pa[r] = fmaf(gs, fminf(pa[r], 0.f), gs);  // x ELU'(y) = 1 + min(y, 0)
          csum[r] += pa[r];
        }
      }
      if (k >= G0 + 8 && k < G0 + 12) store_q(k - G0 - 8);
    }
  };
  static_assert(CL || N_EPI_OPS + 8 <= 32, "epilogue micro-ops must fit before the row stores (steps 6, 7)");

  // ---- several logit maps (CL): the same epilogue with its constants fetched from LDS, laid out
  // slot by slot over the 48 MFMAs of the next tile.  A constant vector is read at least two MFMAs
  // before its use; wqA / wqB alternate between the channel groups q (4 channels of this lane each).
  float4 cbq = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 wqA[CL ? C1 : 1], wqB[CL ? C1 : 1];
  float t3[CL ? C1 : 1] = {}, lg3[CL ? C1 : 1] = {}, ea3[CL ? C1 : 1] = {};
  auto f4c = [](const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
  auto ld_bias = [&](int q) { cbq = *reinterpret_cast<const float4*>(cl + 8 * q); };
  auto ld_w = [&](float4 (&wq)[CL ? C1 : 1], int q) __attribute__((always_inline)) {
#pragma unroll
    for (int oc = 0; oc < (CL ? C1 : 1); ++oc)
      wq[oc] = *reinterpret_cast<const float4*>(cl + 32 + 32 * oc + 8 * q);
  };
  auto cl_elu2 = [&](int k) __attribute__((always_inline)) {  // accumulator registers 2 k, 2 k + 1
    elu_b(2 * k, (k & 1) ? cbq.z : cbq.x);
    elu_b(2 * k + 1, (k & 1) ? cbq.w : cbq.y);
  };
  auto cl_dot = [&](int q, const float4 (&wq)[CL ? C1 : 1]) __attribute__((always_inline)) {
#pragma unroll
    for (int oc = 0; oc < (CL ? C1 : 1); ++oc) {
      float t = (q == 0) ? pa[0] * wq[oc].x : fmaf(pa[4 * q], wq[oc].x, t3[oc]);
      t = fmaf(pa[4 * q + 1], wq[oc].y, t);
      t = fmaf(pa[4 * q + 2], wq[oc].z, t);
      t3[oc] = fmaf(pa[4 * q + 3], wq[oc].w, t);
    }
  };
  auto cl_lg = [&](int ph, int oc) __attribute__((always_inline)) {
    if (ph == 0) {
      lg3[oc] = tp_pairsum32(t3[oc]) + b1r[oc];  // the other 16 channels live in lane ^ 32
      ea3[oc] = odin_exp2(-LOG2E * fabsf(lg3[oc]));
      odin_run_store1s(LG, lg_lane + 4 * oc, tileP_tgt, lg3[oc]);
    }
    if (ph == 1) {
      const float sp = fmaxf(lg3[oc], 0.f) + 0.6931471805599453f * odin_log2(1.f + ea3[oc]);
      llk_lane += half == 0 ? tgtP[oc] * lg3[oc] - sp : 0.f;
    }
    if (ph == 2) {
      const float rr = odin_rcp(1.f + ea3[oc]);
      const float sg = lg3[oc] >= 0.f ? rr : ea3[oc] * rr;
      const float dsig = (sg - tgtP[oc]) * sc;
      db1[oc] += half == 0 ? dsig : 0.f;
      dl[oc] = dsig;
    }
  };
  auto cl_grad2 = [&](int i, const float4 (&wq)[CL ? C1 : 1]) __attribute__((always_inline)) {  // accumulator registers 2 i, 2 i + 1
#pragma unroll
    for (int r = 2 * i; r < 2 * i + 2; ++r) {
      float gs = f4c(wq[0], r & 3) * dl[0];
      dw1[r][0] = fmaf(pa[r], dl[0], dw1[r][0]);
#pragma unroll
      for (int oc = 1; oc < (CL ? C1 : 1); ++oc) {
        gs = fmaf(f4c(wq[oc], r & 3), dl[oc], gs);
        dw1[r][oc] = fmaf(pa[r], dl[oc], dw1[r][oc]);
      }
      pa[r] = fmaf(gs, fminf(pa[r], 0.f), gs);  // x ELU'(y) = 1 + min(y, 0)
      csum[r] += pa[r];
    }
  };
  auto cl_top = [&]() __attribute__((always_inline)) {
    ld_bias(0);
    ld_w(wqA, 0);
  };
  // slots 32-34, 36 and 42 belong to the row fills / operand loads / llk flush of run_tile
  auto cl_slot = [&](auto M) __attribute__((always_inline)) {
    constexpr int m = decltype(M)::value;
    if constexpr (m < 16) {
      if constexpr ((m & 1) == 0) {
        cl_elu2(m >> 1);
        if constexpr ((m & 3) == 2 && m < 14) ld_bias((m >> 2) + 1);
      } else if constexpr ((m & 3) == 3) {
        constexpr int q = m >> 2;
        if constexpr (q & 1) cl_dot(q, wqB); else cl_dot(q, wqA);
        if constexpr (q == 0) ld_w(wqA, 2);
        if constexpr (q == 1) ld_w(wqB, 3);
        if constexpr (q == 3) ld_w(wqA, 0);
      } else if constexpr (m == 1) {
        ld_w(wqB, 1);
      }
    } else if constexpr (m < 16 + 3 * C1) {
      cl_lg((m - 16) / C1, (m - 16) % C1);
      if constexpr (m == 16 + 3 * C1 - 1) ld_w(wqB, 1);
    } else {
      // the free slots behind the logistic terms, in order: 8 gradient slots, then the 4 output stores
      constexpr int n = tp_late_index(16 + 3 * C1, m);
      if constexpr (n >= 0 && n < 8) {
        if constexpr ((n >> 1) & 1) cl_grad2(n, wqB); else cl_grad2(n, wqA);
        if constexpr (n == 1) ld_w(wqA, 2);
        if constexpr (n == 3) ld_w(wqB, 3);
      }
      if constexpr (n >= 8 && n < 12) store_q(n - 8);
    }
  };

  // one log-likelihood partial per tile (a tile lies inside one sample): wave sums through LDS
  auto flush_llk = [&](int T) {
    if (EPI == 3) {
#ifndef ODIN_SIM
      asm volatile("; llk flush" ::: "memory");
#endif
      const float tt = odin_wave_sum64_valu(llk_lane);
      llk_red[T & 1][wave] = tt;  // (every lane holds the sum) summed over the 8 waves behind the next barrier
      llk_lane = 0.f;
    }
  };

  // one tile: loads of tile T + 2's rows, then 8 steps of [6 LDS reads of the next step, 6 MFMAs,
  // slice s of the previous tile's epilogue / of the row stores for tile T + 1]
  // GB (waves 4-7, the SIMD partners of waves 0-3): the same work with the epilogue ops 16 slots later
  // and the row stores early, so that the two waves of a SIMD are not in their VALU-dense / scalar
  // phases at the same time (MI355X_MICROARCH.md, two waves per SIMD, item 9)
  auto run_tile = [&](auto with_epi, auto group_b, int T) {
    constexpr bool WE = decltype(with_epi)::value;
    constexpr bool GB = decltype(group_b)::value && !CL;
    constexpr int ESH = GB ? 16 : 0;                                    // epilogue shift
    constexpr int SF0 = GB ? 4 : 36, SF1 = GB ? 10 : 42, FL = GB ? 47 : 34;  // row stores, llk flush
    int sa = sl0 + roff_a, sb = sa - 1;
    if (sa >= NSLOT) sa -= NSLOT;
    if (sb >= NSLOT) sb -= NSLOT;
    const char* row_a = ring + sa * RB;
    const char* row_b = ring + sb * RB;
    // two accumulator chains: main (h x h) and cross (h x l + l x h, carried times 2^11)
    f32x16 acc = ((EPI == 1 || EPI == 3) && !CL) ? biasv : f32x16_zero(), acx = f32x16_zero();
    u32x4 fa[2][NPL], fb[2][NPL];
    // 8 steps = 4 taps (a/a, a/b, b/a, b/b) x 2 k-halves of 16 channels
    auto loads = [&](int s, u32x4 (&A)[NPL], u32x4 (&Bf)[NPL]) {
      const int tp = s >> 1, kk = s & 1;
      const bool ra = tp < 2, ca = (tp & 1) == 0;
      const char* bp = (ra ? row_a : row_b) + (ca ? offA[kk] : offB[kk]);
      const char* ap = wlane + (ra ? (ca ? tap_aa : tap_ab) : (ca ? tap_ba : tap_bb)) + kk * 1024;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        A[pl] = *reinterpret_cast<const u32x4*>(ap + pl * 2048);
        Bf[pl] = *reinterpret_cast<const u32x4*>(bp + pl * PB);
      }
    };
    loads(0, fa[0], fb[0]);  // first thing after the barrier: everything else waits behind the MFMAs
    if (WE && CL) cl_top();
    ODIN_SCHED_FENCE();
    const unsigned tile_pix = (unsigned)((b_cur * OH + 2 * RP * t_cur) * OW);
    const unsigned tile_out = tile_pix * (unsigned)p.CO * 4u, tile_tgt = tile_pix * (unsigned)C1 * 4u;
    float4 axN[4], pvN[4];
    float tgtN[(EPI == 3) ? C1 : 1] = {};
#pragma unroll
    for (int q = 0; q < 4; ++q) axN[q] = pvN[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    ODIN_SCHED_FENCE();
    // what rides behind the MFMAs, by SLOT: the slot schedule was laid out for the 48 MFMAs of the three-plane form and
    // is kept -- MFMA M of the 24 carries slots 2 M and 2 M + 1
    auto slot = [&](auto MS) __attribute__((always_inline)) {
      constexpr int m = decltype(MS)::value;
      // the 8 activation ops (2 x ~6 VALU + 2 transcendentals each) in every other slot of the first 16, the remaining
      // ops one per slot, the row stores in the last two steps
      constexpr int me = m - ESH;
      constexpr int k = me < 0 ? -1 : (me < 16 ? ((me & 1) ? -1 : me / 2) : me - 8);
      if constexpr (WE && !CL && k >= 0 && k < N_EPI_OPS && !(DBG & 4)) epi_op(k);
      if constexpr (WE && CL && !(DBG & 4)) cl_slot(MS);
      if (m == 20) fill_entries(en, T - T0 + 2);
      if (m == 32) fill_loads(itB, en);  // global loads of tile T + 2's rows
      if (m == 33) {                            // this tile's epilogue operands (used one tile later)
        if (EPI == 2) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            axN[q] = odin_run_load4s(AUX, out_lane + 32 * q, tile_out);
        }
        if (EPI == 3) {
#pragma unroll
          for (int oc = 0; oc < C1; ++oc) tgtN[oc] = odin_run_load1s(TG, tgt_lane + 4 * oc, tile_tgt);
        }
        if (ACC && PL) {
#pragma unroll
          for (int q = 0; q < 4; ++q) pvN[q] = *reinterpret_cast<const float4*>(plds + ((T - T0) * 4 + q) * 8192);
        }
        if (ACC && !PL) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            pvN[q] = odin_run_load4s(OUT, out_lane + 32 * q, tile_out);
        }
      }
      // (the log-likelihood partial of a sample is flushed once, behind its last tile)
      if (WE && m == FL && t_cur == 0) flush_llk(T - 1);
      if (m == SF0) store_fill1(itA[0]);
      if (m == SF1) store_fill1(itA[1]);
    };
    tp_static_for<24>([&](auto MM) __attribute__((always_inline)) {
      constexpr int M = decltype(MM)::value;
      constexpr int s = M / 3, u = M % 3, cur = s & 1, nxt = cur ^ 1;
      if constexpr (u == 0) {
        // the four reads of step s + 1 go out before the MFMAs of step s (a read issued one MFMA before its
        // use exposes the LDS latency)
        if (s + 1 < 8 && !(DBG & 2)) loads(s + 1, fa[nxt], fb[nxt]);
        if (s + 1 < 8 && (DBG & 2)) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) { fa[nxt][pl] = fa[cur][pl]; fb[nxt][pl] = fb[cur][pl]; }
        }
      }
      // plane products (weights x pixels): h*l and l*h into the cross accumulator, h*h into the main one
      if (u == 0) acx = mfma32_f16(fa[cur][0], fb[cur][1], acx);
      if (u == 1) acc = mfma32_f16(fa[cur][0], fb[cur][0], acc);
      if (u == 2) acx = mfma32_f16(fa[cur][1], fb[cur][0], acx);
      slot(std::integral_constant<int, 2 * M>{});
      slot(std::integral_constant<int, 2 * M + 1>{});
      ODIN_SCHED_FENCE();
#ifdef ODIN_DIAG
      if (M == 0 || M == 5 || M == 11 || M == 17 || M == 23) TP_STAMP(12 + (M + 1) / 6);
#endif
    });
    if (SCM == 1 || (SCM == 2 && scl)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pa[r] = fmaf(acx[r], out_sx, acc[r] * out_s);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) pa[r] = fmaf(acx[r], ODIN_LO_UNSCALE, acc[r]);
    }
    tileP_out = tile_out;
    tileP_tgt = tile_tgt;
    tileP_loc = T - T0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { axP[q] = axN[q]; pvP[q] = pvN[q]; }
#pragma unroll
    for (int oc = 0; oc < ((EPI == 3) ? C1 : 1); ++oc) tgtP[oc] = tgtN[oc];
    itA[0] = itB[0];
    itA[1] = itB[1];
  };

  // the persistent tile loop, once per wave group: the branch on the group sits OUTSIDE the loop (a branch
  // per tile made the register allocator keep both variants' state apart: +45 registers)
  auto tile_loop = [&](auto group_b) __attribute__((always_inline)) {
    run_tile(TpNo{}, group_b, T0);
    TP_STAMP(11);
    __syncthreads();
    TP_STAMP(10);
#pragma unroll 1
    for (int T = T0 + 1; T < T1; ++T) {
      sl0 += RP;
      if (++t_cur == p.tiles_per_img) { t_cur = 0; ++b_cur; ++sl0; }
      if (sl0 >= NSLOT) sl0 -= NSLOT;
      run_tile(TpYes{}, group_b, T);
      TP_STAMP(11);
      __syncthreads();  // every wave is past tile T's rows; tile T + 1's rows are stored
      TP_STAMP(10);
      if (EPI == 3 && tid == 0) {
        // tile T - 1 closed a sample: its slot carries the sample's (this workgroup's share of the) sum,
        // the other tiles' slots a zero
        const float* q = llk_red[(T - 1) & 1];
        p.llk_part[T - 1] =
            t_cur == 0 ? ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])) : 0.f;
      }
    }
  };
  if (!CL && wave >= 4) tile_loop(TpYes{}); else tile_loop(TpNo{});
  if (CL) {
    cl_top();
    tp_static_for<48>([&](auto M) __attribute__((always_inline)) { cl_slot(M); });
  } else {
#pragma unroll
    for (int k = 0; k < N_EPI_OPS; ++k) epi_op(k);
  }
  flush_llk(T1 - 1);

#if defined(ODIN_DIAG) && !defined(ODIN_SIM)
  if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    p.stamps[66] = clock64();
    p.stamps[67] = wall_clock64();
  }
#endif
  if (EPI >= 1) {   // (EPI 1: the range word of the activation, odin_conv_desc.y_amax; EPI 2 / 3: of the gradient)
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred, blockIdx.x + gridDim.x * blockIdx.y);
    __syncthreads();  // (cred is reused below)
  }
  // ---- per-workgroup partial sums: 32 pixel lanes by shuffles, then the 8 waves through LDS ----
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
      for (int wv = 0; wv < 8; ++wv) tt += cred[wv * NRED + tid];
      p.colsum[(size_t)blockIdx.x * p.CO + n0 + tid] = tt;
    }
  }
  if (EPI == 3) {
    __syncthreads();
    if (tid == 0) {
      const float* q = llk_red[(T1 - 1) & 1];
      p.llk_part[T1 - 1] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    // slab row: [dW1 (CO, C1) | db1 (C1) | column sums of out (CO)]
    float* row = p.slab + (size_t)blockIdx.x * (p.CO * C1 + C1 + p.CO);
    for (int e = tid; e < 32 + 32 * C1 + C1; e += 512) {
      float tt = 0.f;
      for (int wv = 0; wv < 8; ++wv) tt += cred[wv * NRED + e];
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

template <int EPI, int C1, int W, int DBG = 0, bool ACC = false, int SCM = 0>
__global__ __launch_bounds__(512) void tconv_planes_kernel(TPParams p) {
  tp_body<EPI, C1, W, DBG, ACC, SCM>(p);
}

// 64 reduction channels in ONE launch: both 32-channel passes inside the kernel (round 3 launched them separately: a
// second launch + prologue drain per 64-channel layer, 4 layers per dSprites step).  A tile's raw partial sums of the
// first pass are written to `out` and read back in the second pass by the SAME thread (same wave roles, same tile
// walk); the fence + barrier between the passes also keeps the second prologue's LDS writes behind the first pass's
// last LDS reads.
template <int EPI, int W, int SCM, bool PL>
__global__ __launch_bounds__(512) void tconv_planes2_kernel(TPParams p) {
  {
    TPParams q = p;
    q.colsum = nullptr;
    q.out_amax = nullptr;
    q.ci_off = 0;
    tp_body<0, 1, W, 0, false, SCM, PL>(q);
  }
  // (the same thread reads back what it wrote, through the same CU's write-through L1 and its XCD's L2: a
  // workgroup-scope fence orders it; a device-scope __threadfence() writes back and invalidates the whole L2 of the
  // XCD and cost 50 us per launch)
  odin_wait_vmem();
#ifndef ODIN_SIM
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  __syncthreads();
  p.ci_off = 32;
  tp_body<EPI, 1, W, 0, true, SCM, PL>(p);
}

// LDS: weight planes + row ring + the fill table (rows per fill x 8 bytes per fill, tiles + 3 fills); 4.3 KB are static
constexpr int TP_LDS_MAX = 152 * 1024;   // (dynamic; + up to 2 x 4.3 KB static in the two-pass kernels)
int tp_ring_bytes(int W) { return TP_WBYTES + (2 * (64 / W) + 3) * TP_NPL * (W + 2) * 64; }
int tp_fill_bytes(int W) { return 8 * 2 * (W == 32 ? 2 : W == 16 ? 4 : 8); }
// tiles per workgroup: the chip filled once when the table fits, more workgroups otherwise; -1: does not fit
int tp_tiles_per_wg(int W, int n_tiles, int gy) {
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  int tpw = (n_tiles + cap - 1) / cap;
  const int limit = (TP_LDS_MAX - tp_ring_bytes(W)) / tp_fill_bytes(W) - 3;
  if (tpw > limit) tpw = limit;
  if ((n_tiles + tpw - 1) / tpw > ODIN_MAX_COLSUM_BLOCKS) return -1;
  return tpw;
}

template <int EPI, int C1, bool ACC, int SC>
int tp_launch_w(const TPParams& p, int W, dim3 grid, void* stream) {
  const size_t lds = (size_t)tp_ring_bytes(W) + (size_t)(p.tiles_per_wg + 3) * tp_fill_bytes(W);  // + the fill table
  constexpr int W3 = (EPI == 3 ? 16 : 8);  // (the fused tail has no 8-pixel geometry)
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[3] = {reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 32, 0, ACC, SC>),
                          reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 16, 0, ACC, SC>),
                          reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, W3, 0, ACC, SC>)};
    for (int i = 0; i < 3; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS_MAX) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#ifdef ODIN_DIAG
  // diagnostics build only (make FLAGS_tconv_planes+=-DODIN_DIAG): instances with parts of the kernel
  // switched off, selected by ODIN_TP_DBG -- they compute WRONG results and are not in the product library
  if (W == 32 && EPI == 3 && C1 == 1) {
    static const int dbg = [] { const char* e = ODIN_DIAG_ENV("ODIN_TP_DBG"); return e ? atoi(e) : 0; }();
    static bool dattr = false;
    if (!dattr) {
      const void* dfn[4] = {reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 32, 1, ACC, SC>),
                            reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 32, 2, ACC, SC>),
                            reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 32, 4, ACC, SC>),
                            reinterpret_cast<const void*>(&tconv_planes_kernel<EPI, C1, 32, 7, ACC, SC>)};
      for (int i = 0; i < 4; ++i)
        if (hipFuncSetAttribute(dfn[i], hipFuncAttributeMaxDynamicSharedMemorySize, TP_LDS_MAX) != hipSuccess)
          (void)hipGetLastError();
      dattr = true;
    }
    if (dbg == 1) { ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 32, 1, ACC, SC>), grid, dim3(512), lds, stream, p); return odin_check_launch("tconv_planes(f16x2)"); }
    if (dbg == 2) { ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 32, 2, ACC, SC>), grid, dim3(512), lds, stream, p); return odin_check_launch("tconv_planes(f16x2)"); }
    if (dbg == 4) { ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 32, 4, ACC, SC>), grid, dim3(512), lds, stream, p); return odin_check_launch("tconv_planes(f16x2)"); }
    if (dbg == 7) { ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 32, 7, ACC, SC>), grid, dim3(512), lds, stream, p); return odin_check_launch("tconv_planes(f16x2)"); }
  }
#endif
#endif
  if (W == 32) ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 32, 0, ACC, SC>), grid, dim3(512), lds, stream, p);
  else if (W == 16) ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, 16, 0, ACC, SC>), grid, dim3(512), lds, stream, p);
  else ODIN_LAUNCH((tconv_planes_kernel<EPI, C1, W3, 0, ACC, SC>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("tconv_planes(f16x2)");
}

// the two-pass kernels: dynamic LDS up to 160 000 bytes (2 x 1.2 KB of static arrays on top: 160 KiB per CU)
constexpr int TP2_LDS_MAX = 160000;
// tiles per workgroup whose first-pass partial sums (32 KB per tile) fit in LDS beside the weight planes, the ring and the
// fill table; 0: none
int tp_pl_tiles(int W) {
  const int free_b = TP2_LDS_MAX - tp_ring_bytes(W) - 8 * tp_fill_bytes(W);
  return free_b < 32768 ? 0 : free_b / 32768;
}

template <int EPI, int SC, bool PL>
int tp_launch2_w(const TPParams& p0, int W, dim3 grid, void* stream) {
  TPParams p = p0;
  size_t lds = (size_t)tp_ring_bytes(W) + (size_t)(p.tiles_per_wg + 3) * tp_fill_bytes(W);
  if (PL) {
    lds = (lds + 15) & ~(size_t)15;
    p.part_off = (int)lds;
    lds += (size_t)p.tiles_per_wg * 32768;
  }
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[3] = {reinterpret_cast<const void*>(&tconv_planes2_kernel<EPI, 32, SC, PL>),
                          reinterpret_cast<const void*>(&tconv_planes2_kernel<EPI, 16, SC, PL>),
                          reinterpret_cast<const void*>(&tconv_planes2_kernel<EPI, 8, SC, PL>)};
    for (int i = 0; i < 3; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, TP2_LDS_MAX) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  if (W == 32) ODIN_LAUNCH((tconv_planes2_kernel<EPI, 32, SC, PL>), grid, dim3(512), lds, stream, p);
  else if (W == 16) ODIN_LAUNCH((tconv_planes2_kernel<EPI, 16, SC, PL>), grid, dim3(512), lds, stream, p);
  else ODIN_LAUNCH((tconv_planes2_kernel<EPI, 8, SC, PL>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("tconv_planes(f16x2)");
}

}  // namespace

static long long* g_tp_stamps = nullptr;
void odin_tconv_planes_set_stamps(void* buf) { g_tp_stamps = (long long*)buf; }

// ODIN_SPLIT (any value) and ODIN_NOPLANES select the older instances (gather_conv.hip)
bool odin_tconv_planes_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt,
                                  int pl, int center, int epi, int C1) {
  if (odin_blk_first()) return false;   // (diagnostics: odin_debug_blk_first)
  // (read per call: the A/B tests switch paths inside one process; a captured graph never comes here)
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NOPLANES") || ODIN_DIAG_ENV("ODIN_SPLIT")) return false;
  if (epi == 3 && (CO != 32 || (C1 != 1 && C1 != 3) || CI != 32 || W == 8)) return false;
  return KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && (CI == 32 || CI == 64) && (CO % 32) == 0 &&
         !center && (W == 8 || W == 16 || W == 32) && (H % (64 / W)) == 0 && (size_t)B * H * W * CI * 4 < (1ull << 31) &&
         (size_t)B * H * W * 4 * C1 * 4 < (1ull << 31) && tp_tiles_per_wg(W, B * (H / (64 / W)), CO / 32) > 0;
}

// epi 1: deconv forward (bias + ELU); 2: conv data gradient (x ELU'(aux), column sums); 3: fused tail.
// CI = 64: two reduction passes over 32 channels each (weight planes of 64 channels: 192 KB): the first
// leaves raw partial sums in `out`, the second adds them and runs the epilogue.
int odin_tconv_planes_launch(const float* in, const float* w, const float* bias, const float* aux,
                             float* out, float* colsum, int* rows_out, const float* w1, const float* b1,
                             const float* target, float* logits, float* llk_part, int* n_part_out,
                             float* slab, const float* scale, int C1, int B, int H, int W, int CI,
                             int CO, int epi, const uint32_t* in_amax, uint32_t* out_amax, void* stream) {
  TPParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.w1 = w1; p.b1 = b1; p.target = target; p.logits = logits; p.llk_part = llk_part; p.slab = slab;
  p.scale = scale;
  p.B = B; p.H = H; p.CO = CO;
  p.CS = CI; p.ci_off = 0;
  p.stamps = g_tp_stamps;
  const int RP = 64 / W;
  p.tiles_per_img = H / RP;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  p.tiles_per_wg = tp_tiles_per_wg(W, p.n_tiles, gy);
  if (p.tiles_per_wg <= 0) return odin_fail(-2, "tconv_planes: too many tiles for the fill table");
  // 64 reduction channels: the first pass's partial sums stay in LDS when a workgroup's tiles fit there -- with fewer
  // tiles per workgroup (more workgroups than CUs) if need be, as long as the slab rows allow
  bool pl = false;
  if (CI == 64 && !ODIN_DIAG_ENV("ODIN_TP_NOPL")) {
    const int cap = tp_pl_tiles(W);
    if (cap >= 1) {
      int tpw = p.tiles_per_wg < cap ? p.tiles_per_wg : cap;
      if ((p.n_tiles + tpw - 1) / tpw <= ODIN_MAX_COLSUM_BLOCKS) { p.tiles_per_wg = tpw; pl = true; }
    }
  }
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (n_part_out) *n_part_out = p.tiles_per_img;
  if (out == nullptr) return 0;  // dry run
  if (epi == 2) {
    p.in_amax = odin_range_word_of(in, (size_t)B * H * W * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "tconv_planes: no range word for the gradient input");
  }
  p.out_amax = out_amax;
  // forward launches (epi 1, 3): the activation's word where the caller has one (scaled only outside the safe window)
  const bool aw = epi != 2 && in_amax != nullptr;
  if (aw) p.in_amax = in_amax;
  dim3 grid(gx, gy, 1);
  if (CI == 64) {
    if (epi == 3) return odin_fail(-2, "tconv_planes tail: 32 input channels only");
#ifdef ODIN_DIAG  // diagnostics build: A/B against the two-launch form of round 3
    if (ODIN_DIAG_ENV("ODIN_TP_2LAUNCH")) {
      TPParams q = p;
      q.colsum = nullptr;
      q.out_amax = nullptr;
      const int rc = epi == 1 ? tp_launch_w<0, 1, false, 0>(q, W, grid, stream) : tp_launch_w<0, 1, false, 1>(q, W, grid, stream);
      if (rc != 0) return rc;
      p.ci_off = 32;
      return epi == 1 ? tp_launch_w<1, 1, true, 0>(p, W, grid, stream) : tp_launch_w<2, 1, true, 1>(p, W, grid, stream);
    }
#endif
    if (pl) {
      if (epi == 1) return aw ? tp_launch2_w<1, 2, true>(p, W, grid, stream) : tp_launch2_w<1, 0, true>(p, W, grid, stream);
      return tp_launch2_w<2, 1, true>(p, W, grid, stream);
    }
    if (epi == 1) return aw ? tp_launch2_w<1, 2, false>(p, W, grid, stream) : tp_launch2_w<1, 0, false>(p, W, grid, stream);
    return tp_launch2_w<2, 1, false>(p, W, grid, stream);
  }
  if (epi == 1) return aw ? tp_launch_w<1, 1, false, 2>(p, W, grid, stream) : tp_launch_w<1, 1, false, 0>(p, W, grid, stream);
  if (epi == 2) return tp_launch_w<2, 1, false, 1>(p, W, grid, stream);
  if (C1 == 1) return aw ? tp_launch_w<3, 1, false, 2>(p, W, grid, stream) : tp_launch_w<3, 1, false, 0>(p, W, grid, stream);
  if (C1 == 3) return aw ? tp_launch_w<3, 3, false, 2>(p, W, grid, stream) : tp_launch_w<3, 3, false, 0>(p, W, grid, stream);
  return odin_fail(-2, "tconv_planes tail: one or three logit maps only");
}
