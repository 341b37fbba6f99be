// fconv_planes.hip -- STRIDED gather convolution 4x4 / stride 2 over 32 reduction channels (TF `SAME`,
// pads (1, 1)) with fp32 operands carried through the f16 matrix pipe as two f16 planes (odin_device.h:
// x = h + 2^-11 l; three v_mfma_f32_32x32x16_f16 per 16 k-values into a main and a cross accumulator):
//   Conv2D(k4, s2) forward over 32 input channels                    (image_networks.py:462-468)
//   Conv2DTranspose(k4, s2) DATA GRADIENT over 32 output channels     (tape.gradient of decoder3/4)
//       out[b, oh, ow, n] = sum over (kh, kw, c < 32) of in[b, 2 oh - 1 + kh, 2 ow - 1 + kw, c] * W[kh, kw, c, n]
//
// The strided gather reads an input area 4x the output area, so the LDS cannot hold the weight
// planes (96 KB) beside a useful row window.  Instead the REDUCTION is split over the 8 waves of a
// workgroup: wave v owns the taps (kh = v >> 1, kw = 2 (v & 1) + {0, 1}) and keeps their weight
// fragments -- 2 taps x 2 k-halves x 2 planes = 32 registers -- for the whole kernel; all waves
// multiply the same 32-pixel x 32-channel output tile (12 MFMAs each), leave their partial tiles in
// LDS and meet at the tile's barrier; behind it wave v sums the eight partials of accumulator
// registers 2 v, 2 v + 1 and runs their epilogue while the next tile's MFMAs are already going (two
// scratch buffers).  The row window is the one of wgrad_planes.hip (fine rows as two column-parity
// planes of [slot][32 bf16], split once on the way in), here with the 16-byte k-pieces XOR-swizzled by
// (slot >> 2) for conflict-free ds_read_b128.  One workgroup barrier per tile.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct FPParams {
  const float* in;     // [B, 2 OH, 2 OW, 32]
  const float* w;      // [16 taps][32][CO]
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, OH, OW, CO], out *= ELU'(aux)
  float* out;          // [B, OH, OW, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] partial column sums of out (may be null)
  int B, OH, CO;
  int CS, ci_off;      // channels of the input tensor; first of the 32 this pass reduces over
  int tiles_per_img, n_tiles, tiles_per_wg;
  const unsigned* in_amax;  // SC instances: the input is a gradient tensor; its range word (odin_device.h)
  unsigned* out_amax;       // EPI 2: range word of `out` (may be null)
  long long* stamps;   // diagnostics build: s_memtime stamps of workgroup 0, [wave][32]
};

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define FP_STAMP(k) ((void)0)
#else
#define FP_STAMP(k)                                                                               \
  do {                                                                                            \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 32)        \
      p.stamps[32 * wave + stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

struct alignas(8) FpEnt {
  int x, y;
};

struct FpItem {
  float4 v;
  int dst;  // byte offset of the hi-plane store inside the ring; < 0: no item (wave-uniform)
};

constexpr int FP_MAXU = 4;

// EPI 1: bias + ELU (Conv2D forward); EPI 2: x ELU'(aux), column sums (deconv data gradient); EPI 0: raw partial sums
// (first of two reduction passes over 64 input channels); ACC: add the partial sums the previous pass left in `out`
// before the epilogue.
// SCM 1: the input tensor is a gradient (scaled by 2^gk on its way into the planes, the result scaled back); SCM 2: an
// ACTIVATION with its range word: scaled the same way only when its bound leaves [2^-8, 2^15) -- a wave-uniform flag,
// one scalar branch around each split and each scale-back (tconv_planes.hip: why not two bodies behind one branch)
template <int EPI, int OW, bool ACC, int SCM>
__device__ __forceinline__ void fp_body(const FPParams& p) {
  constexpr int NPL = 2;                 // f16 planes per operand
  constexpr int TC = 32 / OW;            // output rows per tile
  constexpr int WU = 2 * OW;             // input row length
  constexpr int SU = OW + 1;             // slots per column-parity plane
  // one parity plane + one spare slot: the two parity planes then sit 128 bytes apart modulo the 256-byte bank
  // row, so the 8 pixels of a row-fill item (alternating parity) spread over all banks -- with SU * 64 the pixel
  // pairs (1, 2), (3, 4), (5, 6) shared their banks (15-19 % of the LDS cycles of fconv_planes / wgrad_planes)
  constexpr int PARB = (SU + 1) * 64;
  constexpr int PBU = 2 * PARB;
  constexpr int RBU = NPL * PBU;
  constexpr int NSU = 4 * TC + 3;        // live input rows (2 TC + 2) + the next tile's (2 TC + 1 at an image seam)
  constexpr int IPU = WU / 8;            // 1 KB load items per input row
  constexpr int RJ = 8 / IPU > 0 ? 8 / IPU : 1;
  constexpr int RED = 8 * 8 * 64 * 8;    // one reduction buffer: [register pair][wave][lane][8 B]
  ODIN_DYN_SMEM(char, smem);
  char* ring = smem;
  char* red = smem + NSU * RBU;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const int HU = 2 * p.OH, HPU = HU + 1;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;
  int stamp_i = 0;
  (void)stamp_i;
  FP_STAMP(1);

  const OdinRangeReq in_rq = odin_range_issue(SCM != 0 ? p.in_amax : nullptr, lane);   // (finished behind the prologue's loads)
  // ---- the loads of the prologue go out FIRST: this wave's weight fragments (taps (kh, kw0), (kh, kw0 + 1); lane =
  // output channel l31, k = 8 half + e) and its items of fill 0 (all rows of tile T0, offsets computed directly) are in
  // flight while the zero fills and the table arithmetic below run -- a cold L2 answers in ~1 us, and the layers with
  // 8- and 16-pixel rows run only 1-4 tiles per workgroup ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1);
  float wv[2][2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int tap = kh * 4 + kw0 + t;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        wv[t][kk][e] = p.w[((size_t)(tap * p.CS + p.ci_off + 16 * kk + 8 * half + e)) * p.CO + n0 + l31];
    }
  const OdinRun RU = odin_run(p.in, (unsigned)((size_t)p.B * HU * WU * p.CS * 4));
  float amx = 0.f;  // running max |out| of this lane (EPI 1 / 2: the range word of `out`)
  constexpr int RPF = FP_MAXU * RJ;      // rows a fill can carry (row r = r0w + RJ j of item j)
  constexpr int DST_NONE = -(1 << 24);   // ring offset of an item without a row: dst stays negative
  constexpr unsigned OFF_NONE = 0x7FFF0000u;  // global offset of a row that is not read (padding row, no image): out of range
  const unsigned u_rowbytes = (unsigned)(WU * p.CS * 4);
  const int ch4 = lane & 7, pxl = lane >> 3;
  const int r0w = wave / IPU, cblk = wave - r0w * IPU;
  const int pcw = 8 * cblk + pxl + 1;  // padded column of this lane's pixel: parity pcw & 1, slot pcw >> 1
  const int u_lds = (pcw & 1) * PARB + (pcw >> 1) * 64 + (((ch4 >> 1) ^ (((pcw >> 1) >> 2) & 3)) << 4) + (ch4 & 1) * 8;
  const unsigned u_g = (unsigned)(((8 * cblk + pxl) * p.CS + p.ci_off + 4 * ch4) * 4);
  FpItem iuA[FP_MAXU], iuB[FP_MAXU], iuC[FP_MAXU];
  {
    const int tpi = p.tiles_per_img;
    const int b0 = odin_div_small(T0, tpi), t0 = T0 - b0 * tpi;
    const int start = HPU * b0 + 2 * TC * t0;
#pragma unroll
    for (int j = 0; j < FP_MAXU; ++j) {
      const int r = r0w + RJ * j, G = start + r;
      const bool valid = r < 2 * TC + 2;
      const int b = b0 + (2 * TC * t0 + r >= HPU ? 1 : 0), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      const int slot = G - odin_div_small(G, NSU) * NSU;
      iuA[j].dst = valid ? slot * RBU + u_lds : -1;
      iuA[j].v = odin_run_load4(RU, real ? (unsigned)(G - b - 1) * u_rowbytes + u_g : ODIN_OOB);
    }
  }
  ODIN_SCHED_FENCE();
  // (the range word of the input: requested at the top, finished in front of the first split below.  A scaled input
  // is carried times 2^gk -- its maximum lands in [2^14, 2^15) -- and the sums are scaled back)
  bool sc = SCM == 1;
  float in_s = 1.f, in_s2k = ODIN_LO_SCALE, out_s = 1.f;   // (set where the word is finished: in front of the first split)

  // ---- SAME-padding slots (parity plane 0 slot 0, parity plane 1 slot OW) of every ring row and plane ----
  for (int e = tid; e < NSU * 8 * NPL; e += 512) {
    const int sl = e / (8 * NPL), rem = e - sl * (8 * NPL);
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(ring + sl * RBU + pl * PBU + (side ? PARB + OW * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }

  for (int e = tid; e < RED / 16; e += 512)
    *reinterpret_cast<float4*>(red + ((T0 - 1) & 1) * RED + e * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- row fills (as wgrad_planes.hip; the k-pieces of a pixel slot are swizzled by the slot) ----
  // Which rows a fill moves, where they land in the ring and where each tile starts is pure index arithmetic with
  // image seams and ring wrap-arounds: ~170 dependent scalar instructions per tile when done between the MFMAs,
  // 14 of the kernel's 71 us (profiles/r03_fconv_planes_bookkeeping.txt).  It is done ONCE here, by all threads in
  // parallel, into two small LDS tables; the tile loop reads its entries (wave-uniform addresses) and adds lane
  // offsets.  Fill f >= 1 brings the rows tile T0 + f needs beyond those of tile T0 + f - 1; fill 0 all of tile T0's.
  const int NF = p.tiles_per_wg + 4;
  FpEnt* tt = reinterpret_cast<FpEnt*>(red + 2 * RED);  // [NF] tile: (first ring slot, byte offset of its first output)
  FpEnt* tr = tt + NF;                                 // [NF][RPF] fill row: (ring byte offset, global byte offset)
  {
    const int tpi = p.tiles_per_img;
    for (int e = tid; e < NF; e += 512) {
      const int T = T0 + e, b = odin_div_small(T, tpi), t = T - b * tpi;
      const int g0 = HPU * b + 2 * TC * t;
      tt[e] = FpEnt{g0 - odin_div_small(g0, NSU) * NSU, (int)(((size_t)(b * p.OH + TC * t) * OW) * p.CO * 4)};
    }
    for (int e = tid; e < NF * RPF; e += 512) {
      const int f = e / RPF, r = e - f * RPF;
      const int T = T0 + f, b1 = odin_div_small(T, tpi), t1 = T - b1 * tpi;
      const int end = HPU * b1 + 2 * TC * t1 + 2 * TC + 2;
      int start = end - (2 * TC + 2);
      if (f > 0) {  // (tile T - 1 lies in image b1 or in the one before it)
        const int b0 = t1 > 0 ? b1 : b1 - 1, t0 = t1 > 0 ? t1 - 1 : tpi - 1;
        start = HPU * b0 + 2 * TC * t0 + 2 * TC + 2;
      }
      const int G = start + r;
      const bool valid = T < T1 && G < end;
      const int b = odin_div_small(G, HPU), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      tr[e] = FpEnt{valid ? (G - odin_div_small(G, NSU) * NSU) * RBU : DST_NONE,
                        real ? (int)((unsigned)(G - b - 1) * u_rowbytes) : (int)OFF_NONE};
    }
  }
  // the table entries of fill f for this wave's items, then the loads themselves
  auto fill_entries = [&](FpEnt (&en)[FP_MAXU], int f) {
#pragma unroll
    for (int j = 0; j < FP_MAXU; ++j) en[j] = tr[f * RPF + r0w + RJ * j];
  };
  auto fill_loads = [&](FpItem (&iu)[FP_MAXU], const FpEnt (&en)[FP_MAXU]) {
#pragma unroll
    for (int j = 0; j < FP_MAXU; ++j) {
      iu[j].dst = en[j].x + u_lds;  // negative: no row
      iu[j].v = odin_run_load4(RU, (unsigned)en[j].y + u_g);
    }
  };
  // An item without a row (dst < 0, wave-uniform) is skipped by ONE scalar branch: the matrix pipe shares its issue
  // port with the vector ALUs (an MFMA takes ~10 issue cycles, every VALU instruction 4: tools/micro/mfma_bf16_split.hip),
  // and with ~190 VALU instructions per 24 MFMAs the tile was bound by VALU issue, not by the MFMAs -- the ~35
  // instructions of a split nobody needs cost more than the branch around them.
  auto store_item = [&](const FpItem& it) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;
#endif
    u32x2 h, l;
    if (SCM == 1 || (SCM == 2 && sc)) odin_split_h4<true>(it.v, in_s, in_s2k, h, l);
    else odin_split_h4<false>(it.v, 1.f, ODIN_LO_SCALE, h, l);
    char* d = ring + it.dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBU) = l;
  };

  // ---- this lane's output pixel inside the tile and its read offsets ----
  const int orow = (OW == 32) ? 0 : (OW == 16) ? (l31 >> 4) : (l31 >> 3);
  const int ocol = (OW == 32) ? l31 : (OW == 16) ? (l31 & 15) : (l31 & 7);
  // B fragment of tap t, k-half kk: slot ocol + (kw >> 1) of parity kw & 1, piece (2 kk + half) ^ ((slot >> 2) & 3)
  int boff[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kw = kw0 + t, slot = ocol + (kw >> 1);
      boff[t][kk] = (kw & 1) * PARB + slot * 64 + (((2 * kk + half) ^ ((slot >> 2) & 3)) << 4);
    }
  // the accumulator registers this wave finishes: r = 2 wave, 2 wave + 1 -> channels c0, c0 + 1
  const int c0 = n0 + ((2 * wave) & 3) + 8 * ((2 * wave) >> 2) + 4 * half;
  float bias2[2] = {0.f, 0.f};
  if (EPI == 1) { bias2[0] = p.bias[c0]; bias2[1] = p.bias[c0 + 1]; }
  float csum[2] = {0.f, 0.f};

  FpEnt en[FP_MAXU];
  FP_STAMP(6);
  u32x4 wf[2][2][NPL];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const float(&v)[8] = wv[t][kk];
      u32x2 h0, l0, h1, l1;
      odin_split_h4<false>(make_float4(v[0], v[1], v[2], v[3]), 1.f, ODIN_LO_SCALE, h0, l0);
      odin_split_h4<false>(make_float4(v[4], v[5], v[6], v[7]), 1.f, ODIN_LO_SCALE, h1, l1);
      wf[t][kk][0][0] = h0.x; wf[t][kk][0][1] = h0.y; wf[t][kk][0][2] = h1.x; wf[t][kk][0][3] = h1.y;
      wf[t][kk][1][0] = l0.x; wf[t][kk][1][1] = l0.y; wf[t][kk][1][2] = l1.x; wf[t][kk][1][3] = l1.y;
    }

  if (SCM != 0) {
    // the input's range word (requested first thing in the kernel): the bound, the scale flag, the powers of two
    const unsigned in_mb = odin_range_finish(in_rq);
    sc = SCM == 1 || odin_act_needs_scale(in_mb);
    const int gk = sc ? odin_range_shift(in_mb) : 0;
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11); out_s = odin_pow2(-gk);
  }
  FP_STAMP(8);
#pragma unroll
  for (int j = 0; j < FP_MAXU; ++j) store_item(iuA[j]);
  FP_STAMP(9);
  __syncthreads();  // ONE barrier: pads, tables and the first tile's rows are in LDS
  FP_STAMP(7);
  FpEnt en1[FP_MAXU], en2[FP_MAXU];
  fill_entries(en1, 1);  // (all table reads of the prologue in one batch: one LDS round trip, not four)
  fill_entries(en2, 2);
  fill_entries(en, 3);
  FpEnt thN = tt[0];  // (su0, output offset) of the next tile
  fill_loads(iuA, en1);
  fill_loads(iuB, en2);
  const unsigned o_lane = (unsigned)(((orow * OW + ocol) * p.CO + c0) * 4);

  // (the first tile's epilogue pass has no predecessor: it sums a zeroed scratch buffer and its store is out of range)
  const unsigned out_bytes = (unsigned)((size_t)p.B * p.OH * OW * p.CO * 4);
  const OdinRun RO = odin_run(p.out, out_bytes);
  const OdinRun RX = odin_run(EPI == 2 ? p.aux : p.out, out_bytes);
  unsigned ooffP = ODIN_OOB;
  float2 auxP = make_float2(0.f, 0.f), pvP = make_float2(0.f, 0.f);

  // sums the eight partial tiles of registers 2 wave, 2 wave + 1 of tile T - 1 and finishes them
  // scratch layout [register pair][source wave][lane][8 B]: this wave reads pair `wave` of all eight sources --
  // 512 contiguous bytes per read (the round-2 layout [wave][r4][lane][16 B] made these reads 8-byte pieces at
  // a 16-byte stride: 29 % of the kernel's LDS cycles were bank conflicts, profiles/r03_kpmc_planes_8wave.txt)
  auto finish_load = [&](int buf, float2 (&q8)[8]) {
    const char* q = red + buf * RED + ((wave * 8 * 64 + lane) << 3);
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) q8[wv] = *reinterpret_cast<const float2*>(q + wv * (64 * 8));
  };
  auto finish_done = [&](const float2 (&q8)[8]) {
    float2 s = q8[0];
#pragma unroll
    for (int wv = 1; wv < 8; ++wv) { s.x += q8[wv].x; s.y += q8[wv].y; }
    float v[2] = {s.x, s.y};
    if (SCM == 1 || (SCM == 2 && sc)) { v[0] *= out_s; v[1] *= out_s; }
    if (ACC) { v[0] += pvP.x; v[1] += pvP.y; }
    if (EPI == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float tt = v[k] + bias2[k];
        v[k] = fmaxf(tt, 0.f) + (odin_exp2(fminf(tt, 0.f) * 1.44269504088896341f) - 1.f);
      }
      amx = odin_amax3(amx, v[0], v[1]);   // (the range word of the activation: odin_conv_desc.y_amax)
    } else if (EPI == 2) {
      v[0] = fmaf(v[0], fminf(auxP.x, 0.f), v[0]);  // x ELU'(aux) = 1 + min(aux, 0)
      v[1] = fmaf(v[1], fminf(auxP.y, 0.f), v[1]);
      csum[0] += v[0];
      csum[1] += v[1];
      // (the first tile's pass has no predecessor and sums zeros: nothing to mask)
      amx = odin_amax3(amx, v[0], v[1]);
    }
    odin_run_store2(RO, ooffP, make_float2(v[0], v[1]));  // (range-checked: the first tile's pass has no tile T - 1)
  };

  // One tile.  The row fills of tile T + 3 (loads), the split and LDS stores of tile T + 1's rows and the epilogue of
  // tile T - 1 ride between the MFMAs; the only scalar work left in there is one branch per fill item (the index walk
  // that used to sit here made the 24 MFMAs of a wave take 2.1-2.9 k cycles; a bare MFMA chain takes 0.85 k:
  // in-kernel stamps and ablations, profiles/r03_fconv_planes_bookkeeping.txt).
  auto run_tile = [&](int T, FpItem (&ldu)[FP_MAXU], const FpItem (&stu)[FP_MAXU]) {
    const FpEnt th = thN;
    int su = th.x + 2 * orow + kh;  // the wave's tap row kh of this lane's output row
    su -= su >= NSU ? NSU : 0;
    const char* rowp = ring + su * RBU;
    u32x4 fb[2][2][NPL];
    FP_STAMP(2);
    // (the fragments of the first products first)
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) fb[t][kk][pl] = *reinterpret_cast<const u32x4*>(rowp + boff[t][kk] + pl * PBU);
    ODIN_SCHED_FENCE();
    const unsigned ooff = (unsigned)th.y + o_lane;
    float2 auxN = make_float2(0.f, 0.f), pvN = make_float2(0.f, 0.f);
    float2 q8[8];
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();  // main (h x h) and cross (h x l + l x h, times 2^11) sums
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int t = (m >> 1) & 1, kk = m & 1, pp = m >> 2;
      // plane products: weights x pixels h*l, l*h (cross accumulator), h*h (main accumulator)
      if (pp == 0) acx = mfma32_f16(wf[t][kk][0], fb[t][kk][1], acx);
      if (pp == 1) acx = mfma32_f16(wf[t][kk][1], fb[t][kk][0], acx);
      if (pp == 2) acc = mfma32_f16(wf[t][kk][0], fb[t][kk][0], acc);
      if (m == 0) FP_STAMP(3);
      if (m == 11) FP_STAMP(4);
      // the matrix pipe starts as soon as the first fragments are there; everything else rides between MFMAs
      if (m == 0) {
        fill_loads(ldu, en);  // fill T - T0 + 3: its table entries were read a tile ago
        if (EPI == 2) auxN = odin_run_load2(RX, ooff);
        if (ACC) pvN = odin_run_load2(RO, ooff);
      }
      if (m == 10) {  // (behind every other LDS read of this tile)
        fill_entries(en, T - T0 + 4);
        thN = tt[T - T0 + 1];
      }
      if ((m & 1) == 1 && (m >> 1) < FP_MAXU) store_item(stu[m >> 1]);  // rows of tile T + 1
      if (m == 8) finish_load((T - 1) & 1, q8);  // tile T - 1: its partials are complete behind the last barrier
      if (m == 9) finish_done(q8);
      ODIN_SCHED_FENCE();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(acx[r], ODIN_LO_UNSCALE, acc[r]);
    // this wave's partial tile -> scratch [T & 1][register pair][wave][lane]
    char* d = red + (T & 1) * RED + ((wave * 64 + lane) << 3);
#pragma unroll
    for (int pr = 0; pr < 8; ++pr)
      *reinterpret_cast<float2*>(d + pr * (8 * 64 * 8)) = make_float2(acc[2 * pr], acc[2 * pr + 1]);
    FP_STAMP(5);
    ooffP = ooff;
    auxP = auxN;
    pvP = pvN;
    __syncthreads();  // partial tiles complete; every wave is past tile T's rows; tile T + 1's rows are stored
  };
#pragma unroll 1
  for (int T = T0; T < T1; T += 3) {
    run_tile(T, iuC, iuA);
    if (T + 1 < T1) run_tile(T + 1, iuA, iuB);
    if (T + 2 < T1) run_tile(T + 2, iuB, iuC);
  }
  {
    float2 q8[8];
    finish_load((T1 - 1) & 1, q8);
    finish_done(q8);
  }

  if (EPI == 2 || EPI == 1) {
    __syncthreads();  // (the partial-tile scratch is free: every wave is past its last finish pass)
    odin_amax_commit_wg(p.out_amax, amx, tid, 512, reinterpret_cast<float*>(red), blockIdx.x + gridDim.x * blockIdx.y);
  }
  if (EPI == 2 && p.colsum != nullptr) {
    // column sums of this workgroup's outputs: the 32 pixel lanes of each half by shuffles; every
    // (wave, half) owns its own two channels
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v = csum[k];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (l31 == 0) p.colsum[(size_t)blockIdx.x * p.CO + c0 + k] = v;
    }
  }
}

template <int EPI, int OW, bool ACC, int SCM>
__global__ __launch_bounds__(512) void fconv_planes_kernel(FPParams p) {
  fp_body<EPI, OW, ACC, SCM>(p);
}

// 64 reduction channels in ONE launch: both 32-channel passes inside the kernel (tconv_planes.hip: tconv_planes2_kernel);
// the partial sums a thread leaves in `out` are read back by the same thread (same wave / register-pair ownership)
template <int EPI, int OW, int SCM>
__global__ __launch_bounds__(512) void fconv_planes2_kernel(FPParams p) {
  {
    FPParams q = p;
    q.colsum = nullptr;
    q.out_amax = nullptr;
    q.ci_off = 0;
    fp_body<0, OW, false, SCM>(q);
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
  fp_body<EPI, OW, true, SCM>(p);
}

// LDS: row ring + two partial-tile buffers + the fill tables ((1 + rows per fill) x 8 bytes per fill, tiles + 4 fills)
constexpr int FP_LDS_MAX = 160 * 1024;
int fp_ring_bytes(int OW) { return (4 * (32 / OW) + 3) * 2 * 2 * (OW + 2) * 64 + 2 * (8 * 4 * 64 * 16); }
int fp_fill_bytes(int OW) {
  const int ipu = 2 * OW / 8, rj = 8 / ipu > 0 ? 8 / ipu : 1;
  return 8 * (1 + FP_MAXU * rj);
}
// tiles per workgroup: one workgroup per CU when the tables fit, more workgroups otherwise; -1: does not fit
int fp_tiles_per_wg(int OW, int n_tiles, int gy) {
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  int tpw = (n_tiles + cap - 1) / cap;
  const int limit = (FP_LDS_MAX - fp_ring_bytes(OW)) / fp_fill_bytes(OW) - 4;
  if (tpw > limit) tpw = limit;
  if ((n_tiles + tpw - 1) / tpw > ODIN_MAX_COLSUM_BLOCKS) return -1;
  return tpw;
}

template <int EPI, int OW, bool ACC, int SC>
int fp_launch(const FPParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)fp_ring_bytes(OW) + (size_t)(p.tiles_per_wg + 4) * fp_fill_bytes(OW);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fconv_planes_kernel<EPI, OW, ACC, SC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((fconv_planes_kernel<EPI, OW, ACC, SC>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("fconv_planes(f16x2)");
}

template <int EPI, int SC>
int fp_launch2_w(const FPParams& p, int OW, dim3 grid, void* stream) {
  const size_t lds = (size_t)fp_ring_bytes(OW) + (size_t)(p.tiles_per_wg + 4) * fp_fill_bytes(OW);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[3] = {reinterpret_cast<const void*>(&fconv_planes2_kernel<EPI, 32, SC>),
                          reinterpret_cast<const void*>(&fconv_planes2_kernel<EPI, 16, SC>),
                          reinterpret_cast<const void*>(&fconv_planes2_kernel<EPI, 8, SC>)};
    for (int i = 0; i < 3; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, FP_LDS_MAX) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  if (OW == 32) ODIN_LAUNCH((fconv_planes2_kernel<EPI, 32, SC>), grid, dim3(512), lds, stream, p);
  else if (OW == 16) ODIN_LAUNCH((fconv_planes2_kernel<EPI, 16, SC>), grid, dim3(512), lds, stream, p);
  else ODIN_LAUNCH((fconv_planes2_kernel<EPI, 8, SC>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("fconv_planes(f16x2)");
}

}  // namespace

static long long* g_fp_stamps = nullptr;
void odin_fconv_planes_set_stamps(void* buf) { g_fp_stamps = (long long*)buf; }

bool odin_fconv_planes_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                                  int pt, int pl, int center) {
  if (odin_blk_first()) return false;   // (diagnostics: odin_debug_blk_first)
  // (read per call: the A/B tests switch paths inside one process; a captured graph never comes here)
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NOPLANES") || ODIN_DIAG_ENV("ODIN_SPLIT") || ODIN_DIAG_ENV("ODIN_NOFPLANES")) return false;
  return KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && (CI == 32 || (CI == 64 && !ODIN_DIAG_ENV("ODIN_FP_NO64"))) && (CO % 32) == 0 && !center &&
         H == 2 * OH && W == 2 * OW && (OW == 8 || OW == 16 || OW == 32) && (OH % (32 / OW)) == 0 &&
         (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * OH * OW * CO * 4 < (1ull << 31) &&
         fp_tiles_per_wg(OW, B * (OH / (32 / OW)), CO / 32) > 0;
}

template <int EPI, bool ACC, int SC>
int fp_launch_w(const FPParams& p, int OW, dim3 grid, void* stream) {
  if (OW == 32) return fp_launch<EPI, 32, ACC, SC>(p, grid, stream);
  if (OW == 16) return fp_launch<EPI, 16, ACC, SC>(p, grid, stream);
  return fp_launch<EPI, 8, ACC, SC>(p, grid, stream);
}

// epi 1: Conv2D forward (bias + ELU); epi 2: Conv2DTranspose data gradient (x ELU'(aux), column sums).
// CI = 64: two reduction passes over 32 channels each (a wave keeps the weight fragments of 32 channels in
// registers): the first leaves raw partial sums in `out`, the second adds them and runs the epilogue.
int odin_fconv_planes_launch(const float* in, const float* w, const float* bias, const float* aux,
                             float* out, float* colsum, int* rows_out, int B, int OH, int OW, int CI, int CO,
                             int epi, const uint32_t* in_amax, uint32_t* out_amax, void* stream) {
  FPParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.OH = OH; p.CO = CO;
  p.CS = CI; p.ci_off = 0;
  p.stamps = g_fp_stamps;
  const int TC = 32 / OW;
  p.tiles_per_img = OH / TC;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  p.tiles_per_wg = fp_tiles_per_wg(OW, p.n_tiles, gy);
  if (p.tiles_per_wg <= 0) return odin_fail(-2, "fconv_planes: too many tiles for the fill tables");
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  if (epi == 2) {
    p.in_amax = odin_range_word_of(in, (size_t)B * 2 * OH * 2 * OW * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "fconv_planes: no range word for the gradient input");
    p.out_amax = out_amax;
  } else {
    // forward: the activation's word, where the caller has one (scaled only outside the safe window); its output's
    p.in_amax = in_amax;
    p.out_amax = out_amax;
  }
  const bool aw = epi == 1 && in_amax != nullptr;
  dim3 grid(gx, gy, 1);
  if (CI == 64) {
#ifdef ODIN_DIAG  // diagnostics build: A/B against the two-launch form of round 3
    if (ODIN_DIAG_ENV("ODIN_FP_2LAUNCH")) {
      FPParams q = p;
      q.colsum = nullptr;
      q.out_amax = nullptr;
      const int rc = epi == 1 ? fp_launch_w<0, false, 0>(q, OW, grid, stream) : fp_launch_w<0, false, 1>(q, OW, grid, stream);
      if (rc != 0) return rc;
      p.ci_off = 32;
      return epi == 1 ? fp_launch_w<1, true, 0>(p, OW, grid, stream) : fp_launch_w<2, true, 1>(p, OW, grid, stream);
    }
#endif
    if (epi == 1) return aw ? fp_launch2_w<1, 2>(p, OW, grid, stream) : fp_launch2_w<1, 0>(p, OW, grid, stream);
    return fp_launch2_w<2, 1>(p, OW, grid, stream);
  }
  if (epi == 1) return aw ? fp_launch_w<1, false, 2>(p, OW, grid, stream) : fp_launch_w<1, false, 0>(p, OW, grid, stream);
  return fp_launch_w<2, false, 1>(p, OW, grid, stream);
}
