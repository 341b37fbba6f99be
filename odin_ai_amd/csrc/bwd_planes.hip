// bwd_planes.hip -- the WHOLE backward pass of a Conv2DTranspose(k4, s2, `SAME`) over 32 output channels in ONE launch:
// its weight gradient (wgrad_planes.hip) and its data gradient (fconv_planes.hip) read the same gradient tensor
// dy [B, 2h, 2w, 32] -- the largest tensor of the layer (134 MB for the last deconvolution of the 64x64 decoders,
// image_networks.py:499-513) -- through the same rolling row window with the same tap ownership of the 8 waves.  Here dy
// is fetched, range-scaled, split into its two f16 planes (odin_device.h: x = h + 2^-11 l) and stored to LDS ONCE:
//       dW[kh][kw][co][ci] = sum over (b, i, j) of dy[b, 2 i - 1 + kh, 2 j - 1 + kw, co] * x[b, i, j, ci]
//       dx[b, i, j, ci]    = ELU'(aux) * sum over (kh, kw, co) of dy[b, 2 i - 1 + kh, 2 j - 1 + kw, co] * W[kh, kw, co, ci]
// (tape.gradient of the step, base_networks.py:549).  A tile = 32 coarse pixels (1, 2 or 4 rows of x); wave v owns the
// taps (kh = v >> 1, kw = 2 (v & 1) + {0, 1}) for both products:
//   weight gradient: the pixel is the MFMA reduction index -- transposed LDS reads (ds_read_b64_tr_b16) of dy and x, the
//       wave's two 32 x 32 (co x ci) accumulator pairs stay in registers over the whole persistent tile loop and go to
//       this workgroup's slab row at the end;
//   data gradient: the channel co is the reduction index -- the wave's weight fragments (32 registers) stay resident,
//       its partial 32-pixel x 32-channel tile goes through LDS, wave v finishes accumulator registers 2 v, 2 v + 1 of the
//       previous tile (x ELU'(aux), column sums, range word of dx) behind the barrier.
// Both halves execute exactly the MFMA sequences of the stand-alone kernels (same tiles per workgroup, same order): the
// results are bit-identical to odin_deconv2d_wgrad + odin_deconv2d_dgrad.
// The dy window is the one of fconv_planes.hip (two column-parity planes per fine row, 16-byte k-pieces XOR-swizzled by
// (slot >> 2) for the data gradient's conflict-free ds_read_b128); the transposed reads supply per-lane addresses, so
// the swizzle costs them four precomputed offsets.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct BPParams {
  const float* U;      // dy [B, 2h, 2w, CUt]: this pass reduces over / differentiates channels cu_off .. cu_off + 31
  const float* V;      // x  [B, h, w, CVt]   (the layer's input)
  const float* w;      // [16 taps][CUt][CVt]
  const float* aux;    // [B, h, w, CVt]: dx *= ELU'(aux)
  float* dx;           // [B, h, w, CVt]
  float* colsum;       // [gridDim.x][CVt] partial column sums of dx (may be null)
  float* slab;         // [gridDim.x][16 * CUt * CVt]
  int B, h, CVt;
  int CUt, cu_off;
  int slab_stride;
  int tiles_per_img, n_tiles, tiles_per_wg;
  const unsigned* g_amax;  // range word of dy
  const unsigned* a_amax;  // optional range word of x (scaled only outside [2^-8, 2^15): odin_act_needs_scale)
  unsigned* out_amax;      // range word of dx (may be null)
};

struct alignas(8) BpEnt {
  int x, y;
};

struct BpItem {
  float4 v;
  int dst;  // WAVE-UNIFORM part of the byte offset of the hi-plane store inside the LDS image (a scalar register; the
            // lane's part is added at the store); < 0: no item
};

__device__ __forceinline__ int bp_uniform(int v) {
#ifdef ODIN_SIM
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

constexpr int BP_MAXU = 4;  // 1 KB load items (8 pixels x 32 channels) of fine rows per wave and fill

// byte offset of channel c of pixel slot `slot` inside a column-parity plane of the dy window
__device__ __forceinline__ int bp_uoff(int slot, int c) {
  return slot * 64 + ((((c >> 3)) ^ ((slot >> 2) & 3)) << 4) + (c & 7) * 2;
}

// ds_read_b64_tr_b16 over the swizzled dy window: a block of 4 pixel slots (slot0 ..) x 16 channels (16 g ..); lane l16 of
// the 16-lane group receives channel 16 g + l16 of the 4 pixels.  On the hardware lane 4 q + p supplies the address of
// pixel q, channels 4 p .. 4 p + 3 (`lane_off`, precomputed with the swizzle of ITS pixel's slot).
__device__ __forceinline__ u32x2 bp_tr_read_u(const char* plane, int slot0, int g, int l16, int lane_off) {
#ifdef ODIN_SIM
  (void)lane_off;
  unsigned short e[4];
  for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const unsigned short*>(plane + bp_uoff(slot0 + q, 16 * g + l16));
  return odin_u2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
#else
  (void)slot0; (void)g; (void)l16;
  typedef short bp_s4 __attribute__((ext_vector_type(4)));
  const bp_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bp_s4*)(plane + lane_off));
  return __builtin_bit_cast(u32x2, v);
#endif
}

// the unswizzled x window: as wgrad_planes.hip
__device__ __forceinline__ u32x2 bp_tr_read_v(const char* blk, int l16) {
#ifdef ODIN_SIM
  unsigned short e[4];
  for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const unsigned short*>(blk + q * 64 + 2 * l16);
  return odin_u2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
#else
  typedef short bp_s4 __attribute__((ext_vector_type(4)));
  const char* a = blk + (l16 >> 2) * 64 + (l16 & 3) * 8;
  const bp_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bp_s4*)a);
  return __builtin_bit_cast(u32x2, v);
#endif
}

// W = coarse row length (8, 16 or 32); NSETS = register sets of row-fill loads in flight (loads run NSETS - 1 tiles ahead
// of their LDS stores)
// PASS 0: a layer with 32 output channels.  64 output channels take two passes inside one launch (bwd_planes2_kernel, as
// fconv_planes2_kernel): PASS 1 (channels 0-31) writes its weight-gradient block and leaves the data gradient's raw
// partial sums in dx, PASS 2 (channels 32-63) adds them in front of the epilogue.
template <int W, int NSETS, int DBG = 0, int PASS = 0>
__device__ __forceinline__ void bp_body(const BPParams& p) {
  constexpr int NPL = 2;                 // f16 planes per operand
  constexpr int TC = 32 / W;             // coarse rows per tile
  constexpr int WU = 2 * W;              // fine row length
  constexpr int SU = W + 1;              // slots per column-parity plane of a fine row
  constexpr int PARB = (SU + 1) * 64;    // + one spare slot (fconv_planes.hip: the row fills' stores spread over all banks)
  constexpr int PBU = 2 * PARB;          // one f16 plane of a fine row
  constexpr int RBU = NPL * PBU;
  constexpr int NSU = 4 * TC + 3;        // live fine rows (2 TC + 2) + the next tile's (2 TC + 1 at an image seam)
  constexpr int PBV = W * 64;
  constexpr int RBV = NPL * PBV;
  constexpr int NSV = 2 * TC;
  constexpr int IPU = WU / 8;            // load items per fine row: 8, 4, 2
  constexpr int IPV = W / 8;             // per coarse row: 4, 2, 1
  constexpr int RJ = 8 / IPU > 0 ? 8 / IPU : 1;
  constexpr int RED = 8 * 8 * 64 * 8;    // one partial-tile buffer: [register pair][wave][lane][8 B]
  ODIN_DYN_SMEM(char, smem);
  char* uring = smem;
  char* vring = smem + NSU * RBU;
  char* red = vring + NSV * RBV;
  const int tid = threadIdx.x, lane = tid & 63;
  // the two range words: requested first thing, finished in front of the first split (odin_device.h: odin_range_issue)
  const OdinRangeReq g_rq = odin_range_issue(p.g_amax, lane), a_rq = odin_range_issue(p.a_amax, lane);
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15;
  const int cv0 = blockIdx.y * 32;
  const int HU = 2 * p.h, HPU = HU + 1;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  // ---- the loads of the prologue go out FIRST: the wave's weight fragments of the data gradient (taps (kh, kw0),
  // (kh, kw0 + 1); lane = output channel l31, k = 8 half + e) and its items of fill 0 ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1), kws = kw0 >> 1;
  float wv[2][2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int tap = kh * 4 + kw0 + t;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        wv[t][kk][e] = p.w[((size_t)(tap * p.CUt + p.cu_off + 16 * kk + 8 * half + e)) * p.CVt + cv0 + l31];
    }
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * HU * WU * p.CUt * 4));
  const OdinRun RV = odin_run(p.V, (unsigned)((size_t)p.B * p.h * W * p.CVt * 4));
  const int ch4 = lane & 7, pxl = lane >> 3;
  // item j of this wave: row r0 + RJ j of the fill, 8-pixel column block cu_blk (wave constants; 32-bit offsets)
  const int r0 = wave / IPU, cu_blk = wave - r0 * IPU;
  const int pcw = 8 * cu_blk + pxl + 1;  // padded column of this lane's pixel: parity pcw & 1, slot pcw >> 1
  const int u_lds = (pcw & 1) * PARB + bp_uoff(pcw >> 1, 4 * ch4);
  const unsigned u_g = (unsigned)(((8 * cu_blk + pxl) * p.CUt + p.cu_off + 4 * ch4) * 4);
  const unsigned u_rowbytes = (unsigned)(WU * p.CUt * 4), v_rowbytes = (unsigned)(W * p.CVt * 4);
  const int vr = (wave & 3) / IPV, vc = (wave & 3) - vr * IPV;
  const unsigned v_g = (unsigned)(((8 * vc + pxl) * p.CVt + cv0 + 4 * ch4) * 4);
  const int v_lds = NSU * RBU + (8 * vc + pxl) * 64 + ch4 * 8;
  const int n_vrows = p.B * p.h;
  BpItem iu[NSETS][BP_MAXU], iv[NSETS];
  {
    const int tpi = p.tiles_per_img;
    const int b0 = odin_div_small(T0, tpi), t0 = T0 - b0 * tpi;
    const int start = HPU * b0 + 2 * TC * t0;
#pragma unroll
    for (int j = 0; j < BP_MAXU; ++j) {
      const int r = r0 + RJ * j, G = start + r;
      const bool valid = r < 2 * TC + 2;
      const int b = b0 + (2 * TC * t0 + r >= HPU ? 1 : 0), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      iu[0][j].dst = bp_uniform(valid ? (G - odin_div_small(G, NSU) * NSU) * RBU : -(1 << 24));
      iu[0][j].v = odin_run_load4(RU, real ? (unsigned)(G - b - 1) * u_rowbytes + u_g : ODIN_OOB);
    }
    const int grow = TC * T0 + vr;
    iv[0].dst = bp_uniform(wave < 4 ? (grow & (NSV - 1)) * RBV : -(1 << 24));
    iv[0].v = odin_run_load4(RV, (wave < 4 && grow < n_vrows) ? (unsigned)grow * v_rowbytes + v_g : ODIN_OOB);
  }
  ODIN_SCHED_FENCE();

  // ---- SAME-padding slots of every fine ring row and plane (parity plane 0 slot 0, parity plane 1 slot W) ----
  for (int e = tid; e < NSU * 8 * NPL; e += 512) {
    const int sl = e / (8 * NPL), rem = e - sl * (8 * NPL);
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(uring + sl * RBU + pl * PBU + (side ? PARB + W * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // (the first tile's finish pass has no predecessor: it sums a zeroed buffer and its store is out of range)
  for (int e = tid; e < RED / 16; e += 512)
    *reinterpret_cast<float4*>(red + ((T0 - 1) & 1) * RED + e * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  // ---- fill tables (wgrad_planes.hip / fconv_planes.hip: the index walk over image seams and ring wrap-arounds is done
  // once, by all threads, into LDS; the tile loop reads wave-uniform entries and adds lane offsets) ----
  constexpr int RPF = BP_MAXU * RJ;           // fine rows a fill can carry (row r = r0 + RJ j of item j)
  constexpr int DST_NONE = -(1 << 24);
  constexpr unsigned OFF_NONE = 0x7FFF0000u;
  const int NF = p.tiles_per_wg + NSETS + 1;
  BpEnt* tt = reinterpret_cast<BpEnt*>(red + 2 * RED);  // [NF] tile: (first fine ring slot, byte offset of its first output)
  BpEnt* tr = tt + NF;                                 // [NF][RPF] fine row: (LDS byte offset, global byte offset)
  BpEnt* tv = tr + NF * RPF;                           // [NF][TC] coarse row: the same
  {
    const int tpi = p.tiles_per_img;
    for (int e = tid; e < NF; e += 512) {
      const int T = T0 + e, b = odin_div_small(T, tpi), t = T - b * tpi;
      const int g0 = HPU * b + 2 * TC * t;
      tt[e] = BpEnt{g0 - odin_div_small(g0, NSU) * NSU, (int)(((size_t)(b * p.h + TC * t) * W) * p.CVt * 4)};
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
      const int G = start + r;  // global padded fine row HPU * b + gi; gi == 0: the zero row between images
      const bool valid = T < T1 && G < end;
      const int b = odin_div_small(G, HPU), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;
      tr[e] = BpEnt{valid ? (G - odin_div_small(G, NSU) * NSU) * RBU : DST_NONE,
                    real ? (int)((unsigned)(G - b - 1) * u_rowbytes) : (int)OFF_NONE};
    }
    for (int e = tid; e < NF * TC; e += 512) {
      const int f = e / TC, q = e - f * TC;
      const int T = T0 + f, grow = TC * T + q;  // global coarse row h * b + i
      const bool valid = T < T1;
      tv[e] = BpEnt{valid ? (grow & (NSV - 1)) * RBV : DST_NONE,
                    valid && grow < n_vrows ? (int)((unsigned)grow * v_rowbytes) : (int)OFF_NONE};
    }
  }
  // waves 4-7 carry no coarse-row item
  const int v_none_dst = wave < 4 ? 0 : (int)0x80000000;
  const unsigned v_none_off = wave < 4 ? 0u : OFF_NONE;
  struct FillEnt { BpEnt u[BP_MAXU]; BpEnt v; };
  // (wave-uniform entries: moved to scalar registers)
  auto fill_entries = [&](FillEnt& en, int f) {
#pragma unroll
    for (int j = 0; j < BP_MAXU; ++j) {
      const BpEnt e = tr[f * RPF + r0 + RJ * j];
      en.u[j] = BpEnt{bp_uniform(e.x), bp_uniform(e.y)};
    }
    const BpEnt e = tv[f * TC + vr];
    en.v = BpEnt{bp_uniform(e.x), bp_uniform(e.y)};
  };
  // (unconditional loads -- an absent item reads zeros through the range check -- keep the number in flight constant)
  auto fill_loads = [&](BpItem (&u)[BP_MAXU], BpItem& v, const FillEnt& en) {
#pragma unroll
    for (int j = 0; j < BP_MAXU; ++j) {
      u[j].dst = en.u[j].x;  // negative: no row
      u[j].v = odin_run_load4(RU, (unsigned)en.u[j].y + u_g);
    }
    v.dst = en.v.x | v_none_dst;
    v.v = odin_run_load4(RV, ((unsigned)en.v.y + v_g) | v_none_off);
  };
  // dy is carried times 2^gk (its maximum lands in [2^14, 2^15)), x times 2^ak when its bound leaves the safe window;
  // both sums are scaled back at the end (set by finish_words(), in front of the first split)
  int gk = 0, ak = 0;
  bool as = false;
  float g_s = 1.f, g_s2k = ODIN_LO_SCALE, a_s = 1.f, a_s2k = ODIN_LO_SCALE, out_s = 1.f;
  auto finish_words = [&]() {
    gk = odin_range_shift(odin_range_finish(g_rq));
    g_s = odin_pow2(gk); g_s2k = odin_pow2(gk + 11); out_s = odin_pow2(-gk);
    const unsigned a_mb = p.a_amax != nullptr ? odin_range_finish(a_rq) : 0u;
    as = odin_act_needs_scale(a_mb);
    ak = as ? odin_range_shift(a_mb) : 0;
    a_s = odin_pow2(ak); a_s2k = odin_pow2(ak + 11);
  };
  auto store_u = [&](const BpItem& it) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;  // wave-uniform: a scalar branch
#endif
    u32x2 h, l;
    odin_split_h4<true>(it.v, g_s, g_s2k, h, l);
    char* d = smem + (it.dst + u_lds);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBU) = l;
  };
  auto store_v = [&](const BpItem& it) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;
#endif
    u32x2 h, l;
    if (as) odin_split_h4<true>(it.v, a_s, a_s2k, h, l);
    else odin_split_h4<false>(it.v, 1.f, ODIN_LO_SCALE, h, l);
    char* d = smem + (it.dst + v_lds);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBV) = l;
  };
  auto store_fill_item = [&](const BpItem (&u)[BP_MAXU], const BpItem& v, int k) {
    if (k < BP_MAXU) store_u(u[k]);
    else store_v(v);
  };

  // ---- weight gradient: this lane's part of a transposed read ----
  // pixel k of a 16-pixel chunk supplied by this lane: block blk (0, 1) of its half, row q = l16 >> 2
  const int q4 = l16 >> 2, g16 = (lane >> 4) & 1;
  int krow[2], kcol[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int kpix = 8 * half + 4 * blk + q4;
    krow[blk] = (W >= 16) ? 0 : (kpix >> 3);
    kcol[blk] = (W >= 16) ? kpix : (kpix & 7);
  }
  const int colb = (16 * g16 + 4 * (l16 & 3)) * 2;  // byte offset of this lane's 4 channels in the x window
  // dy window: lane offsets of its pixel (slot j0 + kcol + kws) and channels 16 g + 4 p .., per (chunk, block); the
  // wave's two taps differ by the parity plane only
  constexpr int NCH = (W == 32) ? 2 : 1;   // the chunk changes the slot only when a chunk is half a row
  int uoff[NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) uoff[c][blk] = bp_uoff(16 * c + kcol[blk] + kws, 16 * g16 + 4 * (l16 & 3));

  // ---- data gradient: this lane's output pixel inside the tile and its read offsets ----
  const int orow = (W == 32) ? 0 : (W == 16) ? (l31 >> 4) : (l31 >> 3);
  const int ocol = (W == 32) ? l31 : (W == 16) ? (l31 & 15) : (l31 & 7);
  // B fragment of tap t, k-half kk: slot ocol + kws of parity t, piece (2 kk + half) ^ ((slot >> 2) & 3)
  int boff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) boff[kk] = bp_uoff(ocol + kws, 8 * (2 * kk + half));
  // the accumulator registers this wave finishes: r = 2 wave, 2 wave + 1 -> channels c0, c0 + 1
  const int c0 = cv0 + ((2 * wave) & 3) + 8 * ((2 * wave) >> 2) + 4 * half;
  float csum[2] = {0.f, 0.f};
  float amx = 0.f;  // running max |dx| of this lane

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

  f32x16 wacc[2] = {f32x16_zero(), f32x16_zero()};  // weight gradient: main sums (h x h) of the wave's two taps
  f32x16 wacx[2] = {f32x16_zero(), f32x16_zero()};  // cross sums (h x l + l x h), times 2^11
  // ---- prologue: rows of the first tile into LDS; ONE barrier publishes them with the pads and the tables; then the
  // next tiles' rows into registers ----
  finish_words();
#pragma unroll
  for (int k = 0; k <= BP_MAXU; ++k) store_fill_item(iu[0], iv[0], k);
  __syncthreads();
  FillEnt en;
  {
    FillEnt e1;
    fill_entries(e1, 1);
    if (NSETS == 3) {
      FillEnt e2;
      fill_entries(e2, 2);
      fill_entries(en, 3);
      fill_loads(iu[0], iv[0], e1);
      fill_loads(iu[1], iv[1], e2);
    } else {
      fill_entries(en, 2);
      fill_loads(iu[0], iv[0], e1);
    }
  }
  BpEnt thN = tt[0];  // (first fine ring slot, output offset) of the next tile
  int su0 = 0, sv0 = 0;

  const unsigned out_bytes = (unsigned)((size_t)p.B * p.h * W * p.CVt * 4);
  const OdinRun RO = odin_run(p.dx, out_bytes);
  const OdinRun RX = odin_run(p.aux, out_bytes);
  const unsigned o_lane = (unsigned)(((orow * W + ocol) * p.CVt + c0) * 4);
  unsigned ooffP = ODIN_OOB;
  float2 auxP = make_float2(0.f, 0.f), pvP = make_float2(0.f, 0.f);

  // fragments of one 16-pixel chunk of the weight gradient: x (2 planes) and dy for the wave's two taps
  struct Frags { u32x4 fv[NPL]; u32x4 fu[2][NPL]; };
  auto read_chunk = [&](int c, Frags& F) {
    // chunk c: coarse rows row0 (+ krow), columns j0 + kcol
    const int row0 = (W == 32) ? 0 : (W == 16) ? c : 2 * c;
    const int j0 = (W == 32) ? 16 * c : 0;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      int sv = sv0 + row0 + krow[blk];
      if (sv >= NSV) sv -= NSV;
      int su = su0 + 2 * (row0 + krow[blk]) + kh;
      if (su >= NSU) su -= NSU;
      const char* vb = vring + sv * RBV + (j0 + kcol[blk] - q4) * 64 + colb - (l16 & 3) * 8;
      const char* ub = uring + su * RBU;
      const int uo = uoff[W == 32 ? c : 0][blk];
      const int slot0 = j0 + kcol[blk] - q4 + kws;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x2 tvv = bp_tr_read_v(vb + pl * PBV, l16);
        F.fv[pl][2 * blk] = tvv.x; F.fv[pl][2 * blk + 1] = tvv.y;
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // tap kw0 + t: padded column 2 j + kw -> parity t, slot j + kws
          const u32x2 tu = bp_tr_read_u(ub + pl * PBU + t * PARB, slot0, g16, l16, uo);
          F.fu[t][pl][2 * blk] = tu.x; F.fu[t][pl][2 * blk + 1] = tu.y;
        }
      }
    }
  };
  // the 6 MFMAs of a chunk (h*l, l*h into the cross accumulator, h*h into the main one; the two taps alternate);
  // behind every other MFMA one item of the next tile's rows is split and stored
  auto mfma_chunk = [&](const Frags& F, int item0, const BpItem (&stu)[BP_MAXU], const BpItem& stv) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int t = m & 1, pp = m >> 1;
      if (!(DBG & 1)) {
        if (pp == 0) wacx[t] = mfma32_f16(F.fu[t][0], F.fv[1], wacx[t]);
        if (pp == 1) wacx[t] = mfma32_f16(F.fu[t][1], F.fv[0], wacx[t]);
        if (pp == 2) wacc[t] = mfma32_f16(F.fu[t][0], F.fv[0], wacc[t]);
      } else {
        wacc[t][m] += odin_bitsf(F.fu[t][pp & 1][0] ^ F.fv[pp & 1][1]);
      }
      if ((m & 1) == 1) {
        const int k = item0 + (m >> 1);
        if (k <= BP_MAXU && !(DBG & 4)) store_fill_item(stu, stv, k);
        if (k <= BP_MAXU && (DBG & 4)) { const float4 tq = k < BP_MAXU ? stu[k].v : stv.v; csum[0] += tq.x + tq.y + tq.z + tq.w; }
      }
      ODIN_SCHED_FENCE();
    }
  };
  // data gradient: sums the eight partial tiles of registers 2 wave, 2 wave + 1 of tile T - 1 and finishes them
  auto finish_load = [&](int buf, float2 (&q8)[8]) {
    const char* q = red + buf * RED + ((wave * 8 * 64 + lane) << 3);
#pragma unroll
    for (int wvv = 0; wvv < 8; ++wvv) q8[wvv] = *reinterpret_cast<const float2*>(q + wvv * (64 * 8));
  };
  auto finish_done = [&](const float2 (&q8)[8]) {
    float2 s = q8[0];
#pragma unroll
    for (int wvv = 1; wvv < 8; ++wvv) { s.x += q8[wvv].x; s.y += q8[wvv].y; }
    float v[2] = {s.x * out_s, s.y * out_s};
    if (PASS == 2) { v[0] += pvP.x; v[1] += pvP.y; }
    if (PASS != 1) {
      v[0] = fmaf(v[0], fminf(auxP.x, 0.f), v[0]);  // x ELU'(aux) = 1 + min(aux, 0)
      v[1] = fmaf(v[1], fminf(auxP.y, 0.f), v[1]);
      csum[0] += v[0];
      csum[1] += v[1];
      amx = odin_amax3(amx, v[0], v[1]);
    }
    odin_run_store2(RO, ooffP, make_float2(v[0], v[1]));  // (range-checked: the first tile's pass has no tile T - 1)
  };

  // the data gradient's fragments of this wave's tap row
  auto dgrad_frags = [&](u32x4 (&fb)[2][2][NPL]) {
    int sud = su0 + 2 * orow + kh;
    sud -= sud >= NSU ? NSU : 0;
    const char* rowp = uring + sud * RBU;
    if (DBG & 2) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) fb[t][kk][pl] = wf[t][kk][pl];
    } else {
#pragma unroll
      for (int pl = NPL - 1; pl >= 0; --pl)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
            fb[t][kk][pl] = *reinterpret_cast<const u32x4*>(rowp + t * PARB + boff[kk] + pl * PBU);
    }
  };
  // its 12 MFMAs (the previous tile's finish pass rides between them), the partial tile -> scratch [T & 1]
  auto dgrad_tile = [&](int T, const u32x4 (&fb)[2][2][NPL]) {
    float2 q8[8];
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int t = (m >> 1) & 1, kk = m & 1, pp = m >> 2;
      if (!(DBG & 1)) {
        if (pp == 0) acx = mfma32_f16(wf[t][kk][0], fb[t][kk][1], acx);
        if (pp == 1) acx = mfma32_f16(wf[t][kk][1], fb[t][kk][0], acx);
        if (pp == 2) acc = mfma32_f16(wf[t][kk][0], fb[t][kk][0], acc);
      } else {
        acc[m] += odin_bitsf(wf[t][kk][pp & 1][0] ^ fb[t][kk][pp & 1][1]);
      }
      if (!(DBG & 16)) {
        if (m == 4) finish_load((T - 1) & 1, q8);  // tile T - 1: its partials are complete behind the last barrier
        if (m == 8) finish_done(q8);
      }
      ODIN_SCHED_FENCE();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(acx[r], ODIN_LO_UNSCALE, acc[r]);
    char* d = red + (T & 1) * RED + ((wave * 64 + lane) << 3);
    if (!(DBG & 16)) {
#pragma unroll
      for (int pr = 0; pr < 8; ++pr)
        *reinterpret_cast<float2*>(d + pr * (8 * 64 * 8)) = make_float2(acc[2 * pr], acc[2 * pr + 1]);
    } else {
      float t2 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t2 += acc[r];
      csum[1] += t2;
    }
  };

  // One tile.  Waves w and w + 4 share a SIMD, and a wave's MFMAs do not overlap with its own waits (LDS fragment
  // reads, the barrier): ablations showed the matrix-pipe time of a tile ADDED to everything else.  So the two waves of
  // a SIMD run the tile's two halves in OPPOSITE order -- waves 0-3 weight gradient first (36 transposed reads, then
  // MFMAs), waves 4-7 data gradient first (8 reads, then MFMAs) -- and one wave's reads fall into the other's MFMAs.
  Frags F0, F1;
  // (the two-pass instances keep one order: the second body's registers do not fit beside the partial sums it adds)
  const bool stag = ((DBG & 64) || PASS != 0) ? false : wave >= 4;
  auto run_tile = [&](auto dfirst_c, int T, BpItem (&ldu)[BP_MAXU], BpItem& ldv, const BpItem (&stu)[BP_MAXU], const BpItem& stv) {
    constexpr bool dfirst = decltype(dfirst_c)::value;
    const BpEnt th = thN;
    su0 = th.x;
    sv0 = (TC * T) & (NSV - 1);
    const unsigned ooff = (unsigned)th.y + o_lane;
    float2 auxN = make_float2(0.f, 0.f), pvN = make_float2(0.f, 0.f);
    auto loads = [&]() {
      if (!(DBG & 8)) fill_loads(ldu, ldv, en);  // fill T - T0 + NSETS: its table entries were read a tile ago
      else {
#pragma unroll
        for (int j = 0; j < BP_MAXU; ++j) ldu[j].dst = en.u[j].x;
        ldv.dst = en.v.x | v_none_dst;
      }
      if (PASS != 1) auxN = (DBG & 8) ? auxP : odin_run_load2(RX, ooff);
      if (PASS == 2) pvN = odin_run_load2(RO, ooff);
    };
    if (!dfirst) {
      if (!(DBG & 2) || T == T0) read_chunk(0, F0);   // first thing behind the barrier
      ODIN_SCHED_FENCE();
      loads();
      if (!(DBG & 2) || T == T0) read_chunk(1, F1);
      ODIN_SCHED_FENCE();
      mfma_chunk(F0, 0, stu, stv);   // items 0, 1, 2
      u32x4 fb[2][2][NPL];
      dgrad_frags(fb);               // (F0's registers are free)
      ODIN_SCHED_FENCE();
      mfma_chunk(F1, 3, stu, stv);   // items 3, 4
      fill_entries(en, T - T0 + NSETS + 1);
      thN = tt[T - T0 + 1];
      dgrad_tile(T, fb);
    } else {
      u32x4 fb[2][2][NPL];
      dgrad_frags(fb);
      ODIN_SCHED_FENCE();
      loads();
      dgrad_tile(T, fb);
      ODIN_SCHED_FENCE();
      if (!(DBG & 2) || T == T0) read_chunk(0, F0);   // (the partner wave of this SIMD is in its MFMAs meanwhile)
      if (!(DBG & 2) || T == T0) read_chunk(1, F1);
      ODIN_SCHED_FENCE();
      mfma_chunk(F0, 0, stu, stv);
      fill_entries(en, T - T0 + NSETS + 1);
      thN = tt[T - T0 + 1];
      mfma_chunk(F1, 3, stu, stv);
    }
    ooffP = ooff;
    auxP = auxN;
    pvP = pvN;
    if (!(DBG & 32)) __syncthreads();  // partial tiles complete; every wave is past tile T's rows; tile T + 1's rows are stored
  };
  // (two loops, one per order: a per-tile branch between the two bodies cost the register allocator ~50 spills)
  if (stag) {
#pragma unroll 1
    for (int T = T0; T < T1; T += 2) {
      run_tile(std::true_type{}, T, iu[1 % NSETS], iv[1 % NSETS], iu[0], iv[0]);
      if (T + 1 < T1) run_tile(std::true_type{}, T + 1, iu[0], iv[0], iu[1 % NSETS], iv[1 % NSETS]);
    }
  } else {
#pragma unroll 1
    for (int T = T0; T < T1; T += 2) {
      run_tile(std::false_type{}, T, iu[1 % NSETS], iv[1 % NSETS], iu[0], iv[0]);
      if (T + 1 < T1) run_tile(std::false_type{}, T + 1, iu[0], iv[0], iu[1 % NSETS], iv[1 % NSETS]);
    }
  }
  {
    float2 q8[8];
    finish_load((T1 - 1) & 1, q8);
    finish_done(q8);
  }

  // ---- this workgroup's slab row: dW[tap][co][cv0 + ci], lane = column ci = l31 ----
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11), a_o = odin_pow2(-ak);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = kh * 4 + kw0 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = fmaf(wacx[t][r], o_sx, wacc[t][r] * o_s);
      row[((size_t)tap * p.CUt + p.cu_off + cu) * p.CVt + cv0 + l31] = v * a_o;   // (a_o = 1 for an unscaled activation)
    }
  }
  if (PASS == 1) return;
  __syncthreads();  // (the partial-tile scratch is free: every wave is past its last finish pass)
  odin_amax_commit_wg(p.out_amax, amx, tid, 512, reinterpret_cast<float*>(red), blockIdx.x + gridDim.x * blockIdx.y);
  if (p.colsum != nullptr) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v = csum[k];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (l31 == 0) p.colsum[(size_t)blockIdx.x * p.CVt + c0 + k] = v;
    }
  }
}

// ---- producer / consumer form (round 5): 12 waves.  Waves 0-7 are the MFMA waves of bp_body (same tap ownership, same
// MFMA sequences, same results) and do NOTHING else: transposed / plain LDS reads, 24 MFMAs per tile, the partial data
// gradient tile -> LDS.  Waves 8-11 (one per SIMD, beside two MFMA waves) own everything with vector-ALU or memory
// work in it: the global loads of the row fills, split + LDS stores, the finish pass of the previous tile (sum of the
// eight partial tiles, ELU', column sums, range word, store of dx).  Why: in bp_body the matrix pipe's own time of a
// tile ADDS to everything else (profiles/r05_bwd_planes_ablations.txt) -- a wave's MFMAs overlap with the VALU
// instructions between them but not with its waits, and every wave had both kinds of work; a VALU-only partner wave
// runs at full speed beside an MFMA stream.  168 registers per wave (3 waves per SIMD).
template <int W, int PASS = 0>
__device__ __forceinline__ void bp_pc_body(const BPParams& p) {
  constexpr int NPL = 2;
  constexpr int TC = 32 / W;
  constexpr int WU = 2 * W;
  constexpr int SU = W + 1;
  constexpr int PARB = (SU + 1) * 64;
  constexpr int PBU = 2 * PARB;
  constexpr int RBU = NPL * PBU;
  constexpr int NSU = 4 * TC + 3;
  constexpr int PBV = W * 64;
  constexpr int RBV = NPL * PBV;
  constexpr int NSV = 2 * TC;
  constexpr int IPU = WU / 8;
  constexpr int IPV = W / 8;
  constexpr int RJ = 8 / IPU > 0 ? 8 / IPU : 1;
  constexpr int RED = 8 * 8 * 64 * 8;
  constexpr int NT = 768;
  constexpr int NPU = 8;                 // fine-row items of a producer wave per fill: item j = linear item 4 j + pw
  ODIN_DYN_SMEM(char, smem);
  char* uring = smem;
  char* vring = smem + NSU * RBU;
  char* red = vring + NSV * RBV;
  const int tid = threadIdx.x, lane = tid & 63;
  const OdinRangeReq g_rq = odin_range_issue(p.g_amax, lane), a_rq = odin_range_issue(p.a_amax, lane);
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const bool producer = wave >= 8;
  const int pw = wave & 3;               // producer index (waves 8-11)
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15;
  const int cv0 = blockIdx.y * 32;
  const int HU = 2 * p.h, HPU = HU + 1;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  constexpr int RPF = BP_MAXU * RJ;           // fine rows a fill can carry
  constexpr int DST_NONE = -(1 << 24);
  constexpr unsigned OFF_NONE = 0x7FFF0000u;
  const int NF = p.tiles_per_wg + 3;
  BpEnt* tt = reinterpret_cast<BpEnt*>(red + 2 * RED);  // [NF] tile: (first fine ring slot, byte offset of its first output)
  BpEnt* tr = tt + NF;                                 // [NF][RPF] fine row: (LDS byte offset, global byte offset)
  BpEnt* tv = tr + NF * RPF;                           // [NF][TC] coarse row: the same
  const unsigned u_rowbytes = (unsigned)(WU * p.CUt * 4), v_rowbytes = (unsigned)(W * p.CVt * 4);
  const int n_vrows = p.B * p.h;

  // ---- producers: lane constants of their items; the loads of fill 0 go out first ----
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * HU * WU * p.CUt * 4));
  const OdinRun RV = odin_run(p.V, (unsigned)((size_t)p.B * p.h * W * p.CVt * 4));
  const int ch4 = lane & 7, pxl = lane >> 3;
  // item j of producer pw: linear item L = 4 j + pw of the fill -> fill row L / IPU, column block L % IPU
  int u_lds[NPU];
  unsigned u_g[NPU];
#pragma unroll
  for (int j = 0; j < NPU; ++j) {
    const int L = 4 * j + pw, cb = L % IPU;
    const int pcw = 8 * cb + pxl + 1;
    u_lds[j] = (pcw & 1) * PARB + bp_uoff(pcw >> 1, 4 * ch4);
    u_g[j] = (unsigned)(((8 * cb + pxl) * p.CUt + p.cu_off + 4 * ch4) * 4);
  }
  const int vr = pw / IPV, vc = pw - vr * IPV;
  const unsigned v_g = (unsigned)(((8 * vc + pxl) * p.CVt + cv0 + 4 * ch4) * 4);
  const int v_lds = NSU * RBU + (8 * vc + pxl) * 64 + ch4 * 8;
  BpItem iu[2][NPU], iv[2];   // (three sets -- loads two tiles ahead of their LDS stores -- change nothing: 66.7 vs 63.6 us)
  if (producer) {
    const int tpi = p.tiles_per_img;
    const int b0 = odin_div_small(T0, tpi), t0 = T0 - b0 * tpi;
    const int start = HPU * b0 + 2 * TC * t0;
#pragma unroll
    for (int j = 0; j < NPU; ++j) {
      const int r = (4 * j + pw) / IPU, G = start + r;
      const bool valid = r < 2 * TC + 2;
      const int b = b0 + (2 * TC * t0 + r >= HPU ? 1 : 0), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;
      iu[0][j].dst = bp_uniform(valid ? (G - odin_div_small(G, NSU) * NSU) * RBU : DST_NONE);
      iu[0][j].v = odin_run_load4(RU, real ? (unsigned)(G - b - 1) * u_rowbytes + u_g[j] : ODIN_OOB);
    }
    const int grow = TC * T0 + vr;
    iv[0].dst = bp_uniform((grow & (NSV - 1)) * RBV);
    iv[0].v = odin_run_load4(RV, grow < n_vrows ? (unsigned)grow * v_rowbytes + v_g : ODIN_OOB);
  }
  // ---- MFMA waves: the weight fragments of the data gradient ----
  const int kh = (wave & 7) >> 1, kw0 = 2 * (wave & 1), kws = kw0 >> 1;
  float wv[2][2][8];
  if (!producer) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int tap = kh * 4 + kw0 + t;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          wv[t][kk][e] = p.w[((size_t)(tap * p.CUt + p.cu_off + 16 * kk + 8 * half + e)) * p.CVt + cv0 + l31];
      }
  }
  ODIN_SCHED_FENCE();

  // ---- all threads: pads, the zeroed partial-tile buffer of "tile T0 - 1", the fill tables ----
  for (int e = tid; e < NSU * 8 * NPL; e += NT) {
    const int sl = e / (8 * NPL), rem = e - sl * (8 * NPL);
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(uring + sl * RBU + pl * PBU + (side ? PARB + W * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int e = tid; e < RED / 16; e += NT)
    *reinterpret_cast<float4*>(red + ((T0 - 1) & 1) * RED + e * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int tpi = p.tiles_per_img;
    for (int e = tid; e < NF; e += NT) {
      const int T = T0 + e, b = odin_div_small(T, tpi), t = T - b * tpi;
      const int g0 = HPU * b + 2 * TC * t;
      tt[e] = BpEnt{g0 - odin_div_small(g0, NSU) * NSU, (int)(((size_t)(b * p.h + TC * t) * W) * p.CVt * 4)};
    }
    for (int e = tid; e < NF * RPF; e += NT) {
      const int f = e / RPF, r = e - f * RPF;
      const int T = T0 + f, b1 = odin_div_small(T, tpi), t1 = T - b1 * tpi;
      const int end = HPU * b1 + 2 * TC * t1 + 2 * TC + 2;
      int start = end - (2 * TC + 2);
      if (f > 0) {
        const int b0 = t1 > 0 ? b1 : b1 - 1, t0 = t1 > 0 ? t1 - 1 : tpi - 1;
        start = HPU * b0 + 2 * TC * t0 + 2 * TC + 2;
      }
      const int G = start + r;
      const bool valid = T < T1 && G < end;
      const int b = odin_div_small(G, HPU), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;
      tr[e] = BpEnt{valid ? (G - odin_div_small(G, NSU) * NSU) * RBU : DST_NONE,
                    real ? (int)((unsigned)(G - b - 1) * u_rowbytes) : (int)OFF_NONE};
    }
    for (int e = tid; e < NF * TC; e += NT) {
      const int f = e / TC, q = e - f * TC;
      const int T = T0 + f, grow = TC * T + q;
      const bool valid = T < T1;
      tv[e] = BpEnt{valid ? (grow & (NSV - 1)) * RBV : DST_NONE,
                    valid && grow < n_vrows ? (int)((unsigned)grow * v_rowbytes) : (int)OFF_NONE};
    }
  }

  // scales (both roles: the producers split with them, the MFMA waves scale their sums back)
  int gk = 0, ak = 0;
  bool as = false;
  float g_s = 1.f, g_s2k = ODIN_LO_SCALE, a_s = 1.f, a_s2k = ODIN_LO_SCALE, out_s = 1.f;
  {
    gk = odin_range_shift(odin_range_finish(g_rq));
    g_s = odin_pow2(gk); g_s2k = odin_pow2(gk + 11); out_s = odin_pow2(-gk);
    const unsigned a_mb = p.a_amax != nullptr ? odin_range_finish(a_rq) : 0u;
    as = odin_act_needs_scale(a_mb);
    ak = as ? odin_range_shift(a_mb) : 0;
    a_s = odin_pow2(ak); a_s2k = odin_pow2(ak + 11);
  }
  auto store_u = [&](const BpItem& it, int lds_lane) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;  // wave-uniform: a scalar branch
#endif
    u32x2 h, l;
    odin_split_h4<true>(it.v, g_s, g_s2k, h, l);
    char* d = smem + (it.dst + lds_lane);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBU) = l;
  };
  auto store_v = [&](const BpItem& it) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;
#endif
    u32x2 h, l;
    if (as) odin_split_h4<true>(it.v, a_s, a_s2k, h, l);
    else odin_split_h4<false>(it.v, 1.f, ODIN_LO_SCALE, h, l);
    char* d = smem + (it.dst + v_lds);
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBV) = l;
  };
  if (producer) {
#pragma unroll
    for (int j = 0; j < NPU; ++j) store_u(iu[0][j], u_lds[j]);
    store_v(iv[0]);
  }
  __syncthreads();   // pads, tables and the first tile's rows are in LDS

  if (producer) {
    // =========================== producer waves ===========================
    struct FillEnt { BpEnt u[NPU]; BpEnt v; };
    auto fill_entries = [&](FillEnt& en, int f) {
#pragma unroll
      for (int j = 0; j < NPU; ++j) {
        const BpEnt e = tr[f * RPF + (4 * j + pw) / IPU];
        en.u[j] = BpEnt{bp_uniform(e.x), bp_uniform(e.y)};
      }
      const BpEnt e = tv[f * TC + vr];
      en.v = BpEnt{bp_uniform(e.x), bp_uniform(e.y)};
    };
    auto fill_loads = [&](BpItem (&u)[NPU], BpItem& v, const FillEnt& en) {
#pragma unroll
      for (int j = 0; j < NPU; ++j) {
        u[j].dst = en.u[j].x;
        u[j].v = odin_run_load4(RU, (unsigned)en.u[j].y + u_g[j]);
      }
      v.dst = en.v.x;
      v.v = odin_run_load4(RV, (unsigned)en.v.y + v_g);
    };
    // finish pass: this wave owns the accumulator register pairs 2 pw, 2 pw + 1 of every tile
    const int orow = (W == 32) ? 0 : (W == 16) ? (l31 >> 4) : (l31 >> 3);
    const int ocol = (W == 32) ? l31 : (W == 16) ? (l31 & 15) : (l31 & 7);
    int c0[2];
    unsigned o_lane[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pr = 2 * pw + k;
      c0[k] = cv0 + ((2 * pr) & 3) + 8 * ((2 * pr) >> 2) + 4 * half;
      o_lane[k] = (unsigned)(((orow * W + ocol) * p.CVt + c0[k]) * 4);
    }
    const unsigned out_bytes = (unsigned)((size_t)p.B * p.h * W * p.CVt * 4);
    const OdinRun RO = odin_run(p.dx, out_bytes);
    const OdinRun RX = odin_run(p.aux, out_bytes);
    float csum[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float amx = 0.f;
    unsigned tileoffP = ODIN_OOB;   // (the first tile's pass has no predecessor: its stores are out of range)
    float2 auxP[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)}, pvP[2] = {auxP[0], auxP[0]};
    auto finish = [&](int buf) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int pr = 2 * pw + k;
        const char* q = red + buf * RED + ((pr * 8 * 64 + lane) << 3);
        float2 q8[8];
#pragma unroll
        for (int wvv = 0; wvv < 8; ++wvv) q8[wvv] = *reinterpret_cast<const float2*>(q + wvv * (64 * 8));
        float2 s2 = q8[0];
#pragma unroll
        for (int wvv = 1; wvv < 8; ++wvv) { s2.x += q8[wvv].x; s2.y += q8[wvv].y; }
        float v[2] = {s2.x * out_s, s2.y * out_s};
        if (PASS == 2) { v[0] += pvP[k].x; v[1] += pvP[k].y; }
        if (PASS != 1) {
          v[0] = fmaf(v[0], fminf(auxP[k].x, 0.f), v[0]);  // x ELU'(aux) = 1 + min(aux, 0)
          v[1] = fmaf(v[1], fminf(auxP[k].y, 0.f), v[1]);
          csum[k][0] += v[0];
          csum[k][1] += v[1];
          amx = odin_amax3(amx, v[0], v[1]);
        }
        odin_run_store2(RO, tileoffP == ODIN_OOB ? ODIN_OOB : tileoffP + o_lane[k], make_float2(v[0], v[1]));
      }
    };
    FillEnt en;
    {
      FillEnt e1;
      fill_entries(e1, 1);
      fill_entries(en, 2);
      fill_loads(iu[1], iv[1], e1);
    }
    // (set 1 holds fill 1 = the rows of tile T0 + 1, stored during tile T0; set 0 receives fill 2 during tile T0)
    auto tile = [&](int T, BpItem (&ldu)[NPU], BpItem& ldv, const BpItem (&stu)[NPU], const BpItem& stv) {
      const BpEnt th = tt[T - T0];
      fill_loads(ldu, ldv, en);        // fill T - T0 + 2
      float2 auxN[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)}, pvN[2] = {auxN[0], auxN[0]};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (PASS != 1) auxN[k] = odin_run_load2(RX, (unsigned)th.y + o_lane[k]);
        if (PASS == 2) pvN[k] = odin_run_load2(RO, (unsigned)th.y + o_lane[k]);
      }
      finish((T - 1) & 1);             // tile T - 1: its partials are complete behind the last barrier
      fill_entries(en, T - T0 + 3);
#pragma unroll
      for (int j = 0; j < NPU; ++j) store_u(stu[j], u_lds[j]);   // rows of tile T + 1
      store_v(stv);
      tileoffP = (unsigned)th.y;
#pragma unroll
      for (int k = 0; k < 2; ++k) { auxP[k] = auxN[k]; pvP[k] = pvN[k]; }
      __syncthreads();
    };
#pragma unroll 1
    for (int T = T0; T < T1; T += 2) {
      tile(T, iu[0], iv[0], iu[1], iv[1]);
      if (T + 1 < T1) tile(T + 1, iu[1], iv[1], iu[0], iv[0]);
    }
    finish((T1 - 1) & 1);
    if (PASS == 1) return;
    __syncthreads();  // (pairs with the MFMA waves' barrier below: the partial-tile scratch is free)
    odin_amax_commit_wg(p.out_amax, amx, tid, NT, reinterpret_cast<float*>(red), blockIdx.x + gridDim.x * blockIdx.y);
    if (p.colsum != nullptr) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float v = csum[k][c];
#pragma unroll
          for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
          if (l31 == 0) p.colsum[(size_t)blockIdx.x * p.CVt + c0[k] + c] = v;
        }
    }
    return;
  }

  // =========================== MFMA waves (0-7) ===========================
  const int q4 = l16 >> 2, g16 = (lane >> 4) & 1;
  int krow[2], kcol[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int kpix = 8 * half + 4 * blk + q4;
    krow[blk] = (W >= 16) ? 0 : (kpix >> 3);
    kcol[blk] = (W >= 16) ? kpix : (kpix & 7);
  }
  const int colb = (16 * g16 + 4 * (l16 & 3)) * 2;
  constexpr int NCH = (W == 32) ? 2 : 1;
  int uoff[NCH][2];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) uoff[c][blk] = bp_uoff(16 * c + kcol[blk] + kws, 16 * g16 + 4 * (l16 & 3));
  const int orow = (W == 32) ? 0 : (W == 16) ? (l31 >> 4) : (l31 >> 3);
  const int ocol = (W == 32) ? l31 : (W == 16) ? (l31 & 15) : (l31 & 7);
  int boff[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) boff[kk] = bp_uoff(ocol + kws, 8 * (2 * kk + half));
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
  f32x16 wacc[2] = {f32x16_zero(), f32x16_zero()};
  f32x16 wacx[2] = {f32x16_zero(), f32x16_zero()};
  int su0 = 0, sv0 = 0;
  struct Frags { u32x4 fv[NPL]; u32x4 fu[2][NPL]; };
  auto read_chunk = [&](int c, Frags& F) {
    const int row0 = (W == 32) ? 0 : (W == 16) ? c : 2 * c;
    const int j0 = (W == 32) ? 16 * c : 0;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      int sv = sv0 + row0 + krow[blk];
      if (sv >= NSV) sv -= NSV;
      int su = su0 + 2 * (row0 + krow[blk]) + kh;
      if (su >= NSU) su -= NSU;
      const char* vb = vring + sv * RBV + (j0 + kcol[blk] - q4) * 64 + colb - (l16 & 3) * 8;
      const char* ub = uring + su * RBU;
      const int uo = uoff[W == 32 ? c : 0][blk];
      const int slot0 = j0 + kcol[blk] - q4 + kws;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x2 tvv = bp_tr_read_v(vb + pl * PBV, l16);
        F.fv[pl][2 * blk] = tvv.x; F.fv[pl][2 * blk + 1] = tvv.y;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const u32x2 tu = bp_tr_read_u(ub + pl * PBU + t * PARB, slot0, g16, l16, uo);
          F.fu[t][pl][2 * blk] = tu.x; F.fu[t][pl][2 * blk + 1] = tu.y;
        }
      }
    }
  };
  auto mfma_chunk = [&](const Frags& F) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int t = m & 1, pp = m >> 1;
      if (pp == 0) wacx[t] = mfma32_f16(F.fu[t][0], F.fv[1], wacx[t]);
      if (pp == 1) wacx[t] = mfma32_f16(F.fu[t][1], F.fv[0], wacx[t]);
      if (pp == 2) wacc[t] = mfma32_f16(F.fu[t][0], F.fv[0], wacc[t]);
    }
  };
  auto dgrad_frags = [&](u32x4 (&fb)[2][2][NPL]) {
    int sud = su0 + 2 * orow + kh;
    sud -= sud >= NSU ? NSU : 0;
    const char* rowp = uring + sud * RBU;
#pragma unroll
    for (int pl = NPL - 1; pl >= 0; --pl)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          fb[t][kk][pl] = *reinterpret_cast<const u32x4*>(rowp + t * PARB + boff[kk] + pl * PBU);
  };
  auto dgrad_tile = [&](int T, const u32x4 (&fb)[2][2][NPL]) {
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
#pragma unroll
    for (int m = 0; m < 12; ++m) {
      const int t = (m >> 1) & 1, kk = m & 1, pp = m >> 2;
      if (pp == 0) acx = mfma32_f16(wf[t][kk][0], fb[t][kk][1], acx);
      if (pp == 1) acx = mfma32_f16(wf[t][kk][1], fb[t][kk][0], acx);
      if (pp == 2) acc = mfma32_f16(wf[t][kk][0], fb[t][kk][0], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaf(acx[r], ODIN_LO_UNSCALE, acc[r]);
    char* d = red + (T & 1) * RED + (((wave & 7) * 64 + lane) << 3);
#pragma unroll
    for (int pr = 0; pr < 8; ++pr)
      *reinterpret_cast<float2*>(d + pr * (8 * 64 * 8)) = make_float2(acc[2 * pr], acc[2 * pr + 1]);
  };
  Frags F0;
  int suN = tt[0].x;
  // (waves w and w + 4 share a SIMD: they run the tile's halves in opposite order, one's reads in the other's MFMAs)
  const bool dfirst = wave >= 4;
  auto tile = [&](auto dfirst_c, int T) {
    constexpr bool df = decltype(dfirst_c)::value;
    su0 = suN;
    sv0 = (TC * T) & (NSV - 1);
    suN = tt[T - T0 + 1].x;   // (read a tile ahead: no LDS round trip in front of the tile's first fragment reads)
    if (!df) {
      Frags F1;
      read_chunk(0, F0);
      read_chunk(1, F1);
      ODIN_SCHED_FENCE();
      mfma_chunk(F0);
      ODIN_SCHED_FENCE();
      mfma_chunk(F1);
      ODIN_SCHED_FENCE();
      u32x4 fb[2][2][NPL];
      dgrad_frags(fb);
      ODIN_SCHED_FENCE();
      dgrad_tile(T, fb);
    } else {
      {
        u32x4 fb[2][2][NPL];
        dgrad_frags(fb);
        ODIN_SCHED_FENCE();
        dgrad_tile(T, fb);
      }
      ODIN_SCHED_FENCE();
      Frags F1;
      read_chunk(0, F0);
      read_chunk(1, F1);
      ODIN_SCHED_FENCE();
      mfma_chunk(F0);
      ODIN_SCHED_FENCE();
      mfma_chunk(F1);
    }
    __syncthreads();
  };
  if (dfirst) {
#pragma unroll 1
    for (int T = T0; T < T1; ++T) tile(std::true_type{}, T);
  } else {
#pragma unroll 1
    for (int T = T0; T < T1; ++T) tile(std::false_type{}, T);
  }
  // ---- this workgroup's slab row ----
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11), a_o = odin_pow2(-ak);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = kh * 4 + kw0 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = fmaf(wacx[t][r], o_sx, wacc[t][r] * o_s);
      row[((size_t)tap * p.CUt + p.cu_off + cu) * p.CVt + cv0 + l31] = v * a_o;
    }
  }
  if (PASS == 1) return;
  __syncthreads();
  odin_amax_commit_wg(p.out_amax, 0.f, tid, NT, reinterpret_cast<float*>(red), blockIdx.x + gridDim.x * blockIdx.y);
}

template <int W>
__global__ __launch_bounds__(768) void bwd_planes_pc_kernel(BPParams p) {
  bp_pc_body<W, 0>(p);
}

template <int W>
__global__ __launch_bounds__(768) void bwd_planes_pc2_kernel(BPParams p) {
  p.cu_off = 0;
  bp_pc_body<W, 1>(p);
  odin_wait_vmem();
#ifndef ODIN_SIM
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  __syncthreads();
  p.cu_off = 32;
  bp_pc_body<W, 2>(p);
}

template <int W, int NSETS, int DBG = 0>
__global__ __launch_bounds__(512) void bwd_planes_kernel(BPParams p) {
  bp_body<W, NSETS, DBG>(p);
}

// 64 output channels: both 32-channel passes in ONE launch (fconv_planes.hip: fconv_planes2_kernel); the partial sums a
// thread leaves in dx are read back by the same thread
template <int W, int NSETS>
__global__ __launch_bounds__(512) void bwd_planes2_kernel(BPParams p) {
  p.cu_off = 0;
  bp_body<W, NSETS, 0, 1>(p);
  odin_wait_vmem();
#ifndef ODIN_SIM
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
  __syncthreads();
  p.cu_off = 32;
  bp_body<W, NSETS, 0, 2>(p);
}

constexpr int BP_NSETS = 3;   // (upper bound: sizes the fill tables)
constexpr int BP_LDS_MAX = 160 * 1024;
int bp_ring_bytes(int W) {
  const int TC = 32 / W;
  return (4 * TC + 3) * 2 * 2 * (W + 2) * 64 + (2 * TC) * 2 * W * 64 + 2 * (8 * 8 * 64 * 8);
}
int bp_fill_bytes(int W) {
  const int ipu = 2 * W / 8, rj = 8 / ipu > 0 ? 8 / ipu : 1;
  return 8 * (1 + BP_MAXU * rj + 32 / W);
}
// tiles per workgroup: exactly what BOTH stand-alone kernels would choose (the slab rows and the column-sum rows are
// then the same partial sums); -1: they differ or the tables do not fit
int bp_tiles_per_wg(int W, int n_tiles, int gy) {
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  const int cap_slab = cap > ODIN_MAX_SLAB_BLOCKS ? ODIN_MAX_SLAB_BLOCKS : cap;
  const int cap_col = cap > ODIN_MAX_COLSUM_BLOCKS ? ODIN_MAX_COLSUM_BLOCKS : cap;
  if (cap_slab != cap_col) return -1;
  const int tpw = (n_tiles + cap_slab - 1) / cap_slab;
  const int limit = (BP_LDS_MAX - bp_ring_bytes(W)) / bp_fill_bytes(W) - (BP_NSETS + 1);
  if (tpw > limit) return -1;
  return tpw;
}

template <int W, int NS>
int bp_launch_n(const BPParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)bp_ring_bytes(W) + (size_t)(p.tiles_per_wg + BP_NSETS + 1) * bp_fill_bytes(W);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_planes_kernel<W, NS>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((bwd_planes_kernel<W, NS>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("bwd_planes(f16x2)");
}
template <int W>
int bp_launch2(const BPParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)bp_ring_bytes(W) + (size_t)(p.tiles_per_wg + BP_NSETS + 1) * bp_fill_bytes(W);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_planes2_kernel<W, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((bwd_planes2_kernel<W, 2>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("bwd_planes(f16x2)");
}
template <int W>
int bp_launch_pc(const BPParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)bp_ring_bytes(W) + (size_t)(p.tiles_per_wg + BP_NSETS + 1) * bp_fill_bytes(W);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_planes_pc_kernel<W>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_planes_pc2_kernel<W>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  if (p.CUt == 64) ODIN_LAUNCH((bwd_planes_pc2_kernel<W>), grid, dim3(768), lds, stream, p);
  else ODIN_LAUNCH((bwd_planes_pc_kernel<W>), grid, dim3(768), lds, stream, p);
  return odin_check_launch("bwd_planes_pc(f16x2)");
}
template <int W>
int bp_launch(const BPParams& p, dim3 grid, void* stream) {
  // the producer / consumer form (12 waves) is the default; the 8-wave form it grew out of remains for A/Bs (diagnostics
  // build: ODIN_BP_8WAVE) -- both give the same bits
  // (rows of 8 pixels -- 1 to 4 tiles per workgroup -- stay on the 8-wave form: 29.2 vs 30.5 us on decoder2)
  if (W >= 16 && !ODIN_DIAG_ENV("ODIN_BP_8WAVE")) return bp_launch_pc<W>(p, grid, stream);
  if (p.CUt == 64) return bp_launch2<W>(p, grid, stream);
#if !defined(ODIN_SIM) && defined(ODIN_DIAG)
  // diagnostics build only (make diag): instances with parts of the tile switched off -- they compute WRONG results;
  // profiles/r05_bwd_planes_ablations.txt
  const char* e = getenv("ODIN_BP_DBG");
  if (e != nullptr && W == 32) {
    const int dbg = atoi(e);
    const size_t lds = (size_t)bp_ring_bytes(W) + (size_t)(p.tiles_per_wg + BP_NSETS + 1) * bp_fill_bytes(W);
#define BP_DBG_CASE(D) if (dbg == D) { hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_planes_kernel<32, 2, D>), hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_MAX); ODIN_LAUNCH((bwd_planes_kernel<32, 2, D>), grid, dim3(512), lds, stream, p); return odin_check_launch("bwd_planes(dbg)"); }
    BP_DBG_CASE(1) BP_DBG_CASE(2) BP_DBG_CASE(4) BP_DBG_CASE(16) BP_DBG_CASE(32) BP_DBG_CASE(3) BP_DBG_CASE(6) BP_DBG_CASE(22) BP_DBG_CASE(23) BP_DBG_CASE(55) BP_DBG_CASE(8) BP_DBG_CASE(40) BP_DBG_CASE(9) BP_DBG_CASE(64)
  }
#endif
  return bp_launch_n<W, 2>(p, grid, stream);
}

}  // namespace

// Conv2DTranspose(k4, s2) with Cout = 32: x [B, H, W, Cin] -> dy [B, 2H, 2W, 32]; aux: ELU activations below
bool odin_bwd_planes_applicable(int B, int H, int W, int Cin, int Cout) {
  if (odin_blk_first()) return false;   // (diagnostics: odin_debug_blk_first)
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NOPLANES") || ODIN_DIAG_ENV("ODIN_NOBWDPLANES")) return false;
  if (!((Cout == 32 || Cout == 64) && (Cin % 32) == 0 && (W == 8 || W == 16 || W == 32) && (H % (32 / W)) == 0)) return false;
  if (!((size_t)B * 2 * H * 2 * W * Cout * 4 < 0x7FFF0000ull && (size_t)B * H * W * Cin * 4 < 0x7FFF0000ull)) return false;
  return bp_tiles_per_wg(W, B * (H / (32 / W)), Cin / 32) > 0;
}

int odin_bwd_planes_rows(int B, int H, int W, int Cin) {
  const int n_tiles = B * (H / (32 / W));
  const int tpw = bp_tiles_per_wg(W, n_tiles, Cin / 32);
  return tpw > 0 ? (n_tiles + tpw - 1) / tpw : 0;
}

int odin_bwd_planes_launch(const float* x, const float* dy, const float* w, const float* aux, float* dx, float* colsum,
                           float* wslab, int B, int H, int W, int Cin, int Cout, const uint32_t* dy_amax,
                           const uint32_t* x_amax, uint32_t* dx_amax, void* stream) {
  BPParams p;
  memset(&p, 0, sizeof(p));
  p.U = dy; p.V = x; p.w = w; p.aux = aux; p.dx = dx; p.colsum = colsum; p.slab = wslab;
  p.B = B; p.h = H; p.CVt = Cin; p.CUt = Cout; p.cu_off = 0;
  p.slab_stride = 16 * Cout * Cin;
  const int TC = 32 / W;
  p.tiles_per_img = H / TC;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = Cin / 32;
  p.tiles_per_wg = bp_tiles_per_wg(W, p.n_tiles, gy);
  if (p.tiles_per_wg <= 0) return odin_fail(-2, "bwd_planes: not applicable");
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  p.g_amax = odin_range_word_of(dy, (size_t)B * 2 * H * 2 * W * Cout, dy_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "bwd_planes: no range word for dy");
  p.a_amax = x_amax;
  p.out_amax = dx_amax;
  dim3 grid(gx, gy, 1);
  if (W == 32) return bp_launch<32>(p, grid, stream);
  if (W == 16) return bp_launch<16>(p, grid, stream);
  return bp_launch<8>(p, grid, stream);
}
