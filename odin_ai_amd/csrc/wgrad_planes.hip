// wgrad_planes.hip -- WEIGHT GRADIENT of the 4x4 / stride-2 layers (Conv2D and Conv2DTranspose, TF
// `SAME`, pads (1, 1)) with BOTH fp32 operands carried through the f16 matrix pipe as two f16 planes
// (odin_device.h: x = h + 2^-11 l; three plane products per 16 k-values into a main and a cross fp32
// accumulator, <= 3 * 2^-22 per product; the gradient operand is carried times 2^gexp).
//
// One formulation serves both layer kinds.  With a FINE tensor U [B, 2h, 2w, CU] and a COARSE tensor
// V [B, h, w, CV] related by coarse pixel (i, j) <-> fine pixel (2 i - 1 + kh, 2 j - 1 + kw):
//       dW[kh][kw][cu][cv] = sum over (b, i, j) of U[b, 2 i - 1 + kh, 2 j - 1 + kw, cu] * V[b, i, j, cv]
//   Conv2D          (image_networks.py:462-468): U = layer input,            V = dL/d pre-activation,
//                                                dW in Keras' (kh, kw, Cin, Cout); db = column sums of V
//   Conv2DTranspose (image_networks.py:499-506): U = dL/d pre-activation,    V = layer input,
//                                                dW in Keras' (kh, kw, Cout, Cin)
// (tape.gradient of the step, base_networks.py:549.)  The MFMA reduction index is the PIXEL: a
// v_mfma_f32_32x32x16_f16 multiplies A = U^T [32 cu][16 pixels] by B = V [16 pixels][32 cv].  Both
// operands are channel-major fragments of pixel-major data: the LDS images stay [pixel][32 channels]
// (written exactly as in tconv_planes.hip: global_load -> split once -> two ds_write_b64) and the
// fragments are fetched with ds_read_b64_tr_b16, the transposing LDS read of gfx950 (a 16-lane group
// reads 4 pixels x 16 channels and every lane receives one channel's 4 pixels).  A shift by one
// coarse column (kw >= 2) is a shift by one 64-byte pixel slot: always aligned.
//
// Structure: 8 waves, all alike; wave v owns the taps (kh = v >> 1, kw = 2 (v & 1) + {0, 1}) and keeps
// their two 32 x 32 accumulators in registers over the whole persistent tile loop.  A tile = 32
// coarse pixels (1, 2 or 4 coarse rows) = 2 chunks of 16: per chunk 6 transposed reads of V (shared by
// the wave's two taps), 12 of U, 12 MFMAs.  Rolling row windows in LDS (slot = global padded row mod
// NS), fine rows as two column-parity planes so that the 16 pixels of a chunk are consecutive slots;
// per chunk 4 transposed reads of V, 8 of U, 6 MFMAs;
// the split + store of the next tile's rows and the global loads of the one after ride in the MFMA
// stream; one workgroup barrier per tile.  No epilogue: the accumulators go to this workgroup's slab
// row once, at the end (odin_slab_reduce sums the rows in fixed order).
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct WPParams {
  const float* U;      // [B, 2h, 2w, CUt]
  const float* V;      // [B, h, w, CVt]
  float* slab;         // [gridDim.x][16 * CUt * CVt (+ CVt)]
  int B, h;
  int CUt, CVt;        // channels per pixel in memory; this workgroup's 32 start at 32 * blockIdx.y / .z
  int want_bias;
  int slab_stride;
  int tiles_per_img, n_tiles, tiles_per_wg;
  const unsigned* g_amax;  // range word (odin_device.h) of the gradient operand (GU: U, else V)
  const unsigned* a_amax;  // optional range word of the ACTIVATION operand (the other one): scaled when outside [2^-8, 2^15)
};

struct WPQueue {
  struct Job {
    WPParams p;
    dim3 grid;
    size_t lds;
    int kind;
  };
  bool defer = false;
  int n = 0;
  void* stream = nullptr;
  Job job[8];
};
thread_local WPQueue g_wpq;

// ds_read_b64_tr_b16: `blk` = byte address of a block of 4 rows (`stride` bytes apart) x 16 bf16 columns;
// lane l16 of the 16-lane group receives column l16 of the 4 rows (row q in element q).  On the
// hardware lane 4 q + p supplies the address of row q, columns 4 p .. 4 p + 3.
__device__ __forceinline__ u32x2 wp_tr_read(const char* blk, int stride, int l16) {
#ifdef ODIN_SIM
  unsigned short e[4];
  for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const unsigned short*>(blk + q * stride + 2 * l16);
  return odin_u2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
#else
  typedef short wp_s4 __attribute__((ext_vector_type(4)));
  const char* a = blk + (l16 >> 2) * stride + (l16 & 3) * 8;
  const wp_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wp_s4*)a);
  return __builtin_bit_cast(u32x2, v);
#endif
}

struct alignas(8) WpEnt {
  int x, y;
};

struct WpItem {
  float4 v;
  int dst;  // byte offset of the hi-plane store inside the LDS image; < 0: no item (wave-uniform)
};

constexpr int WP_MAXU = 4;  // 1 KB load items (8 pixels x 32 channels) of fine rows per wave and fill

// W = coarse row length (8, 16 or 32); DBG (diagnostics, ODIN_WP_DBG): 1 no MFMAs, 2 no LDS reads in the loop,
// 4 no row fills after the prologue
// GU: the gradient operand is U (Conv2DTranspose), else V (Conv2D)
// (bx, by, bz): the workgroup's block coordinates -- blockIdx of a launch of its own, decoded from a linear index in
// the multi-layer launch below
template <int W, bool GU, int DBG = 0>
__device__ __forceinline__ void wp_body(const WPParams& p, const int bx, const int by, const int bz) {
  __shared__ float bred[8 * 32];
  constexpr int NPL = 2;                 // f16 planes per operand
  constexpr int TC = 32 / W;             // coarse rows per tile
  constexpr int WU = 2 * W;              // fine row length
  constexpr int SU = W + 1;              // slots per column-parity plane of a fine row
  // one parity plane.  (A spare slot per plane -- fconv_planes.hip -- spreads the row fills' stores over all banks
  // but moves the parity-1 plane's transposed reads onto shared banks: measured 66.5 -> 70.1 us on decoder4.)
  constexpr int PARB = SU * 64;
  constexpr int PBU = 2 * PARB;          // one bf16 plane of a fine row
  constexpr int RBU = NPL * PBU;
  constexpr int NSU = 4 * TC + 3;        // live fine rows (2 TC + 2) + the next tile's (2 TC + 1 at an image seam)
  constexpr int PBV = W * 64;
  constexpr int RBV = NPL * PBV;
  constexpr int NSV = 2 * TC;
  constexpr int IPU = WU / 8;            // load items per fine row: 8, 4, 2
  constexpr int IPV = W / 8;             // per coarse row: 4, 2, 1
  ODIN_DYN_SMEM(char, smem);
  char* uring = smem;
  char* vring = smem + NSU * RBU;
  const int tid = threadIdx.x, lane = tid & 63;
  // the two range words: requested first thing, finished in front of the first split (odin_device.h: odin_range_issue)
  const OdinRangeReq g_rq = odin_range_issue(p.g_amax, lane), a_rq = odin_range_issue(p.a_amax, lane);
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15;
  const int cu0 = by * 32, cv0 = bz * 32;
  const int HU = 2 * p.h, HPU = HU + 1;
  const int T0 = bx * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  // ---- row fills (wave-uniform walks; a wave moves up to WP_MAXU fine-row items and one coarse-row item) ----
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * HU * WU * p.CUt * 4));
  const OdinRun RV = odin_run(p.V, (unsigned)((size_t)p.B * p.h * W * p.CVt * 4));
  float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);  // column sums of V (channels 4 (lane & 7) ..) for the bias
  // lane-constant parts of an item: pixel inside the 8-pixel item, channel quad, LDS / global offsets
  const int ch4 = lane & 7, pxl = lane >> 3;
  const int pcl = pxl + 1;  // padded column of the item's pixel when the item starts at column 0
  // a fine-row item at columns 8 c ..: padded column 8 c + pcl -> parity pcl & 1, slot 4 c + (pcl >> 1)
  const int u_lds_lane = (pcl & 1) * PARB + (pcl >> 1) * 64 + ch4 * 8;
  const int v_lds_lane = pxl * 64 + ch4 * 8;
  const unsigned u_g_lane = (unsigned)((pxl * p.CUt + cu0 + 4 * ch4) * 4);
  const unsigned v_g_lane = (unsigned)((pxl * p.CVt + cv0 + 4 * ch4) * 4);
  // item j of this wave: row r0 + RJ j of the fill, 8-pixel column block cu_blk (all wave constants;
  // 32-bit offsets throughout -- the applicability test bounds the tensors to 2 GB -- because this is
  // SCALAR work of every wave on every tile: with 64-bit products it cost ~1000 cycles per tile)
  constexpr int RJ = 8 / IPU > 0 ? 8 / IPU : 1;
  const int r0 = wave / IPU, cu_blk = wave - r0 * IPU;
  const unsigned u_rowbytes = (unsigned)(WU * p.CUt * 4), v_rowbytes = (unsigned)(W * p.CVt * 4);
  const unsigned u_colb = (unsigned)(8 * cu_blk * p.CUt * 4) + u_g_lane;
  const int u_lds_item = 4 * cu_blk * 64 + u_lds_lane;
  const int vr = (wave & 3) / IPV, vc = (wave & 3) - vr * IPV;
  const unsigned v_colb = (unsigned)(8 * vc * p.CVt * 4) + v_g_lane;
  const int v_lds_item = NSU * RBU + 8 * vc * 64 + v_lds_lane;
  const int n_vrows = p.B * p.h;
  // ---- the loads of fill 0 (all rows of tile T0, offsets computed directly) go out FIRST: they are in flight while
  // the zero fills and the table arithmetic run (the layers with 8- and 16-pixel rows run few tiles per workgroup) ----
  WpItem iuA[WP_MAXU], iuB[WP_MAXU], iuC[WP_MAXU], ivA, ivB, ivC;
  {
    const int tpi = p.tiles_per_img;
    const int b0 = odin_div_small(T0, tpi), t0 = T0 - b0 * tpi;
    const int start = HPU * b0 + 2 * TC * t0;
#pragma unroll
    for (int j = 0; j < WP_MAXU; ++j) {
      const int r = r0 + RJ * j, G = start + r;
      const bool valid = r < 2 * TC + 2;
      const int b = b0 + (2 * TC * t0 + r >= HPU ? 1 : 0), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      iuA[j].dst = valid ? (G % NSU) * RBU + u_lds_item : -1;
      iuA[j].v = odin_run_load4(RU, real ? (unsigned)(G - b - 1) * u_rowbytes + u_colb : ODIN_OOB);
    }
    const int grow = TC * T0 + vr;
    ivA.dst = wave < 4 ? (grow & (NSV - 1)) * RBV + v_lds_item : -1;
    ivA.v = odin_run_load4(RV, (wave < 4 && grow < n_vrows) ? (unsigned)grow * v_rowbytes + v_colb : ODIN_OOB);
  }
  ODIN_SCHED_FENCE();

  // ---- SAME-padding slots of every fine ring row and plane: parity plane 0 slot 0 (padded column 0)
  // and parity plane 1 slot W (padded column 2 w + 1): zero for ever ----
  for (int e = tid; e < NSU * 8 * NPL; e += 512) {
    const int sl = e / (8 * NPL), rem = e - sl * (8 * NPL);
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(uring + sl * RBU + pl * PBU + (side ? PARB + W * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // Which rows a fill moves and where they land (image seams, ring wrap-arounds) is index arithmetic that cost
  // ~170 dependent scalar instructions per tile inside the MFMA stream (fconv_planes.hip: 14 of 71 us).  It is done
  // once here, by all threads, into LDS tables; the tile loop reads its entries (wave-uniform addresses) and adds
  // lane offsets.  Fill f >= 1 brings the fine rows tile T0 + f needs beyond those of tile T0 + f - 1 (fill 0: all of
  // tile T0's) and the TC coarse rows of tile T0 + f.
  constexpr int RPF = WP_MAXU * RJ;           // fine rows a fill can carry (row r = r0 + RJ j of item j)
  constexpr int DST_NONE = -(1 << 24);        // LDS offset of an item without a row: dst stays negative
  constexpr unsigned OFF_NONE = 0x7FFF0000u;  // global offset of a row that is not read: out of range
  const int NF = p.tiles_per_wg + 4;
  WpEnt* tt = reinterpret_cast<WpEnt*>(vring + NSV * RBV);  // [NF] tile: (first fine ring slot, first coarse ring slot)
  WpEnt* tr = tt + NF;                                      // [NF][RPF] fine row: (LDS byte offset, global byte offset)
  WpEnt* tv = tr + NF * RPF;                                // [NF][TC] coarse row: the same
  {
    const int tpi = p.tiles_per_img;
    for (int e = tid; e < NF; e += 512) {
      const int T = T0 + e, b = odin_div_small(T, tpi), t = T - b * tpi;
      tt[e] = WpEnt{(HPU * b + 2 * TC * t) % NSU, (TC * T) % NSV};
    }
    for (int e = tid; e < NF * RPF; e += 512) {
      const int f = e / RPF, r = e - f * RPF;
      const int T = T0 + f, b1 = odin_div_small(T, tpi), t1 = T - b1 * tpi;
      const int end = HPU * b1 + 2 * TC * t1 + 2 * TC + 2;
      int start = end - (2 * TC + 2);
      if (f > 0) {
        const int b0 = t1 > 0 ? b1 : b1 - 1, t0 = t1 > 0 ? t1 - 1 : tpi - 1;  // tile T - 1: this image or the one before
        start = HPU * b0 + 2 * TC * t0 + 2 * TC + 2;
      }
      const int G = start + r;  // global padded fine row HPU * b + gi; gi == 0: the zero row between images
      const bool valid = T < T1 && G < end;
      const int b = odin_div_small(G, HPU), gi = G - b * HPU;
      const bool real = valid && gi != 0 && b < p.B;
      tr[e] = WpEnt{valid ? (G % NSU) * RBU : DST_NONE, real ? (int)((unsigned)(G - b - 1) * u_rowbytes) : (int)OFF_NONE};
    }
    for (int e = tid; e < NF * TC; e += 512) {
      const int f = e / TC, q = e - f * TC;
      const int T = T0 + f, grow = TC * T + q;  // global coarse row h * b + i
      const bool valid = T < T1;
      tv[e] = WpEnt{valid ? (grow & (NSV - 1)) * RBV : DST_NONE,
                    valid && grow < n_vrows ? (int)((unsigned)grow * v_rowbytes) : (int)OFF_NONE};
    }
  }
  // waves 4-7 carry no coarse-row item
  const int v_none_dst = wave < 4 ? 0 : (int)0x80000000;
  const unsigned v_none_off = wave < 4 ? 0u : OFF_NONE;
  struct FillEnt { WpEnt u[WP_MAXU]; WpEnt v; };
  auto fill_entries = [&](FillEnt& en, int f) {
#pragma unroll
    for (int j = 0; j < WP_MAXU; ++j) en.u[j] = tr[f * RPF + r0 + RJ * j];
    en.v = tv[f * TC + vr];
  };
  // (the loads are unconditional -- an absent item reads zeros through the range check -- so that the number of
  // loads in flight is a compile-time constant: the wait before a store is vmcnt(N), not vmcnt(0))
  auto fill_loads = [&](WpItem (&iu)[WP_MAXU], WpItem& iv, const FillEnt& en) {
#pragma unroll
    for (int j = 0; j < WP_MAXU; ++j) {
      iu[j].dst = en.u[j].x + u_lds_item;  // negative: no row
      iu[j].v = odin_run_load4(RU, (unsigned)en.u[j].y + u_colb);
    }
    iv.dst = (en.v.x + v_lds_item) | v_none_dst;
    iv.v = odin_run_load4(RV, ((unsigned)en.v.y + v_colb) | v_none_off);
  };
  // the gradient operand is carried times 2^gk (its maximum lands in [2^14, 2^15)), the sums are scaled back at the end.
  // The ACTIVATION operand comes with an optional range word (round 5): carried times 2^ak as well when its bound
  // leaves [2^-8, 2^15) (odin_device.h: odin_act_needs_scale) -- a wave-uniform flag, one scalar branch around its
  // splits; the two scales are taken back one after the other (their sum may leave one factor's exponent range).
  // (All set by finish_words(), in front of the first split.)
  int gk = 0, ak = 0;
  bool as = false;
  float g_s = 1.f, g_s2k = ODIN_LO_SCALE, a_s = 1.f, a_s2k = ODIN_LO_SCALE;
  auto finish_words = [&]() {
    gk = odin_range_shift(odin_range_finish(g_rq));
    g_s = odin_pow2(gk); g_s2k = odin_pow2(gk + 11);
    const unsigned a_mb = p.a_amax != nullptr ? odin_range_finish(a_rq) : 0u;
    as = odin_act_needs_scale(a_mb);
    ak = as ? odin_range_shift(a_mb) : 0;
    a_s = odin_pow2(ak); a_s2k = odin_pow2(ak + 11);
  };
  auto store_item = [&](const WpItem& it, int plane_bytes, auto is_grad) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;  // wave-uniform: a scalar branch
#endif
    u32x2 h, l;
    if (decltype(is_grad)::value) odin_split_h4<true>(it.v, g_s, g_s2k, h, l);
    else if (as) odin_split_h4<true>(it.v, a_s, a_s2k, h, l);
    else odin_split_h4<false>(it.v, 1.f, ODIN_LO_SCALE, h, l);
    char* d = smem + it.dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + plane_bytes) = l;
  };
  // item k of a fill: 0 .. WP_MAXU - 1 fine-row items, WP_MAXU: the coarse-row item (+ its bias sums)
  auto store_fill_item = [&](const WpItem (&iu)[WP_MAXU], const WpItem& iv, int k) {
    if (k < WP_MAXU) {
      store_item(iu[k], PBU, std::integral_constant<bool, GU>{});
    } else {
      store_item(iv, PBV, std::integral_constant<bool, !GU>{});
      bsum4.x += iv.v.x; bsum4.y += iv.v.y; bsum4.z += iv.v.z; bsum4.w += iv.v.w;  // (absent items hold zeros)
    }
  };
  auto store_fill = [&](const WpItem (&iu)[WP_MAXU], const WpItem& iv) {
#pragma unroll
    for (int k = 0; k <= WP_MAXU; ++k) store_fill_item(iu, iv, k);
  };

  // ---- this wave's taps and this lane's part of a transposed read ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1);  // taps (kh, kw0) and (kh, kw0 + 1)
  // pixel k of a 16-pixel chunk supplied by this lane: block blk (0, 1) of its half, row q = l16 >> 2
  int kpix[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) kpix[blk] = 8 * half + 4 * blk + (l16 >> 2);
  const int colb = (16 * ((lane >> 4) & 1) + 4 * (l16 & 3)) * 2;  // byte offset of this lane's 4 columns
  // (coarse row inside the chunk, coarse column) of that pixel
  int krow[2], kcol[2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    krow[blk] = (W >= 16) ? 0 : (kpix[blk] >> 3);
    kcol[blk] = (W >= 16) ? kpix[blk] : (kpix[blk] & 7);
  }

  f32x16 acc[2] = {f32x16_zero(), f32x16_zero()};  // main sums (h x h)
  f32x16 acx[2] = {f32x16_zero(), f32x16_zero()};  // cross sums (h x l + l x h), times 2^11
  // ---- prologue: rows of the first tile into LDS; ONE barrier publishes them with the pads and the tables; then
  // the second and third tile's rows into registers ----
  FillEnt en;
  finish_words();
  store_fill(iuA, ivA);
  __syncthreads();
  FillEnt en1, en2;
  fill_entries(en1, 1);  // (all table reads of the prologue in one batch: one LDS round trip, not four)
  fill_entries(en2, 2);
  fill_entries(en, 3);
  WpEnt thN = tt[0];  // ring slots of the next tile's first fine / coarse row
  fill_loads(iuA, ivA, en1);
  fill_loads(iuB, ivB, en2);
  int su0 = 0, sv0 = 0;

  // fragments of one 16-pixel chunk: V (3 planes) and U for the wave's two taps
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
      // the block's first row is this lane's pixel minus its row index q: pass the block base
      const int q = l16 >> 2;
      const char* vb = vring + sv * RBV + (j0 + kcol[blk] - q) * 64 + colb - (l16 & 3) * 8;
      const char* ub = uring + su * RBU + (j0 + kcol[blk] - q) * 64 + colb - (l16 & 3) * 8;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x2 tv = wp_tr_read(vb + pl * PBV, 64, l16);
        F.fv[pl][2 * blk] = tv.x; F.fv[pl][2 * blk + 1] = tv.y;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kw = kw0 + t;  // padded column 2 j + kw: parity kw & 1, slot j + (kw >> 1)
          const u32x2 tu = wp_tr_read(ub + pl * PBU + (kw & 1) * PARB + (kw >> 1) * 64, 64, l16);
          F.fu[t][pl][2 * blk] = tu.x; F.fu[t][pl][2 * blk + 1] = tu.y;
        }
      }
    }
  };
  // the 6 MFMAs of a chunk (plane products h*l, l*h into the cross accumulator, h*h into the main one; the two
  // taps alternate so that consecutive MFMAs are independent); behind every other MFMA one item of the next tile's
  // rows is split and stored (f16 MFMAs run beside plain VALU work)
  auto mfma_chunk = [&](const Frags& F, int item0, const WpItem (&stu)[WP_MAXU], const WpItem& stv) {
#pragma unroll
    for (int m = 0; m < 6; ++m) {
      const int t = m & 1, pp = m >> 1;
      if (!(DBG & 1)) {
        if (pp == 0) acx[t] = mfma32_f16(F.fu[t][0], F.fv[1], acx[t]);
        if (pp == 1) acx[t] = mfma32_f16(F.fu[t][1], F.fv[0], acx[t]);
        if (pp == 2) acc[t] = mfma32_f16(F.fu[t][0], F.fv[0], acc[t]);
      } else {
        acc[t][m] += odin_bitsf(F.fu[t][pp & 1][0] ^ F.fv[pp & 1][1]);
      }
      if ((m & 1) == 1) {
        const int k = item0 + (m >> 1);
        if (k <= WP_MAXU && !(DBG & 4) && !(DBG & 16)) store_fill_item(stu, stv, k);
        if (k <= WP_MAXU && (DBG & 16)) {  // keep the loads alive without the split / LDS stores
          const float4 t = k < WP_MAXU ? stu[k].v : stv.v;
          bsum4.x += t.x; bsum4.y += t.y; bsum4.z += t.z; bsum4.w += t.w;
        }
      }
      ODIN_SCHED_FENCE();
    }
  };

  // one tile: `ld` receives the loads of tile T + 3's rows, the rows of tile T + 1 (loaded two tiles ago into
  // `st`) are split and stored behind the MFMAs.  The two register sets swap roles every tile (no
  // register copies: a copy of a freshly loaded register would wait for the load, vmcnt(0), every tile).
  Frags F0, F1;
  auto run_tile = [&](int T, WpItem (&ldu)[WP_MAXU], WpItem& ldv, const WpItem (&stu)[WP_MAXU], const WpItem& stv) {
    su0 = thN.x;
    sv0 = thN.y;
    if (!(DBG & 2) || T == T0) read_chunk(0, F0);   // first thing behind the barrier
    ODIN_SCHED_FENCE();
    fill_loads(ldu, ldv, en);  // fill T - T0 + 3: its table entries were read a tile ago
    if (!(DBG & 2) || T == T0) read_chunk(1, F1);
    ODIN_SCHED_FENCE();
    mfma_chunk(F0, 0, stu, stv);   // items 0, 1, 2
    fill_entries(en, T - T0 + 4);
    thN = tt[T - T0 + 1];
    mfma_chunk(F1, 3, stu, stv);   // items 3, 4
    __syncthreads();  // every wave is past tile T's rows; tile T + 1's rows are stored
  };
  // (loads run TWO tiles ahead of their stores: one tile, ~2 us, did not cover the HBM latency under
  // load -- 16 of 64 us went to waiting for them)
#pragma unroll 1
  for (int T = T0; T < T1; T += 3) {
    run_tile(T, iuC, ivC, iuA, ivA);
    if (T + 1 < T1) run_tile(T + 1, iuA, ivA, iuB, ivB);
    if (T + 2 < T1) run_tile(T + 2, iuB, ivB, iuC, ivC);
  }

  // ---- this workgroup's slab row: dW[tap][cu0 + cu][cv0 + cv], lane = column cv = l31 ----
  float* row = p.slab + (size_t)bx * p.slab_stride;
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11), a_o = odin_pow2(-ak);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = kh * 4 + kw0 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = fmaf(acx[t][r], o_sx, acc[t][r] * o_s);
      row[((size_t)tap * p.CUt + cu0 + cu) * p.CVt + cv0 + l31] = v * a_o;   // (a_o = 1 for an unscaled activation)
    }
  }
  if (p.want_bias && by == 0) {
    // column sums of V: lanes with the same channel quad (lane & 7), then the 8 waves through LDS
    float s[4] = {bsum4.x, bsum4.y, bsum4.z, bsum4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int m = 8; m <= 32; m <<= 1) s[k] += __shfl_xor(s[k], m);
    }
    if (lane < 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bred[wave * 32 + 4 * lane + k] = s[k];
    }
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int wv = 0; wv < 8; ++wv) t += bred[wv * 32 + tid];
      row[(size_t)16 * p.CUt * p.CVt + cv0 + tid] = t;
    }
  }
}

template <int W, bool GU, int DBG = 0>
__global__ __launch_bounds__(512) void wgrad_planes_kernel(WPParams p) {
  wp_body<W, GU, DBG>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Several layers' weight gradients in ONE launch.  They depend on nothing but their own layer's tensors, so the
// backward pass defers them (odin_wgrad_planes_defer_begin / _end) and issues them together at its end: a launch costs
// ~4.6 us of floor plus a tail in which most CUs wait for the last workgroup, and a step has five of them (dSprites:
// 37.8 + 23.8 + 16.8 + 13.0 + 17.1 us).  Jobs are laid out largest first; a workgroup finds its job by its linear index.
constexpr int WP_MAXJOBS = 8;
struct WPMulti {
  WPParams p[WP_MAXJOBS];
  int start[WP_MAXJOBS + 1];   // first linear workgroup index of job j
  int gx[WP_MAXJOBS], gy[WP_MAXJOBS];
  int kind[WP_MAXJOBS];        // (W == 8 ? 0 : W == 16 ? 1 : 2) * 2 + GU
  int n;
};
__global__ __launch_bounds__(512) void wgrad_planes_multi_kernel(WPMulti m) {
  int j = 0;
  while (j + 1 < m.n && (int)blockIdx.x >= m.start[j + 1]) ++j;
  const int l = (int)blockIdx.x - m.start[j];
  const int bx = l % m.gx[j], r = l / m.gx[j];
  const int by = r % m.gy[j], bz = r / m.gy[j];
  switch (m.kind[j]) {
    case 0: wp_body<8, false>(m.p[j], bx, by, bz); break;
    case 1: wp_body<8, true>(m.p[j], bx, by, bz); break;
    case 2: wp_body<16, false>(m.p[j], bx, by, bz); break;
    case 3: wp_body<16, true>(m.p[j], bx, by, bz); break;
    case 4: wp_body<32, false>(m.p[j], bx, by, bz); break;
    default: wp_body<32, true>(m.p[j], bx, by, bz); break;
  }
}

// LDS: fine-row ring + coarse-row ring + the fill tables ((1 + fine rows per fill + TC) x 8 bytes per fill, tiles + 4 fills)
constexpr int WP_LDS_MAX = 152 * 1024;   // (dynamic; + 1 KB of static scratch per body: 6 in the multi-layer kernel)
int wp_ring_bytes(int W) {
  const int TC = 32 / W;
  return (4 * TC + 3) * 2 * 2 * (W + 1) * 64 + (2 * TC) * 2 * W * 64;
}
int wp_fill_bytes(int W) {
  const int ipu = 2 * W / 8, rj = 8 / ipu > 0 ? 8 / ipu : 1;
  return 8 * (1 + WP_MAXU * rj + 32 / W);
}
// tiles per workgroup: the chip filled once when the tables fit, more workgroups otherwise; -1: does not fit
int wp_tiles_per_wg(int W, int n_tiles, int gyz) {
  int cap = odin_num_cus() / gyz;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_SLAB_BLOCKS) cap = ODIN_MAX_SLAB_BLOCKS;
  int tpw = (n_tiles + cap - 1) / cap;
  const int limit = (WP_LDS_MAX - wp_ring_bytes(W)) / wp_fill_bytes(W) - 4;
  if (tpw > limit) tpw = limit;
  if ((n_tiles + tpw - 1) / tpw > ODIN_MAX_SLAB_BLOCKS) return -1;
  return tpw;
}

template <int W, bool GU>
int wp_launch(const WPParams& p, dim3 grid, void* stream, bool may_defer) {
  const size_t lds = (size_t)wp_ring_bytes(W) + (size_t)(p.tiles_per_wg + 4) * wp_fill_bytes(W);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#ifdef ODIN_DIAG
  // diagnostics build only (make diag): instances with parts of the kernel switched off, selected by
  // ODIN_WP_DBG -- they compute WRONG results and are not in the product library
  if (W == 32 && GU) {
    static const int dbg = [] { const char* e = ODIN_DIAG_ENV("ODIN_WP_DBG"); return e ? atoi(e) : 0; }();
    static bool dattr = false;
    constexpr int D1 = (W == 32 && GU) ? 1 : 0, D2 = (W == 32 && GU) ? 2 : 0, D4 = (W == 32 && GU) ? 4 : 0,
                  D8 = (W == 32 && GU) ? 8 : 0, D16 = (W == 32 && GU) ? 16 : 0;
    if (!dattr) {
      const void* fns[5] = {reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU, D1>),
                            reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU, D2>),
                            reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU, D4>),
                            reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU, D8>),
                            reinterpret_cast<const void*>(&wgrad_planes_kernel<W, GU, D16>)};
      for (const void* f : fns)
        if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS_MAX) != hipSuccess)
          (void)hipGetLastError();
      dattr = true;
    }
    if (dbg == 1) { ODIN_LAUNCH((wgrad_planes_kernel<W, GU, D1>), grid, dim3(512), lds, stream, p); return odin_check_launch("wgrad_planes(f16x2)"); }
    if (dbg == 2) { ODIN_LAUNCH((wgrad_planes_kernel<W, GU, D2>), grid, dim3(512), lds, stream, p); return odin_check_launch("wgrad_planes(f16x2)"); }
    if (dbg == 4) { ODIN_LAUNCH((wgrad_planes_kernel<W, GU, D4>), grid, dim3(512), lds, stream, p); return odin_check_launch("wgrad_planes(f16x2)"); }
    if (dbg == 8) { ODIN_LAUNCH((wgrad_planes_kernel<W, GU, D8>), grid, dim3(512), lds, stream, p); return odin_check_launch("wgrad_planes(f16x2)"); }
    if (dbg == 16) { ODIN_LAUNCH((wgrad_planes_kernel<W, GU, D16>), grid, dim3(512), lds, stream, p); return odin_check_launch("wgrad_planes(f16x2)"); }
  }
#endif
#endif
  // (a job whose range word is a library scratch word -- one of a ring of 16 -- is launched at once: the word would
  // not outlive the deferral)
  if (may_defer && g_wpq.defer && g_wpq.n < WP_MAXJOBS && (g_wpq.n == 0 || g_wpq.stream == stream)) {
    WPQueue::Job& jb = g_wpq.job[g_wpq.n++];
    jb.p = p; jb.grid = grid; jb.lds = lds; jb.kind = (W == 8 ? 0 : W == 16 ? 1 : 2) * 2 + (GU ? 1 : 0);
    g_wpq.stream = stream;
    return odin_check_launch("wgrad_planes(f16x2)");  // (names the family; nothing was launched yet)
  }
  ODIN_LAUNCH((wgrad_planes_kernel<W, GU>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("wgrad_planes(f16x2)");
}

int wp_launch_one(const WPQueue::Job& jb, void* stream) {
  const WPParams& p = jb.p;
  switch (jb.kind) {
    case 0: ODIN_LAUNCH((wgrad_planes_kernel<8, false>), jb.grid, dim3(512), jb.lds, stream, p); break;
    case 1: ODIN_LAUNCH((wgrad_planes_kernel<8, true>), jb.grid, dim3(512), jb.lds, stream, p); break;
    case 2: ODIN_LAUNCH((wgrad_planes_kernel<16, false>), jb.grid, dim3(512), jb.lds, stream, p); break;
    case 3: ODIN_LAUNCH((wgrad_planes_kernel<16, true>), jb.grid, dim3(512), jb.lds, stream, p); break;
    case 4: ODIN_LAUNCH((wgrad_planes_kernel<32, false>), jb.grid, dim3(512), jb.lds, stream, p); break;
    default: ODIN_LAUNCH((wgrad_planes_kernel<32, true>), jb.grid, dim3(512), jb.lds, stream, p); break;
  }
  return odin_check_launch("wgrad_planes(f16x2)");
}

}  // namespace

// ---- deferred weight gradients: between _begin and _end (or the next odin_slab_reduce, which flushes) the plane
// weight-gradient launches of the calling thread are collected and issued as one multi-layer launch ----
extern "C" void odin_wgrad_planes_defer_begin(void) {
  g_wpq.defer = true;
  g_wpq.n = 0;
}

extern "C" int odin_wgrad_planes_defer_end(void* stream) {
  g_wpq.defer = false;
  const int n = g_wpq.n;
  g_wpq.n = 0;
  if (n == 0) return 0;
  void* st = g_wpq.stream;
  (void)stream;
  if (n == 1) return wp_launch_one(g_wpq.job[0], st);
  // largest first: the tail of the launch is then made of the small layers' short workgroups
  int order[WP_MAXJOBS];
  for (int i = 0; i < n; ++i) order[i] = i;
  auto work = [&](int i) {
    const WPQueue::Job& jb = g_wpq.job[i];
    return (long)jb.p.tiles_per_wg * (jb.kind >= 4 ? 4 : jb.kind >= 2 ? 2 : 1);
  };
  for (int a = 0; a < n; ++a)
    for (int b = a + 1; b < n; ++b)
      if (work(order[b]) > work(order[a])) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
  WPMulti m;
  memset(&m, 0, sizeof(m));
  m.n = n;
  size_t lds = 0;
  int total = 0;
  for (int k = 0; k < n; ++k) {
    const WPQueue::Job& jb = g_wpq.job[order[k]];
    m.p[k] = jb.p; m.kind[k] = jb.kind; m.gx[k] = (int)jb.grid.x; m.gy[k] = (int)jb.grid.y;
    m.start[k] = total;
    total += (int)(jb.grid.x * jb.grid.y * jb.grid.z);
    if (jb.lds > lds) lds = jb.lds;
  }
  m.start[n] = total;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_planes_multi_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, WP_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH(wgrad_planes_multi_kernel, dim3(total), dim3(512), lds, st, m);
  return odin_check_launch("wgrad_planes_multi(f16x2)");
}

// (odin_slab_reduce reads the slabs: anything still deferred on this thread is issued first)
int odin_wgrad_planes_flush(void* stream) {
  if (g_wpq.n == 0) return 0;
  const bool was = g_wpq.defer;
  const int rc = odin_wgrad_planes_defer_end(stream);
  g_wpq.defer = was;
  return rc;
}

// fine tensor U [B, H, W, CI], coarse tensor V [B, OH, OW, CO] (the argument order of wgrad.hip's WParams)
bool odin_wgrad_planes_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                                  int S, int pt, int pl, int center) {
  if (odin_blk_first()) return false;   // (diagnostics: odin_debug_blk_first)
  // (read per call: the A/B tests switch paths inside one process; a captured graph never comes here)
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NOPLANES") || ODIN_DIAG_ENV("ODIN_SPLIT") || ODIN_DIAG_ENV("ODIN_NOWPLANES")) return false;
  return KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && !center && (CI % 32) == 0 && (CO % 32) == 0 &&
         H == 2 * OH && W == 2 * OW && (OW == 8 || OW == 16 || OW == 32) && (OH % (32 / OW)) == 0 &&
         (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * OH * OW * CO * 4 < 0x7FFF0000ull &&
         wp_tiles_per_wg(OW, B * (OH / (32 / OW)), (CI / 32) * (CO / 32)) > 0;
}

// grad_u: the gradient operand is U (Conv2DTranspose) rather than V (Conv2D)
int odin_wgrad_planes_launch(const float* U, const float* V, float* slab, int* rows_out, int B, int OH,
                             int OW, int CI, int CO, int want_bias, int grad_u, const uint32_t* g_amax,
                             const uint32_t* a_amax, void* stream) {
  WPParams p;
  memset(&p, 0, sizeof(p));
  p.U = U; p.V = V; p.slab = slab; p.a_amax = a_amax;
  p.B = B; p.h = OH; p.CUt = CI; p.CVt = CO; p.want_bias = want_bias;
  p.slab_stride = 16 * CI * CO + (want_bias ? CO : 0);
  const int TC = 32 / OW;
  p.tiles_per_img = OH / TC;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CI / 32, gz = CO / 32;
  p.tiles_per_wg = wp_tiles_per_wg(OW, p.n_tiles, gy * gz);
  if (p.tiles_per_wg <= 0) return odin_fail(-2, "wgrad_planes: too many tiles for the fill tables");
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (slab == nullptr) return 0;  // dry run
  p.g_amax = grad_u ? odin_range_word_of(U, (size_t)B * 2 * OH * 2 * OW * CI, g_amax, stream)
                    : odin_range_word_of(V, (size_t)B * OH * OW * CO, g_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "wgrad_planes: no range word for the gradient operand");
  dim3 grid(gx, gy, gz);
  const bool md = g_amax != nullptr;
  if (grad_u) {
    if (OW == 32) return wp_launch<32, true>(p, grid, stream, md);
    if (OW == 16) return wp_launch<16, true>(p, grid, stream, md);
    return wp_launch<8, true>(p, grid, stream, md);
  }
  if (OW == 32) return wp_launch<32, false>(p, grid, stream, md);
  if (OW == 16) return wp_launch<16, false>(p, grid, stream, md);
  return wp_launch<8, false>(p, grid, stream, md);
}
