// blk_planes.hip -- the 4x4 / stride-2 layers of the conv stacks (image_networks.py:460-513) for ANY image size, as two-
// plane f16 products (odin_device.h: x = h + 2^-11 l) over BLOCK WINDOWS (round 6).
//
// The row-window kernels (tconv_planes / fconv_planes / wgrad_planes / bwd_planes) stage whole image rows of 8, 16 or 32
// pixels; the audio VAE's maps (examples/vae/vae_audio.py:84-110 on the same stack: 96 x 80 -> 48 x 40 -> 24 x 20 ->
// 12 x 10) fit none of them and ran on igemm_h.hip, which gathers and splits every operand fragment per use (~100 TFLOP/s
// in fp32 FLOPs).  Here a tile is a block of 8 x 8 COARSE pixels (the low-resolution side of the layer) wherever it lies
// in the image: its window -- 10 x 10 coarse pixels, or 18 x 18 fine ones -- is fetched with zero fill outside the image,
// split ONCE into planes on its way into LDS (two buffers: the next tile's window is in flight while this one
// multiplies), and tiles that overhang the image's right / lower edge mask their stores.  No row ring, no seams, no
// tables; a window re-reads its halo from L2 (1.27x of the fine tensor, 1.56x of the coarse one).
//
// The matrix instruction is v_mfma_f32_16x16x32_f16 (k = the 32 channels of ONE tap): eight waves own eight disjoint
// 16 x 16 output blocks, so no partial tiles cross waves, and the WEIGHTS of a wave's blocks live in its registers for
// the whole launch (tconv: 4 taps x 32 k x 16 n; fconv: 16 taps) -- LDS holds pixel windows only.
//
//   tconv_blk  coarse -> fine: Conv2DTranspose forward (bias + activation), Conv2D data gradient (x act'(aux), column
//              sums).  out[b, oh, ow, n] = sum over taps with (oh + 1 - kh), (ow + 1 - kw) even, c of
//              in[b, (oh + 1 - kh) / 2, (ow + 1 - kw) / 2, c] * W[kh, kw, n, c]
//   fconv_blk  fine -> coarse: Conv2D forward, Conv2DTranspose data gradient.
//              out[b, i, j, n] = sum over (kh, kw, c) of in[b, 2 i - 1 + kh, 2 j - 1 + kw, c] * W[kh, kw, c, n]
//   wgrad_blk  fine (x) coarse: the weight gradient of either layer.
//              dW[kh, kw, cu, cv] = sum over (b, i, j) of U[b, 2 i - 1 + kh, 2 j - 1 + kw, cu] * V[b, i, j, cv]
#include "odin_device.h"
#include "odin_internal.h"
#include "blk_common.h"
#include <cstdlib>

namespace {

// =====================================================================================================================
// tconv_blk
// =====================================================================================================================
struct TBParams {
  const float* in;     // [B, H, W, CS]
  const float* w;      // [16 taps][CO][CS]
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, 2H, 2W, CO], out *= act'(aux)
  float* out;          // [B, 2H, 2W, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] partial column sums of out (may be null)
  int B, H, W, CS, CO;
  int act;             // EPI 1: the layer's activation; EPI 2: the activation whose derivative (from aux) multiplies
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* in_amax;   // range word of `in` (optional for an activation)
  unsigned* out_amax;        // range word of `out` (may be null)
  int in_is_grad;
};

// coarse window: [k-pass][plane][row 10][slot 16][32 f16]; the 16-byte k-pieces of a pixel XOR-swizzled so that the 16
// lanes of a ds_read_b128 group (two window rows x 8 columns, one k-piece) hit 16 distinct slots of the bank row
constexpr int TB_PLB = 10 * 16 * 64;     // one plane
constexpr int TB_KPB = 2 * TB_PLB;       // one 32-channel pass

template <int EPI, int NK, int ACT>
__global__ __launch_bounds__(512) void tconv_blk_kernel(TBParams p) {
  constexpr int BUFB = NK * TB_KPB;
  constexpr int NIT = (800 * NK + 511) / 512;   // float4 items per thread and window
  ODIN_DYN_SMEM(char, smem);
  __shared__ float cred[8 * 16 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int cls = wave & 3, nb = wave >> 2;
  const int cpw = cls & 1, rpar = cls >> 1;
  const int n0 = blockIdx.y * 32 + 16 * nb;
  const int OH = 2 * p.H, OW = 2 * p.W;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq in_rq = odin_range_issue(p.in_amax, lane);
  // ---- this wave's weights: 4 taps x NK passes, lane = (output channel l15, k-piece lq) ----
  const int kh_a = rpar ? 0 : 1, kw_a = cpw ? 0 : 1;
  float4 wv[4][NK][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tap = (kh_a + 2 * (t >> 1)) * 4 + kw_a + 2 * (t & 1);
#pragma unroll
    for (int kp = 0; kp < NK; ++kp) {
      const float* src = p.w + ((size_t)(tap * p.CO + n0 + l15) * p.CS + 32 * kp + 8 * lq);
      wv[t][kp][0] = *reinterpret_cast<const float4*>(src);
      wv[t][kp][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  }
  // ---- window items of this thread (constant over the tiles) ----
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * p.CS * 4));
  int it_dst[NIT], it_g[NIT], it_wr[NIT], it_wc[NIT];
#pragma unroll
  for (int j = 0; j < NIT; ++j) {
    const int e = tid + 512 * j;
    const int kp = e >= 800 ? 1 : 0, e2 = e - 800 * kp;
    const int px = e2 >> 3, ch4 = e2 & 7;
    const int wr = odin_div_small(px, 10), wc = px - 10 * wr;
    it_wr[j] = (e < 800 * NK) ? wr : (1 << 20);   // (no item: never inside an image)
    it_wc[j] = wc;
    it_dst[j] = kp * TB_KPB + wr * 1024 + wc * 64 + (((ch4 >> 1) ^ tb_swz(wr, wc)) << 4) + (ch4 & 1) * 8;
    it_g[j] = ((wr * p.W + wc) * p.CS + 32 * kp + 4 * ch4) * 4;
  }
  float4 itv[NIT];
  auto decode = [&](int T, int& b, int& ty, int& tx) {
    const int per = p.nty * p.ntx;
    b = odin_div_small(T, per);
    const int r = T - b * per;
    ty = odin_div_small(r, p.ntx);
    tx = r - ty * p.ntx;
  };
  auto issue = [&](int b, int ty, int tx) {
    const int base = ((b * p.H + 8 * ty - 1) * p.W + 8 * tx - 1) * p.CS * 4;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int gr = 8 * ty - 1 + it_wr[j], gc = 8 * tx - 1 + it_wc[j];
      const bool ok = gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
      itv[j] = odin_run_load4(IN, ok ? (unsigned)(base + it_g[j]) : ODIN_OOB);
    }
  };
  float in_s = 1.f, in_s2k = ODIN_LO_SCALE, out_s = 1.f, out_sx = ODIN_LO_UNSCALE;
  auto stage = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (NIT * 512 == 800 * NK || tid + 512 * j < 800 * NK) {
        u32x2 h, l;
        odin_split_h4<true>(itv[j], in_s, in_s2k, h, l);
        *reinterpret_cast<u32x2*>(buf + it_dst[j]) = h;
        *reinterpret_cast<u32x2*>(buf + it_dst[j] + TB_PLB) = l;
      }
    }
  };
  int b_c, ty_c, tx_c;
  decode(T0, b_c, ty_c, tx_c);
  if (T0 < T1) issue(b_c, ty_c, tx_c);
  {
    const unsigned mb = odin_range_finish(in_rq);
    const int gk = bk_shift(mb, p.in_is_grad);
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11);
    out_s = odin_pow2(-gk); out_sx = odin_pow2(-gk - 11);
  }
  u32x4 wh[4][NK], wl[4][NK];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int kp = 0; kp < NK; ++kp) bk_split8(wv[t][kp][0], wv[t][kp][1], 1.f, ODIN_LO_SCALE, wh[t][kp], wl[t][kp]);
  if (T0 < T1) stage(smem);
  __syncthreads();

  // ---- per-lane constants of the MFMA operand reads and of the epilogue ----
  const int ri0 = l15 >> 3, cj = l15 & 7;   // pixel block pb: tile rows 2 pb + ri0
  int boff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = ri0 + (rpar ? 2 : 1) - (t >> 1), col = cj + (cpw ? 2 : 1) - (t & 1);
    boff[t] = row * 1024 + col * 64 + ((lq ^ tb_swz(row, col)) << 4);
  }
  const unsigned out_bytes = (unsigned)((size_t)p.B * OH * OW * p.CO * 4);
  const OdinRun OUT = odin_run(p.out, out_bytes);
  const OdinRun AUX = odin_run(EPI == 2 ? p.aux : nullptr, EPI == 2 ? out_bytes : 0u);
  const unsigned out_lane = (unsigned)((((2 * ri0 + rpar) * OW + 2 * cj + cpw) * p.CO + n0 + 4 * lq) * 4);
  const unsigned pb_step = (unsigned)(4 * OW * p.CO * 4);   // two tile rows = four fine rows
  float bias_r[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_r[r] = p.bias[n0 + 4 * lq + r];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * BUFB;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * BUFB;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      decode(T + 1, b_n, ty_n, tx_n);
      issue(b_n, ty_n, tx_n);
    }
    const unsigned tile_out = (unsigned)(((b_c * OH + 16 * ty_c) * OW + 16 * tx_c) * p.CO * 4);
    unsigned voff[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      const bool ok = 8 * ty_c + 2 * pb + ri0 < p.H && 8 * tx_c + cj < p.W;
      voff[pb] = ok ? out_lane + pb * pb_step : ODIN_OOB_V;
    }
    float4 ax[4];
    if (EPI == 2) {
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) ax[pb] = odin_run_load4s(AUX, voff[pb], tile_out);
    }
    f32x4 acc[4], acx[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) { acc[pb] = bk_zero4(); acx[pb] = bk_zero4(); }
#pragma unroll
    for (int kp = 0; kp < NK; ++kp)
#pragma unroll
      for (int t = 0; t < 4; ++t)
      {
        // (the three plane products of a tap as three sweeps over the four pixel blocks: two MFMAs into the SAME accumulator
        // are four instructions apart -- back to back, v_mfma_f32_16x16x32_f16's second read of an accumulator it wrote one
        // instruction earlier came out wrong in a few lanes, intermittently: tools/race_layer.py)
        u32x4 xh[4], xl[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const char* a = buf + kp * TB_KPB + boff[t] + pb * 2048;
          xh[pb] = *reinterpret_cast<const u32x4*>(a);
          xl[pb] = *reinterpret_cast<const u32x4*>(a + TB_PLB);
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acx[pb] = mfma16_f16(wh[t][kp], xl[pb], acx[pb]);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acc[pb] = mfma16_f16(wh[t][kp], xh[pb], acc[pb]);
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) acx[pb] = mfma16_f16(wl[t][kp], xh[pb], acx[pb]);
      }
    if (T + 1 < T1) stage(nbuf);
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf(acx[pb][r], out_sx, acc[pb][r] * out_s);
      if (EPI == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = bk_act<ACT>(p.act, v[r] + bias_r[r]);
      } else {
        const bool ok = voff[pb] != ODIN_OOB_V;
        const float a4[4] = {ax[pb].x, ax[pb].y, ax[pb].z, ax[pb].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = ok ? v[r] * bk_act_grad<ACT>(p.act, a4[r]) : 0.f;
          csum[r] += v[r];
        }
      }
      if (EPI == 2 || voff[pb] != ODIN_OOB_V) amx = odin_amax3(odin_amax3(amx, v[0], v[1]), v[2], v[3]);
      odin_run_store4s(OUT, voff[pb], tile_out, make_float4(v[0], v[1], v[2], v[3]));
    }
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
  }

  odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred + 128, blockIdx.x + gridDim.x * blockIdx.y);
  if (EPI == 2 && p.colsum != nullptr) {
    // column sums: the 16 pixel lanes of a k-piece group, then the 4 parity classes of an n-block through LDS
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vv = csum[r];
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
      if (l15 == 0) cred[wave * 16 + 4 * lq + r] = vv;
    }
    __syncthreads();
    if (tid < 32) {
      const int nbk = tid >> 4, ch = tid & 15;
      float tt = 0.f;
      for (int c = 0; c < 4; ++c) tt += cred[(4 * nbk + c) * 16 + ch];
      p.colsum[(size_t)blockIdx.x * p.CO + blockIdx.y * 32 + tid] = tt;
    }
  }
}

constexpr int TB_LDS_MAX = 2 * 2 * TB_KPB;

int tb_tiles_per_wg(int n_tiles, int gy) {
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  return (n_tiles + cap - 1) / cap;
}

template <int EPI, int NK, int ACT>
int tb_launch_a(const TBParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)2 * NK * TB_KPB;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_blk_kernel<EPI, NK, ACT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_MAX) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((tconv_blk_kernel<EPI, NK, ACT>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("tconv_blk(f16x2)");
}
// ELU (every conv layer of the reference's stacks) as a compile-time constant, anything else through the run-time switch
template <int EPI, int NK>
int tb_launch(const TBParams& p, dim3 grid, void* stream) {
  return p.act == ODIN_ACT_ELU ? tb_launch_a<EPI, NK, ODIN_ACT_ELU>(p, grid, stream) : tb_launch_a<EPI, NK, -1>(p, grid, stream);
}

}  // namespace

// the block-window kernels take a layer from this many FLOP per launch (as igemm_h.hip, whose place they take where the
// geometry fits); the tests send their small shapes here with a threshold of 0
static bool g_blk_off = false;
static double g_blk_min_flop = 1.5e9;
extern "C" int odin_debug_blk_planes(int enable) {
  const int old = g_blk_off ? 0 : 1;
  if (enable >= 0) g_blk_off = enable == 0;
  return old;
}
// diagnostics: 1 = the row-window plane kernels step aside wherever a block-window kernel applies (A/B of the two
// families on the 8 / 16 / 32-pixel rows; tools/blkbench.py)
static bool g_blk_first = false;
extern "C" int odin_debug_blk_first(int on) {
  const int old = g_blk_first ? 1 : 0;
  if (on >= 0) g_blk_first = on != 0;
  return old;
}
bool odin_blk_first() { return g_blk_first && !g_blk_off; }
extern "C" double odin_debug_blk_min_flop(double flop) {
  const double old = g_blk_min_flop;
  if (flop >= 0.0) g_blk_min_flop = flop;
  return old;
}
// rows whose width is not a power of two have no tuned small-layer path (the implicit GEMMs of igemm.hip were laid out
// on 4 / 8-pixel rows): those layers come here from a third of the size -- the audio encoder3 (12 x 10 x 64 -> 6 x 5 x 64,
// 1.0 GFLOP): weight + data gradient 61 -> 27 us
static double blk_scale_for_width(int W) { return (W & (W - 1)) == 0 ? 1.0 : 3.0; }
bool odin_blk_enabled(double flop) {
  return !(g_blk_off || odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NOBLK")) && flop >= g_blk_min_flop;
}

// Conv2DTranspose(k4, s2, SAME) forward from CI in {32, 64} channels / Conv2D(k4, s2) data gradient, any H x W
bool odin_tconv_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center) {
  if (!odin_blk_enabled(2.0 * B * H * W * 16.0 * CI * CO * blk_scale_for_width(W))) return false;
  if (!(KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && !center && (CI == 32 || CI == 64) && (CO % 32) == 0))
    return false;
  if (H < 1 || W < 1 || H > 4096 || W > 4096) return false;
  return (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * H * W * 4 * CO * 4 < 0x7FFF0000ull;
}

int odin_tconv_blk_rows(int B, int H, int W, int CO) {
  const int n_tiles = B * ((H + 7) / 8) * ((W + 7) / 8);
  const int tpw = tb_tiles_per_wg(n_tiles, CO / 32);
  return (n_tiles + tpw - 1) / tpw;
}

// epi 1: forward (bias + act); epi 2: data gradient (x act'(aux), column sums into colsum[rows][CO])
int odin_tconv_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int H, int W, int CI, int CO, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream) {
  TBParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.H = H; p.W = W; p.CS = CI; p.CO = CO; p.act = act;
  p.nty = (H + 7) / 8; p.ntx = (W + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = CO / 32;
  p.tiles_per_wg = tb_tiles_per_wg(p.n_tiles, gy);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  p.in_is_grad = epi == 2;
  if (epi == 2) {
    if (aux == nullptr) { p.act = ODIN_ACT_LINEAR; p.aux = out; }
    p.in_amax = odin_range_word_of(in, (size_t)B * H * W * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "tconv_blk: no range word for the gradient input");
  } else {
    p.in_amax = in_amax;
  }
  p.out_amax = out_amax;
  dim3 grid(gx, gy, 1);
  if (epi == 1) return CI == 32 ? tb_launch<1, 1>(p, grid, stream) : tb_launch<1, 2>(p, grid, stream);
  return CI == 32 ? tb_launch<2, 1>(p, grid, stream) : tb_launch<2, 2>(p, grid, stream);
}

// =====================================================================================================================
// the FINE window shared by fconv_blk and wgrad_blk: 18 x 18 fine pixels around a block of 8 x 8 coarse ones, every row as
// two column-parity planes of 9 pixel slots (a tap reads 8 consecutive slots of one parity), [plane][row][parity][slot] x
// 32 f16; k-pieces swizzled by the slot and by the row pair so that the two rows of a 16-lane read group never collide
// =====================================================================================================================
namespace {

constexpr int FW_ROWB = 2 * 9 * 64;        // one window row of one plane
constexpr int FW_PLB = 18 * FW_ROWB;       // one plane: 20736 bytes
constexpr int FW_BYTES = 2 * FW_PLB;
constexpr int FW_NIT = (18 * 18 * 8 + 511) / 512;   // float4 items per thread and window: 6 (the last one mostly empty)
__host__ __device__ constexpr int fw_swz(int row, int slot) { return ((slot >> 2) + 2 * ((row >> 1) & 1)) & 3; }

struct FwItems {
  int dst[FW_NIT], g[FW_NIT], wr[FW_NIT], wc[FW_NIT];
};

// item j of thread tid: fine window pixel (wr, wc), channels 4 ch4 .. + 3 of the pass's 32
__device__ __forceinline__ void fw_items(FwItems& I, int tid, int FWid, int CS, int c_off) {
#pragma unroll
  for (int j = 0; j < FW_NIT; ++j) {
    const int e = tid + 512 * j;
    const int px = e >> 3, ch4 = e & 7;
    const int wr = odin_div_small(px, 18), wc = px - 18 * wr;
    I.wr[j] = (e < 18 * 18 * 8) ? wr : (1 << 20);
    I.wc[j] = wc;
    I.dst[j] = wr * FW_ROWB + (wc & 1) * 576 + (wc >> 1) * 64 + (((ch4 >> 1) ^ fw_swz(wr, wc >> 1)) << 4) + (ch4 & 1) * 8;
    I.g[j] = ((wr * FWid + wc) * CS + c_off + 4 * ch4) * 4;
  }
}
__device__ __forceinline__ void fw_issue(float4 (&v)[FW_NIT], const FwItems& I, const OdinRun& R, int b, int ty, int tx,
                                         int FH, int FWid, int CS) {
  const int base = ((b * FH + 16 * ty - 1) * FWid + 16 * tx - 1) * CS * 4;
#pragma unroll
  for (int j = 0; j < FW_NIT; ++j) {
    const int gr = 16 * ty - 1 + I.wr[j], gc = 16 * tx - 1 + I.wc[j];
    const bool ok = gr >= 0 && gr < FH && gc >= 0 && gc < FWid;
    v[j] = odin_run_load4(R, ok ? (unsigned)(base + I.g[j]) : ODIN_OOB);
  }
}
__device__ __forceinline__ void fw_stage(char* buf, const float4 (&v)[FW_NIT], const FwItems& I, int tid, float s,
                                         float s2k) {
#pragma unroll
  for (int j = 0; j < FW_NIT; ++j) {
    if (tid + 512 * j < 18 * 18 * 8) {
      u32x2 h, l;
      odin_split_h4<true>(v[j], s, s2k, h, l);
      *reinterpret_cast<u32x2*>(buf + I.dst[j]) = h;
      *reinterpret_cast<u32x2*>(buf + I.dst[j] + FW_PLB) = l;
    }
  }
}


// =====================================================================================================================
// fconv_blk
// =====================================================================================================================
struct FBParams {
  const float* in;     // [B, 2 OH, 2 OW, CS]
  const float* w;      // [16 taps][CS][CO]
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, OH, OW, CO], out *= act'(aux)
  float* out;          // [B, OH, OW, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] (may be null)
  int B, OH, OW, CS, CO;
  int act;
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* in_amax;
  unsigned* out_amax;
  int in_is_grad;
};

template <int EPI, int ACT>
__global__ __launch_bounds__(512) void fconv_blk_kernel(FBParams p) {
  ODIN_DYN_SMEM(char, smem);
  __shared__ float cred[8 * 16 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int pb = wave & 3, nb = wave >> 2;
  const int n0 = blockIdx.y * 32 + 16 * nb;
  const int FH = 2 * p.OH, FWid = 2 * p.OW;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq in_rq = odin_range_issue(p.in_amax, lane);
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * FH * FWid * p.CS * 4));
  FwItems I;
  fw_items(I, tid, FWid, p.CS, 0);
  float4 itv[FW_NIT];
  int b_c, ty_c, tx_c;
  bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
  if (T0 < T1) fw_issue(itv, I, IN, b_c, ty_c, tx_c, FH, FWid, p.CS);
  // ---- this wave's weights: all 16 taps, lane = (output channel l15, k-piece lq): 128 registers for the whole launch ----
  u32x4 wh[16], wl[16];
#pragma unroll
  for (int tap = 0; tap < 16; ++tap) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = p.w[((size_t)(tap * p.CS + 8 * lq + j)) * p.CO + n0 + l15];
    bk_split8(make_float4(e[0], e[1], e[2], e[3]), make_float4(e[4], e[5], e[6], e[7]), 1.f, ODIN_LO_SCALE, wh[tap],
              wl[tap]);
  }
  float in_s, in_s2k, out_s, out_sx;
  {
    const unsigned mb = odin_range_finish(in_rq);
    const int gk = bk_shift(mb, p.in_is_grad);
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11);
    out_s = odin_pow2(-gk); out_sx = odin_pow2(-gk - 11);
  }
  if (T0 < T1) fw_stage(smem, itv, I, tid, in_s, in_s2k);
  __syncthreads();

  // ---- per-lane constants: this lane's coarse pixel (ri, cj) of the tile; operand offsets by (kh >> 1, kw >> 1) ----
  const int ri = 2 * pb + (l15 >> 3), cj = l15 & 7;
  int boff[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int row = 2 * ri + 2 * a, slot = cj + c;   // (kh = 2 a + {0, 1} share (row >> 1) & 1; kw = 2 c + {0, 1} the slot)
      boff[a][c] = 2 * ri * FW_ROWB + slot * 64 + ((lq ^ fw_swz(row, slot)) << 4);
    }
  const unsigned out_bytes = (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4);
  const OdinRun OUT = odin_run(p.out, out_bytes);
  const OdinRun AUX = odin_run(EPI == 2 ? p.aux : nullptr, EPI == 2 ? out_bytes : 0u);
  const unsigned out_lane = (unsigned)(((ri * p.OW + cj) * p.CO + n0 + 4 * lq) * 4);
  float bias_r[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_r[r] = p.bias[n0 + 4 * lq + r];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * FW_BYTES;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * FW_BYTES;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
      fw_issue(itv, I, IN, b_n, ty_n, tx_n, FH, FWid, p.CS);
    }
    const unsigned tile_out = (unsigned)(((b_c * p.OH + 8 * ty_c) * p.OW + 8 * tx_c) * p.CO * 4);
    const bool ok = 8 * ty_c + ri < p.OH && 8 * tx_c + cj < p.OW;
    const unsigned voff = ok ? out_lane : ODIN_OOB_V;
    float4 ax = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 2) ax = odin_run_load4s(AUX, voff, tile_out);
    // SIX accumulators -- (main, high x low, low x high) for the even and for the odd taps -- so that two MFMAs into the
    // same accumulator are six instructions apart (bk_mfma16 note in blk_common.h: closer ones went wrong on the MI355X)
    f32x4 acc[2] = {bk_zero4(), bk_zero4()}, acx[2] = {bk_zero4(), bk_zero4()}, acy[2] = {bk_zero4(), bk_zero4()};
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) {
      const int kh = tap >> 2, kw = tap & 3;
      const char* a = buf + boff[kh >> 1][kw >> 1] + kh * FW_ROWB + (kw & 1) * 576;
      const u32x4 xh = *reinterpret_cast<const u32x4*>(a);
      const u32x4 xl = *reinterpret_cast<const u32x4*>(a + FW_PLB);
      acx[tap & 1] = mfma16_f16(wh[tap], xl, acx[tap & 1]);
      acc[tap & 1] = mfma16_f16(wh[tap], xh, acc[tap & 1]);
      acy[tap & 1] = mfma16_f16(wl[tap], xh, acy[tap & 1]);
    }
    if (T + 1 < T1) fw_stage(nbuf, itv, I, tid, in_s, in_s2k);
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = fmaf((acx[0][r] + acx[1][r]) + (acy[0][r] + acy[1][r]), out_sx, (acc[0][r] + acc[1][r]) * out_s);
    if (EPI == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = bk_act<ACT>(p.act, v[r] + bias_r[r]);
    } else {
      const float a4[4] = {ax.x, ax.y, ax.z, ax.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = ok ? v[r] * bk_act_grad<ACT>(p.act, a4[r]) : 0.f;
        csum[r] += v[r];
      }
    }
    if (EPI == 2 || ok) amx = odin_amax3(odin_amax3(amx, v[0], v[1]), v[2], v[3]);
    odin_run_store4s(OUT, voff, tile_out, make_float4(v[0], v[1], v[2], v[3]));
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
  }

  odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred + 128, blockIdx.x + gridDim.x * blockIdx.y);
  if (EPI == 2 && p.colsum != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vv = csum[r];
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
      if (l15 == 0) cred[wave * 16 + 4 * lq + r] = vv;
    }
    __syncthreads();
    if (tid < 32) {
      const int nbk = tid >> 4, ch = tid & 15;
      float tt = 0.f;
      for (int c = 0; c < 4; ++c) tt += cred[(4 * nbk + c) * 16 + ch];
      p.colsum[(size_t)blockIdx.x * p.CO + blockIdx.y * 32 + tid] = tt;
    }
  }
}

template <int EPI, int ACT>
int fb_launch_a(const FBParams& p, dim3 grid, void* stream) {
  const size_t lds = (size_t)2 * FW_BYTES;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fconv_blk_kernel<EPI, ACT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * FW_BYTES) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((fconv_blk_kernel<EPI, ACT>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("fconv_blk(f16x2)");
}
template <int EPI>
int fb_launch(const FBParams& p, dim3 grid, void* stream) {
  return p.act == ODIN_ACT_ELU ? fb_launch_a<EPI, ODIN_ACT_ELU>(p, grid, stream) : fb_launch_a<EPI, -1>(p, grid, stream);
}

// =====================================================================================================================
// wgrad_blk
// =====================================================================================================================
struct WBParams {
  const float* U;      // fine   [B, 2h, 2w, CUt]
  const float* V;      // coarse [B, h, w, CVt]
  float* slab;         // [gridDim.x][slab_stride]: dW [16 taps][CUt][CVt] (+ [CVt] column sums of V when want_bias)
  int B, h, w, CUt, CVt;
  int slab_stride, want_bias;
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* u_amax;
  const unsigned* v_amax;
  int u_is_grad;       // 1: U is the gradient tensor (Conv2DTranspose), 0: V is (Conv2D)
};

constexpr int WB_VPLB = 64 * 64;             // one plane of the coarse block: [pixel 64][32 f16]
constexpr int WB_BUF = FW_BYTES + 2 * WB_VPLB;

// ds_read_b64_tr_b16: the 16 lanes of a group hand in the addresses of 4 pixels x 4 channel quads (lane 4 q + p: pixel q,
// channels 4 p .. 4 p + 3 of the group's 16) and lane i receives channel i of the 4 pixels -- four consecutive k of an
// MFMA operand whose reduction index is the pixel
__device__ __forceinline__ u32x2 wb_tr(const char* addr_hw, const char* base_sim, int pix_stride, int l16) {
#ifdef ODIN_SIM
  (void)addr_hw;
  unsigned short e[4];
  for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const unsigned short*>(base_sim + q * pix_stride + 2 * l16);
  return odin_u2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
#else
  (void)base_sim; (void)pix_stride; (void)l16;
  typedef short wb_s4 __attribute__((ext_vector_type(4)));
  const wb_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wb_s4*)addr_hw);
  return __builtin_bit_cast(u32x2, v);
#endif
}

__global__ __launch_bounds__(512) void wgrad_blk_kernel(WBParams p) {
  ODIN_DYN_SMEM(char, smem);
  __shared__ float bred[8 * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15, g = (lane >> 4) & 1;
  const int cv0 = blockIdx.y * 32, cu0 = blockIdx.z * 32;
  const int FH = 2 * p.h, FWid = 2 * p.w;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq u_rq = odin_range_issue(p.u_amax, lane), v_rq = odin_range_issue(p.v_amax, lane);
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * FH * FWid * p.CUt * 4));
  const OdinRun RV = odin_run(p.V, (unsigned)((size_t)p.B * p.h * p.w * p.CVt * 4));
  FwItems I;
  fw_items(I, tid, FWid, p.CUt, cu0);
  // the coarse block: 64 pixels x 8 float4 = one item per thread
  const int v_px = tid >> 3, v_ch4 = tid & 7;
  const int v_r = v_px >> 3, v_c = v_px & 7;
  const int v_dst = v_px * 64 + v_ch4 * 8;
  const int v_g = ((v_r * p.w + v_c) * p.CVt + cv0 + 4 * v_ch4) * 4;
  float4 itv[FW_NIT], vv;
  auto issue = [&](int b, int ty, int tx) {
    fw_issue(itv, I, RU, b, ty, tx, FH, FWid, p.CUt);
    const bool ok = 8 * ty + v_r < p.h && 8 * tx + v_c < p.w;
    vv = odin_run_load4(RV, ok ? (unsigned)(((b * p.h + 8 * ty) * p.w + 8 * tx) * p.CVt * 4 + v_g) : ODIN_OOB);
  };
  int b_c, ty_c, tx_c;
  bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
  if (T0 < T1) issue(b_c, ty_c, tx_c);
  const unsigned umb = odin_range_finish(u_rq), vmb = odin_range_finish(v_rq);
  const int gu = bk_shift(umb, p.u_is_grad), gv = bk_shift(vmb, !p.u_is_grad);
  const float u_s = odin_pow2(gu), u_s2k = odin_pow2(gu + 11), v_s = odin_pow2(gv), v_s2k = odin_pow2(gv + 11);
  float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage = [&](char* buf) {
    fw_stage(buf, itv, I, tid, u_s, u_s2k);
    u32x2 h, l;
    odin_split_h4<true>(vv, v_s, v_s2k, h, l);
    *reinterpret_cast<u32x2*>(buf + FW_BYTES + v_dst) = h;
    *reinterpret_cast<u32x2*>(buf + FW_BYTES + WB_VPLB + v_dst) = l;
    bsum4.x += vv.x; bsum4.y += vv.y; bsum4.z += vv.z; bsum4.w += vv.w;
  };
  if (T0 < T1) stage(smem);
  __syncthreads();

  // ---- this wave's two taps (kh, kw0), (kh, kw0 + 1): same rows and slots, the two column parities ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1), c1 = kw0 >> 1;
  // transposed reads: lane 4 q + p of a 16-lane group addresses pixel q, channel quad p of the group's 16 channels
  const int tq = l16 >> 2, tp = l16 & 3;
  // U: k-step s covers tile rows 2 s (half 0) and 2 s + 1 (half 1); window row 2 ri + kh, slots c1 + 4 m + q
  int uoff[2];   // [m]: first / second four pixels of the row, k-step 0; a k-step adds 4 window rows
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int row = 2 * half + kh, slot = c1 + 4 * m + tq;
    uoff[m] = row * FW_ROWB + slot * 64 + (((2 * g + (tp >> 1)) ^ fw_swz(row, slot)) << 4) + (tp & 1) * 8;
  }
  // (a k-step moves down 4 window rows: (row >> 1) & 1 is unchanged, so is the swizzle)
  int voff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) voff[m] = (half * 8 + 4 * m + tq) * 64 + (16 * g + 4 * tp) * 2;
#ifdef ODIN_SIM
  int usim[2], vsim[2];   // the simulator's form: base of pixel q = 0 of the block, channel group g
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    usim[m] = (2 * half + kh) * FW_ROWB + (c1 + 4 * m) * 64;
    vsim[m] = (half * 8 + 4 * m) * 64 + 32 * g;
  }
#endif
  f32x16 acc[2] = {f32x16_zero(), f32x16_zero()}, acx[2] = {f32x16_zero(), f32x16_zero()};

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * WB_BUF;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * WB_BUF;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
      issue(b_n, ty_n, tx_n);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // B operand: V[k = pixel][cv], both planes
      u32x4 vb[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const char* vbase = buf + FW_BYTES + pl * WB_VPLB + s * 16 * 64;
#ifdef ODIN_SIM
        const u32x2 lo = wb_tr(nullptr, vbase + vsim[0], 64, l16), hi = wb_tr(nullptr, vbase + vsim[1], 64, l16);
#else
        const u32x2 lo = wb_tr(vbase + voff[0], nullptr, 0, 0), hi = wb_tr(vbase + voff[1], nullptr, 0, 0);
#endif
        vb[pl][0] = lo[0]; vb[pl][1] = lo[1]; vb[pl][2] = hi[0]; vb[pl][3] = hi[1];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        u32x4 ua[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
          const char* ubase = buf + pl * FW_PLB + s * 4 * FW_ROWB + t * 576;
#ifdef ODIN_SIM
          // (the swizzle of a pixel's k-pieces: resolved per element)
          unsigned short e[8];
          for (int k = 0; k < 8; ++k) {
            const int row = 4 * s + 2 * half + kh, slot = c1 + k, c = 16 * g + l16;
            e[k] = *reinterpret_cast<const unsigned short*>(buf + pl * FW_PLB + row * FW_ROWB + t * 576 + slot * 64 +
                                                            (((c >> 3) ^ fw_swz(row, slot)) << 4) + (c & 7) * 2);
          }
          (void)ubase; (void)usim;
          ua[pl][0] = (unsigned)e[0] | ((unsigned)e[1] << 16); ua[pl][1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
          ua[pl][2] = (unsigned)e[4] | ((unsigned)e[5] << 16); ua[pl][3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
#else
          const u32x2 lo = wb_tr(ubase + uoff[0], nullptr, 0, 0), hi = wb_tr(ubase + uoff[1], nullptr, 0, 0);
          ua[pl][0] = lo[0]; ua[pl][1] = lo[1]; ua[pl][2] = hi[0]; ua[pl][3] = hi[1];
#endif
        }
        acx[t] = mfma32_f16(ua[0], vb[1], acx[t]);
        acc[t] = mfma32_f16(ua[0], vb[0], acc[t]);
        acx[t] = mfma32_f16(ua[1], vb[0], acx[t]);
      }
    }
    if (T + 1 < T1) stage(nbuf);
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
  }

  // ---- this workgroup's slab row: dW[tap][cu0 + cu][cv0 + cv], lane = column cv = l31 ----
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const float ou = odin_pow2(-gu), ov = odin_pow2(-gv);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = kh * 4 + kw0 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = fmaf(acx[t][r], ODIN_LO_UNSCALE, acc[t][r]);
      row[((size_t)tap * p.CUt + cu0 + cu) * p.CVt + cv0 + l31] = (v * ou) * ov;
    }
  }
  if (p.want_bias && blockIdx.z == 0) {
    // column sums of V: threads with the same channel quad (tid & 7), then the 8 waves through LDS
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


// =====================================================================================================================
// bwd_blk: the WHOLE backward pass of a Conv2DTranspose(k4, s2) with 32 output channels in one launch (the block-window
// form of bwd_planes.hip): the fine window of dy is fetched, scaled and split ONCE for
//   dW[kh, kw, co, ci] = sum over pixels of dy[2 i - 1 + kh, 2 j - 1 + kw, co] * x[i, j, ci]        (wgrad_blk's waves)
//   dx[i, j, ci] = act'(aux) * sum over (kh, kw, co) of dy[2 i - 1 + kh, 2 j - 1 + kw, co] * W[kh, kw, co, ci]
// dy is the layer's largest tensor (251 MB for the audio VAE's last deconvolution at batch 256) and the two launches
// read it 1.27 x each.  Registers do not hold fconv_blk's 16 taps of weights beside the 64 accumulators of the weight
// gradient, so the data gradient splits its REDUCTION over the waves instead: wave (nb, q) owns the four taps kh = q of
// n-block nb for all four pixel blocks (32 registers of weights), leaves its partial blocks in LDS (32 KB) and, behind a
// barrier, finishes pixel block q.
// =====================================================================================================================
struct BBParams {
  const float* U;      // dy [B, 2h, 2w, 32]
  const float* V;      // x  [B, h, w, CVt]
  const float* wt;     // [16 taps][32][CVt]
  const float* aux;    // [B, h, w, CVt]: dx *= act'(aux)
  float* dx;           // [B, h, w, CVt]
  float* colsum;       // [gridDim.x][CVt] (may be null)
  float* slab;         // [gridDim.x][16 * 32 * CVt]
  int B, h, w, CVt, act;
  int slab_stride;
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* g_amax;   // range word of dy
  const unsigned* a_amax;   // optional range word of x
  unsigned* out_amax;       // range word of dx (may be null)
};

constexpr int BB_RED = 8 * 4 * 64 * 16;   // partial pixel blocks: [wave][pixel block][lane] x 16 bytes

template <int ACT>
__global__ __launch_bounds__(512) void bwd_blk_kernel(BBParams p) {
  ODIN_DYN_SMEM(char, smem);
  __shared__ float cred[8 * 16 + 16];
  char* red = smem + 2 * WB_BUF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15, g = (lane >> 4) & 1;
  const int l15 = lane & 15, lq = lane >> 4;
  const int cv0 = blockIdx.y * 32;
  const int FH = 2 * p.h, FWid = 2 * p.w;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq u_rq = odin_range_issue(p.g_amax, lane), v_rq = odin_range_issue(p.a_amax, lane);
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * FH * FWid * 32 * 4));
  const unsigned v_bytes = (unsigned)((size_t)p.B * p.h * p.w * p.CVt * 4);
  const OdinRun RV = odin_run(p.V, v_bytes);
  FwItems I;
  fw_items(I, tid, FWid, 32, 0);
  const int v_px = tid >> 3, v_ch4 = tid & 7;
  const int v_r = v_px >> 3, v_c = v_px & 7;
  const int v_dst = v_px * 64 + v_ch4 * 8;
  const int v_g = ((v_r * p.w + v_c) * p.CVt + cv0 + 4 * v_ch4) * 4;
  float4 itv[FW_NIT], vv;
  auto issue = [&](int b, int ty, int tx) {
    fw_issue(itv, I, RU, b, ty, tx, FH, FWid, 32);
    const bool ok = 8 * ty + v_r < p.h && 8 * tx + v_c < p.w;
    vv = odin_run_load4(RV, ok ? (unsigned)(((b * p.h + 8 * ty) * p.w + 8 * tx) * p.CVt * 4 + v_g) : ODIN_OOB);
  };
  int b_c, ty_c, tx_c;
  bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
  if (T0 < T1) issue(b_c, ty_c, tx_c);
  // ---- data gradient: this wave's weights, taps (kh = tq, kw = 0 .. 3) of n-block nb; lane = (channel l15, k-piece lq) ----
  const int nb = wave >> 2, tq = wave & 3;
  const int n0 = cv0 + 16 * nb;
  u32x4 dwh[4], dwl[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tap = 4 * tq + t;
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = p.wt[((size_t)(tap * 32 + 8 * lq + j)) * p.CVt + n0 + l15];
    bk_split8(make_float4(e[0], e[1], e[2], e[3]), make_float4(e[4], e[5], e[6], e[7]), 1.f, ODIN_LO_SCALE, dwh[t],
              dwl[t]);
  }
  const unsigned umb = odin_range_finish(u_rq), vmb = odin_range_finish(v_rq);
  const int gu = bk_shift(umb, 1), gv = bk_shift(vmb, 0);
  const float u_s = odin_pow2(gu), u_s2k = odin_pow2(gu + 11), v_s = odin_pow2(gv), v_s2k = odin_pow2(gv + 11);
  const float out_s = odin_pow2(-gu), out_sx = odin_pow2(-gu - 11);
  auto stage = [&](char* buf) {
    fw_stage(buf, itv, I, tid, u_s, u_s2k);
    u32x2 h, l;
    odin_split_h4<true>(vv, v_s, v_s2k, h, l);
    *reinterpret_cast<u32x2*>(buf + FW_BYTES + v_dst) = h;
    *reinterpret_cast<u32x2*>(buf + FW_BYTES + WB_VPLB + v_dst) = l;
  };
  if (T0 < T1) stage(smem);
  __syncthreads();

  // ---- weight gradient: this wave's two taps (kh, kw0), (kh, kw0 + 1) (wgrad_blk_kernel) ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1), c1 = kw0 >> 1;
  const int tq4 = l16 >> 2, tp = l16 & 3;
  int uoff[2], voff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int row = 2 * half + kh, slot = c1 + 4 * m + tq4;
    uoff[m] = row * FW_ROWB + slot * 64 + (((2 * g + (tp >> 1)) ^ fw_swz(row, slot)) << 4) + (tp & 1) * 8;
    voff[m] = (half * 8 + 4 * m + tq4) * 64 + (16 * g + 4 * tp) * 2;
  }
#ifdef ODIN_SIM
  int vsim[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) vsim[m] = (half * 8 + 4 * m) * 64 + 32 * g;
#endif
  f32x16 acc[2] = {f32x16_zero(), f32x16_zero()}, acx[2] = {f32x16_zero(), f32x16_zero()};
  // ---- data gradient: operand offsets of pixel block 0 by kw >> 1 (kh = tq fixed); the block this wave finishes ----
  const int ri0 = l15 >> 3, cj = l15 & 7;
  int boff[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int row = 2 * ri0 + tq, slot = cj + c;
    boff[c] = row * FW_ROWB + slot * 64 + ((lq ^ fw_swz(row, slot)) << 4);
  }
  const int fri = 2 * tq + ri0;   // tile row of this lane's pixel of block tq
  const OdinRun OUT = odin_run(p.dx, v_bytes);
  const OdinRun AUX = odin_run(p.aux, v_bytes);
  const unsigned out_lane = (unsigned)(((fri * p.w + cj) * p.CVt + n0 + 4 * lq) * 4);
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * WB_BUF;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * WB_BUF;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
      issue(b_n, ty_n, tx_n);
    }
    const unsigned tile_out = (unsigned)(((b_c * p.h + 8 * ty_c) * p.w + 8 * tx_c) * p.CVt * 4);
    const bool ok = 8 * ty_c + fri < p.h && 8 * tx_c + cj < p.w;
    const unsigned vo = ok ? out_lane : ODIN_OOB_V;
    const float4 ax = odin_run_load4s(AUX, vo, tile_out);
    // ---- weight gradient: 4 k-steps of 16 pixels ----
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 vb[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const char* vbase = buf + FW_BYTES + pl * WB_VPLB + s * 16 * 64;
#ifdef ODIN_SIM
        const u32x2 lo = wb_tr(nullptr, vbase + vsim[0], 64, l16), hi = wb_tr(nullptr, vbase + vsim[1], 64, l16);
#else
        const u32x2 lo = wb_tr(vbase + voff[0], nullptr, 0, 0), hi = wb_tr(vbase + voff[1], nullptr, 0, 0);
#endif
        vb[pl][0] = lo[0]; vb[pl][1] = lo[1]; vb[pl][2] = hi[0]; vb[pl][3] = hi[1];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        u32x4 ua[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#ifdef ODIN_SIM
          unsigned short e[8];
          for (int k = 0; k < 8; ++k) {
            const int row = 4 * s + 2 * half + kh, slot = c1 + k, c = 16 * g + l16;
            e[k] = *reinterpret_cast<const unsigned short*>(buf + pl * FW_PLB + row * FW_ROWB + t * 576 + slot * 64 +
                                                            (((c >> 3) ^ fw_swz(row, slot)) << 4) + (c & 7) * 2);
          }
          ua[pl][0] = (unsigned)e[0] | ((unsigned)e[1] << 16); ua[pl][1] = (unsigned)e[2] | ((unsigned)e[3] << 16);
          ua[pl][2] = (unsigned)e[4] | ((unsigned)e[5] << 16); ua[pl][3] = (unsigned)e[6] | ((unsigned)e[7] << 16);
#else
          const char* ubase = buf + pl * FW_PLB + s * 4 * FW_ROWB + t * 576;
          const u32x2 lo = wb_tr(ubase + uoff[0], nullptr, 0, 0), hi = wb_tr(ubase + uoff[1], nullptr, 0, 0);
          ua[pl][0] = lo[0]; ua[pl][1] = lo[1]; ua[pl][2] = hi[0]; ua[pl][3] = hi[1];
#endif
        }
        acx[t] = mfma32_f16(ua[0], vb[1], acx[t]);
        acc[t] = mfma32_f16(ua[0], vb[0], acc[t]);
        acx[t] = mfma32_f16(ua[1], vb[0], acx[t]);
      }
    }
    // ---- data gradient: this wave's four taps over the four pixel blocks, partial blocks to LDS ----
    {
      f32x4 dacc[4], dacx[4], dacy[4];   // (main, high x low, low x high: an accumulator is touched once per tap)
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) { dacc[pb] = bk_zero4(); dacx[pb] = bk_zero4(); dacy[pb] = bk_zero4(); }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
          const char* a = buf + boff[t >> 1] + pb * 4 * FW_ROWB + (t & 1) * 576;
          const u32x4 xh = *reinterpret_cast<const u32x4*>(a);
          const u32x4 xl = *reinterpret_cast<const u32x4*>(a + FW_PLB);
          dacx[pb] = mfma16_f16(dwh[t], xl, dacx[pb]);
          dacc[pb] = mfma16_f16(dwh[t], xh, dacc[pb]);
          dacy[pb] = mfma16_f16(dwl[t], xh, dacy[pb]);
        }
#pragma unroll
      for (int pb = 0; pb < 4; ++pb) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(dacx[pb][r] + dacy[pb][r], out_sx, dacc[pb][r] * out_s);
        *reinterpret_cast<f32x4*>(red + ((wave * 4 + pb) * 64 + lane) * 16) = v;
      }
    }
    __syncthreads();
    {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(red + (((4 * nb + j) * 4 + tq) * 64 + lane) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += q[r];
      }
      const float a4[4] = {ax.x, ax.y, ax.z, ax.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = ok ? v[r] * bk_act_grad<ACT>(p.act, a4[r]) : 0.f;
        csum[r] += v[r];
      }
      amx = odin_amax3(odin_amax3(amx, v[0], v[1]), v[2], v[3]);
      odin_run_store4s(OUT, vo, tile_out, make_float4(v[0], v[1], v[2], v[3]));
    }
    if (T + 1 < T1) stage(nbuf);
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
  }

  // ---- weight gradient: this workgroup's slab row ----
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const float ou = odin_pow2(-gu), ov = odin_pow2(-gv);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tap = kh * 4 + kw0 + t;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = fmaf(acx[t][r], ODIN_LO_UNSCALE, acc[t][r]);
      row[((size_t)tap * 32 + cu) * p.CVt + cv0 + l31] = (v * ou) * ov;
    }
  }
  odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred + 128, blockIdx.x + gridDim.x * blockIdx.y);
  if (p.colsum != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = csum[r];
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) t += __shfl_xor(t, m);
      if (l15 == 0) cred[wave * 16 + 4 * lq + r] = t;
    }
    __syncthreads();
    if (tid < 32) {
      const int nbk = tid >> 4, ch = tid & 15;
      float tt = 0.f;
      for (int c = 0; c < 4; ++c) tt += cred[(4 * nbk + c) * 16 + ch];
      p.colsum[(size_t)blockIdx.x * p.CVt + cv0 + tid] = tt;
    }
  }
}

}  // namespace

// Conv2D(k4, s2, SAME) forward over 32 input channels / Conv2DTranspose(k4, s2) data gradient over 32 output channels
bool odin_fconv_blk_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S, int pt, int pl,
                               int center) {
  if (!odin_blk_enabled(2.0 * B * OH * OW * 16.0 * CI * CO * blk_scale_for_width(OW))) return false;
  if (!(KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && !center && CI == 32 && (CO % 32) == 0)) return false;
  if (H != 2 * OH || W != 2 * OW || OH < 1 || OW < 1 || H > 8192 || W > 8192) return false;
  return (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * OH * OW * CO * 4 < 0x7FFF0000ull;
}

// epi 1: forward (bias + act); epi 2: data gradient (x act'(aux), column sums into colsum[rows][CO])
int odin_fconv_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int OH, int OW, int CI, int CO, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream) {
  FBParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.OH = OH; p.OW = OW; p.CS = CI; p.CO = CO; p.act = act;
  p.nty = (OH + 7) / 8; p.ntx = (OW + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = CO / 32;
  p.tiles_per_wg = tb_tiles_per_wg(p.n_tiles, gy);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  p.in_is_grad = epi == 2;
  if (epi == 2) {
    if (aux == nullptr) { p.act = ODIN_ACT_LINEAR; p.aux = out; }
    p.in_amax = odin_range_word_of(in, (size_t)B * 4 * OH * OW * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "fconv_blk: no range word for the gradient input");
  } else {
    p.in_amax = in_amax;
  }
  p.out_amax = out_amax;
  dim3 grid(gx, gy, 1);
  return epi == 1 ? fb_launch<1>(p, grid, stream) : fb_launch<2>(p, grid, stream);
}

// weight gradient of a 4x4 / stride-2 layer: U fine [B, 2 OH, 2 OW, CI], V coarse [B, OH, OW, CO] (names of wgrad_planes.hip)
bool odin_wgrad_blk_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S, int pt, int pl,
                               int center) {
  if (!odin_blk_enabled(2.0 * B * OH * OW * 16.0 * CI * CO * blk_scale_for_width(OW))) return false;
  if (!(KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && !center && (CI % 32) == 0 && (CO % 32) == 0)) return false;
  if (H != 2 * OH || W != 2 * OW || OH < 1 || OW < 1 || H > 8192 || W > 8192) return false;
  return (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * OH * OW * CO * 4 < 0x7FFF0000ull;
}

static int wb_tiles_per_wg(int n_tiles, int gyz) {
  int cap = odin_num_cus() / gyz;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_SLAB_BLOCKS) cap = ODIN_MAX_SLAB_BLOCKS;
  return (n_tiles + cap - 1) / cap;
}

int odin_wgrad_blk_launch(const float* U, const float* V, float* slab, int* rows_out, int B, int OH, int OW, int CI,
                          int CO, int want_bias, int grad_u, const uint32_t* g_amax, const uint32_t* a_amax,
                          void* stream) {
  WBParams p;
  memset(&p, 0, sizeof(p));
  p.U = U; p.V = V; p.slab = slab;
  p.B = B; p.h = OH; p.w = OW; p.CUt = CI; p.CVt = CO; p.want_bias = want_bias;
  p.slab_stride = 16 * CI * CO + (want_bias ? CO : 0);
  p.nty = (OH + 7) / 8; p.ntx = (OW + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = CO / 32, gz = CI / 32;
  p.tiles_per_wg = wb_tiles_per_wg(p.n_tiles, gy * gz);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (slab == nullptr) return 0;  // dry run
  const uint32_t* gw = grad_u ? odin_range_word_of(U, (size_t)B * 4 * OH * OW * CI, g_amax, stream)
                              : odin_range_word_of(V, (size_t)B * OH * OW * CO, g_amax, stream);
  if (gw == nullptr) return odin_fail(-3, "wgrad_blk: no range word for the gradient operand");
  p.u_is_grad = grad_u;
  p.u_amax = grad_u ? gw : a_amax;
  p.v_amax = grad_u ? a_amax : gw;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_blk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * WB_BUF) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  dim3 grid(gx, gy, gz);
  ODIN_LAUNCH((wgrad_blk_kernel), grid, dim3(512), (size_t)2 * WB_BUF, stream, p);
  return odin_check_launch("wgrad_blk(f16x2)");
}

// =====================================================================================================================
// the fused Gaussian tail: Conv2DTranspose(k4, s2, 32 -> 32, act) -> Conv2D 1x1 (2 maps: loc | raw scale) ->
// Independent(Normal(loc, raw | softplus1(raw))).log_prob(target) AND its backward in the epilogue of tconv_blk (the
// audio VAE's decoder4 -> decoder6 -> observation, examples/vae/vae_audio.py:84-110, image_networks.py:505-511): the
// [B, 2H, 2W, 32] activation never reaches HBM -- the layer writes dL/d(pre-activation) where odin_deconv2d_fwd +
// odin_gaussian_head_fwd_bwd wrote the activation, read it back and wrote the gradient (3 x 251 MB at batch 256).
// A wave owns one parity class and TWO pixel blocks with both 16-channel output blocks, so a pixel's 32 channels sit in
// the 4 lanes (lane & 15 = pixel, lane >> 4 = channel quad of either block) that meet by two cross-lane adds.
// =====================================================================================================================
namespace {

struct TGParams {
  const float* in;      // [B, H, W, 32]
  const float* w;       // [16 taps][32][32]
  const float* bias;    // [32]
  const float* w1;      // [32][2]
  const float* b1;      // [2]
  const float* target;  // [B, 2H, 2W, 1]
  float* logits;        // [B, 2H, 2W, 2] (may be null)
  float* out;           // [B, 2H, 2W, 32]: dL/d(pre-activation of the layer), L = -scale[0] * sum llk
  float* llk_part;      // [n_tiles]: one partial per tile (a tile lies inside one sample)
  float* slab;          // [gridDim.x][32 * 2 (dW1) | 2 (db1) | 32 (column sums of out)]
  const float* scale;
  int B, H, W, act;
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* in_amax;
  unsigned* out_amax;
};

constexpr int TG_ROW = 32 * 2 + 2 + 32;

template <int SP1, int ACT>
__global__ __launch_bounds__(512) void tconv_blk_gtail_kernel(TGParams p) {
  ODIN_DYN_SMEM(char, smem);
  __shared__ float cred[8 * TG_ROW + 16];
  __shared__ float llk_red[2][8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int cls = wave & 3, pbh = wave >> 2;
  const int cpw = cls & 1, rpar = cls >> 1;
  const int OH = 2 * p.H, OW = 2 * p.W;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq in_rq = odin_range_issue(p.in_amax, lane);
  const int kh_a = rpar ? 0 : 1, kw_a = cpw ? 0 : 1;
  float4 wv[4][2][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int tap = (kh_a + 2 * (t >> 1)) * 4 + kw_a + 2 * (t & 1);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const float* src = p.w + ((size_t)(tap * 32 + 16 * nb + l15) * 32 + 8 * lq);
      wv[t][nb][0] = *reinterpret_cast<const float4*>(src);
      wv[t][nb][1] = *reinterpret_cast<const float4*>(src + 4);
    }
  }
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * 32 * 4));
  int it_dst[2], it_g[2], it_wr[2], it_wc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int e = tid + 512 * j;
    const int px = e >> 3, ch4 = e & 7;
    const int wr = odin_div_small(px, 10), wc = px - 10 * wr;
    it_wr[j] = (e < 800) ? wr : (1 << 20);
    it_wc[j] = wc;
    it_dst[j] = wr * 1024 + wc * 64 + (((ch4 >> 1) ^ tb_swz(wr, wc)) << 4) + (ch4 & 1) * 8;
    it_g[j] = ((wr * p.W + wc) * 32 + 4 * ch4) * 4;
  }
  float4 itv[2];
  auto issue = [&](int b, int ty, int tx) {
    const int base = ((b * p.H + 8 * ty - 1) * p.W + 8 * tx - 1) * 32 * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gr = 8 * ty - 1 + it_wr[j], gc = 8 * tx - 1 + it_wc[j];
      const bool ok = gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
      itv[j] = odin_run_load4(IN, ok ? (unsigned)(base + it_g[j]) : ODIN_OOB);
    }
  };
  float in_s = 1.f, in_s2k = ODIN_LO_SCALE, out_s = 1.f, out_sx = ODIN_LO_UNSCALE;
  auto stage = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (tid + 512 * j < 800) {
        u32x2 h, l;
        odin_split_h4<true>(itv[j], in_s, in_s2k, h, l);
        *reinterpret_cast<u32x2*>(buf + it_dst[j]) = h;
        *reinterpret_cast<u32x2*>(buf + it_dst[j] + TB_PLB) = l;
      }
    }
  };
  int b_c, ty_c, tx_c;
  bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
  if (T0 < T1) issue(b_c, ty_c, tx_c);
  {
    const unsigned mb = odin_range_finish(in_rq);
    const int gk = bk_shift(mb, 0);
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11);
    out_s = odin_pow2(-gk); out_sx = odin_pow2(-gk - 11);
  }
  u32x4 wh[4][2], wl[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) bk_split8(wv[t][nb][0], wv[t][nb][1], 1.f, ODIN_LO_SCALE, wh[t][nb], wl[t][nb]);
  if (T0 < T1) stage(smem);
  __syncthreads();

  const int ri0 = l15 >> 3, cj = l15 & 7;   // pixel block pb: tile rows 2 pb + ri0
  int boff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = ri0 + (rpar ? 2 : 1) - (t >> 1), col = cj + (cpw ? 2 : 1) - (t & 1);
    boff[t] = row * 1024 + col * 64 + ((lq ^ tb_swz(row, col)) << 4);
  }
  const unsigned npix_bytes = (unsigned)((size_t)p.B * OH * OW * 4);
  const OdinRun OUT = odin_run(p.out, npix_bytes * 32u);
  const OdinRun TG = odin_run(p.target, npix_bytes);
  const OdinRun LG = odin_run(p.logits, p.logits != nullptr ? npix_bytes * 2u : 0u);
  const unsigned pix_lane = (unsigned)((2 * ri0 + rpar) * OW + 2 * cj + cpw);
  const unsigned pix_step = (unsigned)(4 * OW);   // two tile rows = four fine rows
  // this lane's channels: 16 nb + 4 lq + r
  float bias_r[2][4], w1r[2][4][2], dw1[2][4][2], csum[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = 16 * nb + 4 * lq + r;
      bias_r[nb][r] = p.bias[ch];
      w1r[nb][r][0] = p.w1[2 * ch];
      w1r[nb][r][1] = p.w1[2 * ch + 1];
      dw1[nb][r][0] = dw1[nb][r][1] = 0.f;
      csum[nb][r] = 0.f;
    }
  const float b1_0 = p.b1[0], b1_1 = p.b1[1];
  float db1_0 = 0.f, db1_1 = 0.f;
  const float sc = p.scale[0];
  float amx = 0.f;

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * TB_KPB;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * TB_KPB;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
      issue(b_n, ty_n, tx_n);
    }
    const unsigned tile_pix = (unsigned)((b_c * OH + 16 * ty_c) * OW + 16 * tx_c);
    unsigned vpix[2];
    float tgt[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int pb = 2 * pbh + i;
      const bool ok = 8 * ty_c + 2 * pb + ri0 < p.H && 8 * tx_c + cj < p.W;
      vpix[i] = ok ? pix_lane + pb * pix_step : 0x3FFF0000u;   // (x 4 ... x 128 bytes stays out of range, no wrap below 2^32)
      tgt[i] = odin_run_load1s(TG, ok ? vpix[i] * 4u : ODIN_OOB_V, tile_pix * 4u);
    }
    f32x4 acc[2][2], acx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) { acc[i][nb] = bk_zero4(); acx[i][nb] = bk_zero4(); }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      // (three sweeps over the four accumulator pairs: MFMAs into the same accumulator are four instructions apart)
      u32x4 xh[2], xl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* a = buf + boff[t] + (2 * pbh + i) * 2048;
        xh[i] = *reinterpret_cast<const u32x4*>(a);
        xl[i] = *reinterpret_cast<const u32x4*>(a + TB_PLB);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acx[i][nb] = mfma16_f16(wh[t][nb], xl[i], acx[i][nb]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[i][nb] = mfma16_f16(wh[t][nb], xh[i], acc[i][nb]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acx[i][nb] = mfma16_f16(wl[t][nb], xh[i], acx[i][nb]);
    }
    if (T + 1 < T1) stage(nbuf);
    float llk_lane = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = vpix[i] != 0x3FFF0000u;
      float y[2][4];
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          y[nb][r] = bk_act<ACT>(p.act, fmaf(acx[i][nb][r], out_sx, acc[i][nb][r] * out_s) + bias_r[nb][r]);
          t0 = fmaf(y[nb][r], w1r[nb][r][0], t0);
          t1 = fmaf(y[nb][r], w1r[nb][r][1], t1);
        }
      t0 = bk_add_xor32(bk_add_xor16(t0));
      t1 = bk_add_xor32(bk_add_xor16(t1));
      const float loc = t0 + b1_0, raw = t1 + b1_1;
      float sd, dsd;
      if (SP1 == 1) {  // softplus1(raw) = softplus(raw + softplus^-1(1)); its derivative = sigmoid of the same
        const float a = raw + 0.5413248546129181f;
        const float e = odin_exp2(-1.4426950408889634f * fabsf(a));
        const float r = odin_rcp(1.f + e);
        sd = fmaxf(a, 0.f) + 0.6931471805599453f * odin_log2(1.f + e);
        dsd = a >= 0.f ? r : e * r;
      } else {
        sd = raw;
        dsd = 1.f;
      }
      const float inv = 1.f / sd;
      const float d = (tgt[i] - loc) * inv;
      float l1 = -0.5f * d * d - 0.6931471805599453f * odin_log2(sd) - 0.5f * 1.8378770664093453f;
      float dl0 = -(d * inv) * sc, dl1 = -((d * d - 1.f) * inv) * dsd * sc;
      if (!ok) { dl0 = 0.f; dl1 = 0.f; l1 = 0.f; }
      const bool first = lq == 0;
      llk_lane += first ? l1 : 0.f;
      db1_0 += first ? dl0 : 0.f;
      db1_1 += first ? dl1 : 0.f;
      const unsigned lgoff = (ok && first) ? vpix[i] * 8u : ODIN_OOB_V;
      odin_run_store1s(LG, lgoff, tile_pix * 8u, loc);
      odin_run_store1s(LG, lgoff == ODIN_OOB_V ? ODIN_OOB_V : lgoff + 4u, tile_pix * 8u, raw);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float gq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gs = fmaf(w1r[nb][r][1], dl1, w1r[nb][r][0] * dl0);
          dw1[nb][r][0] = fmaf(y[nb][r], dl0, dw1[nb][r][0]);
          dw1[nb][r][1] = fmaf(y[nb][r], dl1, dw1[nb][r][1]);
          gq[r] = gs * bk_act_grad<ACT>(p.act, y[nb][r]);
          csum[nb][r] += gq[r];
        }
        amx = odin_amax3(odin_amax3(amx, gq[0], gq[1]), gq[2], gq[3]);
        odin_run_store4s(OUT, ok ? vpix[i] * 128u + (unsigned)((16 * nb + 4 * lq) * 4) : ODIN_OOB_V, tile_pix * 128u,
                         make_float4(gq[0], gq[1], gq[2], gq[3]));
      }
    }
    {
      const float tt = wave_sum64(llk_lane);
      if (lane == 0) llk_red[(T - T0) & 1][wave] = tt;
    }
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
    if (tid == 0) {
      const float* q = llk_red[(T - T0) & 1];
      p.llk_part[T] = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
  }

  odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred + 8 * TG_ROW, blockIdx.x);
  // ---- slab row [dW1 (32 x 2) | db1 (2) | column sums of out (32)]: the 16 pixel lanes, then the 8 waves through LDS ----
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ch = 16 * nb + 4 * lq + r;
      float a0 = dw1[nb][r][0], a1 = dw1[nb][r][1], cs = csum[nb][r];
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) { a0 += __shfl_xor(a0, m); a1 += __shfl_xor(a1, m); cs += __shfl_xor(cs, m); }
      if (l15 == 0) {
        cred[wave * TG_ROW + 2 * ch] = a0;
        cred[wave * TG_ROW + 2 * ch + 1] = a1;
        cred[wave * TG_ROW + 66 + ch] = cs;
      }
    }
  {
    float d0 = db1_0, d1 = db1_1;
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) { d0 += __shfl_xor(d0, m); d1 += __shfl_xor(d1, m); }
    if (lane == 0) { cred[wave * TG_ROW + 64] = d0; cred[wave * TG_ROW + 65] = d1; }
  }
  __syncthreads();
  if (tid < TG_ROW) {
    float tt = 0.f;
    for (int wv = 0; wv < 8; ++wv) tt += cred[wv * TG_ROW + tid];
    p.slab[(size_t)blockIdx.x * TG_ROW + tid] = tt;
  }
}

}  // namespace

// 1: odin_gaussian_tail_fwd_bwd takes this layer (Conv2DTranspose k4 s2 32 -> 32 + 1x1 head of 2 maps, C = 1)
extern "C" int odin_gaussian_tail_applicable(const odin_conv_desc* d, int C) {
  return (C == 1 && d->Cin == 32 && d->Cout == 32 && d->OH == 2 * d->H && d->OW == 2 * d->W &&
          odin_tconv_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l,
                                    d->center) &&
          (size_t)d->B * d->OH * d->OW < 0x3FFF0000ull / 32) ? 1 : 0;
}

extern "C" int odin_gaussian_tail_fwd_bwd(const float* x, const float* w, const float* bias, const float* w1,
                                          const float* b1, const float* target, float* logits, float* g_out,
                                          float* llk_part, int* n_part_out, float* tail_slab, int* slab_rows_out,
                                          const float* scale, const odin_conv_desc* d, int C, int softplus1,
                                          void* stream) {
  if (!odin_gaussian_tail_applicable(d, C) || (softplus1 != 0 && softplus1 != 1))
    return odin_fail(-2, "gaussian_tail: shapes outside the kernel");
  TGParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.w = w; p.bias = bias; p.w1 = w1; p.b1 = b1; p.target = target; p.logits = logits; p.out = g_out;
  p.llk_part = llk_part; p.slab = tail_slab; p.scale = scale;
  p.B = d->B; p.H = d->H; p.W = d->W; p.act = d->act;
  p.nty = (d->H + 7) / 8; p.ntx = (d->W + 7) / 8;
  p.n_tiles = d->B * p.nty * p.ntx;
  p.tiles_per_wg = tb_tiles_per_wg(p.n_tiles, 1);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (slab_rows_out) *slab_rows_out = gx;
  if (n_part_out) *n_part_out = p.nty * p.ntx;
  if (g_out == nullptr) return 0;  // dry run
  p.in_amax = d->x_amax;
  p.out_amax = d->dy_amax;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[4] = {reinterpret_cast<const void*>(&tconv_blk_gtail_kernel<0, ODIN_ACT_ELU>),
                          reinterpret_cast<const void*>(&tconv_blk_gtail_kernel<1, ODIN_ACT_ELU>),
                          reinterpret_cast<const void*>(&tconv_blk_gtail_kernel<0, -1>),
                          reinterpret_cast<const void*>(&tconv_blk_gtail_kernel<1, -1>)};
    for (int i = 0; i < 4; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_KPB) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  const size_t lds = (size_t)2 * TB_KPB;
  if (d->act == ODIN_ACT_ELU) {
    if (softplus1 == 1) ODIN_LAUNCH((tconv_blk_gtail_kernel<1, ODIN_ACT_ELU>), dim3(gx), dim3(512), lds, stream, p);
    else ODIN_LAUNCH((tconv_blk_gtail_kernel<0, ODIN_ACT_ELU>), dim3(gx), dim3(512), lds, stream, p);
  } else {
    if (softplus1 == 1) ODIN_LAUNCH((tconv_blk_gtail_kernel<1, -1>), dim3(gx), dim3(512), lds, stream, p);
    else ODIN_LAUNCH((tconv_blk_gtail_kernel<0, -1>), dim3(gx), dim3(512), lds, stream, p);
  }
  return odin_check_launch("tconv_blk_gtail(f16x2)");
}

// the whole backward pass of a Conv2DTranspose(k4, s2) with 32 output channels (x [B, H, W, Cin] -> dy [B, 2H, 2W, 32])
bool odin_bwd_blk_applicable(int B, int H, int W, int Cin, int Cout) {
  if (!odin_blk_enabled(2.0 * B * H * W * 16.0 * Cin * Cout * blk_scale_for_width(W))) return false;
  if (!(Cout == 32 && (Cin % 32) == 0 && H >= 1 && W >= 1 && H <= 4096 && W <= 4096)) return false;
  return (size_t)B * 4 * H * W * Cout * 4 < 0x7FFF0000ull && (size_t)B * H * W * Cin * 4 < 0x7FFF0000ull;
}

int odin_bwd_blk_rows(int B, int H, int W, int Cin) {
  const int n_tiles = B * ((H + 7) / 8) * ((W + 7) / 8);
  const int tpw = wb_tiles_per_wg(n_tiles, Cin / 32);
  return (n_tiles + tpw - 1) / tpw;
}

int odin_bwd_blk_launch(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                        float* colsum, float* wslab, int B, int H, int W, int Cin, int Cout, const uint32_t* dy_amax,
                        const uint32_t* x_amax, uint32_t* dx_amax, void* stream) {
  BBParams p;
  memset(&p, 0, sizeof(p));
  p.U = dy; p.V = x; p.wt = w; p.aux = aux; p.dx = dx; p.colsum = colsum; p.slab = wslab;
  p.B = B; p.h = H; p.w = W; p.CVt = Cin; p.act = aux_act;
  p.slab_stride = 16 * Cout * Cin;
  p.nty = (H + 7) / 8; p.ntx = (W + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = Cin / 32;
  p.tiles_per_wg = wb_tiles_per_wg(p.n_tiles, gy);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (aux == nullptr || aux_act == ODIN_ACT_LINEAR) { p.aux = x; p.act = ODIN_ACT_LINEAR; }
  p.g_amax = odin_range_word_of(dy, (size_t)B * 4 * H * W * Cout, dy_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "bwd_blk: no range word for dy");
  p.a_amax = x_amax;
  p.out_amax = dx_amax;
  const size_t lds = (size_t)2 * WB_BUF + BB_RED;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[2] = {reinterpret_cast<const void*>(&bwd_blk_kernel<ODIN_ACT_ELU>),
                          reinterpret_cast<const void*>(&bwd_blk_kernel<-1>)};
    for (int i = 0; i < 2; ++i)
      if (hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  dim3 grid(gx, gy, 1);
  if (p.act == ODIN_ACT_ELU) ODIN_LAUNCH((bwd_blk_kernel<ODIN_ACT_ELU>), grid, dim3(512), lds, stream, p);
  else ODIN_LAUNCH((bwd_blk_kernel<-1>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("bwd_blk(f16x2)");
}
