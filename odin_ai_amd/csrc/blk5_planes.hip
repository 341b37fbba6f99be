// blk5_planes.hip -- the 5x5 / stride-1 `SAME` convolutions of the MNIST conv stack (image_networks.py:244-271:
// Conv2D(32, 5, 1) / Conv2D(64, 5, 1) between the strided layers; 14 x 14 and 28 x 28 maps) as two-plane f16 products over
// BLOCK WINDOWS, the 25-tap sibling of blk_planes.hip (round 6; VERDICT r5 item 3: "rows of 28 / 14, k5").
//
// On igemm_h.hip these layers ran at 55-100 TFLOP/s in fp32 FLOPs -- 58-92 us for 5.1 GFLOP at batch 128, 0.53 ms of the
// 0.99 ms step in three layers x three roles -- although they move 13-26 MB: a 32 x 32 tile per wave there pays a row
// decode, a gather and a split per fragment for 25 x 2 steps.  Here a tile is a block of 8 x 8 output pixels; its 12 x 12
// input window is fetched one tile ahead (zero fill outside the image), scaled by the tensor's range word and split ONCE
// into planes on its way into LDS (two buffers); the weights of the workgroup's 32 output channels over 32 reduction
// channels sit in LDS as well, split once per pass ([tap][plane][k-piece][n] x 16 bytes = 100 KB: 25 taps do not fit in
// registers).  v_mfma_f32_16x16x32_f16: an MFMA step is one tap over 32 channels; the eight waves own the tile's eight
// 16-pixel x 16-channel blocks.  64 reduction channels take two passes over the workgroup's tiles -- the second adds the
// partial sums the first left in `out` (the same thread reads back what it wrote; 13-26 MB, L2 / Infinity Cache
// resident) -- because the second half of the weights needs the same LDS.
//
//   conv5_blk  EPI 1: Conv2D(k5, s1) forward (bias + activation)                 out[y, x, n] = sum in[y + kh - 2, x + kw - 2, c] W[kh, kw, c, n]
//              EPI 2: its DATA GRADIENT (x act'(aux), column sums): the same gather over dy with the taps reversed and
//              the weight matrix transposed                                       dx[y, x, c] = sum dy[y + 2 - kh, x + 2 - kw, n] W[kh, kw, c, n]
//   wgrad5_blk the weight gradient                                               dW[kh, kw, c, n] = sum x[y + kh - 2, x + kw - 2, c] dy[y, x, n]
#include "odin_device.h"
#include "odin_internal.h"
#include "blk_common.h"
#include <cstdlib>

namespace {

struct C5Params {
  const float* in;     // [B, H, W, CS]
  const float* w;      // wmode 0: [25][CS][CO]; wmode 1 (data gradient): the layer's [25][CO][CS], taps reversed
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, H, W, CO], out *= act'(aux)
  float* out;          // [B, H, W, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] (may be null)
  int B, H, W, CS, CO;
  int act, wmode, nk;
  int pad;             // rows / columns of padding before the first pixel: (K - 1) / 2 forward, K - 1 - (K - 1) / 2 for the data gradient
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* in_amax;
  unsigned* out_amax;
  int in_is_grad;
};

constexpr int C5_TAPB = 2 * 4 * 32 * 16;     // one tap of the weight planes: [plane][k-piece][n 32][8 f16]
constexpr int C5_PLB = 12 * 16 * 64;         // one plane of a window: [row <= 12][slot 16][32 f16]
constexpr int C5_WIN = 2 * C5_PLB;
// K = 5 (MNIST) or 4 (CelebA's Conv2D(64, 4, 1): `SAME` pads (1, 2)): K x K taps, a window of 8 + K - 1 pixels a side
__host__ __device__ constexpr int c5_wb(int K) { return K * K * C5_TAPB; }              // weight planes: 100 KB / 64 KB
__host__ __device__ constexpr int c5_lds(int K) { return c5_wb(K) + 2 * C5_WIN; }        // 151552 / 114688 bytes
__host__ __device__ constexpr int c5_nit(int K) { return ((8 + K - 1) * (8 + K - 1) * 8 + 511) / 512; }

template <int EPI, int ACT, int K>
__global__ __launch_bounds__(512) void conv5_blk_kernel(C5Params p) {
  constexpr int WS = 8 + K - 1, NT = K * K, C5_NIT = c5_nit(K), C5_WB = c5_wb(K);
  ODIN_DYN_SMEM(char, smem);
  __shared__ float cred[8 * 16 + 16];
  char* wl = smem;
  char* win = smem + C5_WB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l15 = lane & 15, lq = lane >> 4;
  const int pb = wave & 3, nb = wave >> 2;
  const int n0w = blockIdx.y * 32, n0 = n0w + 16 * nb;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq in_rq = odin_range_issue(p.in_amax, lane);
  const OdinRun IN = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * p.CS * 4));
  int it_dst[C5_NIT], it_g[C5_NIT], it_wr[C5_NIT], it_wc[C5_NIT];
#pragma unroll
  for (int j = 0; j < C5_NIT; ++j) {
    const int e = tid + 512 * j;
    const int px = e >> 3, ch4 = e & 7;
    const int wr = odin_div_small(px, WS), wc = px - WS * wr;
    it_wr[j] = (e < WS * WS * 8) ? wr : (1 << 20);
    it_wc[j] = wc;
    it_dst[j] = wr * 1024 + wc * 64 + (((ch4 >> 1) ^ tb_swz(wr, wc)) << 4) + (ch4 & 1) * 8;
    it_g[j] = ((wr * p.W + wc) * p.CS + 4 * ch4) * 4;
  }
  float4 itv[C5_NIT];
  int kp = 0;
  auto issue = [&](int b, int ty, int tx) {
    const int base = ((b * p.H + 8 * ty - p.pad) * p.W + 8 * tx - p.pad) * p.CS * 4 + kp * 128;
#pragma unroll
    for (int j = 0; j < C5_NIT; ++j) {
      const int gr = 8 * ty - p.pad + it_wr[j], gc = 8 * tx - p.pad + it_wc[j];
      const bool ok = gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
      itv[j] = odin_run_load4(IN, ok ? (unsigned)(base + it_g[j]) : ODIN_OOB);
    }
  };
  float in_s, in_s2k, out_s;
  {
    const unsigned mb = odin_range_finish(in_rq);
    const int gk = bk_shift(mb, p.in_is_grad);
    in_s = odin_pow2(gk); in_s2k = odin_pow2(gk + 11);
    out_s = odin_pow2(-gk);
  }
  auto stage = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < C5_NIT; ++j) {
      if (tid + 512 * j < WS * WS * 8) {
        u32x2 h, l;
        odin_split_h4<true>(itv[j], in_s, in_s2k, h, l);
        *reinterpret_cast<u32x2*>(buf + it_dst[j]) = h;
        *reinterpret_cast<u32x2*>(buf + it_dst[j] + C5_PLB) = l;
      }
    }
  };
  // ---- per-lane constants: the lane's output pixel (ri, cj) of the tile, operand offsets by (kh parity, kw) ----
  const int ri = 2 * pb + (l15 >> 3), cj = l15 & 7;
  int boff[2][K];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr)
#pragma unroll
    for (int kw = 0; kw < K; ++kw)
      boff[pr][kw] = ri * 1024 + (cj + kw) * 64 + ((lq ^ tb_swz(ri + pr, cj + kw)) << 4);
  const char* wlane = wl + lq * 512 + (16 * nb + l15) * 16;
  const unsigned out_bytes = (unsigned)((size_t)p.B * p.H * p.W * p.CO * 4);
  const OdinRun OUT = odin_run(p.out, out_bytes);
  const OdinRun AUX = odin_run(EPI == 2 ? p.aux : nullptr, EPI == 2 ? out_bytes : 0u);
  const unsigned out_lane = (unsigned)(((ri * p.W + cj) * p.CO + n0 + 4 * lq) * 4);
  float bias_r[4] = {0.f, 0.f, 0.f, 0.f};
  if (EPI == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bias_r[r] = p.bias[n0 + 4 * lq + r];
  }
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;

#pragma unroll 1
  for (kp = 0; kp < p.nk; ++kp) {
    const bool last = kp == p.nk - 1;
    if (kp > 0) {
      // the partial sums of the previous pass: written and read back by the SAME thread (workgroup-scope fence, as
      // tconv_planes2_kernel); the barrier also keeps this pass's weight stores behind the last pass's LDS reads
      odin_wait_vmem();
#ifndef ODIN_SIM
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
      __syncthreads();
    }
    int b_c, ty_c, tx_c;
    bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
    if (T0 < T1) issue(b_c, ty_c, tx_c);
    // ---- this pass's weights -> planes: item = (tap, channel n, k-piece lq): eight reduction channels ----
    for (int e = tid; e < NT * 32 * 4; e += 512) {
      const int q = e & 3, n = (e >> 2) & 31, tap = e >> 7;
      float4 a, b;
      if (p.wmode == 0) {
        const float* src = p.w + ((size_t)(tap * p.CS + 32 * kp + 8 * q)) * p.CO + n0w + n;
        a = make_float4(src[0], src[(size_t)p.CO], src[(size_t)2 * p.CO], src[(size_t)3 * p.CO]);
        b = make_float4(src[(size_t)4 * p.CO], src[(size_t)5 * p.CO], src[(size_t)6 * p.CO], src[(size_t)7 * p.CO]);
      } else {
        const float* src = p.w + ((size_t)((NT - 1 - tap) * p.CO + n0w + n)) * p.CS + 32 * kp + 8 * q;
        a = make_float4(src[0], src[1], src[2], src[3]);
        b = make_float4(src[4], src[5], src[6], src[7]);
      }
      u32x4 h, l;
      bk_split8(a, b, 1.f, ODIN_LO_SCALE, h, l);
      char* d = wl + tap * C5_TAPB + q * 512 + n * 16;
      *reinterpret_cast<u32x4*>(d) = h;
      *reinterpret_cast<u32x4*>(d + 2048) = l;
    }
    if (T0 < T1) stage(win);
    __syncthreads();

#pragma unroll 1
    for (int T = T0; T < T1; ++T) {
      const char* buf = win + ((T - T0) & 1) * C5_WIN;
      char* nbuf = win + (((T - T0) & 1) ^ 1) * C5_WIN;
      int b_n = 0, ty_n = 0, tx_n = 0;
      if (T + 1 < T1) {
        bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
        issue(b_n, ty_n, tx_n);
      }
      const unsigned tile_out = (unsigned)(((b_c * p.H + 8 * ty_c) * p.W + 8 * tx_c) * p.CO * 4);
      const bool ok = 8 * ty_c + ri < p.H && 8 * tx_c + cj < p.W;
      const unsigned voff = ok ? out_lane : ODIN_OOB_V;
      float4 ax = make_float4(0.f, 0.f, 0.f, 0.f), pv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (EPI == 2 && last) ax = odin_run_load4s(AUX, voff, tile_out);
      if (kp > 0) pv = odin_run_load4s(OUT, voff, tile_out);
      // (six accumulators: main, high x low, low x high for the even and the odd taps -- MFMAs into the same accumulator
      // six instructions apart: blk_common.h, bk_mfma16 note)
      f32x4 acc[2] = {bk_zero4(), bk_zero4()}, acx[2] = {bk_zero4(), bk_zero4()}, acy[2] = {bk_zero4(), bk_zero4()};
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        const int kh = tap / K, kw = tap - K * kh;
        const char* a = wlane + tap * C5_TAPB;
        const char* bq = buf + boff[kh & 1][kw] + kh * 1024;
        const u32x4 wh = *reinterpret_cast<const u32x4*>(a);
        const u32x4 wlo = *reinterpret_cast<const u32x4*>(a + 2048);
        const u32x4 xh = *reinterpret_cast<const u32x4*>(bq);
        const u32x4 xl = *reinterpret_cast<const u32x4*>(bq + C5_PLB);
        acx[tap & 1] = mfma16_f16(wh, xl, acx[tap & 1]);
        acc[tap & 1] = mfma16_f16(wh, xh, acc[tap & 1]);
        acy[tap & 1] = mfma16_f16(wlo, xh, acy[tap & 1]);
      }
      if (T + 1 < T1) stage(nbuf);
      float v[4];
      const float p4[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = fmaf((acx[0][r] + acx[1][r]) + (acy[0][r] + acy[1][r]), ODIN_LO_UNSCALE, acc[0][r] + acc[1][r]) + p4[r];
      if (last) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= out_s;
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
      }
      odin_run_store4s(OUT, voff, tile_out, make_float4(v[0], v[1], v[2], v[3]));
      b_c = b_n; ty_c = ty_n; tx_c = tx_n;
      __syncthreads();
    }
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
      p.colsum[(size_t)blockIdx.x * p.CO + n0w + tid] = tt;
    }
  }
}

template <int EPI, int ACT, int K>
int c5_launch_a(const C5Params& p, dim3 grid, void* stream) {
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv5_blk_kernel<EPI, ACT, K>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, c5_lds(K)) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((conv5_blk_kernel<EPI, ACT, K>), grid, dim3(512), (size_t)c5_lds(K), stream, p);
  return odin_check_launch(K == 5 ? "conv5_blk(f16x2)" : "conv4s1_blk(f16x2)");
}
template <int EPI, int K>
int c5_launch(const C5Params& p, dim3 grid, void* stream) {
  if (p.act == ODIN_ACT_ELU) return c5_launch_a<EPI, ODIN_ACT_ELU, K>(p, grid, stream);
  if (K == 5 && p.act == ODIN_ACT_RELU) return c5_launch_a<EPI, ODIN_ACT_RELU, 5>(p, grid, stream);
  return c5_launch_a<EPI, -1, K>(p, grid, stream);
}

int c5_tiles_per_wg(int n_tiles, int gy) {
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  return (n_tiles + cap - 1) / cap;
}


// =====================================================================================================================
// wgrad5_blk: dW[kh, kw, c, n] = sum over (b, y, x) of X[b, y + kh - 2, x + kw - 2, c] * DY[b, y, x, n] (+ column sums of
// DY for the bias).  The pixel is the reduction index: both operands are read TRANSPOSED from LDS (ds_read_b64_tr_b16;
// windows unswizzled -- no ds_read_b128 here).  Wave w owns taps w, w + 8, w + 16 (wave 0 also tap 24): 32 x 32 (c x n)
// accumulator pairs in registers over the whole tile walk, one slab row per workgroup.
// =====================================================================================================================
struct W5Params {
  const float* U;      // x  [B, H, W, CUt]
  const float* V;      // dy [B, H, W, CVt]
  float* slab;         // [gridDim.x][25 * CUt * CVt (+ CVt)]
  int B, H, W, CUt, CVt;
  int slab_stride, want_bias;
  int nty, ntx, n_tiles, tiles_per_wg;
  const unsigned* u_amax;   // optional range word of x
  const unsigned* v_amax;   // range word of dy
};

constexpr int W5_UPLB = 12 * 16 * 64;            // one plane of the x window
constexpr int W5_VPLB = 64 * 64;                 // one plane of the dy block
constexpr int W5_BUF = 2 * W5_UPLB + 2 * W5_VPLB;

template <int K>
__global__ __launch_bounds__(512) void wgrad5_blk_kernel(W5Params p) {
  constexpr int WS = 8 + K - 1, NT = K * K, C5_NIT = c5_nit(K);
  ODIN_DYN_SMEM(char, smem);
  __shared__ float bred[8 * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15, g = (lane >> 4) & 1;
  const int cv0 = blockIdx.y * 32, cu0 = blockIdx.z * 32;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;

  const OdinRangeReq u_rq = odin_range_issue(p.u_amax, lane), v_rq = odin_range_issue(p.v_amax, lane);
  const OdinRun RU = odin_run(p.U, (unsigned)((size_t)p.B * p.H * p.W * p.CUt * 4));
  const OdinRun RV = odin_run(p.V, (unsigned)((size_t)p.B * p.H * p.W * p.CVt * 4));
  int it_dst[C5_NIT], it_g[C5_NIT], it_wr[C5_NIT], it_wc[C5_NIT];
#pragma unroll
  for (int j = 0; j < C5_NIT; ++j) {
    const int e = tid + 512 * j;
    const int px = e >> 3, ch4 = e & 7;
    const int wr = odin_div_small(px, WS), wc = px - WS * wr;
    it_wr[j] = (e < WS * WS * 8) ? wr : (1 << 20);
    it_wc[j] = wc;
    it_dst[j] = wr * 1024 + wc * 64 + ch4 * 8;
    it_g[j] = ((wr * p.W + wc) * p.CUt + cu0 + 4 * ch4) * 4;
  }
  const int v_px = tid >> 3, v_ch4 = tid & 7;
  const int v_r = v_px >> 3, v_c = v_px & 7;
  const int v_dst = v_px * 64 + v_ch4 * 8;
  const int v_g = ((v_r * p.W + v_c) * p.CVt + cv0 + 4 * v_ch4) * 4;
  float4 itv[C5_NIT], vv;
  auto issue = [&](int b, int ty, int tx) {
    const int base = ((b * p.H + 8 * ty - (K - 1) / 2) * p.W + 8 * tx - (K - 1) / 2) * p.CUt * 4;
#pragma unroll
    for (int j = 0; j < C5_NIT; ++j) {
      const int gr = 8 * ty - (K - 1) / 2 + it_wr[j], gc = 8 * tx - (K - 1) / 2 + it_wc[j];
      const bool ok = gr >= 0 && gr < p.H && gc >= 0 && gc < p.W;
      itv[j] = odin_run_load4(RU, ok ? (unsigned)(base + it_g[j]) : ODIN_OOB);
    }
    const bool okv = 8 * ty + v_r < p.H && 8 * tx + v_c < p.W;
    vv = odin_run_load4(RV, okv ? (unsigned)(((b * p.H + 8 * ty) * p.W + 8 * tx) * p.CVt * 4 + v_g) : ODIN_OOB);
  };
  int b_c, ty_c, tx_c;
  bk_decode(T0, p.nty, p.ntx, b_c, ty_c, tx_c);
  if (T0 < T1) issue(b_c, ty_c, tx_c);
  const unsigned umb = odin_range_finish(u_rq), vmb = odin_range_finish(v_rq);
  const int gu = bk_shift(umb, 0), gv = bk_shift(vmb, 1);
  const float u_s = odin_pow2(gu), u_s2k = odin_pow2(gu + 11), v_s = odin_pow2(gv), v_s2k = odin_pow2(gv + 11);
  float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage = [&](char* buf) {
#pragma unroll
    for (int j = 0; j < C5_NIT; ++j) {
      if (tid + 512 * j < WS * WS * 8) {
        u32x2 h, l;
        odin_split_h4<true>(itv[j], u_s, u_s2k, h, l);
        *reinterpret_cast<u32x2*>(buf + it_dst[j]) = h;
        *reinterpret_cast<u32x2*>(buf + it_dst[j] + W5_UPLB) = l;
      }
    }
    u32x2 h, l;
    odin_split_h4<true>(vv, v_s, v_s2k, h, l);
    *reinterpret_cast<u32x2*>(buf + 2 * W5_UPLB + v_dst) = h;
    *reinterpret_cast<u32x2*>(buf + 2 * W5_UPLB + W5_VPLB + v_dst) = l;
    bsum4.x += vv.x; bsum4.y += vv.y; bsum4.z += vv.z; bsum4.w += vv.w;
  };
  if (T0 < T1) stage(smem);
  __syncthreads();

  // transposed reads: lane 4 q + p of a 16-lane group addresses pixel q, channel quad p of the group's 16 channels; a
  // k-step is two tile rows (half 0 / 1) of 8 pixels = two reads of 4 pixels
  const int tq = l16 >> 2, tp = l16 & 3;
  const int lane_u = (half * 16 + tq) * 64 + (16 * g + 4 * tp) * 2;       // + (2 s + kh) * 1024 + (kw + 4 m) * 64
  const int lane_v = (half * 8 + tq) * 64 + (16 * g + 4 * tp) * 2;        // + s * 1024 + m * 256
#ifdef ODIN_SIM
  const int sim_u = half * 1024 + 32 * g, sim_v = half * 512 + 32 * g;
#endif
  const int n_own = (NT - wave + 7) / 8;   // taps wave, wave + 8, ...: 25 taps -> 4 / 3 / 3 ..., 16 taps -> 2 each
  f32x16 acc[4], acx[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { acc[t] = f32x16_zero(); acx[t] = f32x16_zero(); }

#pragma unroll 1
  for (int T = T0; T < T1; ++T) {
    const char* buf = smem + ((T - T0) & 1) * W5_BUF;
    char* nbuf = smem + (((T - T0) & 1) ^ 1) * W5_BUF;
    int b_n = 0, ty_n = 0, tx_n = 0;
    if (T + 1 < T1) {
      bk_decode(T + 1, p.nty, p.ntx, b_n, ty_n, tx_n);
      issue(b_n, ty_n, tx_n);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u32x4 vb[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const char* vbase = buf + 2 * W5_UPLB + pl * W5_VPLB + s * 1024;
#ifdef ODIN_SIM
        const u32x2 lo = bk_tr(nullptr, vbase + sim_v, 64, l16), hi = bk_tr(nullptr, vbase + sim_v + 256, 64, l16);
#else
        const u32x2 lo = bk_tr(vbase + lane_v, nullptr, 0, 0), hi = bk_tr(vbase + lane_v + 256, nullptr, 0, 0);
#endif
        vb[pl][0] = lo[0]; vb[pl][1] = lo[1]; vb[pl][2] = hi[0]; vb[pl][3] = hi[1];
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t < n_own) {   // (wave-uniform)
          const int tap = wave + 8 * t, kh = tap / K, kw = tap - K * kh;
          u32x4 ua[2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            const char* ubase = buf + pl * W5_UPLB + (2 * s + kh) * 1024 + kw * 64;
#ifdef ODIN_SIM
            const u32x2 lo = bk_tr(nullptr, ubase + sim_u, 64, l16), hi = bk_tr(nullptr, ubase + sim_u + 256, 64, l16);
#else
            const u32x2 lo = bk_tr(ubase + lane_u, nullptr, 0, 0), hi = bk_tr(ubase + lane_u + 256, nullptr, 0, 0);
#endif
            ua[pl][0] = lo[0]; ua[pl][1] = lo[1]; ua[pl][2] = hi[0]; ua[pl][3] = hi[1];
          }
          acx[t] = mfma32_f16(ua[0], vb[1], acx[t]);
          acc[t] = mfma32_f16(ua[0], vb[0], acc[t]);
          acx[t] = mfma32_f16(ua[1], vb[0], acx[t]);
        }
      }
    }
    if (T + 1 < T1) stage(nbuf);
    b_c = b_n; ty_c = ty_n; tx_c = tx_n;
    __syncthreads();
  }

  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  const float ou = odin_pow2(-gu), ov = odin_pow2(-gv);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (t < n_own) {
      const int tap = wave + 8 * t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = fmaf(acx[t][r], ODIN_LO_UNSCALE, acc[t][r]);
        row[((size_t)tap * p.CUt + cu0 + cu) * p.CVt + cv0 + l31] = (v * ou) * ov;
      }
    }
  }
  if (p.want_bias && blockIdx.z == 0) {
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
      row[(size_t)NT * p.CUt * p.CVt + cv0 + tid] = t;
    }
  }
}

}  // namespace

// Conv2D(k, s1, SAME), k = 5 (pads 2, 2) or 4 (pads 1, 2), over CI in {32, 64} reduction channels: forward and data
// gradient (there CI = the layer's OUTPUT channels, CO its input channels)
bool odin_conv5_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center) {
  if (!odin_blk_enabled(2.0 * B * H * W * (double)(KH * KW) * CI * CO)) return false;
  if (!((KH == 5 || KH == 4) && KW == KH && S == 1 && pt == (KH - 1) / 2 && pl == (KW - 1) / 2 && !center &&
        (CI == 32 || CI == 64) && (CO % 32) == 0))
    return false;
  if (H < 1 || W < 1 || H > 4096 || W > 4096) return false;
  return (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * H * W * CO * 4 < 0x7FFF0000ull;
}

// epi 1: forward (bias + act); epi 2: data gradient (x act'(aux), column sums into colsum[rows][CO])
int odin_conv5_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int H, int W, int CI, int CO, int K, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream) {
  C5Params p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.H = H; p.W = W; p.CS = CI; p.CO = CO; p.act = act; p.nk = CI / 32; p.wmode = epi == 2 ? 1 : 0;
  p.pad = epi == 2 ? K - 1 - (K - 1) / 2 : (K - 1) / 2;
  p.nty = (H + 7) / 8; p.ntx = (W + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = CO / 32;
  p.tiles_per_wg = c5_tiles_per_wg(p.n_tiles, gy);
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  p.in_is_grad = epi == 2;
  if (epi == 2) {
    if (aux == nullptr) { p.act = ODIN_ACT_LINEAR; p.aux = out; }
    p.in_amax = odin_range_word_of(in, (size_t)B * H * W * CI, in_amax, stream);
    if (p.in_amax == nullptr) return odin_fail(-3, "conv5_blk: no range word for the gradient input");
  } else {
    p.in_amax = in_amax;
  }
  p.out_amax = out_amax;
  dim3 grid(gx, gy, 1);
  if (K == 5) return epi == 1 ? c5_launch<1, 5>(p, grid, stream) : c5_launch<2, 5>(p, grid, stream);
  return epi == 1 ? c5_launch<1, 4>(p, grid, stream) : c5_launch<2, 4>(p, grid, stream);
}

// weight gradient of a Conv2D(k, s1), k = 5 or 4: x [B, H, W, CI], dy [B, H, W, CO]
bool odin_wgrad5_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center) {
  if (!odin_blk_enabled(2.0 * B * H * W * (double)(KH * KW) * CI * CO)) return false;
  if (!((KH == 5 || KH == 4) && KW == KH && S == 1 && pt == (KH - 1) / 2 && pl == (KW - 1) / 2 && !center &&
        (CI % 32) == 0 && (CO % 32) == 0))
    return false;
  if (H < 1 || W < 1 || H > 4096 || W > 4096) return false;
  return (size_t)B * H * W * CI * 4 < 0x7FFF0000ull && (size_t)B * H * W * CO * 4 < 0x7FFF0000ull;
}

int odin_wgrad5_blk_launch(const float* x, const float* dy, float* slab, int* rows_out, int B, int H, int W, int CI,
                           int CO, int K, int want_bias, const uint32_t* g_amax, const uint32_t* a_amax, void* stream) {
  W5Params p;
  memset(&p, 0, sizeof(p));
  p.U = x; p.V = dy; p.slab = slab;
  p.B = B; p.H = H; p.W = W; p.CUt = CI; p.CVt = CO; p.want_bias = want_bias;
  p.slab_stride = K * K * CI * CO + (want_bias ? CO : 0);
  p.nty = (H + 7) / 8; p.ntx = (W + 7) / 8;
  p.n_tiles = B * p.nty * p.ntx;
  const int gy = CO / 32, gz = CI / 32;
  int cap = odin_num_cus() / (gy * gz);
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_SLAB_BLOCKS) cap = ODIN_MAX_SLAB_BLOCKS;
  p.tiles_per_wg = (p.n_tiles + cap - 1) / cap;
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (slab == nullptr) return 0;  // dry run
  p.v_amax = odin_range_word_of(dy, (size_t)B * H * W * CO, g_amax, stream);
  if (p.v_amax == nullptr) return odin_fail(-3, "wgrad5_blk: no range word for dy");
  p.u_amax = a_amax;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad5_blk_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * W5_BUF) != hipSuccess)
      (void)hipGetLastError();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad5_blk_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * W5_BUF) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  dim3 grid(gx, gy, gz);
  if (K == 5) ODIN_LAUNCH((wgrad5_blk_kernel<5>), grid, dim3(512), (size_t)2 * W5_BUF, stream, p);
  else ODIN_LAUNCH((wgrad5_blk_kernel<4>), grid, dim3(512), (size_t)2 * W5_BUF, stream, p);
  return odin_check_launch(K == 5 ? "wgrad5_blk(f16x2)" : "wgrad4s1_blk(f16x2)");
}
