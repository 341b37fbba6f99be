// igemm.hip -- the small-spatial layers of the image stacks (encoder3, decoder1 of
// odin/networks/image_networks.py:462-505; CelebA's 8x8 layers, :678-703; the 6x5 layers of the speech
// stack) as implicit GEMMs on the fp32 matrix cores with BOTH operands straight from L2.
//
// These layers are 0.07-0.5 GFLOP at batch 256: on the tiled paths (gather_conv.hip / wgrad.hip) they
// are <= 128 workgroups that first stage a 128 KB weight slice through LDS and then multiply -- 16-28 us
// per launch for 3 % of the step's FLOPs.  Here nothing is staged: a workgroup owns ONE 32 x 32 output
// tile, its NW waves split the reduction, and every wave streams its share of both operands from L2 in
// 4-deep, double-buffered batches of 16-byte loads (the next batch is in flight while the current one
// feeds v_mfma_f32_32x32x2_f32); the partial tiles meet in LDS in wave order (bit-reproducible).
//
//   forward / data gradient (igemm_kernel):  C[m][j] = sum_k A(m, k) * Wt(k, j)
//     rows m   = output pixels; for the transposed gathers (Conv2DTranspose forward, Conv2D data gradient)
//                ordered by stride class (oy % S, ox % S) so that a tile's rows share their valid taps and
//                the reduction runs over those only (4 of the 16 taps of a 4x4/s2 kernel)
//     k        = (tap, channel): 8 consecutive channels of one tap per k-group, the gather address is one
//                per-lane base + one wave-uniform tap offset, SAME padding is a per-lane tap bit mask
//   weight gradient (igemm_wgrad_kernel):    dW[(tap, cu)][cv] = sum_m U(pix(m, tap), cu) * V(m, cv)
//     U = the fine tensor (Conv2D: input, Conv2DTranspose: output gradient), V = the coarse one; the
//     reduction over the pixels m is split over gridDim.z workgroups (one slab row each) and their waves
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

#ifdef ODIN_SIM
#define IG_UNIFORM(x) (x)
#else
#define IG_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// k-groups (8 k-values each) per batch.  8 until round 4: two double-buffered batches of 8 groups are 128 operand
// registers (164-168 in all: three waves per SIMD); with 4 the kernels fit five and the launches of 1000-1500
// workgroups hide their L2 round trips behind other waves instead -- same-box A/B (two rounds, -DODIN_IG_U builds):
// dSprites step 0.528 -> 0.522 ms, speech 1.550 -> 1.520, FactorVAE 0.860 -> 0.851, Shapes3D 0.574 -> 0.572; 2 groups:
// speech / FactorVAE slower again (1.537 / 0.857), CelebA slightly faster (1.311 vs 1.324)
#ifndef ODIN_IG_U
#define ODIN_IG_U 4
#endif
constexpr int IG_U = ODIN_IG_U;

struct IGParams {
  const float* in;    // gathered tensor [B, H, W, CI]
  const float* w;
  float* out;         // [B, OH, OW, CO]
  const float* bias;  // forward
  const float* aux;   // data gradient: multiply by act'(aux), aux shaped like out
  float* colsum;      // data gradient: slab [gridDim.y][CO] of column sums (may be null)
  unsigned* out_amax; // data gradient: range word of `out` (odin_device.h; may be null)
  int B, H, W, CI, OH, OW, CO, KH, KW, S, pt, pl;
  int gpt;            // k-groups per tap = CI / 8
  unsigned mg_gpt;    // ceil(2^32 / gpt) (0: gpt == 1)
  int act, aux_act;
  int Mc;             // rows per stride class
  int tpc;            // 32-row tiles per stride class
  unsigned mg_tpc, mg_img, mg_row;  // ceil(2^32 / d) for d = tpc, (OH/S)(OW/S), OW/S (0: d == 1)
  unsigned dbg_a, dbg_b;  // diagnostics (ODIN_IG_DBG bit 0 / 1): all A / B loads out of range (no traffic)
  long long* stamps;      // diagnostics build: clock stamps of workgroup (0, 0), wave 0
};

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)
#define IG_STAMP(k) ((void)0)
#define IG_WSTAMP(k) ((void)0)
#else
#define IG_STAMP(k)                                                                        \
  do {                                                                                     \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)             \
      p.stamps[k] = (long long)clock64();                                                  \
  } while (0)
#define IG_WSTAMP(k)                                                                       \
  do {                                                                                     \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)             \
      p.stamps[k] = (long long)wall_clock64();                                             \
  } while (0)
#endif

// q / d through the host-computed magic = ceil(2^32 / d): exact for q * d < 2^32 (launcher checks)
__device__ __forceinline__ int ig_magicdiv(int q, unsigned magic) {
  return magic == 0u ? q : (int)__umulhi((unsigned)q, magic);
}
// t / d for 0 <= t < 64, 1 <= d <= 8 on the scalar unit
__device__ __forceinline__ int ig_smalldiv(int t, int d) { return (t * (256 / d + 1)) >> 8; }

// (bx, by): the tile's block coordinates -- blockIdx of a launch of its own, decoded from a linear index in the
// paired launch below; gx = blocks along x
template <int NW, bool TMODE, bool BKC>
__device__ __forceinline__ void igemm_body(const IGParams& p, const int bx, const int by, const int gx) {
  __shared__ float red[NW > 1 ? (NW - 1) * 16 * 64 : 64];
  __shared__ int rowoff[32];
  const int tid = threadIdx.x, lane = tid & 63;
  IG_STAMP(0);
  IG_WSTAMP(8);
  const int wave = IG_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // ---- this tile's stride class and its taps ----
  const int cls = TMODE ? ig_magicdiv(by, p.mg_tpc) : 0;
  const int tile = TMODE ? by - cls * p.tpc : by;
  const int SS = TMODE ? p.S : 1;     // 1 or 2
  const int ssh = SS >> 1;            // x / SS = x >> ssh,  x % SS = x & (SS - 1)
  const int cy = cls >> ssh, cx = cls - (cy << ssh);
  const int kh0 = TMODE ? (cy + p.pt) & (SS - 1) : 0, kw0 = TMODE ? (cx + p.pl) & (SS - 1) : 0;
  const int nkh = (p.KH - kh0 + SS - 1) >> ssh, nkw = (p.KW - kw0 + SS - 1) >> ssh;
  const int ntap = nkh * nkw;
  // ---- this lane's A row: one output pixel ----
  const int q = tile * 32 + l31;
  const bool a_ok = q < p.Mc;
  const int ohs = p.OH >> ssh, ows = p.OW >> ssh;
  const int b = ig_magicdiv(q, p.mg_img), r = q - b * (ohs * ows);
  const int ys = ig_magicdiv(r, p.mg_row), xs = r - ys * ows;
  const int oy = ys * SS + cy, ox = xs * SS + cx;
  // gathered pixel of tap (a, c): (Y0 + sg a, X0 + sg c)
  const int sg = TMODE ? -1 : 1;
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
  const int lanebase = ((b * p.H + Y0) * p.W + X0) * p.CI + 4 * h;  // floats
  if (h == 0) rowoff[l31] = a_ok ? ((b * p.OH + oy) * p.OW + ox) : -1;
  const int j = bx * 32 + l31;
  const bool b_ok = j < p.CO;
  const unsigned bl = b_ok ? (unsigned)((BKC ? j * p.CI + 4 * h : 4 * h * p.CO + j) * 4) : ODIN_OOB_V;
  const OdinRun RA = odin_run(p.in, (unsigned)((size_t)p.B * p.H * p.W * p.CI * 4));
  const OdinRun RB = odin_run(p.w, (unsigned)((size_t)p.KH * p.KW * p.CI * p.CO * 4));
  const int gpt = p.gpt;                // k-groups (8 channels) per tap
  const int ngroups = ntap * gpt;
  f32x16 acc = f32x16_zero();
  float av0[IG_U][4], bv0[IG_U][4], av1[IG_U][4], bv1[IG_U][4];
  // (the epilogue's bias is fetched now; its store offsets and act'(aux) factors after the main loop, see there)
  float auxv[16];
  unsigned ooff[16];  // byte offset of (row, column j) in `out`; out of range for rows / columns beyond the tensor
  __syncthreads();    // rowoff
  const float bj = (p.bias != nullptr && b_ok) ? p.bias[j] : 0.f;

  auto load_group = [&](int gr, float (&av)[4], float (&bv)[4]) {
    // branch-free (a wave-uniform `cond ? a : b` becomes a scalar branch that cuts the batch of loads in
    // two): groups beyond the reduction are clamped to the last one and their lanes read out of range
    const unsigned live = (unsigned)(gr - ngroups) >> 31;   // 1 / 0   (gr is wave-uniform)
    const unsigned dead = live - 1u;                          // 0 / 0xFFFFFFFF
    const int g = gr < ngroups ? gr : ngroups - 1;
    const int t = ig_magicdiv(g, p.mg_gpt);     // tap index within the stride class
    const int ci0 = (g - t * gpt) << 3;
    const int a = ig_smalldiv(t, nkw), c = t - a * nkw;
    const int tapoff = sg * (a * p.W + c) * p.CI;
    const unsigned va = live & (mask >> t) & 1u;
    const float4 x = odin_run_load4(RA, (unsigned)((lanebase + tapoff + ci0) * 4) | (va - 1u) | p.dbg_a);
    av[0] = x.x; av[1] = x.y; av[2] = x.z; av[3] = x.w;
    const int wt = (kh0 + a * SS) * p.KW + kw0 + c * SS;  // weight tap
    if constexpr (BKC) {
      const unsigned so = (unsigned)(wt * p.CO * p.CI + ci0) * 4u;
      const float4 y = odin_run_load4s(RB, bl | dead | p.dbg_b, so);
      bv[0] = y.x; bv[1] = y.y; bv[2] = y.z; bv[3] = y.w;
    } else {
      const unsigned so = (unsigned)((wt * p.CI + ci0) * p.CO) * 4u;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        bv[e] = odin_run_load1s(RB, bl | dead | p.dbg_b, so + (unsigned)(e * p.CO * 4));
    }
  };
  // the MFMAs of the current batch with the loads of the next one BETWEEN them: a wave issues in order and
  // its dependent MFMA chain occupies the issue slot for 64 cycles per instruction -- loads placed behind the
  // chain would only start once it has drained (measured: 4 k cycles per batch instead of 2 k).  The fences
  // pin the interleaving.
  auto mul_load = [&](const float (&av)[IG_U][4], const float (&bv)[IG_U][4], int gnext,
                      float (&nav)[IG_U][4], float (&nbv)[IG_U][4]) {
#pragma unroll
    for (int u = 0; u < IG_U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = mfma32(av[u][e], bv[u][e], acc);
      ODIN_SCHED_FENCE();
      load_group(gnext + u, nav[u], nbv[u]);
      ODIN_SCHED_FENCE();
    }
  };

  // ---- this wave's batches: wave, wave + NW, ... ----
  const int stride = NW * IG_U;
  int g = wave * IG_U;
  IG_STAMP(1);
  if (g < ngroups) {
#pragma unroll
    for (int u = 0; u < IG_U; ++u) load_group(g + u, av0[u], bv0[u]);
    IG_STAMP(2);
    for (;;) {
      mul_load(av0, bv0, g + stride, av1, bv1);  // (beyond the reduction: zeros, no memory traffic)
      g += 2 * stride;
      if (g - stride >= ngroups) break;
      mul_load(av1, bv1, g, av0, bv0);
      if (g >= ngroups) break;
    }
  }
  IG_STAMP(3);
  // ---- combine the NW partial tiles in wave order ----
  if (NW > 1) {
    if (wave > 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) red[((wave - 1) * 16 + rr) * 64 + lane] = acc[rr];
    }
  }
  __syncthreads();
  IG_STAMP(4);
  if (wave != 0) return;
  // the epilogue's operands (store offsets, act'(aux) factors): fetched HERE by wave 0.  Until round 4 every wave
  // fetched them before the main loop (one exposed cold-L2 round trip less in a workgroup's life, measured then as
  // 2 us per launch) -- at the price of 32 registers through the loop; with the 4-group batches above the kernels are
  // at 76 + 32 registers = four waves per SIMD without them, and other workgroups cover the round trip: same-box A/B
  // dSprites 0.525 -> 0.517 ms, Shapes3D 0.574 -> 0.567
  {
    const OdinRun RX = odin_run(p.aux != nullptr ? p.aux : p.in,
                                p.aux != nullptr ? (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4) : 0u);
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int po = rowoff[(rr & 3) + 8 * (rr >> 2) + 4 * h];
      const unsigned ok = (unsigned)b_ok & (((unsigned)po >> 31) ^ 1u);
      ooff[rr] = (unsigned)((po * p.CO + j) * 4) | (ok - 1u);
      auxv[rr] = odin_run_load1(RX, ooff[rr]);
    }
  }
  if (NW > 1) {
    for (int w = 1; w < NW; ++w) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[rr] += red[((w - 1) * 16 + rr) * 64 + lane];
    }
  }
  IG_STAMP(5);
  // ---- epilogue: lane holds column j, rows (rr & 3) + 8 (rr >> 2) + 4 h ----
  float cs = 0.f, amx = 0.f;
  const OdinRun RO = odin_run(p.out, (unsigned)((size_t)p.B * p.OH * p.OW * p.CO * 4));
  const bool has_aux = p.aux != nullptr;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    float v = odin_act(p.act, acc[rr] + bj);
    if (has_aux) v *= odin_act_grad(p.aux_act, auxv[rr]);
    odin_run_store1(RO, ooff[rr], v);           // range-checked: nothing is written for masked rows
    cs += ((int)ooff[rr] >= 0) ? v : 0.f;
    amx = fmaxf(amx, ((int)ooff[rr] >= 0 && b_ok) ? fabsf(v) : 0.f);
  }
  odin_amax_commit_wave(p.out_amax, amx, lane, (unsigned)(bx + gx * by));  // (wave 0 speaks for the tile)
  if (p.colsum != nullptr) {
    cs += __shfl_xor(cs, 32);
    if (h == 0 && b_ok) p.colsum[(size_t)by * p.CO + j] = cs;
  }
  IG_STAMP(6);
  IG_WSTAMP(9);
}

template <int NW, bool TMODE, bool BKC>
__global__ __launch_bounds__(NW * 64) void igemm_kernel(IGParams p) {
  igemm_body<NW, TMODE, BKC>(p, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x);
}

// --------------------------------------------------------------------------------------------------
struct IWParams {
  const float* u;   // fine tensor   [B, FH, FW, CU]
  const float* v;   // coarse tensor [B, h, w, CV]
  float* slab;      // [gridDim.z][slab_stride]: (dW [KH*KW*CU][CV] | column sums of V [CV])
  int slab_stride;
  int B, FH, FW, CU, h, w, CV, KH, KW, S, pt, pl;
  unsigned mg_cu;   // ceil(2^32 / CU)
  int M;            // B * h * w
  int chunk;        // pixels per workgroup (multiple of 8, <= IW_CHUNK)
  int want_bias;
};

constexpr int IW_CHUNK = 1024;

template <int NW>
__device__ __forceinline__ void igemm_wgrad_body(const IWParams& p, const int bx, const int by, const int bz) {
  __shared__ float red[NW > 1 ? (NW - 1) * 16 * 64 : 64];
  __shared__ int tb_base[IW_CHUNK + 8];   // float offset of the fine pixel (y S - pt, x S - pl) of coarse pixel m
  __shared__ int tb_yx[IW_CHUNK + 8];     // (y S - pt + 64) << 16 | (x S - pl + 64); -1: beyond this workgroup's pixels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = IG_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int I = p.KH * p.KW * p.CU;
  const int i = by * 32 + l31, j = bx * 32 + l31;
  const bool i_ok = i < I, j_ok = j < p.CV;
  const int tap = ig_magicdiv(i, p.mg_cu), cu = i - tap * p.CU;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int rowc = (kh * p.FW + kw) * p.CU + cu;
  const int mlo = bz * p.chunk;
  const int mhi = (mlo + p.chunk < p.M) ? mlo + p.chunk : p.M;
  for (int e = tid; e < IW_CHUNK + 8; e += NW * 64) {
    const int m = mlo + e;
    if (m < mhi && e < p.chunk) {
      const int b = m / (p.h * p.w), r = m - b * (p.h * p.w), y = r / p.w, x = r - y * p.w;
      const int fy = y * p.S - p.pt, fx = x * p.S - p.pl;
      tb_base[e] = ((b * p.FH + fy) * p.FW + fx) * p.CU;
      tb_yx[e] = ((fy + 64) << 16) | (fx + 64);
    } else {
      tb_base[e] = 0;
      tb_yx[e] = -1;
    }
  }
  __syncthreads();
  const OdinRun RU = odin_run(p.u, (unsigned)((size_t)p.B * p.FH * p.FW * p.CU * 4));
  const OdinRun RV = odin_run(p.v, (unsigned)((size_t)p.M * p.CV * 4));
  const int ngroups = (mhi - mlo + 7) >> 3;
  f32x16 acc = f32x16_zero();
  float csum = 0.f;
  float av0[IG_U][4], bv0[IG_U][4], av1[IG_U][4], bv1[IG_U][4];

  auto load_group = [&](int gr, float (&av)[4], float (&bv)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // pixel of this lane's half within the chunk (branch-free: entries [chunk, IW_CHUNK + 8) are -1)
      const int mr = gr * 8 + 4 * h + e;
      const int ml = mr < IW_CHUNK ? mr : IW_CHUNK;
      const int base = tb_base[ml];
      const int yx = tb_yx[ml];
      const int fy = (yx >> 16) - 64 + kh, fx = (yx & 0xFFFF) - 64 + kw;
      const unsigned inb = ((unsigned)yx >> 31) ^ 1u;   // 1: a pixel of this workgroup
      const unsigned va = (unsigned)i_ok & inb & ((unsigned)fy < (unsigned)p.FH) & ((unsigned)fx < (unsigned)p.FW);
      av[e] = odin_run_load1(RU, (unsigned)((base + rowc) * 4) | (va - 1u));
      const unsigned vb = (unsigned)j_ok & inb;
      bv[e] = odin_run_load1(RV, (unsigned)(((mlo + ml) * p.CV + j) * 4) | (vb - 1u));
    }
  };
  // (MFMAs of the current batch with the loads of the next between them, as in igemm_kernel)
  auto mul_load = [&](const float (&av)[IG_U][4], const float (&bv)[IG_U][4], int gnext,
                      float (&nav)[IG_U][4], float (&nbv)[IG_U][4]) {
#pragma unroll
    for (int u = 0; u < IG_U; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc = mfma32(av[u][e], bv[u][e], acc);
        csum += bv[u][e];
      }
      ODIN_SCHED_FENCE();
      load_group(gnext + u, nav[u], nbv[u]);
      ODIN_SCHED_FENCE();
    }
  };

  const int stride = NW * IG_U;
  int g = wave * IG_U;
  if (g < ngroups) {
#pragma unroll
    for (int u = 0; u < IG_U; ++u) load_group(g + u, av0[u], bv0[u]);
    for (;;) {
      mul_load(av0, bv0, g + stride, av1, bv1);
      g += 2 * stride;
      if (g - stride >= ngroups) break;
      mul_load(av1, bv1, g, av0, bv0);
      if (g >= ngroups) break;
    }
  }
  if (NW > 1) {
    if (wave > 0) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) red[((wave - 1) * 16 + rr) * 64 + lane] = acc[rr];
    }
  }
  __syncthreads();
  if (NW > 1 && wave == 0) {
    for (int w = 1; w < NW; ++w) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) acc[rr] += red[((w - 1) * 16 + rr) * 64 + lane];
    }
  }
  float* row = p.slab + (size_t)bz * p.slab_stride;
  if (p.want_bias && by == 0) {
    // column sums of V over this workgroup's pixels: halves h = 0 / 1 and the NW waves, fixed order
    __syncthreads();
    const float t = csum + __shfl_xor(csum, 32);
    if (h == 0) red[wave * 32 + l31] = t;
    __syncthreads();
    if (wave == 0 && h == 0 && j_ok) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += red[w * 32 + l31];
      row[(size_t)I * p.CV + j] = s;
    }
  }
  if (wave != 0 || !j_ok) return;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int ii = by * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
    if (ii < I) row[(size_t)ii * p.CV + j] = acc[rr];
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void igemm_wgrad_kernel(IWParams p) {
  igemm_wgrad_body<NW>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// A layer's data gradient and weight gradient in ONE launch: both read dy, neither reads the other's result, and each
// alone is a latency-bound launch of a few hundred single- to four-wave workgroups (12-18 us for 0.07-0.5 GFLOP:
// DESIGN 3.8).  Workgroups [0, na) run the data gradient, the rest the weight gradient; waves beyond a role's count
// leave at once (a finished wave does not take part in s_barrier).
template <int NWA, bool TMODE, bool BKC, int NWB>
__global__ __launch_bounds__((NWA > NWB ? NWA : NWB) * 64) void igemm_pair_kernel(IGParams a, IWParams b, int na,
                                                                                 int gax, int gbx, int gby) {
  const int id = (int)blockIdx.x;
  if (id < na) {
    if (NWA < NWB && (int)threadIdx.x >= NWA * 64) return;
    const int by = id / gax;
    igemm_body<NWA, TMODE, BKC>(a, id - by * gax, by, gax);
  } else {
    if (NWB < NWA && (int)threadIdx.x >= NWB * 64) return;
    const int r = id - na, bz = r / (gbx * gby), q = r - bz * (gbx * gby), by = q / gbx;
    igemm_wgrad_body<NWB>(b, q - by * gbx, by, bz);
  }
}

// TWO weight gradients in one launch (round 6): the neck's backward pass (neck.hip) leaves conv3's and the projection's
// weight gradients -- reductions over the whole batch, 14.6 and 7.3 us as launches of their own -- with nothing between
// them; neither reads the other's result.  Workgroups [0, na) run the first, the rest the second.
template <int NWA, int NWB>
__global__ __launch_bounds__((NWA > NWB ? NWA : NWB) * 64) void igemm_wpair_kernel(IWParams a, IWParams b, int na, int gax,
                                                                                  int gay, int gbx, int gby) {
  const int id = (int)blockIdx.x;
  if (id < na) {
    if (NWA < NWB && (int)threadIdx.x >= NWA * 64) return;
    const int bz = id / (gax * gay), q = id - bz * (gax * gay), by = q / gax;
    igemm_wgrad_body<NWA>(a, q - by * gax, by, bz);
  } else {
    if (NWB < NWA && (int)threadIdx.x >= NWB * 64) return;
    const int r = id - na, bz = r / (gbx * gby), q = r - bz * (gbx * gby), by = q / gbx;
    igemm_wgrad_body<NWB>(b, q - by * gbx, by, bz);
  }
}

// (read on every call: the A/B tests of the other paths switch it at run time; graph replays never get here)
bool igemm_enabled() { return ODIN_DIAG_ENV("ODIN_NOIGEMM") == nullptr; }
// Largest layer routed here, in FLOP (ODIN_IG_MAXGF overrides, GFLOP).  Measured against the tiled paths
// (CelebA B=512, speech B=256; profiles/r03_igemm_cap.txt): strided gathers, stride-1 layers and Dense layers win up
// to ~5 GFLOP (Dense 4096 -> 512: 114 / 66 / 96 -> 32 / 38 / 52 us; Conv2D 64 -> 64 k4 s1 on 8x8: 94 / 99 -> 52 / 66 us),
// the transposed stride-2 gathers and the convolution weight gradients only while the layer is launch-bound.
double igemm_max_flop(bool wide) {
  const char* e = ODIN_DIAG_ENV("ODIN_IG_MAXGF");
  return e ? atof(e) * 1e9 : (wide ? 5.0e9 : 1.2e9);
}

// ---- deferred weight-gradient launch (odin_igemm_pair_begin / _end): the *_bwd entry points issue the weight
// gradient first; when it lands here it waits for the data gradient of the same layer and shares its launch ----
struct PendingW {
  bool defer = false, pending = false;
  int depth = 0;   // brackets nest (odin_wgrad_pair_begin around a *_bwd entry point): only the outermost one opens / flushes
  IWParams p;
  dim3 grid;
  int nw = 0;
  void* stream = nullptr;
};
thread_local PendingW g_pw;

int iw_launch_now(const IWParams& p, dim3 grid, int nw, void* stream) {
  if (nw >= 8) ODIN_LAUNCH((igemm_wgrad_kernel<8>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) ODIN_LAUNCH((igemm_wgrad_kernel<4>), grid, dim3(256), 0, stream, p);
  else if (nw == 2) ODIN_LAUNCH((igemm_wgrad_kernel<2>), grid, dim3(128), 0, stream, p);
  else ODIN_LAUNCH((igemm_wgrad_kernel<1>), grid, dim3(64), 0, stream, p);
  return odin_check_launch("igemm_wgrad");
}

template <int NWA>
int iw_pair_b(const IWParams& a, dim3 ga, const IWParams& b, dim3 gb, int nwb, void* stream) {
  const int na = (int)(ga.x * ga.y * ga.z), nb = (int)(gb.x * gb.y * gb.z);
  const dim3 grid((unsigned)(na + nb));
  if (nwb == 4) ODIN_LAUNCH((igemm_wpair_kernel<NWA, 4>), grid, dim3((NWA > 4 ? NWA : 4) * 64), 0, stream, a, b, na, (int)ga.x, (int)ga.y, (int)gb.x, (int)gb.y);
  else if (nwb == 2) ODIN_LAUNCH((igemm_wpair_kernel<NWA, 2>), grid, dim3((NWA > 2 ? NWA : 2) * 64), 0, stream, a, b, na, (int)ga.x, (int)ga.y, (int)gb.x, (int)gb.y);
  else ODIN_LAUNCH((igemm_wpair_kernel<NWA, 1>), grid, dim3(NWA * 64), 0, stream, a, b, na, (int)ga.x, (int)ga.y, (int)gb.x, (int)gb.y);
  return odin_check_launch("igemm_wgrad+igemm_wgrad");
}

template <int NWA, bool TMODE, bool BKC>
int ig_pair_b(const IGParams& a, dim3 ga, const IWParams& b, dim3 gb, int nwb, void* stream) {
  const int na = (int)(ga.x * ga.y), nb = (int)(gb.x * gb.y * gb.z);
  const dim3 grid((unsigned)(na + nb));
  if (nwb == 4) ODIN_LAUNCH((igemm_pair_kernel<NWA, TMODE, BKC, 4>), grid, dim3((NWA > 4 ? NWA : 4) * 64), 0, stream, a, b, na, (int)ga.x, (int)gb.x, (int)gb.y);
  else if (nwb == 2) ODIN_LAUNCH((igemm_pair_kernel<NWA, TMODE, BKC, 2>), grid, dim3((NWA > 2 ? NWA : 2) * 64), 0, stream, a, b, na, (int)ga.x, (int)gb.x, (int)gb.y);
  else ODIN_LAUNCH((igemm_pair_kernel<NWA, TMODE, BKC, 1>), grid, dim3(NWA * 64), 0, stream, a, b, na, (int)ga.x, (int)gb.x, (int)gb.y);
  return odin_check_launch("igemm+igemm_wgrad");
}

template <bool TMODE, bool BKC>
int ig_launch_t(IGParams& p, dim3 grid, int nw, void* stream) {
#ifndef ODIN_SIM  // (the simulator's barrier counts every thread of the block: the roles keep their own launches there)
  if (g_pw.pending && g_pw.stream == stream && nw <= 4 && g_pw.nw <= 4 && p.stamps == nullptr) {
    g_pw.pending = false;
    if (nw == 4) return ig_pair_b<4, TMODE, BKC>(p, grid, g_pw.p, g_pw.grid, g_pw.nw, stream);
    if (nw == 2) return ig_pair_b<2, TMODE, BKC>(p, grid, g_pw.p, g_pw.grid, g_pw.nw, stream);
    return ig_pair_b<1, TMODE, BKC>(p, grid, g_pw.p, g_pw.grid, g_pw.nw, stream);
  }
#endif
  if (nw >= 8) ODIN_LAUNCH((igemm_kernel<8, TMODE, BKC>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) ODIN_LAUNCH((igemm_kernel<4, TMODE, BKC>), grid, dim3(256), 0, stream, p);
  else if (nw == 2) ODIN_LAUNCH((igemm_kernel<2, TMODE, BKC>), grid, dim3(128), 0, stream, p);
  else ODIN_LAUNCH((igemm_kernel<1, TMODE, BKC>), grid, dim3(64), 0, stream, p);
  return odin_check_launch("igemm");
}

long long* g_ig_stamps = nullptr;

}  // namespace

void odin_igemm_set_stamps(void* buf) { g_ig_stamps = (long long*)buf; }

// The small-layer regime: a reduction of whole 8-channel groups (channel count divisible by 8), at most
// 25 taps, strides 1 / 2 (transposed gathers: output extents divisible by the stride), <= 1.2 GFLOP, and
// tensors below 2^29 elements (32-bit byte offsets).  `tmode`: transposed gather.  (H, W, CI) = gathered
// tensor, (OH, OW, CO) = produced tensor.
bool odin_igemm_applicable(int tmode, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                           int S, int center) {
  if (!igemm_enabled() || center) return false;
  if (CI < 8 || (CI & 7) != 0 || CI > 8192 || KH * KW > 25 || KH < 1 || KW < 1 || KH > 8 || KW > 8 || S < 1 || S > 2)
    return false;
  if (tmode && (OH % S || OW % S || KH < S || KW < S)) return false;
  const double flop = 2.0 * B * (tmode ? (double)H * W : (double)OH * OW) * KH * KW * CI * CO;
  if (flop > igemm_max_flop(!tmode || S == 1)) return false;
  if ((double)B * OH * OW * OH * OW >= 2e9) return false;  // exactness of the magic-number row decoding
  if ((long)B * H * W * CI >= (1L << 29) || (long)B * OH * OW * CO >= (1L << 29) ||
      (long)KH * KW * CI * CO >= (1L << 29))
    return false;
  return true;
}

// rows of column sums the data-gradient launch writes (= its 32-row tiles)
int odin_igemm_tiles(int tmode, int B, int OH, int OW, int S) {
  const int SS = tmode ? S : 1;
  const int Mc = B * (OH / SS) * (OW / SS);
  return SS * SS * ((Mc + 31) / 32);
}

int odin_igemm_launch(int tmode, const float* in, const float* w, const float* bias, const float* aux,
                      int aux_act, float* out, float* colsum, int B, int H, int W, int CI, int OH, int OW,
                      int CO, int KH, int KW, int S, int pt, int pl, int act, uint32_t* out_amax, void* stream) {
  IGParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.out = out; p.bias = bias; p.out_amax = out_amax;
  p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act; p.colsum = colsum;
  p.B = B; p.H = H; p.W = W; p.CI = CI; p.OH = OH; p.OW = OW; p.CO = CO;
  p.KH = KH; p.KW = KW; p.S = S; p.pt = pt; p.pl = pl; p.gpt = CI / 8; p.act = act;
  const int SS = tmode ? S : 1;
  p.Mc = B * (OH / SS) * (OW / SS);
  p.tpc = (p.Mc + 31) / 32;
  auto magic = [](long d) { return d <= 1 ? 0u : (unsigned)(((1L << 32) + d - 1) / d); };
  p.mg_tpc = magic(p.tpc); p.mg_img = magic((long)(OH / SS) * (OW / SS)); p.mg_row = magic(OW / SS);
  p.mg_gpt = magic(p.gpt);
  dim3 grid((CO + 31) / 32, SS * SS * p.tpc, 1);
  // waves per tile: enough k-groups per wave to amortise the launch, enough waves to fill the chip
  const int ngroups = (KH / SS) * (KW / SS) * CI / 8;
  const long tiles = (long)grid.x * grid.y;
  int nw = 1;
  // (4 waves at most: the 8-wave variant measured slower on every layer -- the serial 7-tile sum of wave 0)
  while (nw < 4 && tiles * nw < 4 * 256 && ngroups / (nw * 2) >= IG_U) nw *= 2;
  if (const char* e = ODIN_DIAG_ENV("ODIN_IG_NW")) nw = atoi(e);
  if (const char* e = ODIN_DIAG_ENV("ODIN_IG_DBG")) {
    const int f = atoi(e);
    p.dbg_a = (f & 1) ? 0xFFFFFFFFu : 0u;
    p.dbg_b = (f & 2) ? 0xFFFFFFFFu : 0u;
  }
  p.stamps = g_ig_stamps;
  // Conv2D forward / Conv2DTranspose data gradient: weights [tap][k][j]; the transposed gathers: [tap][j][k]
  if (tmode) return ig_launch_t<true, true>(p, grid, nw, stream);
  return ig_launch_t<false, false>(p, grid, nw, stream);
}

// weight gradient: fine tensor (FH, FW, CU) gathered around the pixels of the coarse one (h, w, CV)
bool odin_igemm_wgrad_applicable(int B, int FH, int FW, int CU, int h, int w, int CV, int KH, int KW, int S,
                                 int center) {
  if (!igemm_enabled() || center) return false;
  if (CU < 8 || (CU & 7) != 0 || CU > 8192 || KH * KW > 64 || KH < 1 || KW < 1 || S < 1 || S > 4) return false;
  if (FH > 8192 || FW > 8192 || (long)B * h * w > 65536) return false;
  if ((double)KH * KW * CU * CU >= 4e9) return false;  // exactness of the magic-number row decoding
  const double flop = 2.0 * B * h * w * KH * KW * CU * CV;
  if (flop > igemm_max_flop(h * w == 1 && FH * FW == 1)) return false;
  if ((long)B * FH * FW * CU >= (1L << 29) || (long)B * h * w * CV >= (1L << 29)) return false;
  return true;
}

// reduction splits (= slab rows) of the weight-gradient launch
int odin_igemm_wgrad_rows(int B, int h, int w, int KH, int KW, int CU, int CV) {
  const int M = B * h * w;
  const long tiles = (long)((KH * KW * CU + 31) / 32) * ((CV + 31) / 32);
  int R = (M + IW_CHUNK - 1) / IW_CHUNK;
  // enough workgroups to fill the chip, at least 64 pixels each
  while (tiles * R < 512 && M / (R * 2) >= 64 && R * 2 <= ODIN_MAX_SLAB_BLOCKS) R *= 2;
  if (const char* e = ODIN_DIAG_ENV("ODIN_IG_R")) {
    const int r = atoi(e);
    if (r >= 1 && r <= ODIN_MAX_SLAB_BLOCKS && (M + r - 1) / r <= IW_CHUNK) R = r;
  }
  return R;
}

int odin_igemm_wgrad_launch(const float* u, const float* v, float* slab, int slab_stride, int B, int FH,
                            int FW, int CU, int h, int w, int CV, int KH, int KW, int S, int pt, int pl,
                            int want_bias, void* stream) {
  IWParams p;
  memset(&p, 0, sizeof(p));
  p.u = u; p.v = v; p.slab = slab; p.slab_stride = slab_stride;
  p.B = B; p.FH = FH; p.FW = FW; p.CU = CU; p.h = h; p.w = w; p.CV = CV;
  p.KH = KH; p.KW = KW; p.S = S; p.pt = pt; p.pl = pl;
  p.mg_cu = CU <= 1 ? 0u : (unsigned)(((1L << 32) + CU - 1) / CU);
  p.M = B * h * w; p.want_bias = want_bias;
  const int R = odin_igemm_wgrad_rows(B, h, w, KH, KW, CU, CV);
  p.chunk = (((p.M + R - 1) / R) + 7) & ~7;
  if (p.chunk > IW_CHUNK) return odin_fail(-2, "igemm wgrad: reduction chunk too long");
  dim3 grid((CV + 31) / 32, (KH * KW * CU + 31) / 32, R);
  const int ngroups = p.chunk / 8;
  const long wgs = (long)grid.x * grid.y * grid.z;
  // measured (enc3 / dec1 weight gradients): 4 waves beat 1, 2 and 8 even with one batch per wave
  (void)wgs;
  int nw = ngroups >= 16 ? 4 : ngroups >= 8 ? 2 : 1;
  if (const char* e = ODIN_DIAG_ENV("ODIN_IG_NW")) nw = atoi(e);
  if (g_pw.defer && !g_pw.pending) {  // wait for the data gradient of the same layer (odin_igemm_pair_end flushes)
    g_pw.pending = true;
    g_pw.p = p; g_pw.grid = grid; g_pw.nw = nw; g_pw.stream = stream;
    return 0;
  }
#ifndef ODIN_SIM  // (the simulator's barrier counts every thread of the block: the roles keep their own launches there)
  if (g_pw.defer && g_pw.pending && g_pw.stream == stream && nw <= 4 && g_pw.nw <= 4) {
    // a second weight gradient behind a pending one (odin_wgrad_pair_begin / _end): both in one launch
    g_pw.pending = false;
    if (g_pw.nw == 4) return iw_pair_b<4>(g_pw.p, g_pw.grid, p, grid, nw, stream);
    if (g_pw.nw == 2) return iw_pair_b<2>(g_pw.p, g_pw.grid, p, grid, nw, stream);
    return iw_pair_b<1>(g_pw.p, g_pw.grid, p, grid, nw, stream);
  }
#endif
  return iw_launch_now(p, grid, nw, stream);
}

// C ABI: the caller announces that the weight-gradient calls up to odin_wgrad_pair_end are independent of one another
// (include/odin_hip.h); those that land on this kernel family share a launch two by two
extern "C" void odin_wgrad_pair_begin(void) { odin_igemm_pair_begin(); }
extern "C" int odin_wgrad_pair_end(void) { return odin_igemm_pair_end(); }

void odin_igemm_pair_begin() {
  if (g_pw.depth++ == 0) { g_pw.defer = true; g_pw.pending = false; }
}
int odin_igemm_pair_end() {
  if (g_pw.depth > 0 && --g_pw.depth > 0) return 0;
  g_pw.defer = false;
  if (!g_pw.pending) return 0;
  g_pw.pending = false;
  return iw_launch_now(g_pw.p, g_pw.grid, g_pw.nw, g_pw.stream);
}
