// tconv_ring.hip -- TRANSPOSED gather convolution 4x4 / stride 2 over 32 reduction channels with
// the rolling LDS row window / LDS-DMA producer / 16x16x4 fp32 MFMA consumer structure of
// fconv_ring.hip.
//
// Serves (TF `SAME`, pads (1, 1)):
//   Conv2DTranspose(k4, s2) forward from 32 channels          (image_networks.py:503-506, decoder4)
//   Conv2D(k4, s2) DATA GRADIENT into 32-channel inputs       (tape.gradient of encoder1)
//   the fused decoder tail: decoder4 -> Conv2D 1x1 (decoder6, C1 maps) -> Independent(Bernoulli)
//   log-prob AND its backward in the epilogue (image_networks.py:505-511, 87-93;
//   variational_autoencoder.py:528-530): the [B, 2H, 2W, 32] activation never reaches HBM.
// i.e.  out[b, oh, ow, n] = sum over (kh, kw) with (oh + 1 - kh), (ow + 1 - kw) even, c < 32 of
//       in[b, (oh + 1 - kh) / 2, (ow + 1 - kw) / 2, c] * W[kh, kw, n, c]
//
// An output pixel sees 4 of the 16 taps, selected by its (row, column) parity; 16 output pixels
// of one row and one column parity read 16 CONSECUTIVE input pixels per tap.  A wave owns a
// column parity (and a half row / a row parity), walks 4 such 16-pixel groups per tile (64 MFMAs
// each) and finishes each with its own epilogue.  Rows: [zero row][H rows] per image, slot =
// global padded row mod NSLOT; a tile of RP input rows (2 RP output rows) has RP + 2 rows live.
// Everything else (swizzled 128-byte pixel slots, producer walk, one barrier per tile) is as in
// fconv_ring.hip.  All arithmetic is fp32 (v_mfma_f32_16x16x4_f32): this replaces the bf16-split
// instances for these layers.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

// Since round 3 this all-fp32 kernel is the A/B reference of tconv_planes.hip only (ODIN_TRING=1): it is compiled in
// the diagnostics build (`make diag`) and absent from the product library.
#ifndef ODIN_DIAG
void odin_tconv_ring_set_stamps(void*) {}
bool odin_tconv_ring_applicable(int, int, int, int, int, int, int, int, int, int) { return false; }
int odin_tconv_ring_launch(const float*, const float*, const float*, const float*, float*, float*, int*, const float*,
                           const float*, const float*, float*, float*, int*, float*, const float*, int, int, int, int,
                           int, int, void*) {
  return odin_fail(-2, "tconv_ring: diagnostics build only");
}
#else

__device__ float odin_tr_zero_row[1024];  // 4 KB of zeros: DMA source of the SAME-padding rows

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifdef ODIN_SIM
#define TR_UNIFORM(x) (x)
#else
#define TR_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

__device__ __forceinline__ f32x4v tr_mfma16(float a, float b, f32x4v c) {
#ifdef ODIN_SIM
  return sim::mfma_16x16x4(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

struct TRParams {
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
  float* llk_part;     // [n_tiles][8 consumer waves]
  float* slab;         // [gridDim.x][CO * C1 + C1 + CO]
  const float* scale;  // device scalar 1/B
  int B, H, W, CO;
  int RP;              // input rows per tile (RP * W == 64)
  int NSLOT, RB;
  int tiles_per_img, n_tiles, tiles_per_wg;
  long long* stamps;   // diagnostics: s_memtime stamps of workgroup 0 (consumer wave 0: [0,32), producer wave 8: [32,64))
};

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define TR_STAMP(base, k) ((void)0)
#else
#define TR_STAMP(base, k)                                                                        \
  do {                                                                                           \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 31)  \
      p.stamps[(base) + stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

// 8 consumer waves (two per SIMD) + 4 producer waves.  Measured on gfx950 (tools/micro/mfma_valu.hip):
// a wave's own VALU instructions ADD to its MFMA stream (32 -> 45 ticks per MFMA with 4 FMAs in
// the gap), while a second wave on the same SIMD issues ~4 VALU instructions per MFMA of its
// partner at no cost to it.  So the epilogues are hidden by the SIMD's other consumer wave, not by
// instruction scheduling inside one wave.
constexpr int TR_NCW = 8;
// The ring holds TR_DEPTH tiles: no workgroup barrier in the tile loop.  Producers and consumers meet
// through two LDS counters (rows landed / tiles consumed), so the two consumer waves of a SIMD can
// drift into opposite phases (one in its MFMA stream, the other in its epilogue) and stay there; a
// per-tile s_barrier re-aligned them every tile and left the last epilogue of each tile exposed
// (tile period 11.0 k ticks at 8.2 k of MFMA work).
constexpr int TR_DEPTH = 4;

// counter >= target (workgroup-scope acquire); spins with s_sleep
__device__ __forceinline__ void tr_wait_ge(int* ctr, int target) {
#ifdef ODIN_SIM
  while (*reinterpret_cast<volatile int*>(ctr) < target) sim::yield();
#else
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target)
    __builtin_amdgcn_s_sleep(2);
#endif
}
// counter += 1 by one lane of the wave, after everything the wave issued has completed (release:
// vmcnt(0) covers its LDS-DMA pieces, lgkmcnt(0) its LDS reads)
__device__ __forceinline__ void tr_signal(int* ctr, int lane) {
#ifdef ODIN_SIM
  if (lane == 0) *ctr += 1;
#else
  if (lane == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
constexpr int TR_THREADS = (TR_NCW + 4) * 64;

struct TrYes { static constexpr bool value = true; };
struct TrNo { static constexpr bool value = false; };

constexpr int TR_WBYTES = 16 * 8 * 32 * 16;  // weight image [tap][piece][out channel][4]: 64 KB

// hardware log / reciprocal forms of softplus / sigmoid (as in gather_conv.hip's fused tail)
__device__ __forceinline__ float tr_softplus(float x) {
  return fmaxf(x, 0.f) + 0.6931471805599453f * odin_log2(1.f + odin_exp2(-1.4426950408889634f * fabsf(x)));
}
__device__ __forceinline__ float tr_sigmoid(float x) {
  const float e = odin_exp2(-1.4426950408889634f * fabsf(x));
  const float r = odin_rcp(1.f + e);
  return x >= 0.f ? r : e * r;
}

// EPI 1: bias + ELU; EPI 2: x ELU'(aux) + column sums; EPI 3: fused Bernoulli tail with C1 logit maps
template <int EPI, int C1>
__global__ __launch_bounds__(TR_THREADS) void tconv_ring_kernel(TRParams p) {
  ODIN_DYN_SMEM(char, smem);
  char* wl = smem;
  char* ring = smem + TR_WBYTES;
  constexpr int NRED = 32 * (1 + (EPI == 3 ? C1 : 0)) + 4;  // per-wave reduction row
  __shared__ float cred[TR_NCW * NRED];
  __shared__ int tr_sync[2];  // [0]: producer waves x tiles whose rows have landed; [1]: consumer waves x tiles consumed
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = TR_UNIFORM(tid >> 6);
  int stamp_i = 0;
  (void)stamp_i;
  const int n0 = blockIdx.y * 32;
  const int HP = p.H + 1;
  const int OH = 2 * p.H, OW = 2 * p.W;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;
  if (tid == 0) { tr_sync[0] = 0; tr_sync[1] = 0; }
  const int nlive = p.RP + 2;
  const int g00 = HP * (T0 / p.tiles_per_img) + p.RP * (T0 % p.tiles_per_img);

  if (wave >= TR_NCW) {
    // ---------------------------- producers: LDS-DMA (scalar row walk) ----------------------------
    const int pw = wave - TR_NCW;
    const int cpr = p.W / 8;  // 1 KB chunks per row: 4 (W 32) or 2 (W 16)
    // weights: piece (tap, c4, co) = 4 consecutive floats of W[tap][n0 + co][4 c4 ..]: 64 DMA instructions
    {
      const OdinRun WR = odin_run(p.w, (unsigned)((size_t)16 * p.CO * 32 * 4));
      for (int c = pw; c < 64; c += 4) {
        const int e = c * 64 + lane;
        const int co = e & 31, c4 = (e >> 5) & 7, tap = e >> 8;
        odin_run_dma16(WR, reinterpret_cast<float*>(wl + (size_t)c * 1024),
                       (unsigned)(((tap * p.CO + n0 + co) * 32 + 4 * c4) * 4), lane);  // CO % 32 == 0
      }
    }
    // this lane's 16 bytes of chunk pw of a row: slot pc = 1 + (q >> 3), piece at pos q & 7
    const int q = pw * 64 + lane;
    const int pc = 1 + (q >> 3), c4s = (q & 7) ^ ((pc >> 1) & 7);
    const unsigned gofs = (unsigned)((((pc - 1) * 32) + 4 * c4s) * 4);
    const int dofs = 128 + pw * 1024;
    const bool active = pw < cpr;
    const OdinRun ZR = odin_run(odin_tr_zero_row, (unsigned)sizeof(odin_tr_zero_row));
    int g_hi = TR_UNIFORM(g00);
    int gi = TR_UNIFORM(g_hi % HP), bimg = TR_UNIFORM(g_hi / HP), slot = TR_UNIFORM(g_hi % p.NSLOT);
    int g_next0 = g_hi, t_in_img = TR_UNIFORM(T0 % p.tiles_per_img);
    __syncthreads();  // counters initialised, SAME-padding slots zeroed
    for (int T = T0; T < T1; ++T) {
      {
        // the rows of tile T reuse the slots of tiles <= T - TR_DEPTH: all 8 consumer waves must be past them
        if (T - TR_DEPTH >= T0) tr_wait_ge(&tr_sync[1], TR_NCW * (T - TR_DEPTH - T0 + 1));
        if (pw == 0) TR_STAMP(32, 20);
        const int g_need = g_next0 + nlive;
        for (int g = g_hi; g < g_need; ++g) {
          char* rowl = ring + (size_t)slot * p.RB;
          if (active) {
            if (gi == 0) {
              odin_run_dma16(ZR, reinterpret_cast<float*>(rowl + dofs), (unsigned)(lane * 16), lane);
            } else {
              const OdinRun R = odin_run(p.in + (size_t)(bimg * p.H + gi - 1) * p.W * 32,
                                         (unsigned)(p.W * 32 * 4));
              odin_run_dma16(R, reinterpret_cast<float*>(rowl + dofs), gofs, lane);
            }
          }
          if (++gi == HP) { gi = 0; ++bimg; }
          if (++slot == p.NSLOT) slot = 0;
        }
        if (pw == 0) TR_STAMP(32, 21);
        g_hi = g_need;
        g_next0 += p.RP;
        if (++t_in_img == p.tiles_per_img) { t_in_img = 0; g_next0 += 1; }
      }
      odin_wait_vmem();
      tr_signal(&tr_sync[0], lane);  // this wave's share of tile T (and, the first time, of the weights) is in LDS
      if (pw == 0) TR_STAMP(32, 22);
    }
  } else {
    // ---------------------------- consumers ----------------------------
    // SAME-padding slots (pc = 0 and pc = W + 1) of every ring row: zero for ever
    for (int e = tid; e < p.NSLOT * 16; e += TR_NCW * 64) {
      const int sl = e >> 4, qq = e & 15;
      char* rowl = ring + (size_t)sl * p.RB + ((qq & 8) ? (p.W + 1) * 128 : 0);
      reinterpret_cast<float4*>(rowl)[qq & 7] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int l15 = lane & 15, kq = lane >> 4;
    // role (wave & 3): column parity x (half row | row parity); waves w and w + 4 (same SIMD) share a
    // role and take two of the tile's four groups each
    const int role = wave & 3, half = wave >> 2;
    const int cpw = role & 1;                          // column parity of this wave's pixels
    const int hx = (p.W == 32) ? (role >> 1) : 0;      // half row (W 32)
    const int i_in = 16 * hx + l15;                    // input-resolution column index i: ow = 2 i + cpw
    // column taps of this wave: parity 0 -> kw = 1 (padded column pc = i + 1), kw = 3 (pc = i);
    // parity 1 -> kw = 0 (pc = i + 2), kw = 2 (pc = i + 1)
    const int kw_a = cpw ? 0 : 1, kw_b = kw_a + 2;
    const int d_a = cpw ? 2 : 1, d_b = d_a - 1;
    // lane offsets inside a ring row of the two column taps, per channel group g
    int loA[2], loB[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int pa = i_in + d_a, pb = i_in + d_b;
      loA[g] = pa * 128 + (((4 * g + kq) ^ ((pa >> 1) & 7)) << 4);
      loB[g] = pb * 128 + (((4 * g + kq) ^ ((pb >> 1) & 7)) << 4);
    }
    const char* wlane = wl + ((kq * 32 + l15) << 4);
    float4 bias4[2];
    if (EPI == 1 || EPI == 3) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int n = n0 + cb * 16 + 4 * kq;
        bias4[cb] = (p.bias != nullptr && n + 3 < p.CO) ? *reinterpret_cast<const float4*>(p.bias + n)
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // tail constants: the 1x1 weights of this lane's 8 channels
    float w1r[(EPI == 3) ? 8 : 1][(EPI == 3) ? C1 : 1];
    float b1r[(EPI == 3) ? C1 : 1];
    float dw1[(EPI == 3) ? 8 : 1][(EPI == 3) ? C1 : 1];
    float db1[(EPI == 3) ? C1 : 1];
    if (EPI == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int n = n0 + (i >> 2) * 16 + 4 * kq + (i & 3);
#pragma unroll
        for (int oc = 0; oc < C1; ++oc) {
          w1r[i][oc] = n < p.CO ? p.w1[n * C1 + oc] : 0.f;
          dw1[i][oc] = 0.f;
        }
      }
#pragma unroll
      for (int oc = 0; oc < C1; ++oc) { b1r[oc] = p.b1[oc]; db1[oc] = 0.f; }
    }
    float csum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) csum[i] = 0.f;
    const float sc = (EPI == 3) ? p.scale[0] : 0.f;

    // ---- software pipeline over groups: the epilogue of group G - 1 (pure VALU + stores) is issued
    // inside the MFMA stream of group G -- between two 32-cycle MFMAs the wave has ~28 free issue
    // cycles -- and crosses the tile barrier (it does not touch the ring).
    constexpr int NV = (EPI == 1) ? 2 : (EPI == 2) ? 1 : (C1 == 1 ? 3 : 5);  // VALU slots per MFMA
    int b_cur = T0 / p.tiles_per_img, t_cur = T0 - b_cur * p.tiles_per_img;
    int sl0 = (HP * b_cur + p.RP * t_cur) % p.NSLOT;  // ring slot of the tile's first padded row
    const int nG = 2 * (T1 - T0);  // this wave's groups: mt = 2 half + (G & 1)
    f32x4v pa0 = {0.f, 0.f, 0.f, 0.f}, pa1 = {0.f, 0.f, 0.f, 0.f};
    size_t opixP = 0;
    float4 axP[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    float tgtP[(EPI == 3) ? C1 : 1] = {};
    float llk_lane = 0.f;
    // optional logits output: a null pointer becomes an empty range (every store dropped)
    const OdinRun LG = odin_run(EPI == 3 ? p.logits : nullptr,
                                (EPI == 3 && p.logits != nullptr)
                                    ? (unsigned)((size_t)p.B * OH * OW * C1 * 4) : 0u);

    // epilogue of one group: lane = pixel (oh, 2 i + cpw) x channels n0 + cb * 16 + 4 kq + 0..3
    auto epilogue = [&](const f32x4v& e0, const f32x4v& e1, size_t opix, const float4 (&ax)[2],
                        const float (&tgt)[(EPI == 3) ? C1 : 1]) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = e0[k]; v[4 + k] = e1[k]; }
      if (EPI == 1 || EPI == 3) {
        const float bb[8] = {bias4[0].x, bias4[0].y, bias4[0].z, bias4[0].w,
                             bias4[1].x, bias4[1].y, bias4[1].z, bias4[1].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float tt = v[k] + bb[k];
          v[k] = fmaxf(tt, 0.f) + (odin_exp2(fminf(tt, 0.f) * 1.44269504088896341f) - 1.f);
        }
      }
      if (EPI == 2) {
        const float aa[8] = {ax[0].x, ax[0].y, ax[0].z, ax[0].w, ax[1].x, ax[1].y, ax[1].z, ax[1].w};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          v[k] = fmaf(v[k], fminf(aa[k], 0.f), v[k]);
          csum[k] += v[k];
        }
      }
      if (EPI == 3) {
        // 1x1 conv over the pixel's 32 channels: 8 here, the rest in the lanes kq ^ 1, kq ^ 2
        float dl[C1];
#pragma unroll
        for (int oc = 0; oc < C1; ++oc) {
          float tt = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) tt = fmaf(v[k], w1r[k][oc], tt);
          const float l = odin_rowsum4(tt) + b1r[oc];
          const float x = tgt[oc];
          const float dsig = (tr_sigmoid(l) - x) * sc;
          // the kq == 0 lane of a pixel owns its scalar results (selects, not branches: the
          // epilogue must stay one basic block to be scheduled into the MFMA stream)
          llk_lane += kq == 0 ? x * l - tr_softplus(l) : 0.f;
          db1[oc] += kq == 0 ? dsig : 0.f;
          odin_run_store1(LG, kq == 0 ? (unsigned)((opix * C1 + oc) * 4) : ODIN_OOB, l);
          dl[oc] = dsig;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float gsum = 0.f;
#pragma unroll
          for (int oc = 0; oc < C1; ++oc) {
            gsum = fmaf(w1r[k][oc], dl[oc], gsum);
            dw1[k][oc] = fmaf(v[k], dl[oc], dw1[k][oc]);
          }
          v[k] = fmaf(gsum, fminf(v[k], 0.f), gsum);  // x ELU'(y) = 1 + min(y, 0)
          csum[k] += v[k];
        }
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        *reinterpret_cast<float4*>(p.out + opix * p.CO + n0 + cb * 16 + 4 * kq) =
            make_float4(v[4 * cb], v[4 * cb + 1], v[4 * cb + 2], v[4 * cb + 3]);
    };

    // MFMAs of group G (tile (b_cur, t_cur), sub-group mt = G & 3), with the epilogue of group G - 1
    auto run_group = [&](auto with_epi, int mt) {
      constexpr bool WE = decltype(with_epi)::value;
      // input row pair rp, output row parity rpar
      const int rp = (p.W == 32) ? (mt >> 1) : mt;
      const int rpar = (p.W == 32) ? (mt & 1) : (role >> 1);
      const int oh = 2 * (p.RP * t_cur + rp) + rpar;
      // row taps: parity 0 -> kh = 1 (padded row rp + 1), kh = 3 (rp); parity 1 -> kh = 0 (rp + 2), kh = 2 (rp + 1)
      const int kh_a = rpar ? 0 : 1, kh_b = kh_a + 2;
      int sa = sl0 + rp + (rpar ? 2 : 1), sb = sa - 1;
      if (sa >= p.NSLOT) sa -= p.NSLOT;
      if (sb >= p.NSLOT) sb -= p.NSLOT;
      const char* row_a = ring + (size_t)sa * p.RB;
      const char* row_b = ring + (size_t)sb * p.RB;
      const size_t opix = ((size_t)b_cur * OH + oh) * OW + 2 * i_in + cpw;
      // this group's epilogue operands: consumed one group later
      float4 axN[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      float tgtN[(EPI == 3) ? C1 : 1] = {};
      if (EPI == 2) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
          axN[cb] = *reinterpret_cast<const float4*>(p.aux + opix * p.CO + n0 + cb * 16 + 4 * kq);
      }
      if (EPI == 3) {
#pragma unroll
        for (int oc = 0; oc < C1; ++oc) tgtN[oc] = p.target[opix * C1 + oc];
      }
      f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      float4 bq[2], a0[2], a1[2];
      // 8 steps = 4 taps (a/a, a/b, b/a, b/b) x 2 channel groups
      auto loads = [&](int s, float4& vb, float4& va0, float4& va1) {
        const int tp = s >> 1, g = s & 1;
        const bool ra = tp < 2, ca = (tp & 1) == 0;
        const char* rowp = ra ? row_a : row_b;
        vb = *reinterpret_cast<const float4*>(rowp + (ca ? loA[g] : loB[g]));
        const int tap = (ra ? kh_a : kh_b) * 4 + (ca ? kw_a : kw_b);
        const char* wp = wlane + ((tap * 8 + 4 * g) * 32 << 4);
        va0 = *reinterpret_cast<const float4*>(wp);
        va1 = *reinterpret_cast<const float4*>(wp + 256);
      };
      loads(0, bq[0], a0[0], a1[0]);
      ODIN_SCHED_FENCE();
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < 8) loads(s + 1, bq[nxt], a0[nxt], a1[nxt]);
        acc0 = tr_mfma16(a0[cur].x, bq[cur].x, acc0);
        acc1 = tr_mfma16(a1[cur].x, bq[cur].x, acc1);
        acc0 = tr_mfma16(a0[cur].y, bq[cur].y, acc0);
        acc1 = tr_mfma16(a1[cur].y, bq[cur].y, acc1);
        acc0 = tr_mfma16(a0[cur].z, bq[cur].z, acc0);
        acc1 = tr_mfma16(a1[cur].z, bq[cur].z, acc1);
        acc0 = tr_mfma16(a0[cur].w, bq[cur].w, acc0);
        acc1 = tr_mfma16(a1[cur].w, bq[cur].w, acc1);
        if (!WE) {
          if (s + 1 < 8) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
              ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
            }
            ODIN_SCHED_GROUP(ODIN_SG_MFMA, 5);
          }
          ODIN_SCHED_FENCE();
        }
      }
      if constexpr (WE) {
        epilogue(pa0, pa1, opixP, axP, tgtP);
        // one scheduling region: 64 x (MFMA, [LDS read], NV VALU of the previous group's epilogue)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
            if (u < 3 && s + 1 < 8) ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
            ODIN_SCHED_GROUP(ODIN_SG_VALU, NV);
          }
        }
        ODIN_SCHED_FENCE();
      }
      pa0 = acc0; pa1 = acc1; opixP = opix;
      axP[0] = axN[0]; axP[1] = axN[1];
#pragma unroll
      for (int oc = 0; oc < ((EPI == 3) ? C1 : 1); ++oc) tgtP[oc] = tgtN[oc];
    };
    // one log-likelihood partial per (tile, wave): a tile lies inside one sample
    auto flush_llk = [&](int T) {
      if (EPI == 3) {
#ifndef ODIN_SIM
        asm volatile("; llk flush" ::: "memory");  // not speculated into every group's block
#endif
        const float tt = wave_sum64(llk_lane);
        if (lane == 0) p.llk_part[(size_t)T * TR_NCW + wave] = tt;
        llk_lane = 0.f;
      }
    };

    if (wave == 0) TR_STAMP(0, 2);
    __syncthreads();  // counters initialised, SAME-padding slots zeroed
    tr_wait_ge(&tr_sync[0], 4);  // the first tile's rows (and the weights) are in LDS
    if (wave == 0) TR_STAMP(0, 10);
    run_group(TrNo{}, 2 * half);
    if (nG == 1) tr_signal(&tr_sync[1], lane);
    if (wave == 0) TR_STAMP(0, 11);
#pragma unroll 1
    for (int G = 1; G < nG; ++G) {
      const int mt = 2 * half + (G & 1);
      if ((G & 1) == 0) {
        // tile boundary: every ring read of the previous tile has been issued and consumed
        sl0 += p.RP;
        if (++t_cur == p.tiles_per_img) { t_cur = 0; ++b_cur; ++sl0; }
        if (sl0 >= p.NSLOT) sl0 -= p.NSLOT;
        if (wave == 0) TR_STAMP(0, 12);
        tr_wait_ge(&tr_sync[0], 4 * ((G >> 1) + 1));  // rows of tile T0 + (G >> 1)
        if (wave == 0) TR_STAMP(0, 10);
      }
      run_group(TrYes{}, mt);
      if (G & 1) tr_signal(&tr_sync[1], lane);  // every ring read of this tile has returned
      if (wave == 0) TR_STAMP(0, 11);
      if ((G & 1) == 0) flush_llk(T0 + (G >> 1) - 1);
    }
    epilogue(pa0, pa1, opixP, axP, tgtP);
    flush_llk(T1 - 1);
    if (EPI >= 2) {
      // per-workgroup partial sums: 16 pixel lanes by shuffles, then the 4 waves through LDS
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float vv = csum[i];
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
        const int ch = (i >> 2) * 16 + 4 * kq + (i & 3);
        if (l15 == 0) cred[wave * NRED + ch] = vv;
        if (EPI == 3) {
#pragma unroll
          for (int oc = 0; oc < C1; ++oc) {
            float ww = dw1[i][oc];
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) ww += __shfl_xor(ww, m);
            if (l15 == 0) cred[wave * NRED + 32 + ch * C1 + oc] = ww;
          }
        }
      }
    }
    if (EPI == 3) {
#pragma unroll
      for (int oc = 0; oc < C1; ++oc) {
        float vv = db1[oc];  // non-zero in the kq == 0 lanes only
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) vv += __shfl_xor(vv, m);
        if (lane == 0) cred[wave * NRED + 32 + 32 * C1 + oc] = vv;
      }
    }
  }  // consumers
  if (EPI == 2 && p.colsum != nullptr) {
    __syncthreads();
    if (tid < 32 && n0 + tid < p.CO)
      p.colsum[(size_t)blockIdx.x * p.CO + n0 + tid] =
          ((cred[tid] + cred[NRED + tid]) + (cred[2 * NRED + tid] + cred[3 * NRED + tid])) +
          ((cred[4 * NRED + tid] + cred[5 * NRED + tid]) + (cred[6 * NRED + tid] + cred[7 * NRED + tid]));
  }
  if (EPI == 3) {
    __syncthreads();
    // slab row: [dW1 (CO, C1) | db1 (C1) | column sums of out (CO)]
    float* row = p.slab + (size_t)blockIdx.x * (p.CO * C1 + C1 + p.CO);
    for (int e = tid; e < 32 + 32 * C1 + C1; e += TR_THREADS) {
      const float tt = ((cred[e] + cred[NRED + e]) + (cred[2 * NRED + e] + cred[3 * NRED + e])) +
                       ((cred[4 * NRED + e] + cred[5 * NRED + e]) + (cred[6 * NRED + e] + cred[7 * NRED + e]));
      if (e < 32) {
        if (e < p.CO) row[p.CO * C1 + C1 + e] = tt;
      } else if (e < 32 + 32 * C1) {
        const int ch = (e - 32) / C1, oc = (e - 32) - ch * C1;
        if (ch < p.CO) row[ch * C1 + oc] = tt;
      } else {
        row[p.CO * C1 + (e - 32 - 32 * C1)] = tt;
      }
    }
  }
}

}  // namespace

static long long* g_tr_stamps = nullptr;
void odin_tconv_ring_set_stamps(void* buf) { g_tr_stamps = (long long*)buf; }

bool odin_tconv_ring_applicable(int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl,
                                int center) {
  // Opt-in (ODIN_TRING=1): measured on MI355X this all-fp32 form is not faster than the bf16-plane
  // instances of gather_conv.hip (fused tail 105 vs 90 us, encoder1 data gradient 25.7 vs 25.5 us):
  // v_mfma_f32_*_f32 shares the vector ALU's issue with every other VALU instruction (nothing of an
  // epilogue hides behind it -- tools/micro/mfma_fillers.hip), while bf16 MFMAs run beside the VALU.
  const char* e = ODIN_DIAG_ENV("ODIN_TRING");
  if (e == nullptr || e[0] != '1') return false;
  return KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && CI == 32 && (CO % 32) == 0 &&
         !center && (W == 16 || W == 32) && (H % (64 / W)) == 0;
}

// epi 1 / 2 as above; epi 3: fused tail with C1 (1 or 3) logit maps
int odin_tconv_ring_launch(const float* in, const float* w, const float* bias, const float* aux,
                           float* out, float* colsum, int* rows_out, const float* w1, const float* b1,
                           const float* target, float* logits, float* llk_part, int* n_part_out,
                           float* slab, const float* scale, int C1, int B, int H, int W, int CO,
                           int epi, void* stream) {
  TRParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.w1 = w1; p.b1 = b1; p.target = target; p.logits = logits; p.llk_part = llk_part; p.slab = slab;
  p.scale = scale;
  p.B = B; p.H = H; p.W = W; p.CO = CO;
  p.stamps = g_tr_stamps;
  p.RP = 64 / W;
  p.NSLOT = TR_DEPTH * p.RP + 2 + TR_DEPTH;
  p.RB = (W + 2) * 128;
  p.tiles_per_img = H / p.RP;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  p.tiles_per_wg = (p.n_tiles + cap - 1) / cap;
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (n_part_out) *n_part_out = TR_NCW * p.tiles_per_img;
  if (out == nullptr) return 0;  // dry run
  if (epi == 3 && (CO != 32 || (C1 != 1 && C1 != 3)))
    return odin_fail(-2, "tconv_ring tail: needs Cout == 32 and 1 or 3 logit maps");
  const size_t lds = (size_t)TR_WBYTES + (size_t)p.NSLOT * p.RB;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[4] = {reinterpret_cast<const void*>(&tconv_ring_kernel<1, 1>),
                          reinterpret_cast<const void*>(&tconv_ring_kernel<2, 1>),
                          reinterpret_cast<const void*>(&tconv_ring_kernel<3, 1>),
                          reinterpret_cast<const void*>(&tconv_ring_kernel<3, 3>)};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  dim3 grid(gx, gy, 1);
  if (epi == 1) ODIN_LAUNCH((tconv_ring_kernel<1, 1>), grid, dim3(TR_THREADS), lds, stream, p);
  else if (epi == 2) ODIN_LAUNCH((tconv_ring_kernel<2, 1>), grid, dim3(TR_THREADS), lds, stream, p);
  else if (C1 == 1) ODIN_LAUNCH((tconv_ring_kernel<3, 1>), grid, dim3(TR_THREADS), lds, stream, p);
  else ODIN_LAUNCH((tconv_ring_kernel<3, 3>), grid, dim3(TR_THREADS), lds, stream, p);
  return odin_check_launch("tconv_ring");
}
#endif  // ODIN_DIAG
