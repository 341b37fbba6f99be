// gather_conv.hip -- NHWC implicit-GEMM convolution on the gfx950 f32 matrix cores.
//
// One kernel, two gather modes, covers six reference operations (all TF `SAME`):
//   MODE_F (gather conv):   out[b,oh,ow,n] = sum_{kh,kw,c} in[b,oh*S-pt+kh,ow*S-pl+kw,c] * W
//        = Conv2D forward              (odin/networks/image_networks.py:166-169)
//        = Conv2DTranspose data-grad   (tape.gradient, odin/networks/base_networks.py:518)
//        = Dense forward (1x1 image)   (odin/networks/base_networks.py:1002-1014)
//   MODE_T (transposed gather): out[b,oh,ow,n] = sum over (kh,kw,c) with
//        (oh+pt-kh)%S==0: in[b,(oh+pt-kh)/S,(ow+pl-kw)/S,c] * W
//        = Conv2DTranspose forward     (odin/networks/image_networks.py:170-173)
//        = Conv2D data-grad, Dense data-grad
// Weight layouts in HBM (wmode): 0 = [kh][kw][reduce][out]  1 = [kh][kw][out][reduce],
// which are exactly Keras' Conv2D (kh,kw,Cin,Cout) and Conv2DTranspose (kh,kw,Cout,Cin)
// layouts seen from the forward (0 / 1) or the data-grad (1 / 0) side: no weight is ever
// transposed in memory.
//
// Tiling (MI355X): a workgroup owns TR full-width output rows (~128 output pixels) x 32
// output channels.  The input patch those pixels touch (with halo, zero-filled SAME
// padding, optional CenterAt0 fold-in) is staged ONCE into LDS with an odd pixel pitch,
// the weight slice [taps][CIC][32] is staged into LDS (kept resident across the
// workgroup's persistent tile loop when it fits), and every wave runs
// v_mfma_f32_32x32x2_f32 with A = weights (rows = output channels) and B = pixels
// (columns), so each lane ends up with 4 consecutive output channels of one pixel ->
// float4 NHWC stores with the bias / activation / activation-gradient epilogue fused.
// Reduction order per output is the fixed k-ordered fmaf chain of the MFMA: results are
// bit-reproducible run to run.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

enum { MODE_F = 0, MODE_T = 1 };

struct GParams {
  const float* in;
  const float* w;
  const float* bias;
  const float* aux;   // epilogue: out *= act'(aux) (aux_act), same shape as out
  float* out;
  float* colsum_slab;  // optional [gridDim.x][CO] partial column sums of `out`
  int B, H, W, CI, OH, OW, CO;
  int KH, KW, S, pt, pl;
  int wmode, act, aux_act, center;
  // plan
  int TR, RPI, NIMG, n_tiles;  // rows per tile, rows per image in a tile, images per tile
  int NRI, PW, P;              // patch rows per image, patch width (pixels), pixel pitch
  int ih_off, iw_lo;           // F: ih_lo = oh0*S + ih_off ; T: ih_lo = oh0/S + ih_off
  int CIC, n_chunks, WP, w_resident;
  int patch_floats, MT, MTP, SPP;  // M-tiles total / per phase, slots per phase
  int vec, KI, pipelined;          // 16-byte staging items, items per lane per patch row
  int KS;                          // intra-workgroup split of the reduction (1, 2 or 4 waves per M-tile)
  int n_batches;                   // staging batches per patch (1 when pipelined)
  int flat;                        // 1x1 images (Dense): the patch is one contiguous [NIMG, CIC] block
  int wdma;                        // weight slice staged by LDS-DMA (wmode 0, full blocks)
  long long* stamps;  // diagnostic: s_memtime stamps of workgroup 0 / wave 0 (env ODIN_STAMPS)
  int dbg;  // diagnostic ablation mask (env ODIN_DBG): 1 skip MFMA, 2 skip stores, 4 skip patch staging
};

// Weight slice [taps][CIC][32 output channels] -> LDS.  Loads are issued in batches of 8
// per thread (16-byte loads where the layout allows) so that their latencies overlap; a
// one-load-per-iteration loop costs a full memory round trip per element.
__device__ __forceinline__ void stage_weights(const GParams& p, float* wl, int c0, int n0,
                                              int tid, int nthreads) {
  const int ntaps = p.KH * p.KW;
  constexpr int U = 8;
  // one range-checked run over the whole weight tensor: masked items read zeros
  const OdinRun WR = odin_run(p.w, (unsigned)((size_t)ntaps * p.CI * p.CO * 4));
  if (p.wmode == 0) {
    // global [tap][ci][co]: rows of 32 consecutive output channels
    const bool vec = ((p.CO & 3) == 0);
    if (vec && p.wdma) {
      // full 32-channel block, whole reduction in one chunk: the LDS image [tap*CIC + ci][32] is
      // lane-linear in 16-byte pieces, so the slice goes HBM -> LDS by DMA (no registers, no
      // ds_write; for a workgroup that computes one or two tiles the register-staged copy costs
      // as much as the tiles).  The issuing wave's vmcnt covers it before the first barrier.
      const int total = ntaps * p.CIC * 8;  // 16-byte pieces
      const int lane = tid & 63, w0 = tid - lane;
      for (int e0 = w0; e0 < total; e0 += nthreads) {  // wave-uniform
        const int e = e0 + lane;
        if (e < total) {
          const int t2 = e >> 3;
          const int ci = t2 % p.CIC, tap = t2 / p.CIC;
          odin_run_dma16(WR, wl + (size_t)e0 * 4,
                         (unsigned)((((tap * p.CI + c0 + ci) * p.CO) + n0 + (e & 7) * 4) * 4), lane);
        }
      }
      return;
    }
    if (vec) {
      const int total = ntaps * p.CIC * 8;  // float4 items
      for (int e0 = tid; e0 < total; e0 += nthreads * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          int co = (e & 7) * 4, t2 = e >> 3;
          int ci = t2 % p.CIC, tap = t2 / p.CIC;
          int c = c0 + ci, n = n0 + co;
          const bool ok = e < total && c < p.CI && n < p.CO;
          v[u] = odin_run_load4(WR, ok ? (unsigned)(((tap * p.CI + c) * p.CO + n) * 4) : ODIN_OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          if (e < total) {
            int co = (e & 7) * 4, t2 = e >> 3;  // t2 = tap*CIC + ci
            float* d = wl + t2 * p.WP + co;
            d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
          }
        }
      }
    } else {
      const int total = ntaps * p.CIC * 32;
      for (int e0 = tid; e0 < total; e0 += nthreads * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          int co = e & 31, t2 = e >> 5;
          int ci = t2 % p.CIC, tap = t2 / p.CIC;
          int c = c0 + ci, n = n0 + co;
          const bool ok = e < total && c < p.CI && n < p.CO;
          v[u] = odin_run_load1(WR, ok ? (unsigned)(((tap * p.CI + c) * p.CO + n) * 4) : ODIN_OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          if (e < total) wl[(e >> 5) * p.WP + (e & 31)] = v[u];
        }
      }
    }
  } else {
    // global [tap][co][ci]: contiguous along the reduction channel; transposed into LDS
    const bool vec = ((p.CI & 3) == 0) && ((p.CIC & 3) == 0) && ((c0 & 3) == 0);
    if (vec) {
      const int c4n = p.CIC >> 2;
      const int total = ntaps * 32 * c4n;
      for (int e0 = tid; e0 < total; e0 += nthreads * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // item -> (output channel fastest, then channel group, then tap): the four
          // transposed LDS writes of a wave then hit consecutive banks (a channel-group-fastest
          // order put all 64 lanes on 4 banks: 16-way conflicts on every write)
          int e = e0 + u * nthreads;
          int co = e & 31, t2 = e >> 5;
          int ci = (t2 % c4n) * 4, tap = t2 / c4n;
          int c = c0 + ci, n = n0 + co;
          const bool ok = e < total && c < p.CI && n < p.CO;
          v[u] = odin_run_load4(WR, ok ? (unsigned)(((tap * p.CO + n) * p.CI + c) * 4) : ODIN_OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          if (e < total) {
            int co = e & 31, t2 = e >> 5;
            int ci = (t2 % c4n) * 4, tap = t2 / c4n;
            float* d = wl + (tap * p.CIC + ci) * p.WP + co;
            d[0] = v[u].x; d[p.WP] = v[u].y; d[2 * p.WP] = v[u].z; d[3 * p.WP] = v[u].w;
          }
        }
      }
    } else {
      const int total = ntaps * p.CIC * 32;
      for (int e0 = tid; e0 < total; e0 += nthreads * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          int ci = e % p.CIC, t2 = e / p.CIC;
          int co = t2 & 31, tap = t2 >> 5;
          int c = c0 + ci, n = n0 + co;
          const bool ok = e < total && c < p.CI && n < p.CO;
          v[u] = odin_run_load1(WR, ok ? (unsigned)(((tap * p.CO + n) * p.CI + c) * 4) : ODIN_OOB);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          if (e < total) {
            int ci = e % p.CIC, t2 = e / p.CIC;
            int co = t2 & 31, tap = t2 >> 5;
            wl[(tap * p.CIC + ci) * p.WP + co] = v[u];
          }
        }
      }
    }
  }
}

__device__ __forceinline__ float4 center4(float4 v) {
  return make_float4(2.f * v.x - 1.f, 2.f * v.y - 1.f, 2.f * v.z - 1.f, 2.f * v.w - 1.f);
}

// ---- patch staging ---------------------------------------------------------------
// Row-aligned: wave w stages patch rows w, w+NW, ...; inside a row lane l handles items
// l, l+64, ... (an item = 4 consecutive channels of one patch pixel, or 1 float when the
// channel count does not allow 16-byte accesses).  Everything that depends only on
// (lane, k) -- source offset inside the image row, LDS offset inside the patch row, SAME
// padding mask -- is computed ONCE per kernel; per tile a row costs one scalar base, one
// scalar bounds test and KI loads with 32-bit offsets.  The loads of tile t+1 are issued
// into registers before tile t is multiplied and committed to LDS after it.
#ifdef ODIN_SIM
#define ODIN_UNIFORM(x) (x)
#else
#define ODIN_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// Split layout (fp32 through the bf16 matrix pipe, odin_device.h): the patch is three bf16 planes
// [pixel][32 channels], pixel pitch 80 bytes (64 + 16: the 16-byte B-operand reads of 8 lanes
// then cover 8 distinct bank quads for pixel strides 1 and 2).
constexpr int SPLIT_PITCH = 80;

template <int KMAX>
struct LaneStage {
  unsigned gofs[KMAX];  // byte offset of the item inside an input image row
  int ldo[KMAX];        // float offset of the item inside a patch row
  int ldo2[KMAX];       // split (3 x bf16 plane) layout: byte offset inside a patch row
  unsigned jmask;       // bit k: item k exists (inside the patch row)
  unsigned okmask;      // bit k: item k reads real data (not SAME padding)
};

template <int KMAX, bool VEC>
__device__ __forceinline__ LaneStage<KMAX> lane_stage_init(const GParams& p, int lane) {
  LaneStage<KMAX> L;
  L.jmask = 0;
  L.okmask = 0;
  const int cpi = VEC ? (p.CIC >> 2) : p.CIC;
  const int rowlen = p.PW * cpi;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = lane + 64 * k;
    const int pcol = j / cpi;
    const int cc = (j - pcol * cpi) * (VEC ? 4 : 1);
    const int iw = p.iw_lo + pcol;
    const bool jv = (k < p.KI) && (j < rowlen);
    const bool ok = jv && (iw >= 0) && (iw < p.W) && (cc < p.CI);
    L.gofs[k] = ok ? (unsigned)((iw * p.CI + cc) * 4) : ODIN_OOB;
    L.ldo[k] = pcol * p.P + cc;
    L.ldo2[k] = pcol * SPLIT_PITCH + cc * 2;
    if (jv) L.jmask |= 1u << k;
    if (ok) L.okmask |= 1u << k;
  }
  return L;
}

template <bool VEC>
struct StageT { typedef float4 type; };
template <>
struct StageT<false> { typedef float type; };

template <bool VEC>
__device__ __forceinline__ typename StageT<VEC>::type stage_zero() {
  if constexpr (VEC) return make_float4(0.f, 0.f, 0.f, 0.f);
  else return 0.f;
}

// issue the loads of patch row r of the tile starting at (b0, ih_lo); c0 = channel chunk.
// Branch-free: the row is one range-checked run (zero bytes when the row is SAME padding or
// beyond the batch), padding lanes carry an out-of-range offset.
template <int KMAX, bool VEC>
__device__ __forceinline__ void stage_row_issue(const GParams& p, const LaneStage<KMAX>& L, int r,
                                                int b0, int ih_lo, int c0,
                                                typename StageT<VEC>::type* v) {
  const int img = (p.NIMG == 1) ? 0 : r / p.NRI;
  const int prow = r - img * p.NRI;
  const int b = b0 + img, ih = ih_lo + prow;
  const bool row_ok = (b < p.B) && (ih >= 0) && (ih < p.H);
  const float* rowp = p.in + (row_ok ? ((size_t)((b * p.H + ih) * p.W) * p.CI + c0) : (size_t)0);
  const OdinRun R = odin_run(rowp, row_ok ? (unsigned)((p.W * p.CI - c0) * 4) : 0u);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    unsigned off = L.gofs[k];
    // ragged last channel chunk: lanes beyond the real channels read zeros
    if (c0 != 0 && c0 + (int)((off >> 2) % (unsigned)p.CI) >= p.CI) off = ODIN_OOB;
    if constexpr (VEC) v[k] = odin_run_load4(R, off);
    else v[k] = odin_run_load1(R, off);
  }
}

template <int KMAX, bool VEC>
__device__ __forceinline__ void stage_row_commit(const GParams& p, const LaneStage<KMAX>& L, int r,
                                                 int b0, int ih_lo,
                                                 const typename StageT<VEC>::type* v,
                                                 float* patch) {
  float* rowl = patch + r * p.PW * p.P;
  bool cen = false;
  if (p.center) {  // CenterAt0 folded into the first layer: real pixels only, padding stays 0
    const int img = (p.NIMG == 1) ? 0 : r / p.NRI;
    const int prow = r - img * p.NRI;
    const int b = b0 + img, ih = ih_lo + prow;
    cen = (b < p.B) && (ih >= 0) && (ih < p.H);
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if ((L.jmask >> k) & 1u) {
      float* d = rowl + L.ldo[k];
      typename StageT<VEC>::type t = v[k];
      if (cen && ((L.okmask >> k) & 1u)) {
        if constexpr (VEC) t = center4(t);
        else t = 2.f * t - 1.f;
      }
      if constexpr (VEC) { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
      else d[0] = t;
    }
  }
}

// A whole staging batch.  Row mode: rows batch*NW*RPWMAX + wave + NW*q.  Flat mode (Dense:
// every "image" is one pixel, the tile's input is the contiguous block in[b0 .. b0+NIMG)[CI]):
// items e = (batch*PFN + i)*NT + tid -> (image, channel group).
template <int KMAX, int RPWMAX, bool VEC, int NW, bool FLAT>
__device__ __forceinline__ void stage_issue(const GParams& p, const LaneStage<KMAX>& LS, int wave,
                                            int tid, int batch, int b0, int ih_lo, int c0,
                                            typename StageT<VEC>::type* pf) {
  constexpr int PFN = KMAX * RPWMAX, NT = NW * 64;
  if constexpr (FLAT) {
    const int cpi = VEC ? (p.CIC >> 2) : p.CIC;
    const int total = p.NIMG * cpi;
    // the tile's input is the contiguous block in[b0 .. B)[CI]; samples beyond the batch fall
    // outside the run and read zeros
    const long left = ((long)(p.B - b0) * p.CI - c0) * 4;
    const OdinRun R = odin_run(p.in + (b0 < p.B ? (size_t)b0 * p.CI + c0 : (size_t)0),
                               left <= 0 ? 0u : (left > 0x7FFFFFF0L ? 0x7FFFFFF0u : (unsigned)left));
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int e = (batch * PFN + i) * NT + tid;
      const int img = e / cpi;
      const int cc = (e - img * cpi) * (VEC ? 4 : 1);
      const unsigned off = (e < total && c0 + cc < p.CI) ? (unsigned)((img * p.CI + cc) * 4) : ODIN_OOB;
      if constexpr (VEC) pf[i] = odin_run_load4(R, off);
      else pf[i] = odin_run_load1(R, off);
    }
  } else {
    const int nrows_p = p.NIMG * p.NRI;
#pragma unroll
    for (int q = 0; q < RPWMAX; ++q) {
      const int r = batch * NW * RPWMAX + wave + NW * q;
      if (r < nrows_p) stage_row_issue<KMAX, VEC>(p, LS, r, b0, ih_lo, c0, pf + q * KMAX);
    }
  }
}

template <int KMAX, int RPWMAX, bool VEC, int NW, bool FLAT>
__device__ __forceinline__ void stage_commit(const GParams& p, const LaneStage<KMAX>& LS, int wave,
                                             int tid, int batch, int b0, int ih_lo,
                                             const typename StageT<VEC>::type* pf, float* patch) {
  constexpr int PFN = KMAX * RPWMAX, NT = NW * 64;
  if constexpr (FLAT) {
    const int cpi = VEC ? (p.CIC >> 2) : p.CIC;
    const int total = p.NIMG * cpi;
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int e = (batch * PFN + i) * NT + tid;
      if (e < total) {
        const int img = e / cpi;
        const int cc = (e - img * cpi) * (VEC ? 4 : 1);
        float* d = patch + img * p.P + cc;
        if constexpr (VEC) { d[0] = pf[i].x; d[1] = pf[i].y; d[2] = pf[i].z; d[3] = pf[i].w; }
        else d[0] = pf[i];
      }
    }
  } else {
    const int nrows_p = p.NIMG * p.NRI;
#pragma unroll
    for (int q = 0; q < RPWMAX; ++q) {
      const int r = batch * NW * RPWMAX + wave + NW * q;
      if (r < nrows_p) stage_row_commit<KMAX, VEC>(p, LS, r, b0, ih_lo, pf + q * KMAX, patch);
    }
  }
}

// geometry of the output pixel a lane owns inside M-tile `mt`.  Everything except the
// tile's first output row is tile-invariant and computed once per kernel.
struct Slot {
  int base;      // patch float index of tap (0,0) channel 0
  int row, col;  // output row inside the tile / output column (row < 0: masked slot)
  int opix;      // linear output pixel index (b*OH+oh)*OW+ow of the CURRENT tile, or -1
  int kh0, kw0;  // MODE_T: first valid tap of this M-tile's phase
};

template <int MODE>
__device__ __forceinline__ Slot slot_geometry(const GParams& p, int mt, int l31) {
  Slot s;
  s.kh0 = s.kw0 = 0;
  s.opix = -1;
  if (MODE == MODE_F) {
    int sl = mt * 32 + l31;
    int r = sl / p.OW, c = sl - r * p.OW;
    bool valid = (r < p.TR);
    int img = r / p.RPI, rl = r - img * p.RPI;
    s.base = valid ? ((img * p.NRI + rl * p.S) * p.PW + c * p.S) * p.P : 0;
    s.row = valid ? r : -1;
    s.col = c;
  } else {
    const int S = p.S;
    int phase = mt / p.MTP, mtl = mt - phase * p.MTP;
    int ph = phase / S, pw = phase - ph * S;
    int sl = mtl * 32 + l31;
    const int IWs = p.OW / S, RPS = p.RPI / S;
    int img = sl / (RPS * IWs), rem = sl - img * (RPS * IWs);
    int rq = rem / IWs, cq = rem - rq * IWs;
    int row_in_tile = img * p.RPI + ph + S * rq;
    bool valid = (sl < p.SPP);
    s.kh0 = (ph + p.pt) % S;
    s.kw0 = (pw + p.pl) % S;
    int dh = (ph + p.pt - s.kh0) / S, dw = (pw + p.pl - s.kw0) / S;
    // patch row of tap jh=0: rq + dh - lo_h, where lo_h == p.ih_off ; same for columns
    int prow = rq + dh - p.ih_off, pcol = cq + dw - p.iw_lo;
    // masked lanes read the last patch pixel (tap offsets are negative in this mode)
    s.base = valid ? ((img * p.NRI + prow) * p.PW + pcol) * p.P
                   : ((p.NIMG * p.NRI - 1) * p.PW + (p.PW - 1)) * p.P;
    s.row = valid ? row_in_tile : -1;
    s.col = pw + S * cq;
  }
  return s;
}

__device__ __forceinline__ void slot_set_tile(const GParams& p, Slot& s, int gr0) {
  const bool live = (s.row >= 0) && (gr0 + s.row < p.B * p.OH);
  s.opix = live ? (gr0 + s.row) * p.OW + s.col : -1;
}

// One 32(out-channel) x 32(pixel) accumulator tile over one channel chunk.  TK/TS/TCIC
// are compile-time kernel size / stride / chunk (0 = runtime).  The specialised instances
// run an explicit two-stage register pipeline: the 32 LDS operands of step t+1 (one tap x
// 32 channels = 16 MFMAs) are read while the MFMAs of step t execute.
struct TapAddr {
  const float* ap;
  const float* wp;
};

template <int MODE, int TK, int TS, int TCIC, int SG>
__device__ __forceinline__ TapAddr tap_addr(const GParams& p, const float* patch, const float* wl,
                                            const Slot& s, int l31, int h, int step) {
  constexpr int SUB = TCIC / (2 * SG);      // sub-steps (2*SG channels each) per tap
  constexpr int NJ = (MODE == MODE_F) ? TK : TK / TS;
  constexpr int WP = 32;
  constexpr int P = TCIC + 1;
  const int tap = step / SUB, sub = step - tap * SUB;
  const int jh = tap / NJ, jw = tap - jh * NJ;
  int tapoff, wt;
  if (MODE == MODE_F) {
    tapoff = (jh * p.PW + jw) * P;
    wt = jh * TK + jw;
  } else {
    tapoff = -(jh * p.PW + jw) * P;
    wt = (s.kh0 + TS * jh) * TK + (s.kw0 + TS * jw);
  }
  TapAddr t;
  t.ap = patch + s.base + tapoff + h + sub * (2 * SG);
  t.wp = wl + (wt * TCIC + h + sub * (2 * SG)) * WP + l31;
  return t;
}

template <int MODE, int TK, int TS, int TCIC, int SG = 16>
__device__ __forceinline__ f32x16 mtile_compute(const GParams& p, const float* patch,
                                                const float* wl, const Slot& s, int l31, int h,
                                                f32x16 acc, int kc0 = 0, int kc1 = -1) {
  if constexpr (TCIC != 0) {
    static_assert(TCIC % 32 == 0 && TK != 0 && TS != 0 && TK % TS == 0, "specialised shape");
    constexpr int NJ = (MODE == MODE_F) ? TK : TK / TS;
    constexpr int NSTEP = NJ * NJ * (TCIC / (2 * SG));
    constexpr int WP = 32;
    float a0[SG], b0[SG], a1[SG], b1[SG];
    {
      TapAddr t = tap_addr<MODE, TK, TS, TCIC, SG>(p, patch, wl, s, l31, h, 0);
#pragma unroll
      for (int u = 0; u < SG; ++u) { a0[u] = t.wp[u * 2 * WP]; b0[u] = t.ap[2 * u]; }
    }
    ODIN_SCHED_FENCE();
#pragma unroll
    for (int st = 0; st < NSTEP; st += 2) {
      if (st + 1 < NSTEP) {
        TapAddr t = tap_addr<MODE, TK, TS, TCIC, SG>(p, patch, wl, s, l31, h, st + 1);
#pragma unroll
        for (int u = 0; u < SG; ++u) { a1[u] = t.wp[u * 2 * WP]; b1[u] = t.ap[2 * u]; }
      }
#pragma unroll
      for (int u = 0; u < SG; ++u) acc = mfma32(a0[u], b0[u], acc);
      // issue the next step's LDS reads in the shadow of this step's MFMAs
#pragma unroll
      for (int u = 0; u < SG; ++u) {
        ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
        ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 2);
      }
      ODIN_SCHED_FENCE();
      if (st + 2 < NSTEP) {
        TapAddr t = tap_addr<MODE, TK, TS, TCIC, SG>(p, patch, wl, s, l31, h, st + 2);
#pragma unroll
        for (int u = 0; u < SG; ++u) { a0[u] = t.wp[u * 2 * WP]; b0[u] = t.ap[2 * u]; }
      }
      if (st + 1 < NSTEP) {
#pragma unroll
        for (int u = 0; u < SG; ++u) acc = mfma32(a1[u], b1[u], acc);
#pragma unroll
        for (int u = 0; u < SG; ++u) {
          ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
          ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 2);
        }
        ODIN_SCHED_FENCE();
      }
    }
    return acc;
  } else {
    const int KH = p.KH, KW = p.KW, S = p.S, CIC = p.CIC, P = p.CIC + 1, WP = p.WP;
    // tap loops are wave-uniform: every lane of an M-tile shares the phase (kh0, kw0)
    const int njh = (MODE == MODE_F) ? KH : (KH - s.kh0 + S - 1) / S;
    const int njw = (MODE == MODE_F) ? KW : (KW - s.kw0 + S - 1) / S;
    for (int jh = 0; jh < njh; ++jh) {
      for (int jw = 0; jw < njw; ++jw) {
        int tapoff, wt;
        if (MODE == MODE_F) {
          tapoff = (jh * p.PW + jw) * P;
          wt = jh * KW + jw;
        } else {
          tapoff = -(jh * p.PW + jw) * P;
          wt = (s.kh0 + S * jh) * KW + (s.kw0 + S * jw);
        }
        const float* ap = patch + s.base + tapoff + h;
        const float* wp = wl + (wt * CIC + h) * WP + l31;
        int c = kc0;
        const int cend = kc1 < 0 ? CIC : kc1;
        for (; c + 16 <= cend; c += 16) {
          float a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u] = wp[(c + 2 * u) * WP]; b[u] = ap[c + 2 * u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = mfma32(a[u], b[u], acc);
        }
        for (; c < cend; c += 2) acc = mfma32(wp[c * WP], ap[c], acc);
      }
    }
    return acc;
  }
}

// ---- split path: staging commit and compute -------------------------------------------
// One patch row: every float4 item becomes three 8-byte writes (bf16 pieces of 4 channels).
template <int KMAX>
__device__ __forceinline__ void stage_row_commit_split(const GParams& p, const LaneStage<KMAX>& L,
                                                       int r, const float4* v, char* patch,
                                                       int pls) {
  char* rowl = patch + r * p.PW * SPLIT_PITCH;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if ((L.jmask >> k) & 1u) {
      const float4 t = v[k];
      const float4 m = make_float4(odin_bf16_rest(t.x), odin_bf16_rest(t.y), odin_bf16_rest(t.z),
                                   odin_bf16_rest(t.w));
      const float4 l = make_float4(odin_bf16_rest(m.x), odin_bf16_rest(m.y), odin_bf16_rest(m.z),
                                   odin_bf16_rest(m.w));
      char* d = rowl + L.ldo2[k];
      *reinterpret_cast<u32x2*>(d) = odin_u2(odin_pack_bf16(t.x, t.y), odin_pack_bf16(t.z, t.w));
      *reinterpret_cast<u32x2*>(d + pls) = odin_u2(odin_pack_bf16(m.x, m.y), odin_pack_bf16(m.z, m.w));
      *reinterpret_cast<u32x2*>(d + 2 * pls) = odin_u2(odin_pack_bf16(l.x, l.y), odin_pack_bf16(l.z, l.w));
    }
  }
}

// eight fp32 weights (consecutive reduction channels of one (tap, output channel)) -> one
// A fragment per bf16 plane
__device__ __forceinline__ void split_fragment(const float* f, u32x4& a0, u32x4& a1, u32x4& a2) {
  float m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = odin_bf16_rest(f[j]);
    l[j] = odin_bf16_rest(m[j]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a0[q] = odin_pack_bf16(f[2 * q], f[2 * q + 1]);
    a1[q] = odin_pack_bf16(m[2 * q], m[2 * q + 1]);
    a2[q] = odin_pack_bf16(l[2 * q], l[2 * q + 1]);
  }
}

// MODE_T, 4x4 / stride 2, 32 reduction channels: the 4 taps of this wave's phase x 2 k-groups of
// 16 channels; the weight fragments A[plane][tap][kgroup] live in registers for the whole kernel
// (no LDS traffic for weights), per step 3 B reads of 16 bytes feed 6 MFMAs.
__device__ __forceinline__ f32x16 mtile_compute_split(const GParams& p, const char* patch, int pls,
                                                      int pix0, int h, const u32x4 (&A)[3][4][2]) {
  f32x16 accE = f32x16_zero(), accO = f32x16_zero();
  const char* bp = patch + pix0 * SPLIT_PITCH + h * 16;
  u32x4 b0[3], b1[3];
  auto bload = [&](int step, u32x4 (&b)[3]) {
    const int tap = step >> 1, kg = step & 1;
    const int jh = tap >> 1, jw = tap & 1;
    const char* q = bp - (jh * p.PW + jw) * SPLIT_PITCH + kg * 32;
    b[0] = *reinterpret_cast<const u32x4*>(q);
    b[1] = *reinterpret_cast<const u32x4*>(q + pls);
    b[2] = *reinterpret_cast<const u32x4*>(q + 2 * pls);
  };
  auto mm = [&](int step, const u32x4 (&b)[3]) {
    const int tap = step >> 1, kg = step & 1;
    // smallest terms first; two accumulator chains
    accE = mfma32_bf16(A[2][tap][kg], b[0], accE);
    accO = mfma32_bf16(A[0][tap][kg], b[2], accO);
    accE = mfma32_bf16(A[1][tap][kg], b[1], accE);
    accO = mfma32_bf16(A[1][tap][kg], b[0], accO);
    accE = mfma32_bf16(A[0][tap][kg], b[1], accE);
    accO = mfma32_bf16(A[0][tap][kg], b[0], accO);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      ODIN_SCHED_GROUP(ODIN_SG_MFMA, 2);
      ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
    }
    ODIN_SCHED_FENCE();
  };
  bload(0, b0);
  ODIN_SCHED_FENCE();
#pragma unroll
  for (int st = 0; st < 8; st += 2) {
    bload(st + 1, b1);
    mm(st, b0);
    if (st + 2 < 8) bload(st + 2, b0);
    mm(st + 1, b1);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) accE[i] += accO[i];
  return accE;
}

// ---- split path, 8-wave workgroups: the bf16 weight planes live in LDS ---------------------
// [plane 3][tap 16][kgroup 2][k-half 2][out channel 32][8 bf16] = 96 KB, shared by the 8 waves
// (two per SIMD) of the workgroup; fragment (tap, kg, h, co) is one 16-byte read.
constexpr int SPLIT_WPLANE = 16 * 2 * 2 * 32 * 16;  // bytes per plane

__device__ __forceinline__ void stage_weights_split(const GParams& p, char* wlb, int n0, int tid,
                                                    int nthreads) {
  const OdinRun WR = odin_run(p.w, (unsigned)((size_t)p.KH * p.KW * p.CI * p.CO * 4));
  for (int i = tid; i < 16 * 2 * 2 * 32; i += nthreads) {
    const int co = n0 + (i & 31), hh = (i >> 5) & 1, kg = (i >> 6) & 1, wt = i >> 7;
    const int ci0 = kg * 16 + hh * 8;
    float f[8];
    if (p.wmode == 1) {
      const unsigned off = co < p.CO ? (unsigned)(((wt * p.CO + co) * p.CI + ci0) * 4) : ODIN_OOB;
      const float4 u0 = odin_run_load4(WR, off), u1 = odin_run_load4(WR, off == ODIN_OOB ? off : off + 16);
      f[0] = u0.x; f[1] = u0.y; f[2] = u0.z; f[3] = u0.w;
      f[4] = u1.x; f[5] = u1.y; f[6] = u1.z; f[7] = u1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        f[j] = odin_run_load1(WR, co < p.CO ? (unsigned)(((wt * p.CI + ci0 + j) * p.CO + co) * 4) : ODIN_OOB);
    }
    u32x4 a0, a1, a2;
    split_fragment(f, a0, a1, a2);
    *reinterpret_cast<u32x4*>(wlb + i * 16) = a0;
    *reinterpret_cast<u32x4*>(wlb + SPLIT_WPLANE + i * 16) = a1;
    *reinterpret_cast<u32x4*>(wlb + 2 * SPLIT_WPLANE + i * 16) = a2;
  }
}

__device__ __forceinline__ f32x16 mtile_compute_split_lds(const GParams& p, const char* patch, int pls,
                                                          int pix0, int l31, int h, int kh0, int kw0,
                                                          const char* wlb) {
  f32x16 accE = f32x16_zero(), accO = f32x16_zero();
  const char* bp = patch + pix0 * SPLIT_PITCH + h * 16;
  const char* ap = wlb + (h * 32 + l31) * 16;
  u32x4 a0[3], b0[3], a1[3], b1[3];
  auto loads = [&](int step, u32x4 (&a)[3], u32x4 (&b)[3]) {
    const int tap = step >> 1, kg = step & 1;
    const int jh = tap >> 1, jw = tap & 1;
    const int wt = (kh0 + 2 * jh) * 4 + (kw0 + 2 * jw);
    const char* qa = ap + ((wt * 2 + kg) * 2) * 32 * 16;
    const char* qb = bp - (jh * p.PW + jw) * SPLIT_PITCH + kg * 32;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      a[pl] = *reinterpret_cast<const u32x4*>(qa + pl * SPLIT_WPLANE);
      b[pl] = *reinterpret_cast<const u32x4*>(qb + pl * pls);
    }
  };
  auto mm = [&](const u32x4 (&a)[3], const u32x4 (&b)[3]) {
    accE = mfma32_bf16(a[2], b[0], accE);
    accO = mfma32_bf16(a[0], b[2], accO);
    accE = mfma32_bf16(a[1], b[1], accE);
    accO = mfma32_bf16(a[1], b[0], accO);
    accE = mfma32_bf16(a[0], b[1], accE);
    accO = mfma32_bf16(a[0], b[0], accO);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
      ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
    }
    ODIN_SCHED_FENCE();
  };
  loads(0, a0, b0);
  ODIN_SCHED_FENCE();
#pragma unroll
  for (int st = 0; st < 8; st += 2) {
    loads(st + 1, a1, b1);
    mm(a0, b0);
    if (st + 2 < 8) loads(st + 2, a0, b0);
    mm(a1, b1);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) accE[i] += accO[i];
  return accE;
}

// Fused decoder tail (training step): this kernel's layer is a Conv2DTranspose whose
// output feeds a 1x1 linear Conv2D with C1 <= 4 maps that parameterise
// Independent(Bernoulli(logits)).  The epilogue evaluates the 1x1 conv, the Bernoulli
// log-likelihood and ITS BACKWARD in registers, so the [B,OH,OW,32] activation never
// round-trips HBM: what is stored is dL/d(pre-activation) of this layer.
// (odin/networks/image_networks.py:505-511 decoder4 -> decoder6 -> :87-93 Bernoulli;
//  variational_autoencoder.py:528-530 log_prob)
struct TailParams {
  const float* w1;      // [CO][C1]
  const float* b1;      // [C1]
  const float* target;  // [B,OH,OW,C1]
  float* logits;        // optional [B,OH,OW,C1]
  float* llk_part;      // [n_tiles]
  float* slab;          // [gridDim.x][CO*C1 + C1 + CO]
  const float* scale;   // device scalar 1/B
  int C1;
};
constexpr int MAXC1 = 4;

// Hardware log / reciprocal (1 ulp): log(1 + e) for e in (0, 1] has an absolute error below
// 1.2e-7 per pixel (the IEEE log1pf / division sequences are ~50 instructions per pixel, a tenth
// of this kernel's epilogue).
__device__ __forceinline__ float odin_rcp(float x) {
#ifdef ODIN_SIM
  return 1.f / x;
#else
  return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float softplus_g(float x) {
  return fmaxf(x, 0.f) + odin_log(1.f + odin_exp(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_g(float x) {
  float e = odin_exp(-fabsf(x));
  float r = odin_rcp(1.f + e);
  return x >= 0.f ? r : e * r;
}

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define ODIN_STAMP(k) ((void)0)
#else
#define ODIN_STAMP(k)                                                                    \
  do {                                                                                   \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && \
        stamp_i < 60)                                                                    \
      p.stamps[stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

// EPI: 3 = as 0 but with FLAT (Dense) staging; 0 = runtime activation / aux / channel masking; 1 = ELU, no aux, CO % 32 == 0
// (forward of the elu stacks); 2 = linear, aux = ELU derivative, CO % 32 == 0 (data-gradients).
// SPL: split path (three bf16 planes, weights in registers; MODE_T k4/s2 32-channel instances).
template <int MODE, int NW, int TK, int TS, int TCIC, bool VEC, int TAIL, int KMAX, int RPWMAX, int EPI, int NMT,
          bool SPL = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && MODE == MODE_T && TK == 4 && TCIC == 32 && RPWMAX == 1 &&
                                       TAIL <= 1 && !(SPL && TAIL > 0)) ? 2 : 1)
void gather_conv_kernel(GParams p, TailParams tp) {
  static_assert(!SPL || (MODE == MODE_T && TK == 4 && TS == 2 && TCIC == 32 && VEC && NMT == 1 && RPWMAX == 1),
                "split path: transposed 4x4/s2 gather, 32 reduction channels, one M-tile per wave");
  constexpr bool SPLW = SPL && NW == 8;  // 8 waves: bf16 weight planes in LDS instead of registers
  ODIN_DYN_SMEM(float, smem);
  float* patch = smem;
  float* wl = smem + p.patch_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = ODIN_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32;
  constexpr int NT = NW * 64;
  typedef typename StageT<VEC>::type SV;
  int stamp_i = 0;
  (void)stamp_i;
  ODIN_STAMP(1);
  const bool pipelined = p.pipelined != 0;
  const LaneStage<KMAX> LS = lane_stage_init<KMAX, VEC>(p, lane);

  // LDS floats of the weight slice
  const int wfloats = SPLW ? (3 * SPLIT_WPLANE) / 4 : (SPL ? 0 : p.KH * p.KW * p.CIC * p.WP);
  if (p.w_resident && !SPL) stage_weights(p, wl, 0, n0, tid, NT);
  if constexpr (SPLW) stage_weights_split(p, reinterpret_cast<char*>(wl), n0, tid, NT);
  ODIN_STAMP(2);

  float bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = 0.f;
  // fused-tail state
  // TAIL = number of 1x1 output maps evaluated by the fused tail (0: no tail)
  constexpr int NC1 = TAIL > 0 ? TAIL : 1;
  f32x2 dw1[TAIL ? 8 : 1][NC1];  // pairs (i, i+1) of the lane's 16 channels
  float db1[NC1], b1r[NC1];
  // the tail's per-channel constants (bias of this layer, 1x1 weights) stay in LDS and are
  // re-read by every epilogue: 32 + 32*NC1 fewer live registers across the MFMA loop
  float* tailc = wl + wfloats + 64;  // [32 bias | NC1 x 32 w1]
  if (TAIL) {
    for (int e = tid; e < 32 * (1 + NC1); e += NT) {
      float val;
      if (e < 32) {
        val = (p.bias != nullptr && n0 + e < p.CO) ? p.bias[n0 + e] : 0.f;
      } else {
        const int oc = (e - 32) >> 5, n = (e - 32) & 31;
        val = (oc < tp.C1 && n < p.CO) ? tp.w1[n * tp.C1 + oc] : 0.f;
      }
      tailc[e] = val;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int oc = 0; oc < NC1; ++oc) dw1[i][oc] = odin_f2(0.f, 0.f);
    }
#pragma unroll
    for (int oc = 0; oc < NC1; ++oc) {
      db1[oc] = 0.f;
      b1r[oc] = oc < tp.C1 ? tp.b1[oc] : 0.f;
    }
  }

  // tile-invariant lane state: slot geometry, bias of this lane's 16 output channels
  // KS > 1: KS waves share one M-tile and split the reduction channels (small-M layers)
  const int KS = p.KS;
  const int kpart = (KS > 1) ? wave % KS : 0;
  const int kc0 = (KS > 1) ? kpart * (p.CIC / KS) : 0;
  const int kc1 = (KS > 1) ? kc0 + p.CIC / KS : -1;
  const int mt0 = (KS > 1) ? wave / KS : wave, mt1 = wave + NW;
  Slot s0 = slot_geometry<MODE>(p, mt0 < p.MT ? mt0 : 0, l31);
  Slot s1 = slot_geometry<MODE>(p, mt1 < p.MT ? mt1 : 0, l31);
  float bias_r[TAIL ? 1 : 16];
  if constexpr (TAIL == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
      bias_r[i] = (p.bias != nullptr && n < p.CO) ? p.bias[n] : 0.f;
    }
  }
  const bool co_vec = ((p.CO & 3) == 0);

  // split path: this wave's weight fragments (its M-tile has a fixed tap phase) and patch geometry
  u32x4 Afr[(SPL && !SPLW) ? 3 : 1][(SPL && !SPLW) ? 4 : 1][(SPL && !SPLW) ? 2 : 1];
  const int pls = p.NIMG * p.NRI * p.PW * SPLIT_PITCH;  // bytes per bf16 plane
  int pix0 = 0;
  if constexpr (SPL) pix0 = s0.base / p.P;
  if constexpr (SPL && !SPLW) {
    const OdinRun WR = odin_run(p.w, (unsigned)((size_t)p.KH * p.KW * p.CI * p.CO * 4));
    const int co = n0 + l31;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int wt = (s0.kh0 + 2 * (tap >> 1)) * 4 + (s0.kw0 + 2 * (tap & 1));
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        const int ci0 = kg * 16 + h * 8;
        float f[8];
        if (p.wmode == 1) {  // [tap][out][reduce]: 8 consecutive floats
          const unsigned off = co < p.CO ? (unsigned)(((wt * p.CO + co) * p.CI + ci0) * 4) : ODIN_OOB;
          const float4 u0 = odin_run_load4(WR, off), u1 = odin_run_load4(WR, off == ODIN_OOB ? off : off + 16);
          f[0] = u0.x; f[1] = u0.y; f[2] = u0.z; f[3] = u0.w;
          f[4] = u1.x; f[5] = u1.y; f[6] = u1.z; f[7] = u1.w;
        } else {             // [tap][reduce][out]
#pragma unroll
          for (int j = 0; j < 8; ++j)
            f[j] = odin_run_load1(WR, co < p.CO ? (unsigned)(((wt * p.CI + ci0 + j) * p.CO + co) * 4) : ODIN_OOB);
        }
        split_fragment(f, Afr[0][tap][kg], Afr[1][tap][kg], Afr[2][tap][kg]);
      }
    }
  }

  SV pf[RPWMAX * KMAX];
  int tile = blockIdx.x;
  if (pipelined && tile < p.n_tiles) {
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = (MODE == MODE_F) ? oh0 * p.S + p.ih_off : oh0 / p.S + p.ih_off;
    stage_issue<KMAX, RPWMAX, VEC, NW, (EPI == 3)>(p, LS, wave, tid, 0, b0, ih_lo, 0, pf);
  }
  ODIN_STAMP(3);

  for (; tile < p.n_tiles; tile += gridDim.x) {
    ODIN_STAMP(4);
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = (MODE == MODE_F) ? oh0 * p.S + p.ih_off : oh0 / p.S + p.ih_off;
    f32x16 acc0 = f32x16_zero(), acc1 = f32x16_zero();
    slot_set_tile(p, s0, gr0);
    slot_set_tile(p, s1, gr0);
    // epilogue operands that come from HBM (activation-derivative operand of the
    // data-gradients, Bernoulli target of the fused tail) are requested BEFORE the MFMA
    // loop: a load issued in the epilogue exposes its full latency every tile
    float4 axp[(EPI == 2) ? NMT : 1][4];
    float tgt[TAIL ? NMT : 1][NC1];
    if constexpr (EPI == 2 || TAIL > 0) {
#pragma unroll
      for (int mi = 0; mi < NMT; ++mi) {
        const int mt = mi == 0 ? mt0 : mt1;
        const Slot& s = mi == 0 ? s0 : s1;
        const bool lv = mt < p.MT && s.opix >= 0;
        if constexpr (EPI == 2) {
          const float* auxp = p.aux + (lv ? (size_t)((unsigned)s.opix * (unsigned)p.CO) : 0) + n0 + 4 * h;
#pragma unroll
          for (int q = 0; q < 4; ++q) axp[mi][q] = *reinterpret_cast<const float4*>(auxp + 8 * q);
        }
        if constexpr (TAIL > 0) {
#pragma unroll
          for (int oc = 0; oc < NC1; ++oc)
            tgt[mi][oc] = tp.target[(lv ? (size_t)s.opix * tp.C1 : 0) + (oc < tp.C1 ? oc : 0)];
        }
      }
    }
    if (pipelined) {
      __syncthreads();  // everyone is done reading the previous patch
      if constexpr (SPL) {
        if (wave < p.NIMG * p.NRI)
          stage_row_commit_split<KMAX>(p, LS, wave, reinterpret_cast<const float4*>(pf),
                                       reinterpret_cast<char*>(patch), pls);
      } else {
        stage_commit<KMAX, RPWMAX, VEC, NW, (EPI == 3)>(p, LS, wave, tid, 0, b0, ih_lo, pf, patch);
      }
      __syncthreads();
      ODIN_STAMP(5);
      const int nt = tile + gridDim.x;
      if (nt < p.n_tiles) {  // loads of the next tile fly during the MFMAs below
        const int g2 = nt * p.TR;
        const int b2 = g2 / p.OH, o2 = g2 - b2 * p.OH;
        const int ih2 = (MODE == MODE_F) ? o2 * p.S + p.ih_off : o2 / p.S + p.ih_off;
        stage_issue<KMAX, RPWMAX, VEC, NW, (EPI == 3)>(p, LS, wave, tid, 0, b2, ih2, 0, pf);
      }
      ODIN_STAMP(6);
      constexpr int SGK = (TAIL > 0 && RPWMAX == 1) ? 8 : 16;
      if constexpr (SPLW) {
        if (mt0 < p.MT)
          acc0 = mtile_compute_split_lds(p, reinterpret_cast<const char*>(patch), pls, pix0, l31, h,
                                         ODIN_UNIFORM(s0.kh0), ODIN_UNIFORM(s0.kw0),
                                         reinterpret_cast<const char*>(wl));
      } else if constexpr (SPL) {
        if (mt0 < p.MT)
          acc0 = mtile_compute_split(p, reinterpret_cast<const char*>(patch), pls, pix0, h, Afr);
      } else if (mt0 < p.MT) {
        acc0 = mtile_compute<MODE, TK, TS, TCIC, SGK>(p, patch, wl, s0, l31, h, acc0, kc0, kc1);
      }
      if (NMT > 1 && mt1 < p.MT) acc1 = mtile_compute<MODE, TK, TS, TCIC, SGK>(p, patch, wl, s1, l31, h, acc1);
    } else {
      for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int c0 = ch * p.CIC;
        __syncthreads();
        for (int bt = 0; bt < p.n_batches; ++bt) {  // KMAX*RPWMAX items per thread in flight
          stage_issue<KMAX, RPWMAX, VEC, NW, (EPI == 3)>(p, LS, wave, tid, bt, b0, ih_lo, c0, pf);
          stage_commit<KMAX, RPWMAX, VEC, NW, (EPI == 3)>(p, LS, wave, tid, bt, b0, ih_lo, pf, patch);
        }
        if (!p.w_resident) stage_weights(p, wl, c0, n0, tid, NT);
        __syncthreads();
        if (mt0 < p.MT) acc0 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s0, l31, h, acc0, kc0, kc1);
        if (NMT > 1 && mt1 < p.MT) acc1 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s1, l31, h, acc1);
      }
    }
    ODIN_STAMP(7);
    if (TCIC == 0 && KS > 1) {
      // combine the KS partial accumulators of each M-tile through LDS (patch area is free)
      __syncthreads();
      float* red = smem;  // [NW][16][64]
#pragma unroll
      for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc0[i];
      __syncthreads();
      if (kpart == 0) {
        for (int k = 1; k < KS; ++k) {
#pragma unroll
          for (int i = 0; i < 16; ++i) acc0[i] += red[((wave + k) * 16 + i) * 64 + lane];
        }
      }
    }
    // ---- epilogue: bias + activation (+ activation-gradient multiply) + NHWC store ----
    float llk_lane = 0.f;
#pragma unroll
    for (int mi = 0; mi < NMT; ++mi) {
      const int mt = mi == 0 ? mt0 : mt1;
      const Slot& s = mi == 0 ? s0 : s1;
      const f32x16& acc = mi == 0 ? acc0 : acc1;
      if (mt >= p.MT || kpart != 0) continue;  // wave-uniform
      const bool live = s.opix >= 0;
      float* outp = p.out + (live ? (size_t)((unsigned)s.opix * (unsigned)p.CO) : 0) + n0 + 4 * h;
      float v[16];
      float bias_l[16];
      if constexpr (TAIL > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(tailc + 8 * q + 4 * h);
          bias_l[4 * q] = t4.x; bias_l[4 * q + 1] = t4.y; bias_l[4 * q + 2] = t4.z; bias_l[4 * q + 3] = t4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) bias_l[i] = bias_r[i];
      }
      if constexpr (EPI == 1) {
        // branch-free ELU on channel pairs: max(t, 0) + (2^(min(t, 0) * log2 e) - 1); the second
        // term is exactly 0 for t > 0.  (Masked lanes compute finite garbage: every use below is
        // guarded by `live` or multiplied by a zero gradient.)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x2 t = odin_f2(acc[2 * j], acc[2 * j + 1]) + odin_f2(bias_l[2 * j], bias_l[2 * j + 1]);
          const f32x2 u = odin_f2(fminf(t.x, 0.f), fminf(t.y, 0.f)) * 1.44269504088896341f;
          const f32x2 e = odin_f2(odin_exp2(u.x), odin_exp2(u.y)) - 1.f;
          const f32x2 y = odin_f2(fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)) + e;
          v[2 * j] = y.x;
          v[2 * j + 1] = y.y;
        }
      } else if constexpr (EPI == 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[i] + bias_l[i];
      } else {
        const int act = p.act;
        if (act == ODIN_ACT_ELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
            const float t = acc[i] + bias_l[i];
            v[i] = (n < p.CO && live) ? (t > 0.f ? t : odin_exp(t) - 1.f) : 0.f;
          }
        } else if (act == ODIN_ACT_RELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
            const float t = acc[i] + bias_l[i];
            v[i] = (n < p.CO && live) ? fmaxf(t, 0.f) : 0.f;
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
            v[i] = (n < p.CO && live) ? acc[i] + bias_l[i] : 0.f;
          }
        }
      }
      if (TAIL) {
        // 1x1 conv: each pixel's 32 channels live in lanes (l31, h=0) and (l31, h=1)
        float lg[NC1], dl[NC1];
        f32x2 w1r[8][NC1], vp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) vp[j] = odin_f2(v[2 * j], v[2 * j + 1]);
#pragma unroll
        for (int oc = 0; oc < NC1; ++oc) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 t4 = *reinterpret_cast<const float4*>(tailc + 32 + 32 * oc + 8 * q + 4 * h);
            w1r[2 * q][oc] = odin_f2(t4.x, t4.y);
            w1r[2 * q + 1][oc] = odin_f2(t4.z, t4.w);
          }
        }
#pragma unroll
        for (int oc = 0; oc < NC1; ++oc) {
          f32x2 t2 = odin_f2(0.f, 0.f);
#pragma unroll
          for (int j = 0; j < 8; ++j) t2 += vp[j] * w1r[j][oc];
          float t = t2.x + t2.y;
          t += __shfl_xor(t, 32);
          lg[oc] = t + b1r[oc];
        }
        const float sc = tp.scale[0];
#pragma unroll
        for (int oc = 0; oc < NC1; ++oc) {
          dl[oc] = 0.f;
          if (oc < tp.C1 && live) {
            const float x = tgt[TAIL ? mi : 0][oc];
            const float l = lg[oc];
            const float dsig = (sigmoid_g(l) - x) * sc;
            if (h == 0) {
              llk_lane += x * l - softplus_g(l);
              db1[oc] += dsig;
              if (tp.logits != nullptr) tp.logits[(size_t)s.opix * tp.C1 + oc] = l;
            }
            dl[oc] = dsig;
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f32x2 g = odin_f2(0.f, 0.f);
#pragma unroll
          for (int oc = 0; oc < NC1; ++oc) {
            g += w1r[j][oc] * dl[oc];
            dw1[j][oc] += vp[j] * dl[oc];
          }
          if constexpr (EPI == 1) {
            // ELU' from the output y: 1 (y > 0) or y + 1 (y <= 0) == 1 + min(y, 0)
            const f32x2 gp = g * odin_f2(fminf(vp[j].x, 0.f), fminf(vp[j].y, 0.f)) + g;
            v[2 * j] = gp.x;
            v[2 * j + 1] = gp.y;
          } else {
            v[2 * j] = g.x * odin_act_grad(p.act, vp[j].x);
            v[2 * j + 1] = g.y * odin_act_grad(p.act, vp[j].y);
          }
        }
      } else if (EPI == 2 && live) {
        const float4* ax = axp[(EPI == 2) ? mi : 0];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // ELU'(y) = 1 + min(y, 0)
          v[4 * q + 0] = fmaf(v[4 * q + 0], fminf(ax[q].x, 0.f), v[4 * q + 0]);
          v[4 * q + 1] = fmaf(v[4 * q + 1], fminf(ax[q].y, 0.f), v[4 * q + 1]);
          v[4 * q + 2] = fmaf(v[4 * q + 2], fminf(ax[q].z, 0.f), v[4 * q + 2]);
          v[4 * q + 3] = fmaf(v[4 * q + 3], fminf(ax[q].w, 0.f), v[4 * q + 3]);
        }
      } else if ((EPI == 0 || EPI == 3) && p.aux != nullptr && live) {
        const float* auxp = p.aux + (size_t)((unsigned)s.opix * (unsigned)p.CO) + n0 + 4 * h;
        if (co_vec) {
          float4 ax[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            ax[q] = (n0 + 8 * q + 4 * h + 3 < p.CO) ? *reinterpret_cast<const float4*>(auxp + 8 * q)
                                                    : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * q + 0] *= odin_act_grad(p.aux_act, ax[q].x);
            v[4 * q + 1] *= odin_act_grad(p.aux_act, ax[q].y);
            v[4 * q + 2] *= odin_act_grad(p.aux_act, ax[q].z);
            v[4 * q + 3] *= odin_act_grad(p.aux_act, ax[q].w);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int nn = 8 * (i >> 2) + (i & 3);
            if (n0 + 4 * h + nn < p.CO) v[i] *= odin_act_grad(p.aux_act, auxp[nn]);
          }
        }
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bsum[i] += v[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 8 * q + 4 * h;
          if ((EPI == 1 || EPI == 2) || (co_vec && n + 3 < p.CO)) {
            *reinterpret_cast<float4*>(outp + 8 * q) =
                make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n + j < p.CO) outp[8 * q + j] = v[4 * q + j];
          }
        }
      }
    }
    if (TAIL) {
      // one log-likelihood partial per tile (a tile lies inside one sample)
      float t = wave_sum64(llk_lane);
      __syncthreads();
      float* red = wl + wfloats;  // scratch after the weights
      if (lane == 0) red[wave] = t;
      __syncthreads();
      if (tid == 0) {
        float a = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) a += red[w2];
        tp.llk_part[tile] = a;
      }
    }
  }

  ODIN_STAMP(8);
  if (p.colsum_slab != nullptr || TAIL) {
    // per-workgroup partial sums: column sums of `out` (bias gradient of a
    // Conv2DTranspose layer) and, for the fused tail, dW1 / db1 of the 1x1 conv.
    // Reduce the 32 pixel lanes by shuffles, the NW waves through LDS.
    __syncthreads();
    float* red = smem;  // [NW][32 * (1 + MAXC1) + MAXC1]
    constexpr int RW = 32 * (1 + MAXC1) + MAXC1;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (l31 == 0) red[wave * RW + 8 * (i >> 2) + 4 * h + (i & 3)] = v;
    }
    if (TAIL) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int oc = 0; oc < NC1; ++oc) {
          float v = dw1[i >> 1][oc][i & 1];
#pragma unroll
          for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
          if (l31 == 0) red[wave * RW + 32 + (8 * (i >> 2) + 4 * h + (i & 3)) * MAXC1 + oc] = v;
        }
      }
#pragma unroll
      for (int oc = 0; oc < NC1; ++oc) {
        float v = db1[oc];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave * RW + 32 * (1 + MAXC1) + oc] = v;
      }
    }
    __syncthreads();
    if (!TAIL) {
      if (tid < 32) {
        float t = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) t += red[w2 * RW + tid];
        if (n0 + tid < p.CO) p.colsum_slab[(size_t)blockIdx.x * p.CO + n0 + tid] = t;
      }
    } else {
      // slab row: [CO*C1 (dW1, layout [c][oc]) | C1 (db1) | CO (column sums of out)]
      float* row = tp.slab + (size_t)blockIdx.x * (p.CO * tp.C1 + tp.C1 + p.CO);
      for (int e = tid; e < RW; e += NT) {
        float t = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) t += red[w2 * RW + e];
        if (e < 32) {
          if (e < p.CO) row[p.CO * tp.C1 + tp.C1 + e] = t;
        } else if (e < 32 * (1 + MAXC1)) {
          int c = (e - 32) / MAXC1, oc = (e - 32) % MAXC1;
          if (c < p.CO && oc < tp.C1) row[c * tp.C1 + oc] = t;
        } else {
          int oc = e - 32 * (1 + MAXC1);
          if (oc < tp.C1) row[p.CO * tp.C1 + oc] = t;
        }
      }
    }
  }
}

// --------------------------------------------------------------------------------------
// host-side planner
// --------------------------------------------------------------------------------------
constexpr int LDS_BUDGET_FLOATS = (160 * 1024 - 2048) / 4;
constexpr int NW_G = 4;
constexpr int GENERIC_KMAX = 9;
constexpr int W_SCRATCH = 64 + 32 * (1 + 4);  // floats after the weight slice (MAXC1 = 4)

bool plan_gather(GParams& p, int mode, int max_blocks, int* grid_x, size_t* lds_bytes,
                 int target = 32 * NW_G) {
  const int S = p.S;
  if (mode == MODE_T && (p.OW % S != 0 || p.OH % S != 0)) return false;
  const int img_pix = p.OH * p.OW;
  const int TARGET = target;  // 128: one M-tile per wave; 32 / 64: waves split the reduction
  p.KS = 1;
  if (img_pix <= TARGET) {
    p.NIMG = TARGET / img_pix;
    if (p.NIMG > p.B) p.NIMG = p.B;
    if (p.NIMG < 1) p.NIMG = 1;
    p.RPI = p.OH;
    p.TR = p.NIMG * p.OH;
  } else {
    p.NIMG = 1;
    // prefer the largest tile <= TARGET, otherwise the smallest legal one
    int pick = 0;
    for (int tr = 1; tr <= p.OH; ++tr) {
      if (p.OH % tr) continue;
      if (mode == MODE_T && tr % S) continue;
      if (tr * p.OW <= TARGET) pick = tr;
    }
    if (pick == 0) {
      for (int tr = 1; tr <= p.OH && pick == 0; ++tr) {
        if (p.OH % tr) continue;
        if (mode == MODE_T && tr % S) continue;
        pick = tr;
      }
    }
    p.TR = p.RPI = pick;
  }
  const int total_rows = p.B * p.OH;
  p.n_tiles = (total_rows + p.TR - 1) / p.TR;
  if (mode == MODE_F) {
    p.NRI = (p.RPI - 1) * S + p.KH;
    p.PW = (p.OW - 1) * S + p.KW;
    p.ih_off = -p.pt;
    p.iw_lo = -p.pl;
    int slots = p.TR * p.OW;
    p.MT = (slots + 31) / 32;
    p.MTP = p.MT;
    p.SPP = slots;
  } else {
    int lo_h = odin_floordiv(p.pt - p.KH + 1, S), lo_w = odin_floordiv(p.pl - p.KW + 1, S);
    p.ih_off = lo_h;
    p.iw_lo = lo_w;
    p.NRI = odin_floordiv(p.RPI - 1 + p.pt, S) - lo_h + 1;
    p.PW = odin_floordiv(p.OW - 1 + p.pl, S) - lo_w + 1;
    p.SPP = p.NIMG * (p.RPI / S) * (p.OW / S);
    p.MTP = (p.SPP + 31) / 32;
    p.MT = p.MTP * S * S;
  }
  if (p.MT > 2 * NW_G) return false;
  const int CIp = (p.CI + 1) & ~1;
  p.WP = 32;
  const int ntaps = p.KH * p.KW;
  int cic = CIp;
  // keep float4 staging possible when CI % 4 == 0
  const int gran = ((p.CI & 3) == 0) ? 4 : 2;
  const bool vec0 = ((p.CI & 3) == 0);
  while (true) {
    int P = cic + 1;
    long pf = (long)p.NIMG * p.NRI * p.PW * P + 8;
    long wf = (long)ntaps * cic * p.WP + W_SCRATCH;
    long rowlen = (long)p.PW * ((vec0 && (cic & 3) == 0) ? cic / 4 : cic);
    const bool flat0 = (p.PW == 1 && p.NRI == 1);
    // (longest patch row an instance stages: the generic ones KMAX = 9 items per lane, the wide-row 4x4 / stride-2
    // strided-gather instance over 32 channels 12)
    const int kmax_row = (mode == MODE_F && p.KH == 4 && p.KW == 4 && S == 2 && p.CI == 32 && cic == 32 &&
                          !ODIN_DIAG_ENV("ODIN_NOWIDEROWS")) ? 12 : GENERIC_KMAX;
    if (pf + wf <= LDS_BUDGET_FLOATS && (flat0 || rowlen <= 64 * kmax_row)) break;
    if (cic <= gran) return false;
    // next smaller chunk: halve, rounded up to the granularity
    int nc = (CIp + cic - 1) / cic + 1;
    int ncic = ((CIp + nc - 1) / nc + gran - 1) / gran * gran;
    if (ncic >= cic) ncic = cic - gran;
    cic = ncic;
  }
  p.CIC = cic;
  p.P = cic + 1;
  p.n_chunks = (CIp + cic - 1) / cic;
  p.w_resident = (p.n_chunks == 1) ? 1 : 0;
  p.vec = (vec0 && (cic & 3) == 0) ? 1 : 0;
  p.KI = (p.PW * (p.vec ? cic / 4 : cic) + 63) / 64;
  p.pipelined = 0;
  p.flat = (p.PW == 1 && p.NRI == 1) ? 1 : 0;
  if (p.flat) p.KI = 1;
  p.patch_floats = (int)(((long)p.NIMG * p.NRI * p.PW * p.P + 8 + 3) & ~3L);
  long wf = (long)ntaps * cic * p.WP + W_SCRATCH;  // + scratch: per-tile reductions, tail constants
  long total = p.patch_floats + wf;
  if (total < 4096 + 64) total = 4096 + 64;  // room for the split-K / tail reductions
  *lds_bytes = (size_t)total * 4;
  if (max_blocks < 0) {  // slab-producing launch: rows are bounded
    int cap = -max_blocks;
    // one slab row per workgroup; two workgroups per CU only where the LDS allows it
    if (*lds_bytes * 2 > 160 * 1024 && cap > odin_num_cus()) cap = odin_num_cus();
    *grid_x = p.n_tiles < cap ? p.n_tiles : cap;
  } else {
    int per_cu = (int)((160 * 1024) / (*lds_bytes));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int cap = max_blocks * per_cu;
    int gx = p.n_tiles < cap ? p.n_tiles : cap;
    *grid_x = gx < 1 ? 1 : gx;
  }
  return true;
}

long long* g_stamps = nullptr;

template <int MODE, int TK, int TS, int TCIC, bool VEC, int TAIL, int KMAX, int RPWMAX, int EPI, int NMT,
          bool SPL = false, int NWL = 4>
int launch_inst2(GParams& p, const TailParams& tp, dim3 grid, size_t lds, void* stream) {
  if (p.KI > KMAX) return odin_fail(-2, "gather_conv: patch row too long for this instance");
  const int rpw = (p.NIMG * p.NRI + NWL - 1) / NWL;
  if (EPI == 3) {
    const int items = p.NIMG * (p.vec ? p.CIC / 4 : p.CIC);
    p.n_batches = (items + KMAX * RPWMAX * NWL * 64 - 1) / (KMAX * RPWMAX * NWL * 64);
  } else {
    p.n_batches = (rpw + RPWMAX - 1) / RPWMAX;
  }
  p.pipelined = (p.n_chunks == 1 && p.n_batches == 1) ? 1 : 0;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(
            &gather_conv_kernel<MODE, NWL, TK, TS, TCIC, VEC, TAIL, KMAX, RPWMAX, EPI, NMT, SPL>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((gather_conv_kernel<MODE, NWL, TK, TS, TCIC, VEC, TAIL, KMAX, RPWMAX, EPI, NMT, SPL>), grid,
              dim3(NWL * 64), lds, stream, p, tp);
  return odin_check_launch(SPL ? "gather_conv(bf16x3)" : "gather_conv");
}

template <int MODE, int TK, int TS, int TCIC, bool VEC, int TAIL, int KMAX, int RPWMAX, int EPI = 0>
int launch_inst(GParams& p, const TailParams& tp, dim3 grid, size_t lds, void* stream) {
  // one M-tile per wave (tiles of <= 128 pixels) drops the second accumulator and epilogue
  if (p.MT <= NW_G)
    return launch_inst2<MODE, TK, TS, TCIC, VEC, TAIL, KMAX, RPWMAX, EPI, 1>(p, tp, grid, lds, stream);
  return launch_inst2<MODE, TK, TS, TCIC, VEC, TAIL, KMAX, RPWMAX, EPI, 2>(p, tp, grid, lds, stream);
}

// Split path (three bf16 planes, register-resident weights): eligibility and LDS size.
// ODIN_SPLIT=1 selects the first form of the split path (4-wave workgroups, weight fragments in 96
// registers): correct, but slower than the fp32 instances -- the two-workgroups-per-CU variants
// spill (103-122 registers) and the one-workgroup fused tail loses the wait hiding of its partner
// (144.6 vs 121.8 us).  The default is the 8-wave form with the weight planes in LDS (split8 in
// launch_gather: fused tail 116.7 -> 89.4 us).
bool split_ok(const GParams& p) {
  const char* e = ODIN_DIAG_ENV("ODIN_SPLIT");
  return e != nullptr && e[0] == '1' && p.CI == 32 && !p.center && p.MT <= NW_G && p.w_resident;
}
size_t split_lds(GParams& p) {
  p.patch_floats = (int)((((long)3 * p.NIMG * p.NRI * p.PW * SPLIT_PITCH) / 4 + 3) & ~3L);
  long total = p.patch_floats + W_SCRATCH;
  if (total < 4096 + 64) total = 4096 + 64;
  return (size_t)total * 4;
}

int launch_gather(int mode, GParams& p, void* stream, int max_blocks, int* rows_out = nullptr,
                  const TailParams* tail = nullptr) {
  int gx;
  size_t lds;
  if (!plan_gather(p, mode, max_blocks, &gx, &lds)) return odin_fail(-2, "gather_conv: no tiling plan");
  // small-M layers: with 128-pixel tiles only a few workgroups exist while each carries a long
  // reduction -> use 32- (or 64-) pixel tiles and let 4 (2) waves split the channels
  if (mode == MODE_F && tail == nullptr && !ODIN_DIAG_ENV("ODIN_NOKSPLIT")) {
    const int blocks = p.n_tiles * ((p.CO + 31) / 32);
    const long kdepth = (long)p.KH * p.KW * p.CI;
    if (blocks * 2 <= odin_num_cus() && kdepth >= 256) {
      for (int tgt = 32; tgt <= 64; tgt *= 2) {
        GParams q = p;
        int gx2;
        size_t lds2;
        if (!plan_gather(q, mode, max_blocks, &gx2, &lds2, tgt)) continue;
        const int ks = NW_G / (q.MT > 0 ? q.MT : 1);
        if (q.MT * ks != NW_G || ks < 2 || (q.CIC % (2 * ks)) != 0) continue;
        if (q.n_tiles * ((q.CO + 31) / 32) <= blocks) continue;
        q.KS = ks;
        p = q; gx = gx2; lds = lds2;
        break;
      }
    }
  }
  p.wdma = (p.wmode == 0 && p.w_resident && (p.CO % 32) == 0 && p.CI == p.CIC && (p.CIC % 8) == 0 &&
            !ODIN_DIAG_ENV("ODIN_NOWDMA")) ? 1 : 0;
  // opt-in ODIN_SPLIT=8: 8-wave workgroups (two waves per SIMD) on 256-pixel tiles, fp32 through
  // the bf16 pipe with the three weight planes shared in LDS
  int split8 = 0;  // 1: EPI 1, 2: EPI 2, 3: fused tail (EPI 1, one logit map)
  {
    const char* e = ODIN_DIAG_ENV("ODIN_SPLIT");
    const bool on8 = (e == nullptr || e[0] == '8') && !odin_exact_fp32();  // default; ODIN_EXACT_FP32: fp32 MFMA instances
    const bool elu_fwd = p.act == ODIN_ACT_ELU && p.aux == nullptr;
    const bool lin_bwd = p.act == ODIN_ACT_LINEAR && p.aux != nullptr && p.aux_act == ODIN_ACT_ELU;
    if (on8 && mode == MODE_T && p.KH == 4 && p.KW == 4 && p.S == 2 &&
        p.CI == 32 && (p.CO % 32) == 0 && !p.center && (elu_fwd || lin_bwd) &&
        (tail == nullptr || (elu_fwd && (tail->C1 == 1 || tail->C1 == 3) && p.CO == 32))) {
      GParams q = p;
      int gx2;
      size_t l2;
      if (plan_gather(q, mode, max_blocks, &gx2, &l2, 256) && q.MT == 8 && q.n_chunks == 1 && q.vec &&
          q.KI <= 5 && q.NIMG * q.NRI <= 8 && q.NIMG == 1) {
        q.patch_floats = (int)((((long)3 * q.NIMG * q.NRI * q.PW * SPLIT_PITCH) / 4 + 3) & ~3L);
        const long total = (long)q.patch_floats + (3 * SPLIT_WPLANE) / 4 + W_SCRATCH;
        if (total * 4 <= 158 * 1024) {
          int cap = odin_num_cus();
          if (max_blocks < 0 && -max_blocks < cap) cap = -max_blocks;
          p = q;
          gx = p.n_tiles < cap ? p.n_tiles : cap;
          lds = (size_t)total * 4;
          split8 = tail != nullptr ? (tail->C1 == 1 ? 3 : 4) : (elu_fwd ? 1 : 2);
        }
      }
    }
  }
  // 64 reduction channels (weight slice 128 KB, one 4-wave workgroup per CU): an 8-wave workgroup
  // on a 256-pixel tile shares the slice, so each SIMD holds two waves that hide each other's
  // staging and barrier waits (what two workgroups per CU do for the 32-channel instances)
  int wave8 = 0;  // 1: forward (ELU epilogue)
  if (split8 == 0 && tail == nullptr && mode == MODE_T && p.KH == 4 && p.KW == 4 && p.S == 2 &&
      p.CI == 64 && (p.CO % 32) == 0 && !p.center && !ODIN_DIAG_ENV("ODIN_NOWAVE8")) {
    const bool elu_fwd = p.act == ODIN_ACT_ELU && p.aux == nullptr;
    GParams q = p;
    int gx2;
    size_t l2;
    // (the data-gradient epilogue variant does not fit 256 registers without spilling: measured
    // slower, so only the forward instance is routed here)
    if (elu_fwd && plan_gather(q, mode, max_blocks, &gx2, &l2, 256) && q.MT == 8 &&
        q.n_chunks == 1 && q.CIC == 64 && q.vec && q.KI <= 5 && (q.NIMG * q.NRI + 7) / 8 <= 2 &&
        l2 <= 158 * 1024) {
      const int gy = (q.CO + 31) / 32;
      int cap = odin_num_cus() / gy;
      if (cap < 1) cap = 1;
      if (max_blocks < 0 && -max_blocks < cap) cap = -max_blocks;
      p = q;
      gx = p.n_tiles < cap ? p.n_tiles : cap;
      lds = l2;
      wave8 = elu_fwd ? 1 : 2;
    }
  }
  if (rows_out) *rows_out = gx;
  if (p.out == nullptr) return 0;  // dry run: planning only
  {
    static int dbg = -1;
    if (dbg < 0) { const char* e = ODIN_DIAG_ENV("ODIN_DBG"); dbg = e ? atoi(e) : 0; }
    p.dbg = dbg;
    p.stamps = g_stamps;
  }
  dim3 grid(gx, (p.CO + 31) / 32, 1);
  TailParams tp;
  memset(&tp, 0, sizeof(tp));
  const bool k4s2 = (p.KH == 4 && p.KW == 4 && p.S == 2 && p.n_chunks == 1 && p.vec);
  const int rpw = (p.NIMG * p.NRI + NW_G - 1) / NW_G;
  constexpr int GK = GENERIC_KMAX;
  static int noepi = -1;
  if (noepi < 0) { const char* e = ODIN_DIAG_ENV("ODIN_NOEPI"); noepi = e ? atoi(e) : 0; }
  const bool fulln = (p.CO % 32) == 0 && !noepi;
  const int epi = !fulln ? 0
                  : (p.act == ODIN_ACT_ELU && p.aux == nullptr) ? 1
                  : (p.act == ODIN_ACT_LINEAR && p.aux != nullptr && p.aux_act == ODIN_ACT_ELU) ? 2 : 0;
  if (wave8 == 1) return launch_inst2<MODE_T, 4, 2, 64, true, 0, 5, 2, 1, 1, false, 8>(p, tp, grid, lds, stream);
  if (split8 != 0) {
    if (tail != nullptr) tp = *tail;
    if (split8 == 3) return launch_inst2<MODE_T, 4, 2, 32, true, 1, 5, 1, 1, 1, true, 8>(p, tp, grid, lds, stream);
    if (split8 == 4) return launch_inst2<MODE_T, 4, 2, 32, true, 3, 5, 1, 1, 1, true, 8>(p, tp, grid, lds, stream);
    if (split8 == 1) return launch_inst2<MODE_T, 4, 2, 32, true, 0, 5, 1, 1, 1, true, 8>(p, tp, grid, lds, stream);
    return launch_inst2<MODE_T, 4, 2, 32, true, 0, 5, 1, 2, 1, true, 8>(p, tp, grid, lds, stream);
  }
  if (tail != nullptr) {
    tp = *tail;
    if (p.CO > 32 || tp.C1 > MAXC1 || p.NIMG != 1 || !p.vec)
      return odin_fail(-2, "bernoulli tail: needs Cout<=32, C1<=4, Cin%4==0 and one image per tile");
    if (mode == MODE_T && k4s2 && p.CIC == 32 && p.KI <= 5 && rpw <= 1 && epi == 1 && tp.C1 == 1 &&
        !ODIN_DIAG_ENV("ODIN_NOTAIL2WG"))  // two workgroups per CU
    {
#ifdef ODIN_DIAG  // (4-wave split form: A/B reference of the diagnostics build)
      if (split_ok(p))
        return launch_inst2<MODE_T, 4, 2, 32, true, 1, 5, 1, 1, 1, true>(p, tp, grid, split_lds(p), stream);
#endif
      return launch_inst<MODE_T, 4, 2, 32, true, 1, 5, 1, 1>(p, tp, grid, lds, stream);
    }
    if (mode == MODE_T && k4s2 && p.CIC == 32 && p.KI <= 5 && epi == 1 && tp.C1 == 1)
      return launch_inst<MODE_T, 4, 2, 32, true, 1, 5, 2, 1>(p, tp, grid, lds, stream);
#ifndef ODIN_DEV_TAIL_ONLY  // (developer switch: compile only the dSprites tail instances)
    if (mode == MODE_T && k4s2 && p.CIC == 32 && p.KI <= 5 && epi == 1 && tp.C1 == 3)
      return launch_inst<MODE_T, 4, 2, 32, true, 3, 5, 2, 1>(p, tp, grid, lds, stream);
    if (mode == MODE_T && k4s2 && p.CIC == 32 && p.KI <= 5)
      return launch_inst<MODE_T, 4, 2, 32, true, 4, 5, 2, 0>(p, tp, grid, lds, stream);
    if (mode == MODE_T) return launch_inst<MODE_T, 0, 0, 0, true, 4, GK, 2>(p, tp, grid, lds, stream);
    return launch_inst<MODE_F, 0, 0, 0, true, 4, GK, 2>(p, tp, grid, lds, stream);
#endif
  }
#ifdef ODIN_DEV_TAIL_ONLY
  return -1;
#else
  if (mode == MODE_F) {
    if (k4s2 && p.CIC == 32 && p.KI <= 5 && rpw <= 5) {
      if (epi == 1) return launch_inst<MODE_F, 4, 2, 32, true, 0, 5, 5, 1>(p, tp, grid, lds, stream);
      if (epi == 2) return launch_inst<MODE_F, 4, 2, 32, true, 0, 5, 5, 2>(p, tp, grid, lds, stream);
      return launch_inst<MODE_F, 4, 2, 32, true, 0, 5, 5, 0>(p, tp, grid, lds, stream);
    }
    if (k4s2 && p.CIC == 32 && p.KI <= 9) {
      if (epi == 2) return launch_inst<MODE_F, 4, 2, 32, true, 0, 9, 3, 2>(p, tp, grid, lds, stream);
      return launch_inst<MODE_F, 4, 2, 32, true, 0, 9, 3, 0>(p, tp, grid, lds, stream);
    }
    if (k4s2 && p.CIC == 64 && p.KI <= 5) {
      if (epi == 1) return launch_inst<MODE_F, 4, 2, 64, true, 0, 5, 3, 1>(p, tp, grid, lds, stream);
      if (epi == 2) return launch_inst<MODE_F, 4, 2, 64, true, 0, 5, 3, 2>(p, tp, grid, lds, stream);
      return launch_inst<MODE_F, 4, 2, 64, true, 0, 5, 3, 0>(p, tp, grid, lds, stream);
    }
    // wider rows than the image stacks' (the speech stack: 40- and 80-pixel rows; vae_audio.py:84-110): the same
    // 4x4 / stride-2 instances with longer patch rows per lane -- these shapes used to fall to the generic
    // instance (30-40 TFLOP/s)
    if (k4s2 && p.CIC == 32 && p.KI <= 12 && !ODIN_DIAG_ENV("ODIN_NOWIDEROWS")) {
      if (epi == 1) return launch_inst<MODE_F, 4, 2, 32, true, 0, 12, 2, 1>(p, tp, grid, lds, stream);
      if (epi == 2) return launch_inst<MODE_F, 4, 2, 32, true, 0, 12, 2, 2>(p, tp, grid, lds, stream);
      return launch_inst<MODE_F, 4, 2, 32, true, 0, 12, 2, 0>(p, tp, grid, lds, stream);
    }
    if (p.flat && p.vec) return launch_inst<MODE_F, 0, 0, 0, true, 0, 2, 8, 3>(p, tp, grid, lds, stream);
    if (p.flat) return launch_inst<MODE_F, 0, 0, 0, false, 0, 2, 8, 3>(p, tp, grid, lds, stream);
    if (p.vec && p.KI <= 2) return launch_inst<MODE_F, 0, 0, 0, true, 0, 2, 8>(p, tp, grid, lds, stream);
    if (p.vec) return launch_inst<MODE_F, 0, 0, 0, true, 0, GK, 2>(p, tp, grid, lds, stream);
    if (p.KI <= 2) return launch_inst<MODE_F, 0, 0, 0, false, 0, 2, 8>(p, tp, grid, lds, stream);
    return launch_inst<MODE_F, 0, 0, 0, false, 0, GK, 2>(p, tp, grid, lds, stream);
  }
  if (k4s2 && p.CIC == 32 && p.KI <= 5 && rpw <= 1 && epi == 1 && !ODIN_DIAG_ENV("ODIN_NO2WG")) {
#ifdef ODIN_DIAG  // (4-wave split form: A/B reference of the diagnostics build)
    if (split_ok(p))
      return launch_inst2<MODE_T, 4, 2, 32, true, 0, 5, 1, 1, 1, true>(p, tp, grid, split_lds(p), stream);
#endif
    return launch_inst<MODE_T, 4, 2, 32, true, 0, 5, 1, 1>(p, tp, grid, lds, stream);
  }
  if (k4s2 && p.CIC == 32 && p.KI <= 5 && rpw <= 1 && epi == 2 && !ODIN_DIAG_ENV("ODIN_NO2WG")) {
#ifdef ODIN_DIAG  // (4-wave split form: A/B reference of the diagnostics build)
    if (split_ok(p))
      return launch_inst2<MODE_T, 4, 2, 32, true, 0, 5, 1, 2, 1, true>(p, tp, grid, split_lds(p), stream);
#endif
    return launch_inst<MODE_T, 4, 2, 32, true, 0, 5, 1, 2>(p, tp, grid, lds, stream);
  }
  if (k4s2 && p.CIC == 32 && p.KI <= 5) {
    if (epi == 1) return launch_inst<MODE_T, 4, 2, 32, true, 0, 5, 2, 1>(p, tp, grid, lds, stream);
    if (epi == 2) return launch_inst<MODE_T, 4, 2, 32, true, 0, 5, 2, 2>(p, tp, grid, lds, stream);
    return launch_inst<MODE_T, 4, 2, 32, true, 0, 5, 2, 0>(p, tp, grid, lds, stream);
  }
  if (k4s2 && p.CIC == 64 && p.KI <= 5) {
    if (epi == 1) return launch_inst<MODE_T, 4, 2, 64, true, 0, 5, 2, 1>(p, tp, grid, lds, stream);
    if (epi == 2) return launch_inst<MODE_T, 4, 2, 64, true, 0, 5, 2, 2>(p, tp, grid, lds, stream);
    return launch_inst<MODE_T, 4, 2, 64, true, 0, 5, 2, 0>(p, tp, grid, lds, stream);
  }
  if (k4s2 && (p.CIC == 32 || p.CIC == 64) && p.KI <= 8 && !ODIN_DIAG_ENV("ODIN_NOWIDEROWS")) {  // (wider rows: see MODE_F)
    if (p.CIC == 32) {
      if (epi == 1) return launch_inst<MODE_T, 4, 2, 32, true, 0, 8, 2, 1>(p, tp, grid, lds, stream);
      if (epi == 2) return launch_inst<MODE_T, 4, 2, 32, true, 0, 8, 2, 2>(p, tp, grid, lds, stream);
      return launch_inst<MODE_T, 4, 2, 32, true, 0, 8, 2, 0>(p, tp, grid, lds, stream);
    }
    if (epi == 1) return launch_inst<MODE_T, 4, 2, 64, true, 0, 8, 2, 1>(p, tp, grid, lds, stream);
    if (epi == 2) return launch_inst<MODE_T, 4, 2, 64, true, 0, 8, 2, 2>(p, tp, grid, lds, stream);
    return launch_inst<MODE_T, 4, 2, 64, true, 0, 8, 2, 0>(p, tp, grid, lds, stream);
  }
  if (p.vec && p.KI <= 2) return launch_inst<MODE_T, 0, 0, 0, true, 0, 2, 8>(p, tp, grid, lds, stream);
  if (p.vec) return launch_inst<MODE_T, 0, 0, 0, true, 0, GK, 2>(p, tp, grid, lds, stream);
  return launch_inst<MODE_T, 0, 0, 0, false, 0, GK, 2>(p, tp, grid, lds, stream);
#endif
}

void fill_common(GParams& p, const odin_conv_desc* d) {
  memset(&p, 0, sizeof(p));
  p.B = d->B;
  p.KH = d->KH;
  p.KW = d->KW;
  p.S = d->stride;
  p.pt = d->pad_t;
  p.pl = d->pad_l;
}

}  // namespace

extern "C" int odin_max_slab_rows(void) { return ODIN_MAX_COLSUM_BLOCKS; }

// The forward half of the range contract (include/odin_hip.h: odin_conv_desc.x_amax / y_amax): a layer that is handed
// a word for its output leaves a valid bound in it -- from the epilogue of the plane / implicit-GEMM / first-layer
// families, by one pass over y behind the others.
static int track_y(int rc, const float* y, const odin_conv_desc* d, void* stream) {
  if (rc != 0 || y == nullptr || d->y_amax == nullptr) return rc;
  return odin_absmax_fold(y, (size_t)d->B * d->OH * d->OW * d->Cout, d->y_amax, stream);
}

// ---- Conv2D -------------------------------------------------------------------------
extern "C" int odin_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                               const odin_conv_desc* d, void* stream) {
  if (odin_smallc_applicable(d)) return odin_smallc_fwd(x, w, bias, y, d, stream);   // (tracks y itself)
  if (odin_pw1x1_applicable(d)) return track_y(odin_pw1x1_fwd(x, w, bias, y, d, stream), y, d, stream);
  if (d->act == ODIN_ACT_ELU && bias != nullptr &&
      odin_fconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                                   d->pad_t, d->pad_l, d->center))
    return odin_fconv_planes_launch(x, w, bias, nullptr, y, nullptr, nullptr, d->B, d->OH, d->OW, d->Cin, d->Cout,
                                    1, d->x_amax, d->y_amax, stream);
  if (d->act == ODIN_ACT_ELU && bias != nullptr &&
      odin_fconv_ring_applicable(d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                                 d->pad_t, d->pad_l, d->center))
    return track_y(odin_fconv_ring_launch(x, w, bias, nullptr, y, nullptr, nullptr, d->B, d->H, d->W, d->Cin,
                                          d->OH, d->OW, d->Cout, 1, stream), y, d, stream);
  // 5x5 / stride-1 layers (the MNIST conv stack): block windows with the weights in LDS (blk5_planes.hip)
  if (bias != nullptr && d->H == d->OH && d->W == d->OW &&
      odin_conv5_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->center))
    return odin_conv5_blk_launch(x, w, bias, nullptr, y, nullptr, nullptr, d->B, d->H, d->W, d->Cin, d->Cout, d->KH, 1, d->act,
                                 d->x_amax, d->y_amax, stream);
  if (bias != nullptr &&
      odin_fconv_blk_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                d->pad_l, d->center))
    return odin_fconv_blk_launch(x, w, bias, nullptr, y, nullptr, nullptr, d->B, d->OH, d->OW, d->Cin, d->Cout, 1,
                                 d->act, d->x_amax, d->y_amax, stream);
  if (odin_igemm_h_applicable(0, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->center))
    return odin_igemm_h_launch(0, x, w, bias, nullptr, 0, y, nullptr, d->B, d->H, d->W, d->Cin, d->OH, d->OW,
                               d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->act, d->x_amax, 0, d->y_amax,
                               stream);
  if (odin_igemm_applicable(0, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                            d->center))
    return odin_igemm_launch(0, x, w, bias, nullptr, 0, y, nullptr, d->B, d->H, d->W, d->Cin, d->OH, d->OW,
                             d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->act, d->y_amax, stream);
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 0; p.act = d->act; p.center = d->center;
  return track_y(launch_gather(MODE_F, p, stream, odin_num_cus()), y, d, stream);
}

// dx[b,ih,iw,ci] = sum_{kh,kw,co} dy[b,(ih+pt-kh)/S,(iw+pl-kw)/S,co] * W[kh,kw,ci,co];
// optionally multiplied by act'(aux) (aux = this layer's input = previous layer's output)
// THE CONTRACT: a data gradient that is handed a word (d->dx_amax) leaves a valid bound of dx in it, whatever kernel
// family ran.  The plane / implicit-GEMM families fold max|dx| in from their epilogues (free); every other family
// (generic gather, 1x1 stream kernel, the fp32 ring kernels) is followed by ONE absmax pass over dx here.  Round 4
// left those words untouched and told the caller through the *_keeps_range predicates below -- a predicate that
// disagreed with the dispatch (a column-sum slab sends a layer of > 16384 tiles to the generic kernel) handed the
// consumers a ZERO word: they scaled by 2^115 and overflowed.  The predicates remain as "kept without an extra pass".
static int track_dx(int rc, const float* dx, const odin_conv_desc* d, void* stream) {
  if (rc != 0 || dx == nullptr || d->dx_amax == nullptr) return rc;
  return odin_absmax_fold(dx, (size_t)d->B * d->H * d->W * d->Cin, d->dx_amax, stream);
}

// 1: the data gradient of this layer (as dispatched for `aux_act`, with the aux tensor present and NO column-sum slab
// beyond ODIN_MAX_COLSUM_BLOCKS tiles) folds max|dx| into d->dx_amax in its own epilogue; 0: by a pass of its own
extern "C" int odin_conv2d_dgrad_keeps_range(const odin_conv_desc* d, int aux_act) {
  if (odin_pw1x1_applicable(d)) return 0;
  if (aux_act == ODIN_ACT_ELU && d->H == 2 * d->OH && d->W == 2 * d->OW &&
      odin_tconv_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, 0, 2, 1))
    return 1;
  if (aux_act == ODIN_ACT_ELU && d->H == 2 * d->OH && d->W == 2 * d->OW &&
      odin_tconv_ring_applicable(d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0))
    return 0;
  if (d->H == 2 * d->OH && d->W == 2 * d->OW &&
      odin_tconv_blk_applicable(d->B, d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0))
    return 1;
  if (d->H == d->OH && d->W == d->OW &&
      odin_conv5_blk_applicable(d->B, d->H, d->W, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0))
    return 1;
  if (odin_igemm_h_applicable(1, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0)) return 1;
  return odin_igemm_applicable(1, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0) ? 1 : 0;
}
extern "C" int odin_deconv2d_dgrad_keeps_range(const odin_conv_desc* d, int aux_act) {
  if (odin_smalldeconv_applicable(d)) return 1;
  if (aux_act == ODIN_ACT_ELU &&
      odin_fconv_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride,
                                   d->pad_t, d->pad_l, 0))
    return 1;
  if (odin_fconv_blk_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                d->pad_l, 0))
    return 1;
  if (odin_igemm_h_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0)) return 1;
  const bool ring_two_pass_vs_igemm =
      d->Cout == 64 && odin_igemm_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW,
                                             d->stride, 0) &&
      odin_igemm_tiles(0, d->B, d->H, d->W, d->stride) <= ODIN_MAX_COLSUM_BLOCKS;
  if (!ring_two_pass_vs_igemm && aux_act == ODIN_ACT_ELU &&
      odin_fconv_ring_applicable(d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                 d->pad_l, 0))
    return 0;
  return odin_igemm_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0) ? 1 : 0;
}
// 1: the fused tail folds max|g_out| into d->dy_amax itself
extern "C" int odin_bernoulli_tail_keeps_range(int is_deconv, const odin_conv_desc* d, int C1) {
  return (is_deconv && d->act == ODIN_ACT_ELU && d->OH == 2 * d->H && d->OW == 2 * d->W &&
          odin_tconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                       d->pad_l, d->center, 3, C1)) ? 1 : 0;
}

extern "C" int odin_conv2d_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                 float* dx, float* colsum_slab, int* slab_rows_out,
                                 const odin_conv_desc* d, void* stream) {
  if (odin_pw1x1_applicable(d))
    return track_dx(odin_pw1x1_dgrad(dy, w, aux, aux_act, dx, colsum_slab, slab_rows_out, d, stream), dx, d, stream);
  // data gradient of a Conv2D = transposed gather over dY: input (OH, OW, Cout), output (H, W, Cin)
  if (aux_act == ODIN_ACT_ELU && (aux != nullptr || dx == nullptr) && d->H == 2 * d->OH &&
      d->W == 2 * d->OW &&
      odin_tconv_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, 0, 2, 1))
    return odin_tconv_planes_launch(dy, w, nullptr, aux, dx, colsum_slab, slab_rows_out, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d->B, d->OH,
                                    d->OW, d->Cout, d->Cin, 2, d->dy_amax, d->dx_amax, stream);
  if (aux_act == ODIN_ACT_ELU && (aux != nullptr || dx == nullptr) && d->H == 2 * d->OH &&
      d->W == 2 * d->OW &&
      odin_tconv_ring_applicable(d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                 d->pad_l, 0))
    return track_dx(odin_tconv_ring_launch(dy, w, nullptr, aux, dx, colsum_slab, slab_rows_out, nullptr, nullptr,
                                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, d->B, d->OH,
                                           d->OW, d->Cin, 2, stream), dx, d, stream);
  if (d->H == d->OH && d->W == d->OW &&
      odin_conv5_blk_applicable(d->B, d->H, d->W, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0))
    return odin_conv5_blk_launch(dy, w, nullptr, aux_act != 0 ? aux : nullptr, dx, colsum_slab, slab_rows_out, d->B,
                                 d->H, d->W, d->Cout, d->Cin, d->KH, 2, aux_act, d->dy_amax, d->dx_amax, stream);
  // any other image size: 8 x 8 blocks of dy through LDS windows (blk_planes.hip)
  if (d->H == 2 * d->OH && d->W == 2 * d->OW &&
      odin_tconv_blk_applicable(d->B, d->OH, d->OW, d->Cout, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0))
    return odin_tconv_blk_launch(dy, w, nullptr, aux_act != 0 ? aux : nullptr, dx, colsum_slab, slab_rows_out, d->B,
                                 d->OH, d->OW, d->Cout, d->Cin, 2, aux_act, d->dy_amax, d->dx_amax, stream);
  if (odin_igemm_h_applicable(1, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0)) {
    if (slab_rows_out) *slab_rows_out = odin_igemm_h_rows(1, d->B, d->H, d->W, d->stride);
    if (dx == nullptr) return 0;  // dry run
    return odin_igemm_h_launch(1, dy, w, nullptr, aux, aux_act, dx, colsum_slab, d->B, d->OH, d->OW, d->Cout, d->H,
                               d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0, d->dy_amax, 1,
                               d->dx_amax, stream);
  }
  if (odin_igemm_applicable(1, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0) &&
      (odin_igemm_tiles(1, d->B, d->H, d->W, d->stride) <= ODIN_MAX_COLSUM_BLOCKS ||
       (colsum_slab == nullptr && dx != nullptr))) {
    if (slab_rows_out) *slab_rows_out = odin_igemm_tiles(1, d->B, d->H, d->W, d->stride);
    if (dx == nullptr) return 0;  // dry run
    return odin_igemm_launch(1, dy, w, nullptr, aux, aux_act, dx, colsum_slab, d->B, d->OH, d->OW, d->Cout,
                             d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0, d->dx_amax, stream);
  }
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 1;
  return track_dx(launch_gather(MODE_T, p, stream, colsum_slab ? -ODIN_MAX_COLSUM_BLOCKS : odin_num_cus(),
                                slab_rows_out), dx, d, stream);
}

// ---- Conv2DTranspose (desc: H,W,Cin = input; OH=H*S, OW=W*S, Cout = output; pads = the
// SAME pads of the forward conv on the OUTPUT size) ------------------------------------
extern "C" int odin_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                 const odin_conv_desc* d, void* stream) {
  if (bias != nullptr && odin_smalldeconv_applicable(d)) return odin_smalldeconv_fwd(x, w, bias, y, d, stream);
  if (d->act == ODIN_ACT_ELU && bias != nullptr && d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_tconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, d->center, 1, 1))
    return odin_tconv_planes_launch(x, w, bias, nullptr, y, nullptr, nullptr, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, nullptr, nullptr, nullptr, 1, d->B, d->H, d->W,
                                    d->Cin, d->Cout, 1, d->x_amax, d->y_amax, stream);
  if (d->act == ODIN_ACT_ELU && bias != nullptr && d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_tconv_ring_applicable(d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                 d->pad_l, d->center))
    return track_y(odin_tconv_ring_launch(x, w, bias, nullptr, y, nullptr, nullptr, nullptr, nullptr, nullptr,
                                          nullptr, nullptr, nullptr, nullptr, nullptr, 1, d->B, d->H, d->W,
                                          d->Cout, 1, stream), y, d, stream);
  // a thin small image the implicit-GEMM families cannot take (fewer than 8 channels: MNIST's first deconvolution)
  if (bias != nullptr && (d->Cin & 7) != 0 && odin_smalldeconv_gen_applicable(d))
    return odin_smalldeconv_gen_fwd(x, w, bias, y, d, stream);
  if (bias != nullptr && d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_tconv_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->center))
    return odin_tconv_blk_launch(x, w, bias, nullptr, y, nullptr, nullptr, d->B, d->H, d->W, d->Cin, d->Cout, 1, d->act,
                                 d->x_amax, d->y_amax, stream);
  if (odin_igemm_h_applicable(1, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->center))
    return odin_igemm_h_launch(1, x, w, bias, nullptr, 0, y, nullptr, d->B, d->H, d->W, d->Cin, d->OH, d->OW,
                               d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->act, d->x_amax, 0, d->y_amax,
                               stream);
  if (odin_igemm_applicable(1, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                            d->center))
    return odin_igemm_launch(1, x, w, bias, nullptr, 0, y, nullptr, d->B, d->H, d->W, d->Cin, d->OH, d->OW,
                             d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, d->act, d->y_amax, stream);
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 1; p.act = d->act; p.center = d->center;
  return track_y(launch_gather(MODE_T, p, stream, odin_num_cus()), y, d, stream);
}

extern "C" int odin_deconv2d_dgrad(const float* dy, const float* w, const float* aux,
                                   int aux_act, float* dx, float* colsum_slab,
                                   int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  // data gradient of a Conv2DTranspose = strided gather over dY: input (OH, OW, Cout), output (H, W, Cin)
  if (colsum_slab == nullptr && odin_smalldeconv_applicable(d)) {
    if (slab_rows_out) *slab_rows_out = 0;
    if (dx == nullptr) return 0;  // dry run
    return odin_smalldeconv_bwd(nullptr, dy, w, aux, aux_act, dx, nullptr, nullptr, d, stream);
  }
  if (aux_act == ODIN_ACT_ELU && (aux != nullptr || dx == nullptr) &&
      odin_fconv_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride,
                                   d->pad_t, d->pad_l, 0))
    return odin_fconv_planes_launch(dy, w, nullptr, aux, dx, colsum_slab, slab_rows_out, d->B, d->H, d->W,
                                    d->Cout, d->Cin, 2, d->dy_amax, d->dx_amax, stream);
  if (odin_fconv_blk_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                d->pad_l, 0))
    return odin_fconv_blk_launch(dy, w, nullptr, aux_act != 0 ? aux : nullptr, dx, colsum_slab, slab_rows_out, d->B, d->H,
                                 d->W, d->Cout, d->Cin, 2, aux_act, d->dy_amax, d->dx_amax, stream);
  if (odin_igemm_h_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0)) {
    if (slab_rows_out) *slab_rows_out = odin_igemm_h_rows(0, d->B, d->H, d->W, d->stride);
    if (dx == nullptr) return 0;  // dry run
    return odin_igemm_h_launch(0, dy, w, nullptr, aux, aux_act, dx, colsum_slab, d->B, d->OH, d->OW, d->Cout, d->H,
                               d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0, d->dy_amax, 1,
                               d->dx_amax, stream);
  }
  // (64 reduction channels take two fconv_ring passes: where the implicit-GEMM kernel covers the layer it does the
  // same work in one launch -- decoder2 of the dSprites stack: 30.8 us in two launches vs 30.2 us in one)
  const bool ring_two_pass_vs_igemm =
      d->Cout == 64 && odin_igemm_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW,
                                             d->stride, 0) &&
      odin_igemm_tiles(0, d->B, d->H, d->W, d->stride) <= ODIN_MAX_COLSUM_BLOCKS;
  if (!ring_two_pass_vs_igemm && aux_act == ODIN_ACT_ELU && (aux != nullptr || dx == nullptr) &&
      odin_fconv_ring_applicable(d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride,
                                 d->pad_t, d->pad_l, 0))
    return track_dx(odin_fconv_ring_launch(dy, w, nullptr, aux, dx, colsum_slab, slab_rows_out, d->B, d->OH,
                                           d->OW, d->Cout, d->H, d->W, d->Cin, 2, stream), dx, d, stream);
  if (odin_igemm_applicable(0, d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0) &&
      (odin_igemm_tiles(0, d->B, d->H, d->W, d->stride) <= ODIN_MAX_COLSUM_BLOCKS ||
       (colsum_slab == nullptr && dx != nullptr))) {
    if (slab_rows_out) *slab_rows_out = odin_igemm_tiles(0, d->B, d->H, d->W, d->stride);
    if (dx == nullptr) return 0;  // dry run
    return odin_igemm_launch(0, dy, w, nullptr, aux, aux_act, dx, colsum_slab, d->B, d->OH, d->OW, d->Cout,
                             d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t, d->pad_l, 0, d->dx_amax, stream);
  }
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 0;
  return track_dx(launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_COLSUM_BLOCKS : odin_num_cus(),
                                slab_rows_out), dx, d, stream);
}

// Dense layers whose reduction width is a multiple of 8 through the implicit-GEMM kernel (a 1x1 convolution on a
// 1x1 image; FactorVAE's 1000-unit discriminator, the 512-unit default nets; enc4 of the dSprites step:
// 12.6 + 9.2 + 7.7 -> 9.9 + 9.6 + 6.4 us stand-alone, 11 us per step in the graph); ODIN_NODENSEIGEMM: A/B switch
static bool dense_via_igemm() { return ODIN_DIAG_ENV("ODIN_NODENSEIGEMM") == nullptr; }

// ---- Dense: y[B,N] = act(x[B,K] @ w[K,N] + b) ----------------------------------------
extern "C" int odin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                              int K, int N, int act, void* stream) {
  return odin_dense_fwd_ranged(x, w, bias, y, B, K, N, act, nullptr, nullptr, stream);
}

// the same with the activation range words (include/odin_hip.h: the range contract): x_amax is read by the two-plane
// GEMM, y_amax is valid on return whatever family ran
extern "C" int odin_dense_fwd_ranged(const float* x, const float* w, const float* bias, float* y, int B, int K, int N,
                                     int act, const uint32_t* x_amax, uint32_t* y_amax, void* stream) {
  auto fold = [&](int rc) {
    if (rc != 0 || y == nullptr || y_amax == nullptr) return rc;
    return odin_absmax_fold(y, (size_t)B * N, y_amax, stream);
  };
  if (odin_tiny_dense_ok(B, K, N)) return fold(odin_tiny_dense_fwd(x, w, bias, y, B, K, N, act, stream));
  // one thin side (FactorVAE's first / last discriminator layers): streaming kernels, range word kept by the kernel
  if (odin_thin_dense_kind(B, K, N) != 0 && ((((size_t)x | (size_t)w | (size_t)y | (size_t)bias)) & 15) == 0)
    return odin_thin_dense_fwd(x, w, bias, y, B, K, N, act, y_amax, stream);
  if (odin_dense_h_ok(B, K, N)) return odin_dense_h_fwd(x, w, bias, y, B, K, N, act, x_amax, y_amax, stream);
  if (dense_via_igemm() && odin_igemm_applicable(0, B, 1, 1, K, 1, 1, N, 1, 1, 1, 0))
    return odin_igemm_launch(0, x, w, bias, nullptr, 0, y, nullptr, B, 1, 1, K, 1, 1, N, 1, 1, 1, 0, 0, act, y_amax,
                             stream);
  if (odin_dense_gemm_ok(B, K, N)) return fold(odin_dense_gemm_fwd(x, w, bias, y, B, K, N, act, stream));
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.B = B; p.H = 1; p.W = 1; p.CI = K; p.OH = 1; p.OW = 1; p.CO = N;
  p.KH = p.KW = 1; p.S = 1; p.act = act; p.wmode = 0;
  return fold(launch_gather(MODE_F, p, stream, odin_num_cus()));
}

// dx[B,K] = (dy[B,N] @ w[K,N]^T) * act'(aux)
extern "C" int odin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                float* dx, float* colsum_slab, int* slab_rows_out, int B, int K,
                                int N, void* stream) {
  // (no range words through this entry: a plane GEMM bounds dy itself)
  return odin_dense_dgrad_ranged(dy, w, aux, aux_act, dx, colsum_slab, slab_rows_out, B, K, N, nullptr, nullptr,
                                 stream);
}

// the kernel families below (without a column-sum slab) that fold max|dx| into dx_amax
bool odin_dense_dgrad_tracks(int B, int K, int N) {
  if (odin_tiny_dense_ok(B, K, N)) return false;
  return odin_thin_dense_kind(B, K, N) != 0 || odin_dense_h_ok(B, K, N) || (dense_via_igemm() && odin_igemm_applicable(1, B, 1, 1, N, 1, 1, K, 1, 1, 1, 0)) ||
         odin_dense_gemm_ok(B, K, N);
}

int odin_dense_dgrad_ranged(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                            float* colsum_slab, int* slab_rows_out, int B, int K, int N, const uint32_t* dy_amax,
                            uint32_t* dx_amax, void* stream) {
  // (the same contract as the convolutions' track_dx: a word that is handed in is valid on return)
  auto fold = [&](int rc) {
    if (rc != 0 || dx == nullptr || dx_amax == nullptr) return rc;
    return odin_absmax_fold(dx, (size_t)B * K, dx_amax, stream);
  };
  if (odin_tiny_dense_ok(B, K, N))
    return fold(odin_tiny_dense_dgrad(dy, w, aux, aux_act, dx, colsum_slab, slab_rows_out, B, K, N, stream));
  if (colsum_slab == nullptr && odin_thin_dense_kind(B, K, N) != 0 &&
      ((((size_t)dy | (size_t)w | (size_t)dx | (size_t)aux)) & 15) == 0) {
    if (slab_rows_out) *slab_rows_out = 0;
    if (dx == nullptr) return 0;
    return odin_thin_dense_dgrad(dy, w, aux, aux_act, dx, B, K, N, dx_amax, stream);
  }
  if (colsum_slab == nullptr && odin_dense_h_ok(B, K, N)) {
    if (slab_rows_out) *slab_rows_out = 0;
    if (dx == nullptr) return 0;
    return odin_dense_h_dgrad(dy, w, aux, aux_act, dx, B, K, N, dy_amax, dx_amax, stream);
  }
  // (as a transposed 1x1 gather: reduction over the N outputs, weights [k_in][n] with n contiguous)
  if (colsum_slab == nullptr && dense_via_igemm() && odin_igemm_applicable(1, B, 1, 1, N, 1, 1, K, 1, 1, 1, 0)) {
    if (slab_rows_out) *slab_rows_out = 0;
    if (dx == nullptr) return 0;
    return odin_igemm_launch(1, dy, w, nullptr, aux, aux_act, dx, nullptr, B, 1, 1, N, 1, 1, K, 1, 1, 1, 0, 0, 0,
                             dx_amax, stream);
  }
  if (colsum_slab == nullptr && odin_dense_gemm_ok(B, K, N)) {
    if (slab_rows_out) *slab_rows_out = 0;
    if (dx == nullptr) return 0;
    return odin_dense_gemm_dgrad(dy, w, aux, aux_act, dx, B, K, N, dx_amax, stream);
  }
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.B = B; p.H = 1; p.W = 1; p.CI = N; p.OH = 1; p.OW = 1; p.CO = K;
  p.KH = p.KW = 1; p.S = 1; p.wmode = 1;
  return fold(launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_COLSUM_BLOCKS : odin_num_cus(),
                            slab_rows_out));
}

// ---- fused decoder tail: (Conv2DTranspose | Conv2D)(act) -> Conv2D 1x1 linear (C1<=4 maps)
// -> Independent(Bernoulli).log_prob(target), forward + backward in one launch ----------
extern "C" int odin_bernoulli_tail_fwd_bwd(int is_deconv, const float* x, const float* w,
                                           const float* bias, const float* w1, const float* b1,
                                           const float* target, float* logits, float* g_out,
                                           float* llk_part, int* n_part_out, float* tail_slab,
                                           int* slab_rows_out, const float* scale,
                                           const odin_conv_desc* d, int C1, void* stream) {
  // (the range contract: a word handed in as d->dy_amax bounds g_out on return -- the plane kernel folds it in from
  // its epilogue, the other families are followed by one pass)
  auto tail_fold = [&](int rc) {
    if (rc != 0 || g_out == nullptr || d->dy_amax == nullptr) return rc;
    return odin_absmax_fold(g_out, (size_t)d->B * d->OH * d->OW * d->Cout, d->dy_amax, stream);
  };
  if (is_deconv && d->act == ODIN_ACT_ELU && d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_tconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, d->center, 3, C1))
    return odin_tconv_planes_launch(x, w, bias, nullptr, g_out, nullptr, slab_rows_out, w1, b1, target,
                                    logits, llk_part, n_part_out, tail_slab, scale, C1, d->B, d->H, d->W,
                                    d->Cin, d->Cout, 3, d->x_amax, d->dy_amax, stream);
  if (is_deconv && d->act == ODIN_ACT_ELU && d->Cout == 32 && (C1 == 1 || C1 == 3) &&
      d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_tconv_ring_applicable(d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                 d->pad_l, d->center))
    return tail_fold(odin_tconv_ring_launch(x, w, bias, nullptr, g_out, nullptr, slab_rows_out, w1, b1, target,
                                            logits, llk_part, n_part_out, tail_slab, scale, C1, d->B, d->H, d->W,
                                            d->Cout, 3, stream));
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = g_out;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = is_deconv ? 1 : 0; p.act = d->act; p.center = d->center;
  TailParams tp;
  tp.w1 = w1; tp.b1 = b1; tp.target = target; tp.logits = logits; tp.llk_part = llk_part;
  tp.slab = tail_slab; tp.scale = scale; tp.C1 = C1;
  int rc = launch_gather(is_deconv ? MODE_T : MODE_F, p, stream, -ODIN_MAX_COLSUM_BLOCKS,
                         slab_rows_out, &tp);
  if (n_part_out) *n_part_out = p.OH / (p.TR > 0 ? p.TR : 1);  // log-likelihood parts per sample
  return tail_fold(rc);
}

// diagnostics: device buffer (>= 64 int64) receiving s_memtime stamps of workgroup 0
extern "C" int odin_debug_set_stamps(void* buf) {
  g_stamps = (long long*)buf;
  odin_fconv_ring_set_stamps(buf);
  odin_tconv_ring_set_stamps(buf);
  odin_tconv_planes_set_stamps(buf);
  odin_fconv_planes_set_stamps(buf);
  odin_igemm_set_stamps(buf);
  return 0;
}
