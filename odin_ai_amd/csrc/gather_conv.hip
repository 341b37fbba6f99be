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
};

// Weight slice [taps][CIC][32 output channels] -> LDS.  Loads are issued in batches of 8
// per thread (16-byte loads where the layout allows) so that their latencies overlap; a
// one-load-per-iteration loop costs a full memory round trip per element.
__device__ __forceinline__ void stage_weights(const GParams& p, float* wl, int c0, int n0,
                                              int tid, int nthreads) {
  const int ntaps = p.KH * p.KW;
  constexpr int U = 8;
  if (p.wmode == 0) {
    // global [tap][ci][co]: rows of 32 consecutive output channels
    const bool vec = ((p.CO & 3) == 0);
    if (vec) {
      const int total = ntaps * p.CIC * 8;  // float4 items
      for (int e0 = tid; e0 < total; e0 += nthreads * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < total) {
            int co = (e & 7) * 4, t2 = e >> 3;
            int ci = t2 % p.CIC, tap = t2 / p.CIC;
            int c = c0 + ci, n = n0 + co;
            if (c < p.CI && n < p.CO)
              v[u] = *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.CI + c) * p.CO + n);
          }
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
          v[u] = 0.f;
          if (e < total) {
            int co = e & 31, t2 = e >> 5;
            int ci = t2 % p.CIC, tap = t2 / p.CIC;
            int c = c0 + ci, n = n0 + co;
            if (c < p.CI && n < p.CO) v[u] = p.w[((size_t)tap * p.CI + c) * p.CO + n];
          }
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
          int e = e0 + u * nthreads;
          v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (e < total) {
            int ci = (e % c4n) * 4, t2 = e / c4n;
            int co = t2 & 31, tap = t2 >> 5;
            int c = c0 + ci, n = n0 + co;
            if (c < p.CI && n < p.CO)
              v[u] = *reinterpret_cast<const float4*>(p.w + ((size_t)tap * p.CO + n) * p.CI + c);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int e = e0 + u * nthreads;
          if (e < total) {
            int ci = (e % c4n) * 4, t2 = e / c4n;
            int co = t2 & 31, tap = t2 >> 5;
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
          v[u] = 0.f;
          if (e < total) {
            int ci = e % p.CIC, t2 = e / p.CIC;
            int co = t2 & 31, tap = t2 >> 5;
            int c = c0 + ci, n = n0 + co;
            if (c < p.CI && n < p.CO) v[u] = p.w[((size_t)tap * p.CO + n) * p.CI + c];
          }
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

// ---- patch staging ---------------------------------------------------------------
// The patch is NIMG*NRI rows of PW pixels x CIC channels.  Work item e (float4 when the
// channel count allows, else one float) -> (row, pcol, c) with one magic-number division;
// the global loads of tile t+1 are issued into registers (PF) before tile t is computed
// and written to LDS after it, so HBM/L2 latency hides under the MFMAs.

struct StageGeom {
  int vec;       // 1: float4 items, 0: float items
  int cpi;       // items per pixel (CIC/4 or CIC)
  int rowlen;    // items per patch row = PW*cpi
  int total;     // items per patch
  unsigned m_row, m_cpi;  // magic reciprocals
};

__device__ __forceinline__ unsigned magic_of(int d) {
  return d <= 1 ? 0u : (unsigned)(4294967296.0 / d) + 1u;
}
__device__ __forceinline__ int fast_div(int e, int d, unsigned m) {
  if (d <= 1) return e;
  int q = (int)__umulhi((unsigned)e, m);
  if (q * d > e) --q;
  return q;
}

__device__ __forceinline__ StageGeom stage_geom(const GParams& p) {
  StageGeom g;
  g.vec = (((p.CI & 3) == 0) && ((p.CIC & 3) == 0)) ? 1 : 0;
  g.cpi = g.vec ? (p.CIC >> 2) : p.CIC;
  g.rowlen = p.PW * g.cpi;
  g.total = p.NIMG * p.NRI * g.rowlen;
  g.m_row = magic_of(g.rowlen);
  g.m_cpi = magic_of(g.cpi);
  return g;
}

struct StageItem {
  int lds;   // float index into the patch
  long gofs; // float index into `in`, or -1 when the item is SAME padding / out of range
};

__device__ __forceinline__ StageItem stage_item(const GParams& p, const StageGeom& g, int e,
                                                int b0, int ih_lo, int c0) {
  StageItem it;
  int row = fast_div(e, g.rowlen, g.m_row);
  int j = e - row * g.rowlen;
  int pcol = fast_div(j, g.cpi, g.m_cpi);
  int cc = (j - pcol * g.cpi) * (g.vec ? 4 : 1);
  int img = (p.NIMG == 1) ? 0 : row / p.NRI;
  int prow = row - img * p.NRI;
  int b = b0 + img, ih = ih_lo + prow, iw = p.iw_lo + pcol, c = c0 + cc;
  it.lds = (row * p.PW + pcol) * p.P + cc;
  bool ok = (b < p.B) && (ih >= 0) && (ih < p.H) && (iw >= 0) && (iw < p.W) && (c < p.CI);
  it.gofs = ok ? ((((long)b * p.H + ih) * p.W + iw) * p.CI + c) : -1;
  return it;
}

__device__ __forceinline__ float4 center4(float4 v) {
  return make_float4(2.f * v.x - 1.f, 2.f * v.y - 1.f, 2.f * v.z - 1.f, 2.f * v.w - 1.f);
}

// synchronous staging (chunked reductions / patches too large for the register prefetch);
// loads are batched 8 deep per thread so their latencies overlap
__device__ __forceinline__ void stage_patch(const GParams& p, const StageGeom& g, float* patch,
                                            int b0, int ih_lo, int c0, int tid, int nthreads) {
  constexpr int U = 8;
  for (int e0 = tid; e0 < g.total; e0 += nthreads * U) {
    float4 v[U];
    int lds[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int e = e0 + u * nthreads;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      lds[u] = -1;
      if (e < g.total) {
        StageItem it = stage_item(p, g, e, b0, ih_lo, c0);
        lds[u] = it.lds;
        if (it.gofs >= 0) {
          if (g.vec) v[u] = *reinterpret_cast<const float4*>(p.in + it.gofs);
          else v[u].x = p.in[it.gofs];
          if (p.center) v[u] = center4(v[u]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (lds[u] >= 0) {
        float* d = patch + lds[u];
        if (g.vec) { d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
        else d[0] = v[u].x;
      }
    }
  }
}

template <int NT, int MAXV>
__device__ __forceinline__ void prefetch_issue(const GParams& p, const StageGeom& g, float4* pf,
                                               int b0, int ih_lo, int tid) {
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int e = tid + i * NT;
    if (e < g.total) {
      StageItem it = stage_item(p, g, e, b0, ih_lo, 0);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it.gofs >= 0) {
        if (g.vec) v = *reinterpret_cast<const float4*>(p.in + it.gofs);
        else v.x = p.in[it.gofs];
        if (p.center) v = center4(v);
      }
      pf[i] = v;
    }
  }
}

template <int NT, int MAXV>
__device__ __forceinline__ void prefetch_commit(const GParams& p, const StageGeom& g,
                                                const float4* pf, float* patch, int tid) {
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    int e = tid + i * NT;
    if (e < g.total) {
      int row = fast_div(e, g.rowlen, g.m_row);
      int j = e - row * g.rowlen;
      int pcol = fast_div(j, g.cpi, g.m_cpi);
      int cc = (j - pcol * g.cpi) * (g.vec ? 4 : 1);
      float* d = patch + (row * p.PW + pcol) * p.P + cc;
      float4 v = pf[i];
      if (g.vec) { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
      else d[0] = v.x;
    }
  }
}

// geometry of the output pixel a lane owns inside M-tile `mt`
struct Slot {
  int base;    // patch float index of tap (0,0) channel 0
  int opix;    // linear output pixel index (b*OH+oh)*OW+ow, or -1 if masked
  int kh0, kw0;  // MODE_T: first valid tap of this M-tile's phase
};

template <int MODE>
__device__ __forceinline__ Slot slot_geometry(const GParams& p, int mt, int l31, int gr0) {
  Slot s;
  s.kh0 = s.kw0 = 0;
  const int total_rows = p.B * p.OH;
  if (MODE == MODE_F) {
    int sl = mt * 32 + l31;
    int r = sl / p.OW, c = sl - r * p.OW;
    bool valid = (r < p.TR) && (gr0 + r < total_rows);
    int img = r / p.RPI, rl = r - img * p.RPI;
    s.base = valid ? ((img * p.NRI + rl * p.S) * p.PW + c * p.S) * p.P : 0;
    s.opix = valid ? (gr0 + r) * p.OW + c : -1;
  } else {
    const int S = p.S;
    int phase = mt / p.MTP, mtl = mt - phase * p.MTP;
    int ph = phase / S, pw = phase - ph * S;
    int sl = mtl * 32 + l31;
    const int IWs = p.OW / S, RPS = p.RPI / S;
    int img = sl / (RPS * IWs), rem = sl - img * (RPS * IWs);
    int rq = rem / IWs, cq = rem - rq * IWs;
    int row_in_tile = img * p.RPI + ph + S * rq;
    bool valid = (sl < p.SPP) && (gr0 + row_in_tile < total_rows);
    s.kh0 = (ph + p.pt) % S;
    s.kw0 = (pw + p.pl) % S;
    int dh = (ph + p.pt - s.kh0) / S, dw = (pw + p.pl - s.kw0) / S;
    // patch row of tap jh=0: rq + dh - lo_h, where lo_h == p.ih_off ; same for columns
    int prow = rq + dh - p.ih_off, pcol = cq + dw - p.iw_lo;
    // masked lanes read the last patch pixel (tap offsets are negative in this mode)
    s.base = valid ? ((img * p.NRI + prow) * p.PW + pcol) * p.P
                   : ((p.NIMG * p.NRI - 1) * p.PW + (p.PW - 1)) * p.P;
    s.opix = valid ? (gr0 + row_in_tile) * p.OW + (pw + S * cq) : -1;
  }
  return s;
}

// One 32(out-channel) x 32(pixel) accumulator tile over one channel chunk.  TK/TS/TCIC
// are compile-time kernel size / stride / chunk (0 = runtime).  The specialised instances
// run an explicit two-stage register pipeline: the 32 LDS operands of step t+1 (one tap x
// 32 channels = 16 MFMAs) are read while the MFMAs of step t execute.
struct TapAddr {
  const float* ap;
  const float* wp;
};

template <int MODE, int TK, int TS, int TCIC>
__device__ __forceinline__ TapAddr tap_addr(const GParams& p, const float* patch, const float* wl,
                                            const Slot& s, int l31, int h, int step) {
  constexpr int SUB = TCIC / 32;            // 32-channel sub-steps per tap
  constexpr int NJ = (MODE == MODE_F) ? TK : TK / TS;
  constexpr int WP = (MODE == MODE_F) ? 32 : 33;
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
  t.ap = patch + s.base + tapoff + h + sub * 32;
  t.wp = wl + (wt * TCIC + h + sub * 32) * WP + l31;
  return t;
}

template <int MODE, int TK, int TS, int TCIC>
__device__ __forceinline__ f32x16 mtile_compute(const GParams& p, const float* patch,
                                                const float* wl, const Slot& s, int l31, int h,
                                                f32x16 acc) {
  if constexpr (TCIC != 0) {
    static_assert(TCIC % 32 == 0 && TK != 0 && TS != 0 && TK % TS == 0, "specialised shape");
    constexpr int NJ = (MODE == MODE_F) ? TK : TK / TS;
    constexpr int NSTEP = NJ * NJ * (TCIC / 32);
    constexpr int WP = (MODE == MODE_F) ? 32 : 33;
    float a0[16], b0[16], a1[16], b1[16];
    {
      TapAddr t = tap_addr<MODE, TK, TS, TCIC>(p, patch, wl, s, l31, h, 0);
#pragma unroll
      for (int u = 0; u < 16; ++u) { a0[u] = t.wp[u * 2 * WP]; b0[u] = t.ap[2 * u]; }
    }
    ODIN_SCHED_FENCE();
#pragma unroll
    for (int st = 0; st < NSTEP; st += 2) {
      if (st + 1 < NSTEP) {
        TapAddr t = tap_addr<MODE, TK, TS, TCIC>(p, patch, wl, s, l31, h, st + 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) { a1[u] = t.wp[u * 2 * WP]; b1[u] = t.ap[2 * u]; }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = mfma32(a0[u], b0[u], acc);
      // issue the next step's LDS reads in the shadow of this step's MFMAs
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
        ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 2);
      }
      ODIN_SCHED_FENCE();
      if (st + 2 < NSTEP) {
        TapAddr t = tap_addr<MODE, TK, TS, TCIC>(p, patch, wl, s, l31, h, st + 2);
#pragma unroll
        for (int u = 0; u < 16; ++u) { a0[u] = t.wp[u * 2 * WP]; b0[u] = t.ap[2 * u]; }
      }
      if (st + 1 < NSTEP) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = mfma32(a1[u], b1[u], acc);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
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
        int c = 0;
        for (; c + 16 <= CIC; c += 16) {
          float a[8], b[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u] = wp[(c + 2 * u) * WP]; b[u] = ap[c + 2 * u]; }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = mfma32(a[u], b[u], acc);
        }
        for (; c < CIC; c += 2) acc = mfma32(wp[c * WP], ap[c], acc);
      }
    }
    return acc;
  }
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

__device__ __forceinline__ float softplus_g(float x) {
  return fmaxf(x, 0.f) + log1pf(odin_exp(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_g(float x) {
  float e = odin_exp(-fabsf(x));
  float r = 1.f / (1.f + e);
  return x >= 0.f ? r : e * r;
}

template <int MODE, int NW, int TK, int TS, int TCIC, bool TAIL, int MAXV>
__global__ __launch_bounds__(NW * 64) void gather_conv_kernel(GParams p, TailParams tp) {
  ODIN_DYN_SMEM(float, smem);
  float* patch = smem;
  float* wl = smem + p.patch_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32;
  constexpr int NT = NW * 64;
  const StageGeom sg = stage_geom(p);
  const bool pipelined = (p.n_chunks == 1) && (sg.total <= MAXV * NT);

  if (p.w_resident) stage_weights(p, wl, 0, n0, tid, NT);

  float bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = 0.f;
  // fused-tail state
  float w1r[TAIL ? 16 : 1][MAXC1], dw1[TAIL ? 16 : 1][MAXC1], db1[MAXC1], b1r[MAXC1];
  if (TAIL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = 8 * (i >> 2) + 4 * h + (i & 3);
#pragma unroll
      for (int oc = 0; oc < MAXC1; ++oc) {
        w1r[i][oc] = (oc < tp.C1 && n < p.CO) ? tp.w1[n * tp.C1 + oc] : 0.f;
        dw1[i][oc] = 0.f;
      }
    }
#pragma unroll
    for (int oc = 0; oc < MAXC1; ++oc) {
      db1[oc] = 0.f;
      b1r[oc] = oc < tp.C1 ? tp.b1[oc] : 0.f;
    }
  }

  float4 pf[MAXV];
  int tile = blockIdx.x;
  if (pipelined && tile < p.n_tiles) {
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = (MODE == MODE_F) ? oh0 * p.S + p.ih_off : oh0 / p.S + p.ih_off;
    prefetch_issue<NT, MAXV>(p, sg, pf, b0, ih_lo, tid);
  }

  for (; tile < p.n_tiles; tile += gridDim.x) {
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = (MODE == MODE_F) ? oh0 * p.S + p.ih_off : oh0 / p.S + p.ih_off;
    f32x16 acc0 = f32x16_zero(), acc1 = f32x16_zero();
    const int mt0 = wave, mt1 = wave + NW;
    Slot s0 = slot_geometry<MODE>(p, mt0 < p.MT ? mt0 : 0, l31, gr0);
    Slot s1 = slot_geometry<MODE>(p, mt1 < p.MT ? mt1 : 0, l31, gr0);
    if (pipelined) {
      __syncthreads();  // everyone is done reading the previous patch
      prefetch_commit<NT, MAXV>(p, sg, pf, patch, tid);
      __syncthreads();
      const int nt = tile + gridDim.x;
      if (nt < p.n_tiles) {  // loads of the next tile fly during the MFMAs below
        const int g2 = nt * p.TR;
        const int b2 = g2 / p.OH, o2 = g2 - b2 * p.OH;
        const int ih2 = (MODE == MODE_F) ? o2 * p.S + p.ih_off : o2 / p.S + p.ih_off;
        prefetch_issue<NT, MAXV>(p, sg, pf, b2, ih2, tid);
      }
      if (mt0 < p.MT) acc0 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s0, l31, h, acc0);
      if (mt1 < p.MT) acc1 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s1, l31, h, acc1);
    } else {
      for (int ch = 0; ch < p.n_chunks; ++ch) {
        const int c0 = ch * p.CIC;
        __syncthreads();
        stage_patch(p, sg, patch, b0, ih_lo, c0, tid, NT);
        if (!p.w_resident) stage_weights(p, wl, c0, n0, tid, NT);
        __syncthreads();
        if (mt0 < p.MT) acc0 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s0, l31, h, acc0);
        if (mt1 < p.MT) acc1 = mtile_compute<MODE, TK, TS, TCIC>(p, patch, wl, s1, l31, h, acc1);
      }
    }
    // ---- epilogue: bias + activation (+ activation-gradient multiply) + NHWC store ----
    float llk_lane = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int mt = mi == 0 ? mt0 : mt1;
      const Slot& s = mi == 0 ? s0 : s1;
      const f32x16& acc = mi == 0 ? acc0 : acc1;
      if (mt >= p.MT) continue;  // wave-uniform
      const bool live = s.opix >= 0;
      const size_t obase = live ? (size_t)s.opix * p.CO : 0;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
        float t = acc[i];
        if (p.bias != nullptr && n < p.CO) t += p.bias[n];
        v[i] = (n < p.CO && live) ? odin_act(p.act, t) : 0.f;
      }
      if (TAIL) {
        // 1x1 conv: each pixel's 32 channels live in lanes (l31, h=0) and (l31, h=1)
        float lg[MAXC1], dl[MAXC1];
#pragma unroll
        for (int oc = 0; oc < MAXC1; ++oc) {
          float t = 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) t += v[i] * w1r[i][oc];
          t += __shfl_xor(t, 32);
          lg[oc] = t + b1r[oc];
        }
        const float sc = tp.scale[0];
#pragma unroll
        for (int oc = 0; oc < MAXC1; ++oc) {
          dl[oc] = 0.f;
          if (oc < tp.C1 && live) {
            const float x = tp.target[(size_t)s.opix * tp.C1 + oc];
            const float l = lg[oc];
            if (h == 0) {
              llk_lane += x * l - softplus_g(l);
              db1[oc] += (sigmoid_g(l) - x) * sc;
              if (tp.logits != nullptr) tp.logits[(size_t)s.opix * tp.C1 + oc] = l;
            }
            dl[oc] = (sigmoid_g(l) - x) * sc;
          }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float g = 0.f;
#pragma unroll
          for (int oc = 0; oc < MAXC1; ++oc) {
            g += dl[oc] * w1r[i][oc];
            dw1[i][oc] += v[i] * dl[oc];
          }
          v[i] = g * odin_act_grad(p.act, v[i]);
        }
      } else if (p.aux != nullptr && live) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + 8 * (i >> 2) + 4 * h + (i & 3);
          if (n < p.CO) v[i] *= odin_act_grad(p.aux_act, p.aux[obase + n]);
        }
      }
      if (live) {
#pragma unroll
        for (int i = 0; i < 16; ++i) bsum[i] += v[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + 8 * q + 4 * h;
          if (((p.CO & 3) == 0) && n + 3 < p.CO) {
            *reinterpret_cast<float4*>(p.out + obase + n) =
                make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (n + j < p.CO) p.out[obase + n + j] = v[4 * q + j];
          }
        }
      }
    }
    if (TAIL) {
      // one log-likelihood partial per tile (a tile lies inside one sample)
      float t = wave_sum64(llk_lane);
      __syncthreads();
      float* red = wl + (p.KH * p.KW * p.CIC * p.WP);  // scratch after the weights
      if (lane == 0) red[wave] = t;
      __syncthreads();
      if (tid == 0) {
        float a = 0.f;
        for (int w2 = 0; w2 < NW; ++w2) a += red[w2];
        tp.llk_part[tile] = a;
      }
    }
  }

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
        for (int oc = 0; oc < MAXC1; ++oc) {
          float v = dw1[i][oc];
#pragma unroll
          for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
          if (l31 == 0) red[wave * RW + 32 + (8 * (i >> 2) + 4 * h + (i & 3)) * MAXC1 + oc] = v;
        }
      }
#pragma unroll
      for (int oc = 0; oc < MAXC1; ++oc) {
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

bool plan_gather(GParams& p, int mode, int max_blocks, int* grid_x, size_t* lds_bytes) {
  const int S = p.S;
  if (mode == MODE_T && (p.OW % S != 0 || p.OH % S != 0)) return false;
  const int img_pix = p.OH * p.OW;
  const int TARGET = 32 * NW_G;  // one M-tile per wave
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
  p.WP = (p.wmode == 0) ? 32 : 33;
  const int ntaps = p.KH * p.KW;
  int cic = CIp;
  // keep float4 staging possible when CI % 4 == 0
  const int gran = ((p.CI & 3) == 0) ? 4 : 2;
  while (true) {
    int P = cic + 1;
    long pf = (long)p.NIMG * p.NRI * p.PW * P + 8;
    long wf = (long)ntaps * cic * p.WP + 64;
    if (pf + wf <= LDS_BUDGET_FLOATS) break;
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
  p.patch_floats = (int)(((long)p.NIMG * p.NRI * p.PW * p.P + 8 + 3) & ~3L);
  long wf = (long)ntaps * cic * p.WP + 64;  // + scratch for per-tile reductions
  long total = p.patch_floats + wf;
  if (total < 1024) total = 1024;
  *lds_bytes = (size_t)total * 4;
  if (max_blocks < 0) {  // slab-producing launch: rows are bounded
    int cap = -max_blocks;
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

template <int MODE, int TK, int TS, int TCIC, bool TAIL, int MAXV>
int launch_inst(const GParams& p, const TailParams& tp, dim3 grid, size_t lds, void* stream) {
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gather_conv_kernel<MODE, NW_G, TK, TS, TCIC, TAIL, MAXV>),
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((gather_conv_kernel<MODE, NW_G, TK, TS, TCIC, TAIL, MAXV>), grid, dim3(NW_G * 64), lds,
              stream, p, tp);
  return odin_check_launch("gather_conv");
}

int launch_gather(int mode, GParams& p, void* stream, int max_blocks, int* rows_out = nullptr,
                  const TailParams* tail = nullptr) {
  int gx;
  size_t lds;
  if (!plan_gather(p, mode, max_blocks, &gx, &lds)) return odin_fail(-2, "gather_conv: no tiling plan");
  if (rows_out) *rows_out = gx;
  if (p.out == nullptr) return 0;  // dry run: planning only
  dim3 grid(gx, (p.CO + 31) / 32, 1);
  TailParams tp;
  memset(&tp, 0, sizeof(tp));
  const bool k4s2 = (p.KH == 4 && p.KW == 4 && p.S == 2 && p.n_chunks == 1);
  if (tail != nullptr) {
    tp = *tail;
    if (p.CO > 32 || tp.C1 > MAXC1 || p.NIMG != 1)
      return odin_fail(-2, "bernoulli tail: needs Cout<=32, C1<=4 and tiles inside one image");
    if (mode == MODE_T && k4s2 && p.CIC == 32 && p.wmode == 1)
      return launch_inst<MODE_T, 4, 2, 32, true, 6>(p, tp, grid, lds, stream);
    if (mode == MODE_T) return launch_inst<MODE_T, 0, 0, 0, true, 8>(p, tp, grid, lds, stream);
    return launch_inst<MODE_F, 0, 0, 0, true, 8>(p, tp, grid, lds, stream);
  }
  if (mode == MODE_F) {
    if (k4s2 && p.wmode == 0 && p.CIC == 32) return launch_inst<MODE_F, 4, 2, 32, false, 20>(p, tp, grid, lds, stream);
    if (k4s2 && p.wmode == 0 && p.CIC == 64) return launch_inst<MODE_F, 4, 2, 64, false, 12>(p, tp, grid, lds, stream);
    return launch_inst<MODE_F, 0, 0, 0, false, 12>(p, tp, grid, lds, stream);
  }
  if (k4s2 && p.wmode == 1 && p.CIC == 32) return launch_inst<MODE_T, 4, 2, 32, false, 6>(p, tp, grid, lds, stream);
  if (k4s2 && p.wmode == 1 && p.CIC == 64) return launch_inst<MODE_T, 4, 2, 64, false, 6>(p, tp, grid, lds, stream);
  return launch_inst<MODE_T, 0, 0, 0, false, 12>(p, tp, grid, lds, stream);
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

extern "C" int odin_max_slab_rows(void) { return ODIN_MAX_SLAB_BLOCKS; }

// ---- Conv2D -------------------------------------------------------------------------
extern "C" int odin_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                               const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 0; p.act = d->act; p.center = d->center;
  return launch_gather(MODE_F, p, stream, odin_num_cus());
}

// dx[b,ih,iw,ci] = sum_{kh,kw,co} dy[b,(ih+pt-kh)/S,(iw+pl-kw)/S,co] * W[kh,kw,ci,co];
// optionally multiplied by act'(aux) (aux = this layer's input = previous layer's output)
extern "C" int odin_conv2d_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                 float* dx, float* colsum_slab, int* slab_rows_out,
                                 const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 1;
  return launch_gather(MODE_T, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}

// ---- Conv2DTranspose (desc: H,W,Cin = input; OH=H*S, OW=W*S, Cout = output; pads = the
// SAME pads of the forward conv on the OUTPUT size) ------------------------------------
extern "C" int odin_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                 const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 1; p.act = d->act; p.center = d->center;
  return launch_gather(MODE_T, p, stream, odin_num_cus());
}

extern "C" int odin_deconv2d_dgrad(const float* dy, const float* w, const float* aux,
                                   int aux_act, float* dx, float* colsum_slab,
                                   int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 0;
  return launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}

// ---- Dense: y[B,N] = act(x[B,K] @ w[K,N] + b) ----------------------------------------
extern "C" int odin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                              int K, int N, int act, void* stream) {
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.B = B; p.H = 1; p.W = 1; p.CI = K; p.OH = 1; p.OW = 1; p.CO = N;
  p.KH = p.KW = 1; p.S = 1; p.act = act; p.wmode = 0;
  return launch_gather(MODE_F, p, stream, odin_num_cus());
}

// dx[B,K] = (dy[B,N] @ w[K,N]^T) * act'(aux)
extern "C" int odin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                float* dx, float* colsum_slab, int* slab_rows_out, int B, int K,
                                int N, void* stream) {
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.B = B; p.H = 1; p.W = 1; p.CI = N; p.OH = 1; p.OW = 1; p.CO = K;
  p.KH = p.KW = 1; p.S = 1; p.wmode = 1;
  return launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}

// ---- fused decoder tail: (Conv2DTranspose | Conv2D)(act) -> Conv2D 1x1 linear (C1<=4 maps)
// -> Independent(Bernoulli).log_prob(target), forward + backward in one launch ----------
extern "C" int odin_bernoulli_tail_fwd_bwd(int is_deconv, const float* x, const float* w,
                                           const float* bias, const float* w1, const float* b1,
                                           const float* target, float* logits, float* g_out,
                                           float* llk_part, int* n_part_out, float* tail_slab,
                                           int* slab_rows_out, const float* scale,
                                           const odin_conv_desc* d, int C1, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = g_out;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = is_deconv ? 1 : 0; p.act = d->act; p.center = d->center;
  TailParams tp;
  tp.w1 = w1; tp.b1 = b1; tp.target = target; tp.logits = logits; tp.llk_part = llk_part;
  tp.slab = tail_slab; tp.scale = scale; tp.C1 = C1;
  int rc = launch_gather(is_deconv ? MODE_T : MODE_F, p, stream, -ODIN_MAX_SLAB_BLOCKS,
                         slab_rows_out, &tp);
  if (n_part_out) *n_part_out = p.OH / (p.TR > 0 ? p.TR : 1);  // log-likelihood parts per sample
  return rc;
}
