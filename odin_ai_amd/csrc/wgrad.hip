// wgrad.hip -- weight / bias gradients of Conv2D, Conv2DTranspose and Dense on the f32
// matrix cores, plus the deterministic slab reduction that finishes them.
//
//   dW[kh,kw,ci,co] = sum_{b,oh,ow} IN[b, oh*S-pt+kh, ow*S-pl+kw, ci] * DY[b,oh,ow,co]
//
// Conv2D:           IN = layer input x,            DY = dL/d(pre-activation output)
// Conv2DTranspose:  IN = dL/d(pre-act output) (the LARGE image), DY = layer input x, which
//                   yields dW directly in Keras' (kh,kw,Cout,Cin) layout
// Dense:            1x1 images, batch as the image index.
// Replaces tape.gradient(loss, parameters) (odin/networks/base_networks.py:518) for the
// kernels created at odin/networks/image_networks.py:166-173 and base_networks.py:1002-1014.
//
// MI355X mapping: the reduction dimension of the MFMA is the PIXEL index.  A workgroup
// stages the IN patch (F-mode geometry, with halo) and the matching DY rows of ~128
// pixels into LDS, and each wave keeps up to 8 32x32 accumulator tiles of dW
// (rows = (tap,ci), columns = co) resident in registers across its whole persistent
// pixel-tile loop; v_mfma_f32_32x32x2_f32 consumes two pixels per instruction.  The bias
// gradient rides along as one more accumulator whose A operand is the constant 1.
// Partial results go to a per-workgroup slab; odin_slab_reduce sums the slabs in a fixed
// order (bit-reproducible, no float atomics).
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

// 16 KB of zeros in HBM: the DMA source for SAME-padding rows
__device__ float odin_zero_row[4096];

namespace {

constexpr int NW_W = 4;
constexpr int NACC = 8;

struct WParams {
  const float* in;
  const float* dy;
  float* slab;
  int B, H, W, CI, OH, OW, CO;
  int KH, KW, S, pt, pl, center;
  int TR, RPI, NIMG, n_tiles, NRI, PW, P;
  int CIB, COB, DP, slots;
  int nrt, ncot, want_bias;
  int slab_stride;
  int patch_floats, dy_floats;
  int pvec, dvec, KI, pipelined;
  int dlog;             // log2(COB/4) when COB/4 is a power of two (fast DY staging), else -1
  int flat, n_batches;  // Dense: IN tile is one contiguous [NIMG, CIB] block
  long long* stamps;
  int bias_mode;  // 0 none, 1 extra MFMA tile with A = 1, 2 summed while staging DY
  const uint32_t* g_amax;  // range word of the gradient operand (wgrad_planes only; may be null)
  const uint32_t* a_amax;  // range word of the activation operand (plane kernels; may be null)
};

// ---- staging ------------------------------------------------------------------------
// IN patch: row-aligned like gather_conv (wave w owns patch rows w, w+NW, ...; per-lane
// source / LDS offsets and SAME-padding masks are computed once per kernel).  DY rows of a
// tile are one contiguous run in HBM.  Tile t+1 is prefetched into registers while tile t
// is being multiplied.
#ifdef ODIN_SIM
#define W_UNIFORM(x) (x)
#else
#define W_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

template <int KMAX>
struct WLane {
  unsigned gofs[KMAX];
  int ldo[KMAX];
  unsigned jmask, okmask;
};

template <int KMAX>
__device__ __forceinline__ WLane<KMAX> wlane_init(const WParams& p, int lane, int cib) {
  WLane<KMAX> L;
  L.jmask = L.okmask = 0;
  const int cpi = p.pvec ? (p.P >> 2) : p.P;
  const int rowlen = p.PW * cpi;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = lane + 64 * k;
    const int pcol = j / cpi;
    const int cc = (j - pcol * cpi) * (p.pvec ? 4 : 1);
    const int iw = pcol - p.pl;
    const bool jv = (k < p.KI) && (j < rowlen);
    const bool ok = jv && (iw >= 0) && (iw < p.W) && (cc < cib);
    L.gofs[k] = ok ? (unsigned)((iw * p.CI + cc) * 4) : ODIN_OOB;
    L.ldo[k] = pcol * p.P + cc;
    if (jv) L.jmask |= 1u << k;
    if (ok) L.okmask |= 1u << k;
  }
  return L;
}

// Branch-free row staging (odin_device.h, OdinRun): a patch row is one range-checked run, zero
// bytes long when the row is SAME padding or beyond the batch.
template <int KMAX, bool PVEC>
__device__ __forceinline__ void wrow_issue(const WParams& p, const WLane<KMAX>& L, int r, int b0,
                                           int ih_lo, int ci0, float4* v) {
  const int img = (p.NIMG == 1) ? 0 : r / p.NRI;
  const int prow = r - img * p.NRI;
  const int b = b0 + img, ih = ih_lo + prow;
  const bool row_ok = (b < p.B) && (ih >= 0) && (ih < p.H);
  const float* rowp = p.in + (row_ok ? ((size_t)((b * p.H + ih) * p.W) * p.CI + ci0) : (size_t)0);
  const OdinRun R = odin_run(rowp, row_ok ? (unsigned)((p.W * p.CI - ci0) * 4) : 0u);
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if constexpr (PVEC) v[k] = odin_run_load4(R, L.gofs[k]);
    else v[k] = make_float4(odin_run_load1(R, L.gofs[k]), 0.f, 0.f, 0.f);
  }
}

template <int KMAX, bool PVEC>
__device__ __forceinline__ void wrow_commit(const WParams& p, const WLane<KMAX>& L, int r, int b0,
                                            int ih_lo, const float4* v, float* patch) {
  float* rowl = patch + r * p.PW * p.P;
  bool cen = false;
  if (p.center) {  // CenterAt0 on real pixels only (padding stays 0)
    const int img = (p.NIMG == 1) ? 0 : r / p.NRI;
    const int prow = r - img * p.NRI;
    const int b = b0 + img, ih = ih_lo + prow;
    cen = (b < p.B) && (ih >= 0) && (ih < p.H);
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if ((L.jmask >> k) & 1u) {
      float4 t = v[k];
      if (cen && ((L.okmask >> k) & 1u))
        t = make_float4(2.f * t.x - 1.f, 2.f * t.y - 1.f, 2.f * t.z - 1.f, 2.f * t.w - 1.f);
      if constexpr (PVEC) *reinterpret_cast<float4*>(rowl + L.ldo[k]) = t;
      else rowl[L.ldo[k]] = t.x;
    }
  }
}

template <int KMAX, int RPWMAX, bool FLAT, bool PVEC>
__device__ __forceinline__ void wpatch_issue(const WParams& p, const WLane<KMAX>& L, int wave,
                                             int tid, int batch, int b0, int ih_lo, int ci0,
                                             int cib, float4* pf) {
  constexpr int PFN = KMAX * RPWMAX, NT = NW_W * 64;
  if constexpr (FLAT) {
    const int cpi = PVEC ? (p.P >> 2) : p.P;
    const int total = p.NIMG * cpi;
    const long left = ((long)(p.B - b0) * p.CI - ci0) * 4;
    const OdinRun R = odin_run(p.in + (b0 < p.B ? (size_t)b0 * p.CI + ci0 : (size_t)0),
                               left <= 0 ? 0u : (left > 0x7FFFFFF0L ? 0x7FFFFFF0u : (unsigned)left));
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int e = (batch * PFN + i) * NT + tid;
      const int img = e / cpi;
      const int cc = (e - img * cpi) * (PVEC ? 4 : 1);
      const unsigned off = (e < total && cc < cib) ? (unsigned)((img * p.CI + cc) * 4) : ODIN_OOB;
      if constexpr (PVEC) pf[i] = odin_run_load4(R, off);
      else pf[i] = make_float4(odin_run_load1(R, off), 0.f, 0.f, 0.f);
    }
  } else {
    const int nrows_p = p.NIMG * p.NRI;
#pragma unroll
    for (int q = 0; q < RPWMAX; ++q) {
      const int r = batch * NW_W * RPWMAX + wave + NW_W * q;
      if (r < nrows_p) wrow_issue<KMAX, PVEC>(p, L, r, b0, ih_lo, ci0, pf + q * KMAX);
    }
  }
}

template <int KMAX, int RPWMAX, bool FLAT, bool PVEC>
__device__ __forceinline__ void wpatch_commit(const WParams& p, const WLane<KMAX>& L, int wave,
                                              int tid, int batch, int b0, int ih_lo,
                                              const float4* pf, float* patch) {
  constexpr int PFN = KMAX * RPWMAX, NT = NW_W * 64;
  if constexpr (FLAT) {
    const int cpi = PVEC ? (p.P >> 2) : p.P;
    const int total = p.NIMG * cpi;
#pragma unroll
    for (int i = 0; i < PFN; ++i) {
      const int e = (batch * PFN + i) * NT + tid;
      if (e < total) {
        const int img = e / cpi;
        const int cc = (e - img * cpi) * (PVEC ? 4 : 1);
        if constexpr (PVEC) *reinterpret_cast<float4*>(patch + img * p.P + cc) = pf[i];
        else patch[img * p.P + cc] = pf[i].x;
      }
    }
  } else {
    const int nrows_p = p.NIMG * p.NRI;
#pragma unroll
    for (int q = 0; q < RPWMAX; ++q) {
      const int r = batch * NW_W * RPWMAX + wave + NW_W * q;
      if (r < nrows_p) wrow_commit<KMAX, PVEC>(p, L, r, b0, ih_lo, pf + q * KMAX, patch);
    }
  }
}

// DY tile: item e < slots*cpd -> (slot = e / cpd, c = (e % cpd) * (vec ? 4 : 1))
template <int DMAX, int NT, bool DCONT>
__device__ __forceinline__ void wdy_issue(const WParams& p, int gr0, int co0, int tid, float4* v) {
  if constexpr (DCONT) {
    // DP == COB, COB/4 a power of two: item e -> slot e >> lg, channel group e & (cpd-1);
    // with COB == CO the tile's DY rows are one contiguous float4 run
    const int lg = p.dlog, cpd = 1 << lg;
    const int total = p.TR * p.OW * cpd;
    // run = the rest of the DY tensor from this tile's first pixel: rows beyond the tensor
    // (ragged last tile) fall outside it and read zeros
    const long left = (((long)p.B * p.OH * p.OW - (long)gr0 * p.OW) * p.CO - co0) * 4;
    const OdinRun R = odin_run(p.dy + (left > 0 ? (size_t)gr0 * p.OW * p.CO + co0 : (size_t)0),
                               left <= 0 ? 0u : (left > 0x7FFFFFF0L ? 0x7FFFFFF0u : (unsigned)left));
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
      const int e = tid + i * NT;
      const int cl = (e & (cpd - 1)) << 2;
      // channel groups beyond CO stay zero
      const bool ok = e < total && co0 + cl < p.CO;
      v[i] = odin_run_load4(R, ok ? (unsigned)(((e >> lg) * p.CO + cl) * 4) : ODIN_OOB);
    }
    return;
  }
  const int cpd = p.dvec ? (p.COB >> 2) : p.COB;
  const int total = p.slots * cpd;
  const int real_slots = p.TR * p.OW;
  const long left = (((long)p.B * p.OH * p.OW - (long)gr0 * p.OW) * p.CO - co0) * 4;
  const OdinRun R = odin_run(p.dy + (left > 0 ? (size_t)gr0 * p.OW * p.CO + co0 : (size_t)0),
                             left <= 0 ? 0u : (left > 0x7FFFFFF0L ? 0x7FFFFFF0u : (unsigned)left));
#pragma unroll
  for (int i = 0; i < DMAX; ++i) {
    const int e = tid + i * NT;
    const int sl = e / cpd;
    const int cl = (e - sl * cpd) * (p.dvec ? 4 : 1);
    const bool ok = e < total && sl < real_slots && co0 + cl < p.CO;
    const unsigned off = ok ? (unsigned)((sl * p.CO + cl) * 4) : ODIN_OOB;
    if (p.dvec) v[i] = odin_run_load4(R, off);
    else v[i] = make_float4(odin_run_load1(R, off), 0.f, 0.f, 0.f);
  }
}

template <int DMAX, int NT, bool DCONT>
__device__ __forceinline__ void wdy_commit(const WParams& p, int tid, const float4* v, float* dyl,
                                           float4& bsum) {
  if constexpr (DCONT) {
    const int total = p.slots << p.dlog;
#pragma unroll
    for (int i = 0; i < DMAX; ++i) {
      const int e = tid + i * NT;
      if (e < total) {
        bsum.x += v[i].x; bsum.y += v[i].y; bsum.z += v[i].z; bsum.w += v[i].w;
        reinterpret_cast<float4*>(dyl)[e] = v[i];
      }
    }
    return;
  }
  const int cpd = p.dvec ? (p.COB >> 2) : p.COB;
  const int total = p.slots * cpd;
#pragma unroll
  for (int i = 0; i < DMAX; ++i) {
    const int e = tid + i * NT;
    if (e < total) {
      // bias gradient (bias_mode 2): every item of this thread carries the same 4 channels
      bsum.x += v[i].x; bsum.y += v[i].y; bsum.z += v[i].z; bsum.w += v[i].w;
      const int sl = e / cpd;
      const int cl = (e - sl * cpd) * (p.dvec ? 4 : 1);
      if (p.dvec) *reinterpret_cast<float4*>(dyl + sl * p.DP + cl) = v[i];
      else dyl[sl * p.DP + cl] = v[i].x;
    }
  }
}

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define W_STAMP(k) ((void)0)
#else
#define W_STAMP(k)                                                                       \
  do {                                                                                   \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 &&  \
        threadIdx.x == 0 && stamp_i < 60)                                                \
      p.stamps[stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

// FAST: one output-channel tile per workgroup (all accumulators share the DY operand), every
// accumulator row is a real weight row (no padding rows, no MFMA bias tile): the hot loop
// is 1 DY read + TNACC patch reads + TNACC MFMAs per pixel pair.
// TSP > 0 (= S * P floats, FAST only): the tile is walked in chunks of 4 pixel pairs of one output
// row; all LDS operand addresses of a chunk are one per-lane base + compile-time offsets, so the
// hot loop is 16 MFMAs + 20 LDS reads + 5 address adds per chunk (no per-pair control flow).
template <int TNACC, int KMAX, int RPWMAX, int DMAX, bool FLAT, bool PVEC, bool DCONT, bool FAST, int TSP>
__global__ __launch_bounds__(NW_W * 64) void wgrad_kernel(WParams p) {
  ODIN_DYN_SMEM(float, smem);
  float* patch = smem;
  float* dyl = smem + p.patch_floats;
  int* tbl = reinterpret_cast<int*>(smem + p.patch_floats + p.dy_floats);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = W_UNIFORM(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int NT = NW_W * 64;
  const int ci0 = blockIdx.y * p.CIB, co0 = blockIdx.z * p.COB;
  const int cib = (p.CI - ci0) < p.CIB ? (p.CI - ci0) : p.CIB;
  const int ntaps = p.KH * p.KW;
  const int nrows = ntaps * cib;
  const int n_wt = p.nrt * p.ncot;
  const int n_bias = (p.bias_mode == 1 && blockIdx.y == 0) ? p.ncot : 0;
  const int n_tot = n_wt + n_bias;
  int stamp_i = 0;
  (void)stamp_i;
  W_STAMP(1);
  const bool pipelined = p.pipelined != 0;
  const WLane<KMAX> WL = wlane_init<KMAX>(p, lane, cib);

  // slot -> patch base table (identical for every tile of this launch)
  for (int s = tid; s < p.slots + 4; s += NT) {
    int r = s / p.OW, c = s - r * p.OW;
    int img = r / p.RPI, rl = r - img * p.RPI;
    tbl[s] = (s < p.slots && r < p.TR) ? ((img * p.NRI + rl * p.S) * p.PW + c * p.S) * p.P : 0;
  }
  // zero pad rows of the DY tile (read by the pipeline's look-ahead, multiplied into nothing)
  for (int e = tid; e < 4 * p.DP; e += NT) dyl[p.slots * p.DP + e] = 0.f;

  // per-accumulator lane constants.  A operand = patch value * amul + aadd: (1,0) for a
  // weight-gradient row, (0,1) for the MFMA bias tile (bias_mode 1), (0,0) for padding
  // rows / unused accumulators -- branch-free in the hot loop.
  int a_off[TNACC], b_off[TNACC];
  float amul[TNACC], aadd[TNACC];
#pragma unroll
  for (int a = 0; a < TNACC; ++a) {
    int T = wave + a * NW_W;
    a_off[a] = 0;
    b_off[a] = l31;
    amul[a] = 0.f;
    aadd[a] = 0.f;
    if (T < n_wt) {
      int rt = T / p.ncot, cot = T - rt * p.ncot;
      int rl = rt * 32 + l31;
      b_off[a] = cot * 32 + l31;
      if (rl < nrows) {
        int tap = rl / cib, cl = rl - tap * cib;
        int kh = tap / p.KW, kw = tap - kh * p.KW;
        a_off[a] = (kh * p.PW + kw) * p.P + cl;
        amul[a] = 1.f;
      }
    } else if (T < n_tot) {
      b_off[a] = (T - n_wt) * 32 + l31;
      aadd[a] = 1.f;
    }
  }
  float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x16 acc[TNACC];
#pragma unroll
  for (int a = 0; a < TNACC; ++a) acc[a] = f32x16_zero();

  float4 pf[RPWMAX * KMAX], df[DMAX];
  int tile = blockIdx.x;
  if (pipelined && tile < p.n_tiles) {
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    wpatch_issue<KMAX, RPWMAX, FLAT, PVEC>(p, WL, wave, tid, 0, b0, oh0 * p.S - p.pt, ci0, cib, pf);
    wdy_issue<DMAX, NT, DCONT>(p, gr0, co0, tid, df);
  }

  W_STAMP(3);
  for (; tile < p.n_tiles; tile += gridDim.x) {
    W_STAMP(4);
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = oh0 * p.S - p.pt;
    __syncthreads();
    if (pipelined) {
      wpatch_commit<KMAX, RPWMAX, FLAT, PVEC>(p, WL, wave, tid, 0, b0, ih_lo, pf, patch);
      wdy_commit<DMAX, NT, DCONT>(p, tid, df, dyl, bsum4);
      __syncthreads();
      W_STAMP(5);
      const int nt = tile + gridDim.x;
      if (nt < p.n_tiles) {
        const int g2 = nt * p.TR;
        const int b2 = g2 / p.OH, o2 = g2 - b2 * p.OH;
        wpatch_issue<KMAX, RPWMAX, FLAT, PVEC>(p, WL, wave, tid, 0, b2, o2 * p.S - p.pt, ci0, cib, pf);
        wdy_issue<DMAX, NT, DCONT>(p, g2, co0, tid, df);
      }
    } else {
      for (int bt = 0; bt < p.n_batches; ++bt) {
        wpatch_issue<KMAX, RPWMAX, FLAT, PVEC>(p, WL, wave, tid, bt, b0, ih_lo, ci0, cib, pf);
        wpatch_commit<KMAX, RPWMAX, FLAT, PVEC>(p, WL, wave, tid, bt, b0, ih_lo, pf, patch);
      }
      const int cpd = p.dvec ? (p.COB >> 2) : p.COB;
      for (int e0 = 0; e0 < p.slots * cpd; e0 += NT * DMAX) {
        // DY rows in batches of DMAX items per thread (item index = (tid + e0) + i*NT)
        wdy_issue<DMAX, NT, DCONT>(p, gr0, co0, tid + e0, df);
        wdy_commit<DMAX, NT, DCONT>(p, tid + e0, df, dyl, bsum4);
      }
      __syncthreads();
    }
    W_STAMP(6);
    const int npairs = p.slots >> 1;
    // two-stage register pipeline over pixel pairs: operands of pair kp+1 are read from
    // LDS while the MFMAs of pair kp execute
    float av[2][TNACC], bv[2][TNACC];
    // raw LDS operands; the (amul, aadd) fix-up is applied right before the MFMA so that
    // the reads of pair kp+1 stay in flight under the MFMAs of pair kp.
    const bool dense11 = (p.OW == 1 && p.OH == 1);  // Dense: slots are consecutive images
    if constexpr (TSP > 0) {
      constexpr int UN = (TNACC > 4) ? 2 : 4;  // pixel pairs per pipeline stage
      constexpr int PS2 = 2 * TSP;  // patch floats between consecutive pairs of a row
      const int nchunks = npairs / UN;
      const int cpr = p.OW / (2 * UN);          // chunks per output row
      const int row_stride = p.S * p.PW * p.P;  // patch floats between output rows
      const float* abase[TNACC];
#pragma unroll
      for (int a = 0; a < TNACC; ++a) abase[a] = patch + h * TSP + a_off[a];
      const float* bbase = dyl + h * 32 + l31;
      int crow = 0, ccol = 0, rl = 0, img = 0, dyo = 0;  // wave-uniform chunk iterator
      float a0[UN][TNACC], b0[UN], a1[UN][TNACC], b1[UN];
      auto chunk_load = [&](float (&a_)[UN][TNACC], float (&b_)[UN]) {
        const int ao = crow + ccol * (UN * PS2);
#pragma unroll
        for (int u = 0; u < UN; ++u) b_[u] = bbase[dyo + u * 64];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
          for (int a = 0; a < TNACC; ++a) a_[u][a] = abase[a][ao + u * PS2];
        }
        dyo += UN * 64;
        if (++ccol == cpr) {
          ccol = 0;
          crow += row_stride;
          if (++rl == p.RPI) { rl = 0; ++img; crow = img * p.NRI * p.PW * p.P; }
        }
      };
      auto chunk_mfma = [&](const float (&a_)[UN][TNACC], const float (&b_)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
          for (int a = 0; a < TNACC; ++a) acc[a] = mfma32(a_[u][a], b_[u], acc[a]);
        }
        // the next chunk's UN * (TNACC + 1) LDS reads go into the shadow of these MFMAs
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
          ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 2);
#pragma unroll
          for (int a = 1; a < TNACC; ++a) {
            ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
            ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
          }
        }
        ODIN_SCHED_FENCE();
      };
      chunk_load(a0, b0);
      ODIN_SCHED_FENCE();
      for (int ch = 0; ch < nchunks; ch += 2) {
        if (ch + 1 < nchunks) chunk_load(a1, b1);
        chunk_mfma(a0, b0);
        if (ch + 2 < nchunks) chunk_load(a0, b0);
        if (ch + 1 < nchunks) chunk_mfma(a1, b1);
      }
    } else if ((p.OW & 1) == 0 || dense11) {
      // fast path: both pixels of a pair lie in one output row, so the patch base is a
      // wave-uniform running value (no table look-up on the critical path)
      int pb = 0, pc = 0, prl = 0, pimg = 0, pslot = 0;
      const int img_stride = p.NRI * p.PW * p.P;
      const int lane_h = dense11 ? h * img_stride : h * p.S * p.P;
      const int real_slots = p.TR * p.OW;
      auto load_next = [&](float* a_, float* b_) {
        // per-lane: the odd pixel of the last pair may be a pad slot (its DY row is zero, but
        // the patch beyond the staged images must not be read)
        const int base = (pslot + h < real_slots) ? pb + lane_h : 0;
        const float* brow = dyl + (pslot + h) * p.DP;
        if constexpr (FAST) {
          b_[0] = brow[l31];
#pragma unroll
          for (int a = 0; a < TNACC; ++a) a_[a] = patch[base + a_off[a]];
        } else {
#pragma unroll
          for (int a = 0; a < TNACC; ++a) {
            a_[a] = patch[base + a_off[a]];
            b_[a] = brow[b_off[a]];
          }
        }
        pslot += 2;
        if (dense11) {
          pb += 2 * img_stride;
          return;
        }
        pc += 2;
        pb += 2 * p.S * p.P;
        if (pc >= p.OW) {
          pc = 0;
          ++prl;
          pb += (p.S * p.PW - p.OW * p.S) * p.P;
          if (prl == p.RPI) { prl = 0; ++pimg; pb = pimg * p.NRI * p.PW * p.P; }
        }
      };
      auto mfma_all = [&](const float* a_, const float* b_) {
        if constexpr (FAST) {
#pragma unroll
          for (int a = 0; a < TNACC; ++a) acc[a] = mfma32(a_[a], b_[0], acc[a]);
        } else {
#pragma unroll
          for (int a = 0; a < TNACC; ++a)
            acc[a] = mfma32(fmaf(a_[a], amul[a], aadd[a]), b_[a], acc[a]);
        }
#pragma unroll
        for (int a = 0; a < TNACC; ++a) {
          ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
          ODIN_SCHED_GROUP(ODIN_SG_DSREAD, FAST ? 1 : 2);
        }
        ODIN_SCHED_FENCE();
      };
      load_next(av[0], bv[0]);
      for (int kp = 0; kp < npairs; kp += 2) {
        load_next(av[1], bv[1]);
        mfma_all(av[0], bv[0]);
        load_next(av[0], bv[0]);
        mfma_all(av[1], bv[1]);
      }
    } else {
      auto load_pair = [&](int kp, int base, float* a_, float* b_) {
        const float* brow = dyl + (2 * kp + h) * p.DP;
#pragma unroll
        for (int a = 0; a < TNACC; ++a) {
          a_[a] = patch[base + a_off[a]];
          b_[a] = brow[b_off[a]];
        }
      };
      int base_b = tbl[2 + h];
      load_pair(0, tbl[h], av[0], bv[0]);
      for (int kp = 0; kp < npairs; kp += 2) {
        const int base_c = tbl[2 * (kp + 2) + h];
        const int base_d = tbl[2 * (kp + 3) + h];
        load_pair(kp + 1, base_b, av[1], bv[1]);
#pragma unroll
        for (int a = 0; a < TNACC; ++a)
          acc[a] = mfma32(fmaf(av[0][a], amul[a], aadd[a]), bv[0][a], acc[a]);
        load_pair(kp + 2, base_c, av[0], bv[0]);
#pragma unroll
        for (int a = 0; a < TNACC; ++a)
          acc[a] = mfma32(fmaf(av[1][a], amul[a], aadd[a]), bv[1][a], acc[a]);
        base_b = base_d;
      }
    }
  }

  W_STAMP(8);
  // ---- write this workgroup's partial tiles into its slab row ----
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
#pragma unroll
  for (int a = 0; a < TNACC; ++a) {
    int T = wave + a * NW_W;
    if (T < n_wt) {
      int rt = T / p.ncot, cot = T - rt * p.ncot;
      int co = co0 + cot * 32 + l31;
      // (tap, channel) of the tile's first row by ONE division; the 16 rows of this lane then step
      // from it (a per-element `rl / cib` cost ~9 us of integer division on the small layers)
      const int tap0 = (rt * 32) / cib, cl0 = rt * 32 - tap0 * cib;
      const bool col_ok = co < p.CO && cot * 32 + l31 < p.COB;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int off = (r & 3) + 8 * (r >> 2) + 4 * h;
        int tap = tap0, cl = cl0 + off;
        while (cl >= cib) { cl -= cib; ++tap; }
        if (rt * 32 + off < nrows && col_ok)
          row[((size_t)tap * p.CI + ci0 + cl) * p.CO + co] = acc[a][r];
      }
    } else if (T < n_tot) {
      int cot = T - n_wt;
      int co = co0 + cot * 32 + l31;
      if (h == 0 && co < p.CO && cot * 32 + l31 < p.COB)
        row[(size_t)ntaps * p.CI * p.CO + co] = acc[a][0];
    }
  }
  W_STAMP(9);
  if (p.bias_mode == 2 && blockIdx.y == 0) {
    // thread t accumulated channels 4*(t % cpd) .. +3 of this block's output-channel slice
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(smem);
    red[tid] = bsum4;
    __syncthreads();
    const int cpd = p.COB >> 2;
    if (tid < cpd) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int u = tid; u < NT; u += cpd) {
        t.x += red[u].x; t.y += red[u].y; t.z += red[u].z; t.w += red[u].w;
      }
      float* bp = row + (size_t)ntaps * p.CI * p.CO + co0 + 4 * tid;
      if (co0 + 4 * tid + 3 < p.CO) { bp[0] = t.x; bp[1] = t.y; bp[2] = t.z; bp[3] = t.w; }
    }
  }
}

// --------------------------------------------------------------------------------------
// Producer / consumer variant (FAST + row-chunk shapes).  The workgroup has 8 waves, two per
// SIMD: waves 0-3 own the dW accumulators and do nothing but LDS reads + MFMAs; waves 4-7
// stage the NEXT tile (IN patch + DY rows) from HBM into the other half of a double-buffered
// LDS area.  One barrier per tile.  Staging cost and HBM latency disappear behind the matrix
// pipe instead of adding to it (at one wave per SIMD they are ~1/3 of the tile time).
// Tiles are 64 pixels so that two buffers fit the 160 KB LDS.
// --------------------------------------------------------------------------------------
#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define WS_STAMP(base, k) ((void)0)
#else
#define WS_STAMP(base, k)                                                                 \
  do {                                                                                    \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 &&   \
        lane == 0 && stamp_i < 31)                                                        \
      p.stamps[(base) + stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

template <int TNACC, int KMAX, int RPWMAX, int DMAX, int TSP>
__global__ __launch_bounds__(2 * NW_W * 64) void wgrad_ws_kernel(WParams p) {
  ODIN_DYN_SMEM(float, smem);
  const int tid = threadIdx.x, lane = tid & 63;
  // producers are waves 0-3 (dispatched first = older: VALU/LDS issue arbitration on a SIMD goes
  // by age), consumers waves 4-7
  const int wave_hw = W_UNIFORM(tid >> 6);
  const int wave = wave_hw ^ NW_W;  // role index: 0-3 consumer, 4-7 producer
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int NT = NW_W * 64;  // threads per role
  const bool loader = wave >= NW_W;
  const int buf_floats = p.patch_floats + p.dy_floats;
  const int ci0 = blockIdx.y * p.CIB, co0 = blockIdx.z * p.COB;
  const int cib = (p.CI - ci0) < p.CIB ? (p.CI - ci0) : p.CIB;
  const int ntaps = p.KH * p.KW;
  const int nrows = ntaps * cib;
  const int n_wt = p.nrt * p.ncot;
  const int n_my = ((int)blockIdx.x < p.n_tiles)
                       ? (p.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  float4 bsum4 = make_float4(0.f, 0.f, 0.f, 0.f);
  int stamp_i = 0;
  (void)stamp_i;
  float* row = p.slab + (size_t)blockIdx.x * p.slab_stride;
  // SAME-padding columns / unused channel slots of both buffers are zero for the whole kernel
  for (int e = tid; e < (2 * buf_floats) >> 2; e += 2 * NT)
    reinterpret_cast<float4*>(smem)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  if (loader) {
    // ---------------- producer waves: LDS-DMA ----------------
    // (staging through registers starves under the consumers' MFMA stream: the ds_write pass of
    // one tile took ~10k cycles beside it, 2.6k alone.)  The patch row span that holds real
    // pixels, columns [pl, pl + W), is contiguous in LDS, so a row is ceil(W*P/256) DMA
    // instructions; SAME-padding columns and channel slots beyond the block are zeroed once
    // (nothing ever writes them), SAME-padding rows are DMA'd from a zero row.
    const int lw = wave - NW_W, ltid = tid;
    const int c4n = p.P >> 2;                 // 16-byte pieces per pixel
    const int npieces = p.W * c4n;            // pieces of the real-pixel span of a row
    unsigned gofs[KMAX];
    bool gok[KMAX];
#pragma unroll
    for (int i = 0; i < KMAX; ++i) {
      const int j = lane + 64 * i;
      const int iw = j / c4n, cc = (j - iw * c4n) * 4;
      gok[i] = j < npieces && cc < cib;
      gofs[i] = (unsigned)((iw * p.CI + cc) * 4);
    }
    const OdinRun ZR = odin_run(odin_zero_row, (unsigned)sizeof(odin_zero_row));
    const int span0 = p.pl * p.P;  // float offset of the span inside a patch row
    for (int k = 0; k <= n_my; ++k) {
      if (lw == 0) WS_STAMP(32, 20);
      if (k < n_my) {
        float* patch = smem + (k & 1) * buf_floats;
        float* dyl = patch + p.patch_floats;
        const int gr0 = (blockIdx.x + k * gridDim.x) * p.TR;
        const int b0 = gr0 / p.OH;
        const int ih_lo = (gr0 - b0 * p.OH) * p.S - p.pt;
        // DY rows of the tile: slots * 8 pieces, lane-linear
        {
          const OdinRun RD = odin_run(p.dy + (size_t)gr0 * p.OW * p.CO + co0,
                                      (unsigned)((p.TR * p.OW * p.CO - co0) * 4));
#pragma unroll
          for (int i = 0; i < DMAX; ++i) {
            const int j = i * NT + lw * 64 + lane;
            if (j < p.slots * 8)
              odin_run_dma16(RD, dyl + (i * NT + lw * 64) * 4,
                             (unsigned)(((j >> 3) * p.CO + (j & 7) * 4) * 4), lane);
          }
        }
#pragma unroll
        for (int q = 0; q < RPWMAX; ++q) {
          const int r = lw + NW_W * q;
          if (r < p.NRI) {
            const int ih = ih_lo + r;
            const bool row_ok = (ih >= 0) && (ih < p.H);
            float* dst = patch + r * p.PW * p.P + span0;
            if (row_ok) {
              const OdinRun R = odin_run(p.in + ((size_t)((b0 * p.H + ih) * p.W) * p.CI + ci0),
                                         (unsigned)((p.W * p.CI - ci0) * 4));
#pragma unroll
              for (int i = 0; i < KMAX; ++i)
                if (gok[i]) odin_run_dma16(R, dst + i * 256, gofs[i], lane);
            } else {
#pragma unroll
              for (int i = 0; i < KMAX; ++i)
                if (gok[i]) odin_run_dma16(ZR, dst + i * 256, (unsigned)((lane + 64 * i) * 16), lane);
            }
          }
        }
        if (lw == 0) WS_STAMP(32, 21);
        if (p.bias_mode == 2) {
          // bias gradient: read this thread's DY pieces back once they have landed
          odin_wait_vmem();
#pragma unroll
          for (int i = 0; i < DMAX; ++i) {
            const int j = i * NT + ltid;
            if (j < p.slots * 8) {
              const float4 t = reinterpret_cast<const float4*>(dyl)[j];
              bsum4.x += t.x; bsum4.y += t.y; bsum4.z += t.z; bsum4.w += t.w;
            }
          }
        }
        if (lw == 0) WS_STAMP(32, 22);
      }
      __syncthreads();
    }
  } else {
    // ---------------- consumer waves ----------------
    // (the accumulators live only in this role: the producer's staging registers and the
    // consumer's accumulators then share the 256-register budget of a 2-waves-per-SIMD kernel)
    f32x16 acc[TNACC];
#pragma unroll
    for (int a = 0; a < TNACC; ++a) acc[a] = f32x16_zero();
    int a_off[TNACC];
#pragma unroll
    for (int a = 0; a < TNACC; ++a) {
      const int T = wave + a * NW_W;
      a_off[a] = 0;
      if (T < n_wt) {
        const int rl = T * 32 + l31;  // ncot == 1
        if (rl < nrows) {
          const int tap = rl / cib, cl = rl - tap * cib;
          const int kh = tap / p.KW, kw = tap - kh * p.KW;
          a_off[a] = (kh * p.PW + kw) * p.P + cl;
        }
      }
    }
    constexpr int UN = 4;
    constexpr int PS2 = 2 * TSP;
    const int cpr = p.OW / (2 * UN);  // chunks per output row: 1, 2 or 4 (launcher)
    const int cshift = cpr == 1 ? 0 : (cpr == 2 ? 1 : 2);
    const int row_stride = p.S * p.PW * p.P;
    for (int k = 0; k <= n_my; ++k) {
      if (wave == 0) WS_STAMP(0, 10);
      if (k >= 1) {
        const float* patch = smem + ((k - 1) & 1) * buf_floats;
        const float* dyl = patch + p.patch_floats;
        const float* abase[TNACC];
#pragma unroll
        for (int a = 0; a < TNACC; ++a) abase[a] = patch + h * TSP + a_off[a];
        const float* bbase = dyl + h * 32 + l31;
        // The tile is exactly 8 chunks (64 slots, enforced by the launcher) and one image: the
        // chunk loop is fully unrolled and branch-free (chunk c -> row c >> cshift, column block
        // c & (cpr - 1), pure scalar arithmetic), so that the LDS reads of chunk c + 1 really sit
        // between the MFMAs of chunk c; with `if (c + 1 < nchunks)` guards the loads landed in
        // their own basic blocks and the 16 MFMAs ran back to back behind them.
        float a0[UN][TNACC], b0[UN], a1[UN][TNACC], b1[UN];
        auto chunk_load = [&](int c, float (&a_)[UN][TNACC], float (&b_)[UN]) {
          const int ao = (c >> cshift) * row_stride + (c & (cpr - 1)) * (UN * PS2);
          const int dyo = c * (UN * 64);
#pragma unroll
          for (int u = 0; u < UN; ++u) b_[u] = bbase[dyo + u * 64];
#pragma unroll
          for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int a = 0; a < TNACC; ++a) a_[u][a] = abase[a][ao + u * PS2];
          }
        };
        auto chunk_mfma = [&](const float (&a_)[UN][TNACC], const float (&b_)[UN], bool with_loads) {
#pragma unroll
          for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int a = 0; a < TNACC; ++a) acc[a] = mfma32(a_[u][a], b_[u], acc[a]);
          }
          if (with_loads) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
              ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
              ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 2);
#pragma unroll
              for (int a = 1; a < TNACC; ++a) {
                ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
                ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
              }
            }
          }
          ODIN_SCHED_FENCE();
        };
        chunk_load(0, a0, b0);
        ODIN_SCHED_FENCE();
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
          chunk_load(c + 1, a1, b1);
          chunk_mfma(a0, b0, true);
          if (c + 2 < 8) chunk_load(c + 2, a0, b0);
          chunk_mfma(a1, b1, c + 2 < 8);
        }
        if (wave == 0) WS_STAMP(0, 11);
      }
      __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TNACC; ++a) {
      const int T = wave + a * NW_W;
      if (T < n_wt) {
        const int co = co0 + l31;
        // one division per tile instead of one per element (see wgrad_kernel's epilogue)
        const int tap0 = (T * 32) / cib, cl0 = T * 32 - tap0 * cib;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int off = (r & 3) + 8 * (r >> 2) + 4 * h;
          int tap = tap0, cl = cl0 + off;
          while (cl >= cib) { cl -= cib; ++tap; }
          if (T * 32 + off < nrows && co < p.CO)
            row[((size_t)tap * p.CI + ci0 + cl) * p.CO + co] = acc[a][r];
        }
      }
    }
  }
  if (p.bias_mode == 2 && blockIdx.y == 0) {
    // producer thread t accumulated channels 4*(t % cpd) .. +3 while staging DY
    float4* red = reinterpret_cast<float4*>(smem);
    if (loader) red[tid] = bsum4;
    __syncthreads();
    const int cpd = p.COB >> 2;
    if (tid < cpd) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int u = tid; u < NT; u += cpd) {
        t.x += red[u].x; t.y += red[u].y; t.z += red[u].z; t.w += red[u].w;
      }
      float* bp = row + (size_t)ntaps * p.CI * p.CO + co0 + 4 * tid;
      if (co0 + 4 * tid + 3 < p.CO) { bp[0] = t.x; bp[1] = t.y; bp[2] = t.z; bp[3] = t.w; }
    }
  }
}


bool plan_wgrad(WParams& p, int* gx, int* gy, int* gz, size_t* lds_bytes, int target = 128,
                int lds_budget_bytes = 160 * 1024 - 2048, int cib_limit = 1 << 30) {
  const int S = p.S;
  const int img_pix = p.OH * p.OW;
  const int TARGET = target;
  if (img_pix <= TARGET) {
    p.NIMG = TARGET / img_pix;
    if (p.NIMG > p.B) p.NIMG = p.B;
    if (p.NIMG < 1) p.NIMG = 1;
    p.RPI = p.OH;
    p.TR = p.NIMG * p.OH;
  } else {
    p.NIMG = 1;
    int pick = 0;
    for (int tr = 1; tr <= p.OH; ++tr)
      if (p.OH % tr == 0 && tr * p.OW <= TARGET) pick = tr;
    if (pick == 0) pick = 1;
    p.TR = p.RPI = pick;
  }
  p.n_tiles = (p.B * p.OH + p.TR - 1) / p.TR;
  p.NRI = (p.RPI - 1) * S + p.KH;
  p.PW = (p.OW - 1) * S + p.KW;
  p.slots = (p.TR * p.OW + 3) & ~3;
  const int ntaps = p.KH * p.KW;
  const int budget = lds_budget_bytes / 4;
  const int co32 = (p.CO + 31) / 32 * 32;
  // 32 output channels per workgroup first: all accumulators then share the DY operand
  // (FAST loop); the IN patch is re-read once per channel block (L2-resident)
  int cob_c[2] = {32, co32 < 64 ? co32 : 64};
  int cib_c[6] = {p.CI, 256, 128, 64, 32, 16};
  for (int ic = 0; ic < 2; ++ic) {
    int COB = cob_c[ic];
    if (COB > co32) continue;
    int ncot = COB / 32;
    for (int jc = 0; jc < 6; ++jc) {
      int CIB = cib_c[jc];
      if (CIB > p.CI || CIB > cib_limit) continue;
      if (jc > 0 && CIB == p.CI) continue;
      int nrt = (ntaps * CIB + 31) / 32;
      const bool dv = ((p.CO & 3) == 0) && ((COB & 3) == 0);
      const int bmode = !p.want_bias ? 0 : ((dv && (256 % (COB / 4)) == 0) ? 2 : 1);
      if (nrt * ncot + (bmode == 1 ? ncot : 0) > NACC * NW_W) continue;
      int P = CIB;
      if ((p.CI & 3) == 0 && (CIB & 3) != 0) continue;
      const bool pv = ((p.CI & 3) == 0) && ((CIB & 3) == 0);
      const bool flat0 = (p.PW == 1 && p.NRI == 1);
      if (!flat0 && (long)p.PW * (pv ? CIB / 4 : CIB) > 64 * 9) continue;  // <= 9 items per lane per row
      long pf = ((long)p.NIMG * p.NRI * p.PW * P + 3) & ~3L;
      const int cpd0 = COB / 4;
      const bool pow2 = ((p.CO & 3) == 0) && (cpd0 & (cpd0 - 1)) == 0;
      int DP = pow2 ? COB : COB + 4;
      long df = (long)(p.slots + 4) * DP;
      if (pf + df + p.slots + 16 > budget) continue;
      p.CIB = CIB; p.COB = COB; p.P = P; p.DP = DP;
      p.nrt = nrt; p.ncot = ncot;
      p.dlog = -1;
      if (pow2) { int l = 0; while ((1 << l) < cpd0) ++l; p.dlog = l; }
      p.bias_mode = bmode;
      p.pvec = pv ? 1 : 0;
      p.dvec = (((p.CO & 3) == 0) && ((COB & 3) == 0)) ? 1 : 0;
      p.KI = (p.PW * (pv ? CIB / 4 : CIB) + 63) / 64;
      p.flat = flat0 ? 1 : 0;
      if (p.flat) p.KI = 1;
      p.patch_floats = (int)pf;
      p.dy_floats = (int)df;
      *lds_bytes = (size_t)(pf + df + p.slots + 16) * 4;
      if (*lds_bytes < 4352) *lds_bytes = 4352;  // room for the end-of-kernel bias reduction
      int cap = ODIN_MAX_SLAB_BLOCKS / (((p.CI + CIB - 1) / CIB) * ((p.CO + COB - 1) / COB));
      if (cap < 32) cap = 32;
      int g = p.n_tiles < cap ? p.n_tiles : cap;
      *gx = g < 1 ? 1 : g;
      *gy = (p.CI + CIB - 1) / CIB;
      *gz = (p.CO + COB - 1) / COB;
      return true;
    }
  }
  return false;
}

long long* g_wstamps = nullptr;

template <int TNACC, int KMAX, int RPWMAX, int DMAX, bool FLAT, bool PVEC, bool DCONT, bool FAST, int TSP = 0>
int launch_winst3(WParams& p, dim3 grid, size_t lds, void* stream) {
  const int rpw = (p.NIMG * p.NRI + NW_W - 1) / NW_W;
  const int ditems = p.slots * (p.dvec ? p.COB / 4 : p.COB);
  if (FLAT) {
    const int items = p.NIMG * (p.pvec ? p.P / 4 : p.P);
    p.n_batches = (items + KMAX * RPWMAX * NW_W * 64 - 1) / (KMAX * RPWMAX * NW_W * 64);
  } else {
    p.n_batches = (rpw + RPWMAX - 1) / RPWMAX;
  }
  p.pipelined = (p.n_batches == 1 && ditems <= DMAX * NW_W * 64 && p.KI <= KMAX) ? 1 : 0;
  if (p.KI > KMAX) return odin_fail(-2, "wgrad: patch row too long for this instance");
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<TNACC, KMAX, RPWMAX, DMAX, FLAT, PVEC, DCONT, FAST, TSP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((wgrad_kernel<TNACC, KMAX, RPWMAX, DMAX, FLAT, PVEC, DCONT, FAST, TSP>), grid, dim3(NW_W * 64), lds, stream, p);
  return odin_check_launch("wgrad");
}

template <int TNACC, int KMAX, int RPWMAX, int DMAX, bool FLAT, bool PVEC, bool DCONT>
int launch_winst2(WParams& p, dim3 grid, size_t lds, void* stream) {
  // FAST needs only: one output-channel tile per workgroup and no MFMA bias tile.  Padding
  // rows / unused accumulators then hold garbage that is never written to the slab.
  if (PVEC && DCONT && p.ncot == 1 && p.bias_mode != 1 &&
      ((p.OW & 1) == 0 || (p.OW == 1 && p.OH == 1))) {
    if constexpr (PVEC && DCONT && !FLAT) {
      // row-chunk loop: whole chunks of 4 pixel pairs per output row, no pad slots
      static int norow = -1;
      if (norow < 0) norow = ODIN_DIAG_ENV("ODIN_NOROWCHUNK") ? 1 : 0;
      if (!norow && (p.OW % 8) == 0 && p.TR * p.OW == p.slots && p.COB == 32 && p.DP == 32) {
        const int sp = p.S * p.P;
        if (sp == 32) return launch_winst3<TNACC, KMAX, RPWMAX, DMAX, FLAT, true, true, true, 32>(p, grid, lds, stream);
        if (sp == 64) return launch_winst3<TNACC, KMAX, RPWMAX, DMAX, FLAT, true, true, true, 64>(p, grid, lds, stream);
        if (sp == 128) return launch_winst3<TNACC, KMAX, RPWMAX, DMAX, FLAT, true, true, true, 128>(p, grid, lds, stream);
      }
    }
    return launch_winst3<TNACC, KMAX, RPWMAX, DMAX, FLAT, PVEC, DCONT, (PVEC && DCONT)>(
        p, grid, lds, stream);
  }
  return launch_winst3<TNACC, KMAX, RPWMAX, DMAX, FLAT, PVEC, DCONT, false>(p, grid, lds, stream);
}

// fast variant (16-byte patch items, contiguous DY) when the shape allows, generic otherwise
template <int TNACC, int KMAX, int RPWMAX, int DMAX, bool FLAT = false>
int launch_winst(WParams& p, dim3 grid, size_t lds, void* stream) {
  const bool dcont = p.dvec && p.dlog >= 0 && p.DP == p.COB;
  if (p.pvec && dcont) return launch_winst2<TNACC, KMAX, RPWMAX, DMAX, FLAT, true, true>(p, grid, lds, stream);
  if (p.pvec) return launch_winst2<TNACC, KMAX, RPWMAX, DMAX, FLAT, true, false>(p, grid, lds, stream);
  return launch_winst2<TNACC, KMAX, RPWMAX, DMAX, FLAT, false, false>(p, grid, lds, stream);
}

template <int KMAX, int RPWMAX, int TSP>
int launch_ws_inst(WParams& p, dim3 grid, size_t lds, void* stream) {
  const int rpw = (p.NIMG * p.NRI + NW_W - 1) / NW_W;
  p.n_batches = (rpw + RPWMAX - 1) / RPWMAX;
  // (try_launch_ws has checked every qualifier already; these can no longer trigger)
  if (p.n_batches != 1 || p.KI > KMAX || p.NIMG != 1 || (p.OH % p.TR) != 0 ||
      (size_t)p.W * p.P * 4 > 16384 || p.pl < 0 || (p.CO % 32) != 0)
    return odin_fail(-2, "wgrad_ws: plan and launch disagree");
  p.pipelined = 1;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_ws_kernel<4, KMAX, RPWMAX, 2, TSP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((wgrad_ws_kernel<4, KMAX, RPWMAX, 2, TSP>), grid, dim3(2 * NW_W * 64), lds, stream, p);
  return odin_check_launch("wgrad_ws");
}

// Try the producer/consumer kernel: 64-pixel tiles, double-buffered LDS.  Returns 1 when the
// shape does not qualify (caller falls back to the single-role kernel).
int try_launch_ws(const WParams& p0, int* rows_out, void* stream) {
  static int nows = -1;
  if (nows < 0) nows = ODIN_DIAG_ENV("ODIN_NOWS") ? 1 : 0;
  if (nows) return 1;
  WParams p = p0;
  int gx, gy, gz;
  size_t lds1;
  // two buffers must fit: plan against half of the LDS (this may pick a smaller channel block)
  if (!plan_wgrad(p, &gx, &gy, &gz, &lds1, 64, (158 * 1024) / 2)) return 1;
  const bool dcont = p.dvec && p.dlog >= 0 && p.DP == p.COB;
  const int nacc = (p.nrt * p.ncot + NW_W - 1) / NW_W;
  const int sp = p.S * p.P;
  if (!(p.pvec && dcont && !p.flat && p.ncot == 1 && p.bias_mode != 1 && nacc <= 4 &&
        (p.OW % 8) == 0 && p.TR * p.OW == p.slots && p.COB == 32 && p.DP == 32 && p.KI <= 9 &&
        p.slots == 64 && (p.OW == 8 || p.OW == 16 || p.OW == 32) &&
        (sp == 32 || sp == 64 || sp == 128)))
    return 1;
  const size_t lds = (size_t)2 * (p.patch_floats + p.dy_floats) * 4;
  if (lds > 158 * 1024 || lds < 4352) return 1;
  {
    const int rpw = (p.NIMG * p.NRI + NW_W - 1) / NW_W;
    const int rmax = p.KI <= 5 ? 5 : 3;
    if (rpw > rmax) return 1;  // more than one staging batch per tile
  }
  // every disqualifier of launch_ws_inst is decided HERE, before the dry-run return: the dry run and
  // the launch must pick the same plan (the caller sizes the slab from the dry run's row count):
  // one image per tile, whole tiles only (no ragged batch end), zero row long enough
  if (p.NIMG != 1 || (p.OH % p.TR) != 0 || (size_t)p.W * p.P * 4 > 16384 || p.pl < 0 ||
      (p.CO % 32) != 0)
    return 1;
  if (rows_out) *rows_out = gx;
  if (p.slab == nullptr) return 0;  // dry run
  p.stamps = g_wstamps;
  dim3 grid(gx, gy, gz);
  if (p.KI <= 5) {
    if (sp == 32) return launch_ws_inst<5, 5, 32>(p, grid, lds, stream);
    if (sp == 64) return launch_ws_inst<5, 5, 64>(p, grid, lds, stream);
    return launch_ws_inst<5, 5, 128>(p, grid, lds, stream);
  }
  if (sp == 32) return launch_ws_inst<9, 3, 32>(p, grid, lds, stream);
  if (sp == 64) return launch_ws_inst<9, 3, 64>(p, grid, lds, stream);
  return launch_ws_inst<9, 3, 128>(p, grid, lds, stream);
}

int launch_wgrad(WParams& p, int* rows_out, void* stream) {
  int gx, gy, gz;
  size_t lds;
  p.slab_stride = p.KH * p.KW * p.CI * p.CO + (p.want_bias ? p.CO : 0);
  if (odin_wgrad_planes_applicable(p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO, p.KH, p.KW, p.S, p.pt, p.pl,
                                   p.center))
    return odin_wgrad_planes_launch(p.in, p.dy, p.slab, rows_out, p.B, p.OH, p.OW, p.CI, p.CO,
                                    p.want_bias, p.want_bias ? 0 : 1, p.g_amax, p.a_amax, stream);
  if (p.H == p.OH && p.W == p.OW && p.want_bias &&
      odin_wgrad5_blk_applicable(p.B, p.H, p.W, p.CI, p.CO, p.KH, p.KW, p.S, p.pt, p.pl, p.center))
    return odin_wgrad5_blk_launch(p.in, p.dy, p.slab, rows_out, p.B, p.H, p.W, p.CI, p.CO, p.KH, p.want_bias, p.g_amax,
                                  p.a_amax, stream);
  if (odin_wgrad_blk_applicable(p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO, p.KH, p.KW, p.S, p.pt, p.pl, p.center))
    return odin_wgrad_blk_launch(p.in, p.dy, p.slab, rows_out, p.B, p.OH, p.OW, p.CI, p.CO, p.want_bias,
                                 p.want_bias ? 0 : 1, p.g_amax, p.a_amax, stream);
  if (odin_igemm_h_wgrad_applicable(p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO, p.KH, p.KW, p.S, p.center)) {
    if (rows_out) *rows_out = odin_igemm_h_wgrad_rows(p.B, p.OH, p.OW, p.KH, p.KW, p.CI, p.CO);
    if (p.slab == nullptr) return 0;  // dry run
    return odin_igemm_h_wgrad_launch(p.in, p.dy, p.slab, p.slab_stride, p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO,
                                     p.KH, p.KW, p.S, p.pt, p.pl, p.want_bias, p.want_bias ? 0 : 1, p.g_amax, p.a_amax, stream);
  }
  if (odin_igemm_wgrad_applicable(p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO, p.KH, p.KW, p.S, p.center)) {
    if (rows_out) *rows_out = odin_igemm_wgrad_rows(p.B, p.OH, p.OW, p.KH, p.KW, p.CI, p.CO);
    if (p.slab == nullptr) return 0;  // dry run
    return odin_igemm_wgrad_launch(p.in, p.dy, p.slab, p.slab_stride, p.B, p.H, p.W, p.CI, p.OH, p.OW, p.CO,
                                   p.KH, p.KW, p.S, p.pt, p.pl, p.want_bias, stream);
  }
  {
    const int rc = try_launch_ws(p, rows_out, stream);
    if (rc != 1) return rc;
  }
  if (!plan_wgrad(p, &gx, &gy, &gz, &lds)) return odin_fail(-2, "wgrad: no tiling plan");
  // bottleneck layers: 128-pixel tiles give only a few dozen workgroups, each with a long serial
  // reduction.  Smaller pixel tiles split the reduction over more slab rows.
  if (gx * gy * gz < 128 && !ODIN_DIAG_ENV("ODIN_NOWSPLIT")) {
    for (int tgt = 64; tgt >= 32; tgt >>= 1) {
      WParams q = p;
      int gx2, gy2, gz2;
      size_t lds2;
      if (!plan_wgrad(q, &gx2, &gy2, &gz2, &lds2, tgt)) break;
      if (gx2 * gy2 * gz2 <= gx * gy * gz) break;
      p = q; gx = gx2; gy = gy2; gz = gz2; lds = lds2;
      if (gx * gy * gz >= 128) break;
    }
  }
  // tiny images (patch rows of a few pixels: KI <= 2 items per lane): a 128-pixel tile spans many
  // short rows, and staging them RPWMAX rows per wave at a time costs one HBM round trip per
  // batch (enc3 / dec1 weight gradients: 10 batches, ~40 us for 0.07-0.5 GFLOP).  Pick the tile
  // whose rows fit ONE batch of the 12-rows-per-wave instance, so that the staging pipelines.
  if (!p.flat && p.KI <= 2 && (p.NIMG * p.NRI + NW_W - 1) / NW_W > 12 && !ODIN_DIAG_ENV("ODIN_NOWSPLIT")) {
    for (int tgt = 64; tgt >= 16; tgt >>= 1) {
      WParams q = p;
      int gx2, gy2, gz2;
      size_t lds2;
      // (same channel block: a smaller pixel tile must not be traded for a wider channel block)
      if (!plan_wgrad(q, &gx2, &gy2, &gz2, &lds2, tgt, 160 * 1024 - 2048, p.CIB)) break;
      if (q.KI > 2) break;
      p = q; gx = gx2; gy = gy2; gz = gz2; lds = lds2;
      if ((p.NIMG * p.NRI + NW_W - 1) / NW_W <= 12) break;
    }
  }
  if (rows_out) *rows_out = gx;
  if (p.slab == nullptr) return 0;  // dry run: planning only
  p.stamps = g_wstamps;
  dim3 grid(gx, gy, gz);
  const int tiles_per_block = p.nrt * p.ncot + (p.bias_mode == 1 ? p.ncot : 0);
  const int nacc = (tiles_per_block + NW_W - 1) / NW_W;
  const int rpw = (p.NIMG * p.NRI + NW_W - 1) / NW_W;
  const int ditems = p.slots * (p.dvec ? p.COB / 4 : p.COB);
  if (p.flat) {
    if (nacc <= 4) return launch_winst<4, 2, 8, 8, true>(p, grid, lds, stream);
    return launch_winst<NACC, 2, 8, 8, true>(p, grid, lds, stream);
  }
  if (nacc <= 4) {
    if (p.KI <= 2 && rpw > 5 && rpw <= 12 && ditems <= 8 * 256)
      return launch_winst<4, 2, 12, 8>(p, grid, lds, stream);
    if (p.KI <= 5 && rpw <= 5 && ditems <= 4 * 256) return launch_winst<4, 5, 5, 4>(p, grid, lds, stream);
    if (rpw <= 3 && ditems <= 4 * 256) return launch_winst<4, 9, 3, 4>(p, grid, lds, stream);
    return launch_winst<4, 9, 2, 8>(p, grid, lds, stream);
  }
  if (p.KI <= 5 && rpw <= 5) return launch_winst<NACC, 5, 5, 8>(p, grid, lds, stream);
  return launch_winst<NACC, 9, 2, 8>(p, grid, lds, stream);
}

constexpr int MAX_JOBS = 64;
struct ReduceJobs {
  odin_reduce_job j[MAX_JOBS];
  // odin_slab_reduce_sumsq: the squared norm of what the launch writes, as one partial per ACTIVE workgroup --
  // part[poff[j] + blockIdx.x] for blockIdx.x < pcnt[j]; poff[j] < 0: job j's result is not part of the gradient
  // (the range-word reset).  stage_src / stage_dst: `stage_n` floats copied by workgroup (0, 0) (the step's
  // hyper-parameter row, read by the Adam launch after `hyper` itself was advanced).
  float* part;
  int poff[MAX_JOBS];
  int pcnt[MAX_JOBS];
  const float* stage_src;
  float* stage_dst;
  int stage_n;
};

// Vector jobs (n, stride multiples of 4, 16-byte aligned): a block owns 64 consecutive floats of
// the result; thread (tx = tid & 15, ty = tid >> 4) sums rows ty, ty + 16, ... of float4 column tx
// with four independent partial sums, the 16 row phases are combined through LDS in a fixed
// order.  Parallelism is rows x n, not n: a 16 K-float gradient over 256 slab rows is 256 blocks
// of 16-byte loads instead of 64 blocks of serial 4-byte loads.  Other jobs (bias rows, the
// tail's 65-float row): one thread per result element.
__global__ __launch_bounds__(256) void slab_reduce_kernel(ReduceJobs jobs) {
  const odin_reduce_job jb = jobs.j[blockIdx.y];
  const size_t st = jb.stride > 0 ? (size_t)jb.stride : (size_t)jb.n;
  if (jobs.stage_dst != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < jobs.stage_n)
    jobs.stage_dst[threadIdx.x] = jobs.stage_src[threadIdx.x];
  float ss = 0.f;   // squares of what this thread writes (odin_slab_reduce_sumsq)
  const bool vec = ((jb.n & 3) == 0) && ((st & 3) == 0) &&
                   ((((size_t)jb.src | (size_t)jb.dst) & 15) == 0);
  if (vec && (jb.n >> 2) >= 1024) {
    // wide jobs (weight tensors): a block owns 256 consecutive floats; every wave-instruction
    // reads one full KB of a slab row, 8 rows in flight per lane, the 4 waves split the rows
    __shared__ float4 partw[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n4 = jb.n >> 2;
    const size_t st4 = st >> 2;
    for (int c0 = blockIdx.x * 64; c0 < n4; c0 += gridDim.x * 64) {  // block-uniform
      const int c = c0 + tx;
      float4 a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < n4) {
        const float4* s = reinterpret_cast<const float4*>(jb.src) + c;
        int g = ty;
        for (; g + 28 < jb.rows; g += 32) {
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(g + 4 * u) * st4];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            a[u].x += v[u].x; a[u].y += v[u].y; a[u].z += v[u].z; a[u].w += v[u].w;
          }
        }
        for (; g < jb.rows; g += 4) {
          const float4 v0 = s[(size_t)g * st4];
          a[0].x += v0.x; a[0].y += v0.y; a[0].z += v0.z; a[0].w += v0.w;
        }
      }
      float4 t;
      t.x = ((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x));
      t.y = ((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y));
      t.z = ((a[0].z + a[1].z) + (a[2].z + a[3].z)) + ((a[4].z + a[5].z) + (a[6].z + a[7].z));
      t.w = ((a[0].w + a[1].w) + (a[2].w + a[3].w)) + ((a[4].w + a[5].w) + (a[6].w + a[7].w));
      partw[threadIdx.x] = t;
      __syncthreads();
      if (ty == 0 && c < n4) {
        const float4 q1 = partw[64 + tx], q2 = partw[128 + tx], q3 = partw[192 + tx];
        t.x = (t.x + q1.x) + (q2.x + q3.x);
        t.y = (t.y + q1.y) + (q2.y + q3.y);
        t.z = (t.z + q1.z) + (q2.z + q3.z);
        t.w = (t.w + q1.w) + (q2.w + q3.w);
        reinterpret_cast<float4*>(jb.dst)[c] = t;
        ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
      }
      __syncthreads();
    }
  } else if (vec) {
    __shared__ float4 part[256];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n4 = jb.n >> 2;
    for (int c0 = blockIdx.x * 16; c0 < n4; c0 += gridDim.x * 16) {  // block-uniform
      const int c = c0 + tx;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      if (c < n4) {
        const float4* s = reinterpret_cast<const float4*>(jb.src) + c;
        const size_t st4 = st >> 2;
        int g = ty;
        for (; g + 112 < jb.rows; g += 128) {  // 8 rows in flight
          const float4 v0 = s[(size_t)g * st4], v1 = s[(size_t)(g + 16) * st4];
          const float4 v2 = s[(size_t)(g + 32) * st4], v3 = s[(size_t)(g + 48) * st4];
          const float4 v4 = s[(size_t)(g + 64) * st4], v5 = s[(size_t)(g + 80) * st4];
          const float4 v6 = s[(size_t)(g + 96) * st4], v7 = s[(size_t)(g + 112) * st4];
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
          a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
          a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
          a0.x += v4.x; a0.y += v4.y; a0.z += v4.z; a0.w += v4.w;
          a1.x += v5.x; a1.y += v5.y; a1.z += v5.z; a1.w += v5.w;
          a2.x += v6.x; a2.y += v6.y; a2.z += v6.z; a2.w += v6.w;
          a3.x += v7.x; a3.y += v7.y; a3.z += v7.z; a3.w += v7.w;
        }
        for (; g + 48 < jb.rows; g += 64) {
          const float4 v0 = s[(size_t)g * st4], v1 = s[(size_t)(g + 16) * st4];
          const float4 v2 = s[(size_t)(g + 32) * st4], v3 = s[(size_t)(g + 48) * st4];
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
          a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
          a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
          a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
        }
        for (; g < jb.rows; g += 16) {
          const float4 v0 = s[(size_t)g * st4];
          a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
        }
      }
      part[threadIdx.x] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                      (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
      __syncthreads();
      if (ty == 0 && c < n4) {
        float4 t = part[tx];
#pragma unroll
        for (int u = 1; u < 16; ++u) {
          const float4 q = part[u * 16 + tx];
          t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
        }
        reinterpret_cast<float4*>(jb.dst)[c] = t;
        ss += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
      }
      __syncthreads();
    }
  } else
  // odd-sized jobs (the fused tail's 65-float rows, 33-float heads): 32 result elements per block,
  // 8 row phases per element, 8 loads in flight per thread -- a one-thread-per-element loop walks
  // 256+ rows serially (dozens of dependent L2 round trips: it set the duration of the whole launch)
  {
    __shared__ float parts[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i0 = blockIdx.x * 32; i0 < jb.n; i0 += gridDim.x * 32) {  // block-uniform
      const int i = i0 + tx;
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = 0.f;
      if (i < jb.n) {
        const float* s = jb.src + i;
        int g = ty;
        for (; g + 56 < jb.rows; g += 64) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(g + 8 * u) * st];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] += v[u];
        }
        for (; g < jb.rows; g += 8) a[0] += s[(size_t)g * st];
      }
      parts[threadIdx.x] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
      __syncthreads();
      if (ty == 0 && i < jb.n) {
        float t = parts[tx];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += parts[u * 32 + tx];
        jb.dst[i] = t;
        ss += t * t;
      }
      __syncthreads();
    }
  }
  if (jobs.part != nullptr && jobs.poff[blockIdx.y] >= 0 && (int)blockIdx.x < jobs.pcnt[blockIdx.y]) {
    // one partial per active workgroup, fixed order inside the workgroup: wave sums by shuffles, then the four waves
    __shared__ float ssw[4];
    float v = ss;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    if ((threadIdx.x & 63) == 0) ssw[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) jobs.part[jobs.poff[blockIdx.y] + blockIdx.x] = (ssw[0] + ssw[1]) + (ssw[2] + ssw[3]);
  }
}

}  // namespace

extern "C" int odin_conv2d_wgrad(const float* x, const float* dy, float* slab,
                                 int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  if (odin_smallc_applicable(d)) return odin_smallc_wgrad(x, dy, slab, slab_rows_out, d, stream);
  if (odin_pw1x1_applicable(d)) return odin_pw1x1_wgrad(x, dy, slab, slab_rows_out, d, stream);
  WParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.dy = dy; p.slab = slab;
  p.B = d->B; p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.KH = d->KH; p.KW = d->KW; p.S = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.center = d->center; p.want_bias = 1;
  p.g_amax = d->dy_amax;
  p.a_amax = d->x_amax;
  return launch_wgrad(p, slab_rows_out, stream);
}

// x = deconv input [B,H,W,Cin], dy = grad wrt deconv pre-activation output [B,OH,OW,Cout]
extern "C" int odin_deconv2d_wgrad(const float* x, const float* dy, float* slab,
                                   int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  if (odin_smalldeconv_applicable(d))   // (dry run: slab == NULL only reports the rows)
    return odin_smalldeconv_bwd(x, dy, nullptr, nullptr, 0, nullptr, slab, slab_rows_out, d, stream);
  WParams p;
  memset(&p, 0, sizeof(p));
  p.in = dy; p.dy = x; p.slab = slab;
  p.B = d->B; p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.KH = d->KH; p.KW = d->KW; p.S = d->stride; p.pt = d->pad_t; p.pl = d->pad_l;
  p.center = 0; p.want_bias = 0;
  p.g_amax = d->dy_amax;
  p.a_amax = d->x_amax;
  return launch_wgrad(p, slab_rows_out, stream);
}

// 1: some launch of this layer (forward or weight gradient, as dispatched now) is a two-plane kernel that READS the
// range word of the layer input (odin_conv_desc.x_amax) -- a caller uses it to decide whether the layer below is asked
// to keep that word at all (a wrong answer is harmless: a plane kernel without a word carries x unscaled, as in round 4)
extern "C" int odin_conv2d_reads_x_range(const odin_conv_desc* d) {
  if (odin_smallc_applicable(d) || odin_pw1x1_applicable(d)) return 0;
  return (odin_fconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                                       d->pad_t, d->pad_l, d->center) ||
          odin_igemm_h_applicable(0, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->center) ||
          odin_fconv_blk_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                    d->pad_l, d->center) ||
          (d->H == d->OH && d->W == d->OW &&
           (odin_conv5_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l,
                                      d->center) ||
            odin_wgrad5_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l,
                                       d->center))) ||
          odin_wgrad_blk_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->pad_t,
                                    d->pad_l, d->center) ||
          odin_wgrad_planes_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                                       d->pad_t, d->pad_l, d->center) ||
          odin_igemm_h_wgrad_applicable(d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride,
                                        d->center)) ? 1 : 0;
}
extern "C" int odin_deconv2d_reads_x_range(const odin_conv_desc* d) {
  return ((d->OH == 2 * d->H && d->OW == 2 * d->W &&
           odin_tconv_planes_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l,
                                        d->center, 1, 1)) ||
          (d->OH == 2 * d->H && d->OW == 2 * d->W &&
           odin_tconv_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad_t, d->pad_l,
                                     d->center)) ||
          odin_igemm_h_applicable(1, d->B, d->H, d->W, d->Cin, d->OH, d->OW, d->Cout, d->KH, d->KW, d->stride, d->center) ||
          odin_wgrad_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride,
                                       d->pad_t, d->pad_l, 0) ||
          odin_wgrad_blk_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                    d->pad_l, 0) ||
          odin_igemm_h_wgrad_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, 0))
             ? 1 : 0;
}
extern "C" int odin_dense_reads_x_range(int B, int K, int N) { return odin_dense_h_ok(B, K, N) ? 1 : 0; }

extern "C" int odin_dense_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out,
                                int B, int K, int N, void* stream) {
  if (odin_dense_h_ok(B, K, N)) {  // both widths >= 256: the two-plane GEMM, ONE complete slab row
    if (slab_rows_out) *slab_rows_out = 1;
    if (slab == nullptr) return 0;  // dry run
    return odin_dense_h_wgrad(x, dy, slab, B, K, N, nullptr, nullptr, stream);
  }
  if (odin_thin_dense_wgrad_rows(B, K, N) > 0 && (slab == nullptr || ((((size_t)x | (size_t)dy | (size_t)slab)) & 15) == 0)) {
    // one thin side: streaming kernel, slab rows = row chunks of the batch (a dry run cannot see the pointers: callers
    // allocate at least 16-byte aligned tensors)
    if (slab_rows_out) *slab_rows_out = odin_thin_dense_wgrad_rows(B, K, N);
    if (slab == nullptr) return 0;  // dry run
    return odin_thin_dense_wgrad(x, dy, slab, B, K, N, stream);
  }
  // (also the tiny layers: their forward / data gradient run on the vector ALUs, but the weight gradient
  // through the generic kernel was a 14.5 us launch for 0.001 GFLOP)
  if (!ODIN_DIAG_ENV("ODIN_NODENSEIGEMM") && !odin_tiny_dense_ok(B, K, N) &&
      odin_igemm_wgrad_applicable(B, 1, 1, K, 1, 1, N, 1, 1, 1, 0)) {
    if (slab_rows_out) *slab_rows_out = odin_igemm_wgrad_rows(B, 1, 1, 1, 1, K, N);
    if (slab == nullptr) return 0;  // dry run
    return odin_igemm_wgrad_launch(x, dy, slab, K * N + N, B, 1, 1, K, 1, 1, N, 1, 1, 1, 0, 0, 1, stream);
  }
  if (odin_dense_gemm_ok(B, K, N) && !ODIN_DIAG_ENV("ODIN_NOTINYWGRADGEMM")) {
    // small GEMM: the waves of a workgroup split the batch, the result is complete: ONE slab row
    if (slab_rows_out) *slab_rows_out = 1;
    if (slab == nullptr) return 0;  // dry run
    return odin_dense_gemm_wgrad(x, dy, slab, B, K, N, stream);
  }
  WParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.dy = dy; p.slab = slab;
  p.B = B; p.H = 1; p.W = 1; p.CI = K; p.OH = 1; p.OW = 1; p.CO = N;
  p.KH = p.KW = 1; p.S = 1; p.want_bias = 1;
  return launch_wgrad(p, slab_rows_out, stream);
}

// ---- a layer's whole backward pass in one call: weight gradient + data gradient.  Where both run on the
// implicit-GEMM kernels (igemm.hip) they share ONE launch; otherwise exactly the two calls above. ----
extern "C" int odin_conv2d_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act,
                               float* dx, float* colsum_slab, int* colsum_rows_out, float* wslab,
                               int* wslab_rows_out, const odin_conv_desc* d, void* stream) {
  odin_igemm_pair_begin();
  int rc = odin_conv2d_wgrad(x, dy, wslab, wslab_rows_out, d, stream);
  if (rc == 0) rc = odin_conv2d_dgrad(dy, w, aux, aux_act, dx, colsum_slab, colsum_rows_out, d, stream);
  const int rc2 = odin_igemm_pair_end();
  return rc != 0 ? rc : rc2;
}

extern "C" int odin_deconv2d_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act,
                                 float* dx, float* colsum_slab, int* colsum_rows_out, float* wslab,
                                 int* wslab_rows_out, const odin_conv_desc* d, void* stream) {
  if (colsum_slab == nullptr && dx != nullptr && wslab != nullptr && odin_smalldeconv_applicable(d)) {
    // the decoders' first Conv2DTranspose: weight and data gradient in ONE launch that stages dy once (smalldeconv.hip)
    if (colsum_rows_out) *colsum_rows_out = 0;
    return odin_smalldeconv_bwd(x, dy, w, aux, aux_act, dx, wslab, wslab_rows_out, d, stream);
  }
  // (a dry run -- dx == NULL and wslab == NULL -- reports the rows of the ONE-call form: with 64 output channels the
  // fused launch writes more slab rows than odin_deconv2d_wgrad alone; callers size their slab for both)
  const bool dry = dx == nullptr && wslab == nullptr;
  if (((dx != nullptr && wslab != nullptr && aux != nullptr) || dry) && aux_act == ODIN_ACT_ELU && d->KH == 4 &&
      d->KW == 4 && d->stride == 2 && d->pad_t == 1 && d->pad_l == 1 && d->OH == 2 * d->H && d->OW == 2 * d->W &&
      odin_bwd_planes_applicable(d->B, d->H, d->W, d->Cin, d->Cout) &&
      odin_wgrad_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, 0) &&
      odin_fconv_planes_applicable(d->B, d->OH, d->OW, d->Cout, d->H, d->W, d->Cin, d->KH, d->KW, d->stride, d->pad_t,
                                   d->pad_l, 0)) {
    // dy is fetched, scaled and split ONCE for both gradients (bwd_planes.hip); with 32 output channels the same
    // partial sums as the two launches
    const int rows = odin_bwd_planes_rows(d->B, d->H, d->W, d->Cin);
    if (colsum_rows_out) *colsum_rows_out = rows;
    if (wslab_rows_out) *wslab_rows_out = rows;
    if (dry) return 0;
    return odin_bwd_planes_launch(x, dy, w, aux, dx, colsum_slab, wslab, d->B, d->H, d->W, d->Cin, d->Cout, d->dy_amax,
                                  d->x_amax, d->dx_amax, stream);
  }
  // any other image size: the block-window form of the same launch (blk_planes.hip), 32 output channels
  if (((dx != nullptr && wslab != nullptr) || dry) && d->KH == 4 && d->KW == 4 && d->stride == 2 && d->pad_t == 1 &&
      d->pad_l == 1 && d->OH == 2 * d->H && d->OW == 2 * d->W && !d->center &&
      odin_bwd_blk_applicable(d->B, d->H, d->W, d->Cin, d->Cout)) {
    const int rows = odin_bwd_blk_rows(d->B, d->H, d->W, d->Cin);
    if (colsum_rows_out) *colsum_rows_out = rows;
    if (wslab_rows_out) *wslab_rows_out = rows;
    if (dry) return 0;
    return odin_bwd_blk_launch(x, dy, w, aux, aux_act, dx, colsum_slab, wslab, d->B, d->H, d->W, d->Cin, d->Cout,
                               d->dy_amax, d->x_amax, d->dx_amax, stream);
  }
  odin_igemm_pair_begin();
  int rc = odin_deconv2d_wgrad(x, dy, wslab, wslab_rows_out, d, stream);
  if (rc == 0) rc = odin_deconv2d_dgrad(dy, w, aux, aux_act, dx, colsum_slab, colsum_rows_out, d, stream);
  const int rc2 = odin_igemm_pair_end();
  return rc != 0 ? rc : rc2;
}

// want_wgrad / want_dgrad: either half may be left out (FactorVAE's TC term back-propagates through the discriminator
// without touching its weights).  dy_amax / dx_amax: the range words of dy (read) and dx (written when
// odin_dense_dgrad_keeps_range says so), both optional.
extern "C" int odin_dense_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act,
                              float* dx, float* colsum_slab, int* colsum_rows_out, float* wslab,
                              int* wslab_rows_out, int B, int K, int N, int want_wgrad, int want_dgrad,
                              const uint32_t* dy_amax, uint32_t* dx_amax, void* stream) {
  return odin_dense_bwd_ranged(x, dy, w, aux, aux_act, dx, colsum_slab, colsum_rows_out, wslab, wslab_rows_out, B, K, N,
                               want_wgrad, want_dgrad, dy_amax, dx_amax, nullptr, stream);
}

// + x_amax: the range word of the activation x (the weight gradient's other operand on the two-plane GEMM)
extern "C" int odin_dense_bwd_ranged(const float* x, const float* dy, const float* w, const float* aux, int aux_act,
                                     float* dx, float* colsum_slab, int* colsum_rows_out, float* wslab,
                                     int* wslab_rows_out, int B, int K, int N, int want_wgrad, int want_dgrad,
                                     const uint32_t* dy_amax, uint32_t* dx_amax, const uint32_t* x_amax,
                                     void* stream) {
  if (odin_dense_h_ok(B, K, N) && colsum_slab == nullptr) {
    int rc = 0;
    // (a dy without a word is bounded ONCE for both halves)
    if (dy_amax == nullptr && ((want_wgrad && wslab != nullptr) || (want_dgrad && dx != nullptr))) {
      dy_amax = odin_range_word_of(dy, (size_t)B * N, nullptr, stream);
      if (dy_amax == nullptr) return odin_fail(-3, "dense_bwd: no range word for dy");
    }
    if (want_wgrad && want_dgrad && wslab != nullptr && dx != nullptr) {
      // both halves: ONE launch (dense_h.hip: dense_h_pair_kernel), bit-identical to the two
      if (wslab_rows_out) *wslab_rows_out = 1;
      if (colsum_rows_out) *colsum_rows_out = 0;
      return odin_dense_h_bwd_pair(x, dy, w, aux, aux_act, dx, wslab, B, K, N, dy_amax, dx_amax, x_amax, stream);
    }
    if (want_wgrad) {
      if (wslab_rows_out) *wslab_rows_out = 1;
      if (wslab != nullptr) rc = odin_dense_h_wgrad(x, dy, wslab, B, K, N, dy_amax, x_amax, stream);
    }
    if (rc == 0 && want_dgrad) {
      if (colsum_rows_out) *colsum_rows_out = 0;
      if (dx != nullptr) rc = odin_dense_h_dgrad(dy, w, aux, aux_act, dx, B, K, N, dy_amax, dx_amax, stream);
    }
    return rc;
  }
  odin_igemm_pair_begin();
  int rc = 0;
  if (want_wgrad) rc = odin_dense_wgrad(x, dy, wslab, wslab_rows_out, B, K, N, stream);
  if (rc == 0 && want_dgrad)
    rc = odin_dense_dgrad_ranged(dy, w, aux, aux_act, dx, colsum_slab, colsum_rows_out, B, K, N, dy_amax, dx_amax,
                                 stream);
  const int rc2 = odin_igemm_pair_end();
  return rc != 0 ? rc : rc2;
}

// 1: the data gradient of this Dense layer (without a column-sum slab) folds max|dx| into dx_amax itself
extern "C" int odin_dense_dgrad_keeps_range(int B, int K, int N) { return odin_dense_dgrad_tracks(B, K, N) ? 1 : 0; }

// workgroups of job jb that write results (the kernel's three paths), for a launch gx wide
static int reduce_active_blocks(const odin_reduce_job& jb, int gx) {
  const size_t st = jb.stride > 0 ? (size_t)jb.stride : (size_t)jb.n;
  const bool vec = ((jb.n & 3) == 0) && ((st & 3) == 0) && ((((size_t)jb.src | (size_t)jb.dst) & 15) == 0);
  int nb;
  if (vec && (jb.n >> 2) >= 1024) nb = ((jb.n >> 2) + 63) / 64;
  else if (vec) nb = ((jb.n >> 2) + 15) / 16;
  else nb = (jb.n + 31) / 32;
  return nb < gx ? nb : gx;
}

static int slab_reduce_impl(const odin_reduce_job* jobs, int n_jobs, const float* g_lo, size_t g_n, float* part,
                            int* n_parts_out, const float* stage_src, float* stage_dst, int stage_n, void* stream) {
  if (int rc = odin_wgrad_planes_flush(stream)) return rc;  // (deferred weight gradients write the slabs read here)
  if (n_parts_out) *n_parts_out = 0;
  if (n_jobs <= 0) return 0;
  int pnext = 0;
  for (int j0 = 0; j0 < n_jobs; j0 += MAX_JOBS) {
    ReduceJobs rj;
    memset(&rj, 0, sizeof(rj));
    int nj = n_jobs - j0 < MAX_JOBS ? n_jobs - j0 : MAX_JOBS;
    int max_n = 0;
    for (int j = 0; j < nj; ++j) {
      rj.j[j] = jobs[j0 + j];
      if (jobs[j0 + j].n > max_n) max_n = jobs[j0 + j].n;
    }
    int gx = (max_n + 63) / 64;  // vector jobs: 64 floats per block
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    rj.part = part;
    for (int j = 0; j < nj; ++j) {
      const odin_reduce_job& jb = rj.j[j];
      const bool in_g = g_lo != nullptr && jb.dst >= g_lo && jb.dst + jb.n <= g_lo + g_n;
      rj.poff[j] = in_g ? pnext : -1;
      rj.pcnt[j] = in_g ? reduce_active_blocks(jb, gx) : 0;
      pnext += rj.pcnt[j];
    }
    if (j0 == 0) { rj.stage_src = stage_src; rj.stage_dst = stage_dst; rj.stage_n = stage_n; }
    if (part == nullptr && g_lo != nullptr) continue;  // dry run: only the number of partials
    dim3 grid(gx, nj, 1), block(256);
    ODIN_LAUNCH(slab_reduce_kernel, grid, block, 0, stream, rj);
    int rc = odin_check_launch("slab_reduce");
    if (rc) return rc;
  }
  if (n_parts_out) *n_parts_out = pnext;
  return 0;
}

extern "C" int odin_slab_reduce(const odin_reduce_job* jobs, int n_jobs, void* stream) {
  return slab_reduce_impl(jobs, n_jobs, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

// odin_slab_reduce that also leaves the squared norm of the gradient it writes: one partial per active workgroup in
// part[0 .. *n_parts_out) -- the stage-1 launch of the gradient norm (odin_sumsq_adam_*) rides in the reduction.  Jobs
// whose dst lies inside [g, g + g_n) count (the caller guarantees that they tile the gradient buffer exactly once);
// others (the range-word reset) do not.  stage_src / stage_dst (optional): stage_n <= 256 floats copied by the launch
// (the hyper-parameter row the Adam launch reads after it advanced `hyper`).  part == NULL: dry run (*n_parts_out only).
extern "C" int odin_slab_reduce_sumsq(const odin_reduce_job* jobs, int n_jobs, const float* g, size_t g_n, float* part,
                                      int* n_parts_out, const float* stage_src, float* stage_dst, int stage_n,
                                      void* stream) {
  if (g == nullptr || stage_n < 0 || stage_n > 256) return odin_fail(-2, "odin_slab_reduce_sumsq: bad arguments");
  return slab_reduce_impl(jobs, n_jobs, g, g_n, part, n_parts_out, stage_src, stage_dst, stage_n, stream);
}

extern "C" int odin_debug_set_wgrad_stamps(void* buf) {
  g_wstamps = (long long*)buf;
  return 0;
}
