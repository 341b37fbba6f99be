// dense_h.hip -- Dense layers with BOTH dimensions >= 256 (keras.layers.Dense, odin/networks/base_networks.py:1002-1014:
// FactorDiscriminator's 1000-unit stack, factor_vae.py:150-153; CelebA's Dense 4096 -> 512, image_networks.py:688; the
// 512-unit default nets, variational_autoencoder.py:181-185) as GEMMs on the f16 matrix pipe with fp32 operands carried
// as two planes (odin_device.h: x = h + 2^-11 l, three v_mfma_f32_32x32x16_f16 per 16 k-values into a main and a cross
// accumulator, <= 3 * 2^-22 per product), operands straight from L2.
//
// Why: at batch 128-512 these are 0.25-2 GFLOP products.  On the fp32 implicit-GEMM kernel (igemm.hip) a
// [128, 1000] x [1000, 1000] launch is 128 workgroups whose four waves each walk a dependent chain of 128
// v_mfma_f32_32x32x2_f32 (64 cycles each: 4.3 us) behind a 6 us launch floor: 12-20 us per launch, 16 launches per
// FactorVAE iteration.  Here a wave's chain is 8-16 k-steps of 3 MFMAs (32 cycles each) + the two-plane split of its own
// operand fragments (48 VALU per step), and 8 waves split the reduction of a tile (1024 waves for 128 tiles).
//
//   C[i][j] = sum_k A(i, k) * B(k, j)                                   (the arrangements of dense_gemm.hip)
//   forward :  A = x  [B, K]  (k contiguous)   B = w  [K, N] (j contiguous)        C = y  [B, N]
//   dgrad   :  A = dy [B, N]  (k contiguous)   B = w  [K, N] (k contiguous: row j)  C = dx [B, K]   A is a gradient
//   wgrad   :  A = x  [B, K]  (i contiguous)   B = dy [B, N] (j contiguous)        C = dW [K, N]   B is a gradient
// A gradient operand is scaled by the power of two of its range word on its way into the planes (odin_range_shift); the
// data gradient keeps the range word of its output.  Reduction lengths are multiples of 8.
#include "blk_common.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct DHParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;        // forward: [N]
  const float* aux;         // dgrad: act'(aux) multiplier, same shape as C
  float* colsum;            // wgrad: db[j] = sum_k B(k, j), written by tile row 0 (may be null)
  const unsigned* g_amax;   // range word of the gradient operand (SCA: A, SCB: B)
  unsigned* out_amax;       // forward / dgrad: range word of C (may be null)
  const unsigned* a_amax;   // AS instances (weight gradient): range word of the activation operand A = x
  int g_cond;               // the g_amax operand is an ACTIVATION (forward): scaled only outside the safe window
  int M, N, K;              // C is [M, N]; reduction length K (multiple of 8)
  int lda, ldb, ldc;
  int act, aux_act;
};

// the 8 reduction values k0 + 8 h + 0..7 of this lane's row (KC: contiguous) or column (strided by ld)
template <bool KC>
__device__ __forceinline__ void dh_load8(const OdinRun& R, bool ok, int idx, int ld, int k, int K, float (&v)[8]) {
  if constexpr (KC) {
    const unsigned off = (ok && k < K) ? (unsigned)((idx * ld + k) * 4) : ODIN_OOB;  // K % 8 == 0: all or nothing
    const float4 t0 = odin_run_load4(R, off), t1 = odin_run_load4(R, off == ODIN_OOB ? ODIN_OOB : off + 16);
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w; v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
      v[e] = odin_run_load1(R, (ok && k < K) ? (unsigned)(((k + e) * ld + idx) * 4) : ODIN_OOB);
  }
}

template <bool SC>
__device__ __forceinline__ void dh_split8(const float (&v)[8], float s, float s2k, u32x4& hi, u32x4& lo) {
  u32x2 h0, l0, h1, l1;
  odin_split_h4<SC>(make_float4(v[0], v[1], v[2], v[3]), s, s2k, h0, l0);
  odin_split_h4<SC>(make_float4(v[4], v[5], v[6], v[7]), s, s2k, h1, l1);
  hi[0] = h0.x; hi[1] = h0.y; hi[2] = h1.x; hi[3] = h1.y;
  lo[0] = l0.x; lo[1] = l0.y; lo[2] = l1.x; lo[3] = l1.y;
}

// NW waves split the k-steps of one 32 x 32 tile (step s -> wave s % NW); SCA / SCB: that operand is a gradient
// AS (weight gradient only): the activation operand A comes with its own range word and power of two
template <int NW, bool A_KC, bool B_KC, bool SCA, bool SCB, bool AS = false>
__device__ __forceinline__ void dense_h_body(const DHParams& p, int bx, int by, int wg, float* red, float* cred) {
  // red: [NW * 16 * 64], cred: [NW * 32 + 16] floats of LDS (the kernel's)
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  const int i0 = by * 32, j0 = bx * 32;
  const int ia = i0 + l31, jb = j0 + l31;
  const bool a_ok = ia < p.M, b_ok = jb < p.N;
  const int nsteps = (p.K + 15) >> 4;
  const OdinRun RA = odin_run(p.A, (unsigned)((size_t)(A_KC ? p.M * p.lda : p.K * p.lda) * 4));
  const OdinRun RB = odin_run(p.B, (unsigned)((size_t)(B_KC ? p.N * p.ldb : p.K * p.ldb) * 4));
  // the gradient operand is carried times 2^gk (its maximum lands in [2^14, 2^15)), the sums are scaled back.  The
  // range words are read BEHIND the first operand loads (round 5): their scalar loads then answer beside the vector
  // loads instead of in front of them -- these launches last 8 us, a serial L2 round trip is 1 us of that
  int gk = 0, ak = 0;
  float g_s = 1.f, g_s2k = ODIN_LO_SCALE, a_s = 1.f, a_s2k = ODIN_LO_SCALE;
  // (activation operands are scaled only when their bound leaves [2^-8, 2^15): wave-uniform flags, one scalar branch
  // around the split -- the unscaled split is 4 VALU instructions per 4 values cheaper)
  bool g_on = SCA || SCB, a_on = AS;
  const OdinRangeReq g_rq = odin_range_issue((SCA || SCB) ? p.g_amax : nullptr, lane);
  const OdinRangeReq a_rq = odin_range_issue(AS ? p.a_amax : nullptr, lane);
  auto read_words = [&]() {
    if (SCA || SCB) {
      const unsigned mb = odin_range_finish(g_rq);
      g_on = !p.g_cond || odin_act_needs_scale(mb);
      gk = g_on ? odin_range_shift(mb) : 0;
      g_s = odin_pow2(gk); g_s2k = odin_pow2(gk + 11);
    }
    if (AS) {
      const unsigned mb = odin_range_finish(a_rq);
      a_on = odin_act_needs_scale(mb);
      ak = a_on ? odin_range_shift(mb) : 0;
      a_s = odin_pow2(ak); a_s2k = odin_pow2(ak + 11);
    }
  };
  f32x16 acc = f32x16_zero(), acx = f32x16_zero();
  float csum = 0.f;  // wgrad bias: column sum of B over this wave's k-steps (lane j = l31, half h)
  // the epilogue's own operands -- the bias of this lane's column, act'(aux) of the elements this wave finishes -- are
  // requested HERE, in front of the reduction: behind the barrier they were one more dependent L2 round trip of a launch
  // that lasts 8 us (round 6)
  constexpr int RPW = 16 / NW > 0 ? 16 / NW : 1;  // registers a wave finishes (NW <= 16)
  const float bj = (p.bias != nullptr && b_ok) ? p.bias[jb] : 0.f;
  const OdinRun RX = odin_run(p.aux != nullptr ? p.aux : p.C, p.aux != nullptr ? (unsigned)((size_t)p.M * p.ldc * 4) : 0u);
  float auxv[RPW];
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int rr = wave * RPW + q;
    const int row = i0 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
    auxv[q] = (p.aux != nullptr && rr < 16) ? odin_run_load1(RX, (row < p.M && b_ok) ? (unsigned)((row * p.ldc + jb) * 4) : ODIN_OOB) : 0.f;
  }
  float a0[8], b0[8], a1[8], b1[8];
  auto mul = [&](const float (&av)[8], const float (&bv)[8]) {
    u32x4 ah, al, bh, bl;
    if (SCA && g_on) dh_split8<true>(av, g_s, g_s2k, ah, al);
    else if (AS && a_on) dh_split8<true>(av, a_s, a_s2k, ah, al);
    else dh_split8<false>(av, 1.f, ODIN_LO_SCALE, ah, al);
    if (SCB && g_on) dh_split8<true>(bv, g_s, g_s2k, bh, bl);
    else dh_split8<false>(bv, 1.f, ODIN_LO_SCALE, bh, bl);
    acx = mfma32_f16(ah, bl, acx);
    acc = mfma32_f16(ah, bh, acc);
    acx = mfma32_f16(al, bh, acx);
    if (!A_KC && !B_KC) {  // (wgrad) bias gradient
#pragma unroll
      for (int e = 0; e < 8; ++e) csum += bv[e];
    }
  };
  int s = wave;
  if (s < nsteps) {
    dh_load8<A_KC>(RA, a_ok, ia, p.lda, 16 * s + 8 * h, p.K, a0);
    dh_load8<B_KC>(RB, b_ok, jb, p.ldb, 16 * s + 8 * h, p.K, b0);
    // (loads beyond the reduction read zeros through the range check: no branch around them)
    dh_load8<A_KC>(RA, a_ok, ia, p.lda, 16 * (s + NW) + 8 * h, p.K, a1);
    dh_load8<B_KC>(RB, b_ok, jb, p.ldb, 16 * (s + NW) + 8 * h, p.K, b1);
    ODIN_SCHED_FENCE();
    read_words();   // (two batches of operand loads are in flight)
    for (;;) {
      ODIN_SCHED_FENCE();
      mul(a0, b0);
      ODIN_SCHED_FENCE();
      s += 2 * NW;
      if (s - NW >= nsteps) break;
      dh_load8<A_KC>(RA, a_ok, ia, p.lda, 16 * s + 8 * h, p.K, a0);
      dh_load8<B_KC>(RB, b_ok, jb, p.ldb, 16 * s + 8 * h, p.K, b0);
      ODIN_SCHED_FENCE();
      mul(a1, b1);
      ODIN_SCHED_FENCE();
      if (s >= nsteps) break;
      dh_load8<A_KC>(RA, a_ok, ia, p.lda, 16 * (s + NW) + 8 * h, p.K, a1);
      dh_load8<B_KC>(RB, b_ok, jb, p.ldb, 16 * (s + NW) + 8 * h, p.K, b1);
    }
  }
  // ---- the NW partial tiles meet in LDS (main + 2^-11 cross, scaled back); wave w finishes registers
  // [w * 16 / NW, (w + 1) * 16 / NW) in wave order: fixed summation order, bit reproducible ----
  if (s == wave) read_words();   // (a wave without a k-step: its partial tile is zero, its scales still the tile's)
  const float o_s = (SCA || SCB) ? odin_pow2(-gk) : 1.f, o_sx = (SCA || SCB) ? odin_pow2(-gk - 11) : ODIN_LO_UNSCALE;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const float t = fmaf(acx[rr], o_sx, acc[rr] * o_s);
    red[(wave * 16 + rr) * 64 + lane] = AS ? t * odin_pow2(-ak) : t;   // (the two scales one after the other)
  }
  if (!A_KC && !B_KC && p.colsum != nullptr) {
    const float t = csum + __shfl_xor(csum, 32);
    if (h == 0) cred[wave * 32 + l31] = t;
  }
  __syncthreads();
  const OdinRun RC = odin_run(p.C, (unsigned)((size_t)p.M * p.ldc * 4));
  float amx = 0.f;
#pragma unroll
  for (int q = 0; q < RPW; ++q) {
    const int rr = wave * RPW + q;
    if (rr < 16) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 16 + rr) * 64 + lane];
      const int row = i0 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
      const unsigned off = (row < p.M && b_ok) ? (unsigned)((row * p.ldc + jb) * 4) : ODIN_OOB;
      v = odin_act(p.act, v + bj);
      if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, auxv[q]);
      odin_run_store1(RC, off, v);
      amx = fmaxf(amx, off != ODIN_OOB ? fabsf(v) : 0.f);
    }
  }
  if (p.out_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, NW * 64, cred, (unsigned)wg);
  } else if (!A_KC && !B_KC && p.colsum != nullptr && by == 0 && wave == 0 && h == 0 && b_ok) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += cred[w * 32 + l31];
    p.colsum[jb] = t;  // (sums of the raw fp32 values: no plane scale)
  }
}

template <int NW, bool A_KC, bool B_KC, bool SCA, bool SCB, bool AS = false>
__global__ __launch_bounds__(NW * 64) void dense_h_kernel(DHParams p) {
  __shared__ float red[NW * 16 * 64];
  __shared__ float cred[NW * 32 + 16];
  dense_h_body<NW, A_KC, B_KC, SCA, SCB, AS>(p, (int)blockIdx.x, (int)blockIdx.y, (int)(blockIdx.x + gridDim.x * blockIdx.y), red,
                                             cred);
}

// A layer's data gradient and weight gradient in ONE launch (round 6): both read dy, neither reads the other's result;
// as launches of their own they took 9.5 + 11.5 us per 1000 x 1000 layer of FactorVAE's discriminator step, mostly
// launch floor and the first round trip of a dependent chain.  Workgroups [0, nd) run the data gradient (NWD waves per
// tile), the rest the weight gradient (NWW waves; the surplus waves of the wider block leave at once: a finished wave
// does not take part in s_barrier).  The same bodies on the same tiles: results bit-identical to the two launches.
template <int NWD, int NWW, bool AS>
__global__ __launch_bounds__((NWD > NWW ? NWD : NWW) * 64) void dense_h_pair_kernel(DHParams pd, DHParams pw, int nd, int gdx,
                                                                                    int gwx) {
  constexpr int NWM = NWD > NWW ? NWD : NWW;
  __shared__ float red[NWM * 16 * 64];
  __shared__ float cred[NWM * 32 + 16];
  const int id = (int)blockIdx.x;
  if (id < nd) {
    if (NWD < NWW && (int)threadIdx.x >= NWD * 64) return;
    const int by = id / gdx;
    dense_h_body<NWD, true, true, true, false, false>(pd, id - by * gdx, by, id, red, cred);
  } else {
    if (NWW < NWD && (int)threadIdx.x >= NWW * 64) return;
    const int r = id - nd, by = r / gwx;
    dense_h_body<NWW, false, false, false, true, AS>(pw, r - by * gwx, by, r, red, cred);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dense_hw (round 6): the weight gradient dW[K, N] = x^T dy over 64 x 64 tiles, both operands staged ONCE through LDS.
// The 32 x 32 tiles above give a 1000 x 1000 gradient over a 256-row batch to 1024 workgroups whose two waves each walk
// 8 dependent (16 dword loads from L2 -> split -> 3 MFMA) steps with two steps in flight: 64 MB of L2 -> CU traffic
// for 4 MB of operands, each 4-byte load instruction a wave of its own in the texture unit, 15 us of a 25 us
// dense_h_pair launch (profiles/r06_step_timeline_factorvae_shapes3d_b256.txt).  Here a workgroup of eight waves
// (2 x 2 blocks of 32 x 32, times the two halves of a chunk's k-steps) owns a 64 x 64 tile: the 64 batch rows of a chunk
// arrive as 16-byte loads (two float4 per thread and operand), are scaled and split into the two f16 planes once, and
// wait in LDS as [row k][column] planes
// (128 bytes per row; the 32-byte column groups XOR-swizzled by (k >> 1) & 1 so that the four rows of a transposed read
// land in four bank quarters).  The MFMA operands want 8 consecutive k per lane with the column as the lane index:
// ds_read_b64_tr_b16 (blk_common.h: bk_tr) hands a lane column l of 4 rows, two of them make an operand.  Two chunks of
// loads are in flight beside the one being multiplied; one barrier per chunk; the two k-halves meet in LDS at the end
// (fixed order: bit reproducible).  LDS: 2 x 32 KB (the first half is the data-gradient role's reduction array in the
// paired launch), <= 128 registers: two workgroups of either role per CU.
constexpr int HW_ROWB = 128;           // a k-row of one plane: 64 f16
constexpr int HW_PLB = 64 * HW_ROWB;   // one plane of a 64-row chunk
constexpr int HW_BUF = 4 * HW_PLB;     // A high, A low, B high, B low: 32 KB
__host__ __device__ constexpr int hw_swz(int k) { return 2 * ((k >> 1) & 1); }

template <bool AS>
__device__ __forceinline__ void dense_hw_body(const DHParams& p, int bx, int by, char* buf0, char* buf1, float* cred) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int kh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int l31 = lane & 31, half = lane >> 5, l16 = lane & 15, cg = (lane >> 4) & 1;
  const int i0 = by * 64, j0 = bx * 64;
  const int n = (p.K + 63) >> 6;   // chunks of 64 batch rows
  const OdinRun RA = odin_run(p.A, (unsigned)((size_t)p.K * p.lda * 4));
  const OdinRun RB = odin_run(p.B, (unsigned)((size_t)p.K * p.ldb * 4));
  // staging items: rows r0 + 32 q (q = 0, 1) of the chunk, columns 4 c4 .. 4 c4 + 3 (widths are multiples of 8)
  const int r0 = tid >> 4, c4 = tid & 15;
  const bool ca_ok = i0 + 4 * c4 < p.M, cb_ok = j0 + 4 * c4 < p.N;
  const unsigned ga = (unsigned)((r0 * p.lda + i0 + 4 * c4) * 4), gb = (unsigned)((r0 * p.ldb + j0 + 4 * c4) * 4);
  const int sdst = r0 * HW_ROWB + (((c4 >> 2) ^ hw_swz(r0)) << 5) + (c4 & 3) * 8;   // (+ 32 rows: the same swizzle)
  float4 ra0[2], rb0[2], ra1[2], rb1[2];
  auto issue = [&](int c, float4 (&va)[2], float4 (&vb)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = 64 * c + 32 * q;
      const bool ok = k + r0 < p.K;
      va[q] = odin_run_load4(RA, (ok && ca_ok) ? ga + (unsigned)(k * p.lda * 4) : ODIN_OOB);
      vb[q] = odin_run_load4(RB, (ok && cb_ok) ? gb + (unsigned)(k * p.ldb * 4) : ODIN_OOB);
    }
  };
  issue(0, ra0, rb0);
  if (n > 1) issue(1, ra1, rb1);
  // the range words behind the first operand loads (dense_h_body): dy always scaled, x only outside the f16 window
  const OdinRangeReq g_rq = odin_range_issue(p.g_amax, lane);
  const OdinRangeReq a_rq = odin_range_issue(AS ? p.a_amax : nullptr, lane);
  const int gk = odin_range_shift(odin_range_finish(g_rq));
  const float g_s = odin_pow2(gk), g_s2k = odin_pow2(gk + 11);
  int ak = 0;
  bool a_on = false;
  if (AS) {
    const unsigned mb = odin_range_finish(a_rq);
    a_on = odin_act_needs_scale(mb);
    ak = a_on ? odin_range_shift(mb) : 0;
  }
  const float a_s = odin_pow2(ak), a_s2k = odin_pow2(ak + 11);
  const bool want_cs = p.colsum != nullptr && by == 0;
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto stage = [&](char* buf, const float4 (&va)[2], const float4 (&vb)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      char* d = buf + sdst + q * 32 * HW_ROWB;
      u32x2 h, l;
      if (AS && a_on) odin_split_h4<true>(va[q], a_s, a_s2k, h, l);
      else odin_split_h4<false>(va[q], 1.f, ODIN_LO_SCALE, h, l);
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + HW_PLB) = l;
      odin_split_h4<true>(vb[q], g_s, g_s2k, h, l);
      *reinterpret_cast<u32x2*>(d + 2 * HW_PLB) = h;
      *reinterpret_cast<u32x2*>(d + 3 * HW_PLB) = l;
      if (want_cs) { bs.x += vb[q].x; bs.y += vb[q].y; bs.z += vb[q].z; bs.w += vb[q].w; }
    }
  };
  // transposed reads: lane 4 q + t of a 16-lane group addresses row q, column quad t of the group's 16 columns
  const int tq = l16 >> 2, tp = l16 & 3;
  int aoff[2], boff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int k = 8 * half + 4 * m + tq;   // (+ 16 per k-step: the same swizzle)
    aoff[m] = k * HW_ROWB + (((2 * wi + cg) ^ hw_swz(k)) << 5) + tp * 8;
    boff[m] = 2 * HW_PLB + k * HW_ROWB + (((2 * wj + cg) ^ hw_swz(k)) << 5) + tp * 8;
  }
  f32x16 acc = f32x16_zero(), acx = f32x16_zero();
  auto compute = [&](const char* buf) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int s = 2 * t + kh;   // this wave's k-steps of the chunk
      u32x4 a[2], b[2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const char* base = buf + pl * HW_PLB + s * 16 * HW_ROWB;
#ifdef ODIN_SIM
        unsigned short ea[8], eb[8];
        for (int j = 0; j < 8; ++j) {
          const int k = 8 * half + j, ca = wi * 32 + l31, cb = wj * 32 + l31;
          ea[j] = *reinterpret_cast<const unsigned short*>(base + k * HW_ROWB + (((ca >> 4) ^ hw_swz(k)) << 5) + (ca & 15) * 2);
          eb[j] = *reinterpret_cast<const unsigned short*>(base + 2 * HW_PLB + k * HW_ROWB + (((cb >> 4) ^ hw_swz(k)) << 5) +
                                                           (cb & 15) * 2);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[pl][e] = (unsigned)ea[2 * e] | ((unsigned)ea[2 * e + 1] << 16);
          b[pl][e] = (unsigned)eb[2 * e] | ((unsigned)eb[2 * e + 1] << 16);
        }
        (void)aoff; (void)boff; (void)tp;
#else
        const u32x2 al = bk_tr(base + aoff[0], nullptr, 0, 0), ah = bk_tr(base + aoff[1], nullptr, 0, 0);
        const u32x2 bl = bk_tr(base + boff[0], nullptr, 0, 0), bh = bk_tr(base + boff[1], nullptr, 0, 0);
        a[pl][0] = al[0]; a[pl][1] = al[1]; a[pl][2] = ah[0]; a[pl][3] = ah[1];
        b[pl][0] = bl[0]; b[pl][1] = bl[1]; b[pl][2] = bh[0]; b[pl][3] = bh[1];
#endif
      }
      acx = mfma32_f16(a[0], b[1], acx);
      acc = mfma32_f16(a[0], b[0], acc);
      acx = mfma32_f16(a[1], b[0], acx);
    }
  };
  stage(buf0, ra0, rb0);
  if (n > 2) issue(2, ra0, rb0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < n; c += 2) {
    compute(buf0);
    ODIN_SCHED_FENCE();
    if (c + 1 < n) {
      stage(buf1, ra1, rb1);
      if (c + 3 < n) issue(c + 3, ra1, rb1);
    }
    __syncthreads();
    if (c + 1 >= n) break;
    compute(buf1);
    ODIN_SCHED_FENCE();
    if (c + 2 < n) {
      stage(buf0, ra0, rb0);
      if (c + 4 < n) issue(c + 4, ra0, rb0);
    }
    __syncthreads();
  }
  // ---- this wave pair's 32 x 32 block of the slab row (main + 2^-11 cross, the two scales one after the other): the
  // second k-half through LDS (every wave is behind its last read of the chunk buffers) ----
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11), o_a = odin_pow2(-ak);
  float* redl = reinterpret_cast<float*>(buf0) + (wave & 3) * 16 * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const float t = fmaf(acx[rr], o_sx, acc[rr] * o_s);
      redl[rr * 64] = AS ? t * o_a : t;
    }
  }
  __syncthreads();
  if (kh == 0) {
    const OdinRun RC = odin_run(p.C, (unsigned)((size_t)p.M * p.ldc * 4));
    const int jc = j0 + wj * 32 + l31;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const float t = fmaf(acx[rr], o_sx, acc[rr] * o_s);
      const int row = i0 + wi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      odin_run_store1(RC, (row < p.M && jc < p.N) ? (unsigned)((row * p.ldc + jc) * 4) : ODIN_OOB,
                      (AS ? t * o_a : t) + redl[rr * 64]);
    }
  }
  if (want_cs) {
    // db[j] = column sums of dy (raw fp32 values): the 4 row groups of a wave by lane swaps, the 8 waves through LDS
    float sv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sv[e] += __shfl_xor(sv[e], 16);
      sv[e] += __shfl_xor(sv[e], 32);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) cred[wave * 64 + 4 * lane + e] = sv[e];
    }
    __syncthreads();
    if (tid < 64 && j0 + tid < p.N) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += cred[w * 64 + tid];
      p.colsum[j0 + tid] = t;
    }
  }
}

template <bool AS>
__global__ __launch_bounds__(512, 4) void dense_hw_kernel(DHParams p) {
  __shared__ __attribute__((aligned(16))) float buf0[HW_BUF / 4];
  __shared__ float cred[512];
  ODIN_DYN_SMEM(char, buf1);
  dense_hw_body<AS>(p, (int)blockIdx.x, (int)blockIdx.y, reinterpret_cast<char*>(buf0), buf1, cred);
}

// ---------------------------------------------------------------------------------------------------------------------
// dense_hd (round 6): the DATA gradient dx[B, K] = (dy[B, N] w^T) act'(aux) over 64 x 64 tiles, both operands staged once
// through LDS -- the sibling of dense_hw for the arrangement in which BOTH operands are k-contiguous (rows of dy, rows of
// w).  On the 32 x 32 tiles CelebA's 4096 -> 512 projection (image_networks.py:688) at batch 512 is 2048 tiles of ONE wave
// each that walks 32 dependent k-steps: 34-38 us (profiles/r06_dense_hw_bench.txt).  Here a chunk of 64 k of 64 rows of
// each operand arrives as 16-byte loads, is scaled / split once and waits in LDS as [row][k] planes with a row pitch of
// 144 bytes (the 16-byte operand reads of 16 consecutive rows then cover all 64 banks); a lane's 8 consecutive k are ONE
// ds_read_b128.  Waves, chunk pipeline, the meeting of the two k-halves as in dense_hw; max |dx| is folded into the range word.
constexpr int HD_ROWB = 144;           // a row of one plane: 64 f16 + 16 bytes
constexpr int HD_PLB = 64 * HD_ROWB;   // 9216
constexpr int HD_BUF = 4 * HD_PLB;     // A high, A low, B high, B low: 36 KB

__global__ __launch_bounds__(512, 4) void dense_hd_kernel(DHParams p) {
  ODIN_DYN_SMEM(char, smem);   // 2 x HD_BUF
  __shared__ float cred[16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = bk_uniform(tid >> 6);
  const int kh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
  const int i0 = by * 64, j0 = bx * 64;
  const int n = (p.K + 63) >> 6;   // chunks of 64 k
  const OdinRun RA = odin_run(p.A, (unsigned)((size_t)p.M * p.lda * 4));
  const OdinRun RB = odin_run(p.B, (unsigned)((size_t)p.N * p.ldb * 4));
  // staging items: rows r0 + 32 q (q = 0, 1) of the tile, k = 4 c4 .. 4 c4 + 3 of the chunk (reduction lengths are multiples of 8)
  const int r0 = tid >> 4, c4 = tid & 15;
  unsigned ga[2], gb[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int ia = i0 + r0 + 32 * q, jb = j0 + r0 + 32 * q;
    ga[q] = ia < p.M ? (unsigned)((ia * p.lda + 4 * c4) * 4) : ODIN_OOB;
    gb[q] = jb < p.N ? (unsigned)((jb * p.ldb + 4 * c4) * 4) : ODIN_OOB;
  }
  const int sdst = r0 * HD_ROWB + c4 * 8;
  float4 ra0[2], rb0[2], ra1[2], rb1[2];
  auto issue = [&](int c, float4 (&va)[2], float4 (&vb)[2]) {
    const bool ok = 64 * c + 4 * c4 < p.K;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      va[q] = odin_run_load4(RA, (ok && ga[q] != ODIN_OOB) ? ga[q] + (unsigned)(256 * c) : ODIN_OOB);
      vb[q] = odin_run_load4(RB, (ok && gb[q] != ODIN_OOB) ? gb[q] + (unsigned)(256 * c) : ODIN_OOB);
    }
  };
  issue(0, ra0, rb0);
  if (n > 1) issue(1, ra1, rb1);
  const int jc = j0 + wj * 32 + l31;
  const OdinRangeReq g_rq = odin_range_issue(p.g_amax, lane);
  const int gk = odin_range_shift(odin_range_finish(g_rq));
  const float g_s = odin_pow2(gk), g_s2k = odin_pow2(gk + 11);
  auto stage = [&](char* buf, const float4 (&va)[2], const float4 (&vb)[2]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      char* d = buf + sdst + q * 32 * HD_ROWB;
      u32x2 h, l;
      odin_split_h4<true>(va[q], g_s, g_s2k, h, l);
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + HD_PLB) = l;
      odin_split_h4<false>(vb[q], 1.f, ODIN_LO_SCALE, h, l);
      *reinterpret_cast<u32x2*>(d + 2 * HD_PLB) = h;
      *reinterpret_cast<u32x2*>(d + 3 * HD_PLB) = l;
    }
  };
  const int aoff = (wi * 32 + l31) * HD_ROWB + half * 16, boff = 2 * HD_PLB + (wj * 32 + l31) * HD_ROWB + half * 16;
  f32x16 acc = f32x16_zero(), acx = f32x16_zero();
  auto compute = [&](const char* buf) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int s = 2 * t + kh;   // this wave's k-steps of the chunk
      const u32x4 ah = *reinterpret_cast<const u32x4*>(buf + aoff + s * 32), al = *reinterpret_cast<const u32x4*>(buf + HD_PLB + aoff + s * 32);
      const u32x4 bh = *reinterpret_cast<const u32x4*>(buf + boff + s * 32), bl = *reinterpret_cast<const u32x4*>(buf + HD_PLB + boff + s * 32);
      acx = mfma32_f16(ah, bl, acx);
      acc = mfma32_f16(ah, bh, acc);
      acx = mfma32_f16(al, bh, acx);
    }
  };
  char* buf0 = smem;
  char* buf1 = smem + HD_BUF;
  stage(buf0, ra0, rb0);
  if (n > 2) issue(2, ra0, rb0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < n; c += 2) {
    compute(buf0);
    ODIN_SCHED_FENCE();
    if (c + 1 < n) {
      stage(buf1, ra1, rb1);
      if (c + 3 < n) issue(c + 3, ra1, rb1);
    }
    __syncthreads();
    if (c + 1 >= n) break;
    compute(buf1);
    ODIN_SCHED_FENCE();
    if (c + 2 < n) {
      stage(buf0, ra0, rb0);
      if (c + 4 < n) issue(c + 4, ra0, rb0);
    }
    __syncthreads();
  }
  const float o_s = odin_pow2(-gk), o_sx = odin_pow2(-gk - 11);
  // the epilogue's act'(aux) of the elements this wave finishes (kh == 0): requested in front of the last barrier (16
  // registers that the main loop does not have to carry at 128 per wave)
  const OdinRun RX = odin_run(p.aux != nullptr ? p.aux : p.C, p.aux != nullptr ? (unsigned)((size_t)p.M * p.ldc * 4) : 0u);
  float auxv[16];
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) {
    const int row = i0 + wi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
    auxv[rr] = (p.aux != nullptr && kh == 0) ? odin_run_load1(RX, (row < p.M && jc < p.N) ? (unsigned)((row * p.ldc + jc) * 4) : ODIN_OOB) : 0.f;
  }
  float* redl = reinterpret_cast<float*>(buf0) + (wave & 3) * 16 * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) redl[rr * 64] = fmaf(acx[rr], o_sx, acc[rr] * o_s);
  }
  __syncthreads();
  float amx = 0.f;
  if (kh == 0) {
    const OdinRun RC = odin_run(p.C, (unsigned)((size_t)p.M * p.ldc * 4));
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      const int row = i0 + wi * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
      const unsigned off = (row < p.M && jc < p.N) ? (unsigned)((row * p.ldc + jc) * 4) : ODIN_OOB;
      float v = fmaf(acx[rr], o_sx, acc[rr] * o_s) + redl[rr * 64];
      if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, auxv[rr]);
      odin_run_store1(RC, off, v);
      amx = fmaxf(amx, off != ODIN_OOB ? fabsf(v) : 0.f);
    }
  }
  if (p.out_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, 512, cred, (unsigned)(bx + (int)gridDim.x * by));
  }
}

// the paired launch with the weight gradient on dense_hw: workgroups [0, nd) = data gradient (8 waves per 32 x 32 tile),
// the rest = 64 x 64 tiles of the weight gradient (8 waves)
template <bool AS>
__global__ __launch_bounds__(512, 4) void dense_h_pair64_kernel(DHParams pd, DHParams pw, int nd, int gdx, int gwx) {
  __shared__ __attribute__((aligned(16))) float red[8 * 16 * 64];   // 32 KB = HW_BUF
  __shared__ float cred[512];
  ODIN_DYN_SMEM(char, buf1);
  const int id = (int)blockIdx.x;
  if (id < nd) {
    const int by = id / gdx;
    dense_h_body<8, true, true, true, false, false>(pd, id - by * gdx, by, id, red, cred);
  } else {
    const int r = id - nd, by = r / gwx;
    dense_hw_body<AS>(pw, r - by * gwx, by, reinterpret_cast<char*>(red), buf1, cred);
  }
}

template <bool A_KC, bool B_KC, bool SCA, bool SCB, bool AS = false>
int dh_launch(const DHParams& p, int nw, void* stream) {
  dim3 grid((p.N + 31) / 32, (p.M + 31) / 32, 1);
  if (nw >= 8) ODIN_LAUNCH((dense_h_kernel<8, A_KC, B_KC, SCA, SCB, AS>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) ODIN_LAUNCH((dense_h_kernel<4, A_KC, B_KC, SCA, SCB, AS>), grid, dim3(256), 0, stream, p);
  else if (nw == 2) ODIN_LAUNCH((dense_h_kernel<2, A_KC, B_KC, SCA, SCB, AS>), grid, dim3(128), 0, stream, p);
  else ODIN_LAUNCH((dense_h_kernel<1, A_KC, B_KC, SCA, SCB, AS>), grid, dim3(64), 0, stream, p);
  return odin_check_launch("dense_h(f16x2)");
}

// waves per tile: enough waves to fill the chip (two per SIMD: 2048), at least 2 k-steps each
int dh_waves(int M, int N, int K, int kind = 0) {
  const long tiles = (long)((M + 31) / 32) * ((N + 31) / 32);
  const int steps = (K + 15) / 16;
  int nw = 1;
  while (nw < 8 && tiles * nw < 2048 && steps / (nw * 2) >= 2) nw *= 2;
  if (const char* e = ODIN_DIAG_ENV("ODIN_DH_NW")) nw = atoi(e);
  if (const char* e = ODIN_DIAG_ENV(kind == 0 ? "ODIN_DH_NW_F" : kind == 1 ? "ODIN_DH_NW_D" : "ODIN_DH_NW_W")) nw = atoi(e);
  return nw;
}

// dense_hw instead of the 32 x 32 tiles for a weight gradient [M, N]: enough 64 x 64 tiles to put one on (about) every
// second CU -- below that the small tiles' parallelism wins
int g_hw_min_tiles = 128;
bool dh_hw_ok(int M, int N) { return (long)((M + 63) / 64) * ((N + 63) / 64) >= g_hw_min_tiles; }
void dh_hw_attr() {
#ifndef ODIN_SIM
  static bool done = false;
  if (done) return;
  const void* ks[4] = {reinterpret_cast<const void*>(&dense_hw_kernel<true>), reinterpret_cast<const void*>(&dense_hw_kernel<false>),
                       reinterpret_cast<const void*>(&dense_h_pair64_kernel<true>),
                       reinterpret_cast<const void*>(&dense_h_pair64_kernel<false>)};
  for (const void* k : ks)
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, HW_BUF) != hipSuccess) (void)hipGetLastError();
  done = true;
#endif
}
// dense_hd instead of the 32 x 32 tiles for a data gradient [M, N]: the same tile count, and a reduction long enough that
// the small tiles would walk it with one or two waves
// (odin_debug_dense_hw_min_tiles(1), the tests' setting: every shape)
bool dh_hd_ok(int M, int N, int K) { return dh_hw_ok(M, N) && (g_hw_min_tiles <= 1 || dh_waves(M, N, K, 1) <= 2); }
int dh_hd_launch(const DHParams& p, void* stream) {
#ifndef ODIN_SIM
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&dense_hd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HD_BUF) != hipSuccess)
      (void)hipGetLastError();
    done = true;
  }
#endif
  dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, 1);
  ODIN_LAUNCH((dense_hd_kernel), grid, dim3(512), (size_t)2 * HD_BUF, stream, p);
  return odin_check_launch("dense_hd(f16x2)");
}
int dh_hw_launch(const DHParams& p, void* stream) {
  dh_hw_attr();
  dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, 1);
  if (p.a_amax != nullptr) ODIN_LAUNCH((dense_hw_kernel<true>), grid, dim3(512), (size_t)HW_BUF, stream, p);
  else ODIN_LAUNCH((dense_hw_kernel<false>), grid, dim3(512), (size_t)HW_BUF, stream, p);
  return odin_check_launch("dense_hw(f16x2)");
}

}  // namespace

extern "C" int odin_debug_dense_hw_min_tiles(int tiles) {
  const int old = g_hw_min_tiles;
  if (tiles >= 0) g_hw_min_tiles = tiles;
  return old;
}

// both layer widths >= 256, every reduction length (K, N, B) a multiple of 8
bool odin_dense_h_ok(int B, int K, int N) {
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NODENSEH")) return false;
  return B >= 32 && K >= 256 && N >= 256 && (B & 7) == 0 && (K & 7) == 0 && (N & 7) == 0 &&
         (long)B * K < (1L << 29) && (long)B * N < (1L << 29) && (long)K * N < (1L << 29);
}

// x_amax (optional): the range word of the activation x -- with it x is carried times its own power of two like a
// gradient operand (any magnitude keeps its 22 bits), without it unscaled (|x| <= 65504); y_amax (optional): the
// range word of y, folded in from the epilogue
int odin_dense_h_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                     const uint32_t* x_amax, uint32_t* y_amax, void* stream) {
  DHParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = w; p.C = y; p.bias = bias;
  p.M = B; p.N = N; p.K = K; p.lda = K; p.ldb = N; p.ldc = N; p.act = act;
  p.out_amax = y_amax;
  if (x_amax != nullptr) {
    p.g_amax = x_amax;
    p.g_cond = 1;
    return dh_launch<true, false, true, false>(p, dh_waves(B, N, K), stream);
  }
  return dh_launch<true, false, false, false>(p, dh_waves(B, N, K), stream);
}

// dx[B, K] = (dy[B, N] w^T) * act'(aux): reduction over N; w row j = its k-contiguous operand
int odin_dense_h_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx, int B, int K, int N,
                       const uint32_t* dy_amax, uint32_t* dx_amax, void* stream) {
  DHParams p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = w; p.C = dx; p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act;
  p.M = B; p.N = K; p.K = N; p.lda = N; p.ldb = N; p.ldc = K;
  p.g_amax = odin_range_word_of(dy, (size_t)B * N, dy_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "dense_h dgrad: no range word for dy");
  p.out_amax = dx_amax;
  if (dh_hd_ok(B, K, N)) return dh_hd_launch(p, stream);
  return dh_launch<true, true, true, false>(p, dh_waves(B, K, N, 1), stream);
}

static void dh_fill_dgrad(DHParams& p, const float* dy, const float* w, const float* aux, int aux_act, float* dx, int B,
                          int K, int N, const uint32_t* g_amax, uint32_t* dx_amax) {
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = w; p.C = dx; p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act;
  p.M = B; p.N = K; p.K = N; p.lda = N; p.ldb = N; p.ldc = K;
  p.g_amax = g_amax; p.out_amax = dx_amax;
}
static void dh_fill_wgrad(DHParams& p, const float* x, const float* dy, float* slab, int B, int K, int N,
                          const uint32_t* g_amax, const uint32_t* x_amax) {
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = dy; p.C = slab; p.colsum = slab + (size_t)K * N;
  p.M = K; p.N = N; p.K = B; p.lda = K; p.ldb = N; p.ldc = N;
  p.g_amax = g_amax; p.a_amax = x_amax;
}

// weight gradient (one complete slab row) + data gradient of a layer in ONE launch; dy_amax as the two calls
int odin_dense_h_bwd_pair(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          float* slab, int B, int K, int N, const uint32_t* dy_amax, uint32_t* dx_amax,
                          const uint32_t* x_amax, void* stream) {
  const uint32_t* g = odin_range_word_of(dy, (size_t)B * N, dy_amax, stream);
  if (g == nullptr) return odin_fail(-3, "dense_h bwd: no range word for dy");
  DHParams pd, pw;
  dh_fill_dgrad(pd, dy, w, aux, aux_act, dx, B, K, N, g, dx_amax);
  dh_fill_wgrad(pw, x, dy, slab, B, K, N, g, x_amax);
  const int nwd = dh_waves(B, K, N, 1), nww = dh_waves(K, N, B, 2);
#ifndef ODIN_SIM  // (the simulator's barrier counts every thread of the block: the roles keep their own launches there)
  if (nwd == 8 && dh_hw_ok(K, N)) {
    dh_hw_attr();
    const int gdx = (K + 31) / 32, gdy = (B + 31) / 32, gwx = (N + 63) / 64, gwy = (K + 63) / 64;
    const int nd = gdx * gdy;
    const dim3 grid((unsigned)(nd + gwx * gwy));
    if (x_amax != nullptr) ODIN_LAUNCH((dense_h_pair64_kernel<true>), grid, dim3(512), (size_t)HW_BUF, stream, pd, pw, nd, gdx, gwx);
    else ODIN_LAUNCH((dense_h_pair64_kernel<false>), grid, dim3(512), (size_t)HW_BUF, stream, pd, pw, nd, gdx, gwx);
    return odin_check_launch("dense_h_pair64(f16x2)");
  }
  if ((nwd == 8 || nwd == 4) && (nww == 2 || nww == 4 || nww == 1)) {
    const int gdx = (K + 31) / 32, gdy = (B + 31) / 32, gwx = (N + 31) / 32, gwy = (K + 31) / 32;
    const int nd = gdx * gdy, nw = gwx * gwy;
    const dim3 grid((unsigned)(nd + nw));
#define DH_PAIR(NWD_, NWW_)                                                                                         \
  do {                                                                                                              \
    if (x_amax != nullptr) ODIN_LAUNCH((dense_h_pair_kernel<NWD_, NWW_, true>), grid, dim3((NWD_ > NWW_ ? NWD_ : NWW_) * 64), 0, stream, pd, pw, nd, gdx, gwx); \
    else ODIN_LAUNCH((dense_h_pair_kernel<NWD_, NWW_, false>), grid, dim3((NWD_ > NWW_ ? NWD_ : NWW_) * 64), 0, stream, pd, pw, nd, gdx, gwx);                  \
    return odin_check_launch("dense_h_pair(f16x2)");                                                                \
  } while (0)
    if (nwd == 8 && nww == 2) DH_PAIR(8, 2);
    if (nwd == 8 && nww == 4) DH_PAIR(8, 4);
    if (nwd == 8 && nww == 1) DH_PAIR(8, 1);
    if (nwd == 4 && nww == 2) DH_PAIR(4, 2);
    if (nwd == 4 && nww == 4) DH_PAIR(4, 4);
    if (nwd == 4 && nww == 1) DH_PAIR(4, 1);
#undef DH_PAIR
  }
#endif
  int rc = dh_hw_ok(K, N)      ? dh_hw_launch(pw, stream)
           : x_amax != nullptr ? dh_launch<false, false, false, true, true>(pw, nww, stream)
                               : dh_launch<false, false, false, true>(pw, nww, stream);
  if (rc == 0) rc = dh_hd_ok(B, K, N) ? dh_hd_launch(pd, stream) : dh_launch<true, true, true, false>(pd, nwd, stream);
  return rc;
}

// slab row 0: dW[K, N] = x^T dy, then db[N] = column sums of dy: reduction over the batch; ONE complete row
int odin_dense_h_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N, const uint32_t* dy_amax,
                       const uint32_t* x_amax, void* stream) {
  DHParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = dy; p.C = slab; p.colsum = slab + (size_t)K * N;
  p.M = K; p.N = N; p.K = B; p.lda = K; p.ldb = N; p.ldc = N;
  p.g_amax = odin_range_word_of(dy, (size_t)B * N, dy_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "dense_h wgrad: no range word for dy");
  if (dh_hw_ok(K, N)) {
    p.a_amax = x_amax;
    return dh_hw_launch(p, stream);
  }
  if (x_amax != nullptr) {
    p.a_amax = x_amax;
    return dh_launch<false, false, false, true, true>(p, dh_waves(K, N, B, 2), stream);
  }
  return dh_launch<false, false, false, true>(p, dh_waves(K, N, B, 2), stream);
}
