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
#include "odin_device.h"
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
  unsigned* out_amax;       // dgrad: range word of C (may be null)
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
template <int NW, bool A_KC, bool B_KC, bool SCA, bool SCB>
__global__ __launch_bounds__(NW * 64) void dense_h_kernel(DHParams p) {
  __shared__ float red[NW * 16 * 64];
  __shared__ float cred[NW * 32 + 16];
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int ia = i0 + l31, jb = j0 + l31;
  const bool a_ok = ia < p.M, b_ok = jb < p.N;
  const int nsteps = (p.K + 15) >> 4;
  const OdinRun RA = odin_run(p.A, (unsigned)((size_t)(A_KC ? p.M * p.lda : p.K * p.lda) * 4));
  const OdinRun RB = odin_run(p.B, (unsigned)((size_t)(B_KC ? p.N * p.ldb : p.K * p.ldb) * 4));
  // the gradient operand is carried times 2^gk (its maximum lands in [2^14, 2^15)), the sums are scaled back
  const int gk = (SCA || SCB) ? odin_range_shift(odin_range_load(p.g_amax)) : 0;
  const float g_s = (SCA || SCB) ? odin_pow2(gk) : 1.f, g_s2k = (SCA || SCB) ? odin_pow2(gk + 11) : ODIN_LO_SCALE;
  f32x16 acc = f32x16_zero(), acx = f32x16_zero();
  float csum = 0.f;  // wgrad bias: column sum of B over this wave's k-steps (lane j = l31, half h)
  float a0[8], b0[8], a1[8], b1[8];
  auto mul = [&](const float (&av)[8], const float (&bv)[8]) {
    u32x4 ah, al, bh, bl;
    dh_split8<SCA>(av, g_s, g_s2k, ah, al);
    dh_split8<SCB>(bv, g_s, g_s2k, bh, bl);
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
    for (;;) {
      // (loads beyond the reduction read zeros through the range check: no branch around them)
      dh_load8<A_KC>(RA, a_ok, ia, p.lda, 16 * (s + NW) + 8 * h, p.K, a1);
      dh_load8<B_KC>(RB, b_ok, jb, p.ldb, 16 * (s + NW) + 8 * h, p.K, b1);
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
    }
  }
  // ---- the NW partial tiles meet in LDS (main + 2^-11 cross, scaled back); wave w finishes registers
  // [w * 16 / NW, (w + 1) * 16 / NW) in wave order: fixed summation order, bit reproducible ----
  const float o_s = (SCA || SCB) ? odin_pow2(-gk) : 1.f, o_sx = (SCA || SCB) ? odin_pow2(-gk - 11) : ODIN_LO_UNSCALE;
#pragma unroll
  for (int rr = 0; rr < 16; ++rr) red[(wave * 16 + rr) * 64 + lane] = fmaf(acx[rr], o_sx, acc[rr] * o_s);
  if (!A_KC && !B_KC && p.colsum != nullptr) {
    const float t = csum + __shfl_xor(csum, 32);
    if (h == 0) cred[wave * 32 + l31] = t;
  }
  __syncthreads();
  constexpr int RPW = 16 / NW > 0 ? 16 / NW : 1;  // registers a wave finishes (NW <= 16)
  const float bj = (p.bias != nullptr && b_ok) ? p.bias[jb] : 0.f;
  const OdinRun RC = odin_run(p.C, (unsigned)((size_t)p.M * p.ldc * 4));
  const OdinRun RX = odin_run(p.aux != nullptr ? p.aux : p.C, p.aux != nullptr ? (unsigned)((size_t)p.M * p.ldc * 4) : 0u);
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
      if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, odin_run_load1(RX, off));
      odin_run_store1(RC, off, v);
      amx = fmaxf(amx, off != ODIN_OOB ? fabsf(v) : 0.f);
    }
  }
  if (p.out_amax != nullptr) {
    __syncthreads();
    odin_amax_commit_wg(p.out_amax, amx, tid, NW * 64, cred, blockIdx.x + gridDim.x * blockIdx.y);
  } else if (!A_KC && !B_KC && p.colsum != nullptr && blockIdx.y == 0 && wave == 0 && h == 0 && b_ok) {
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += cred[w * 32 + l31];
    p.colsum[jb] = t;  // (sums of the raw fp32 values: no plane scale)
  }
}

template <bool A_KC, bool B_KC, bool SCA, bool SCB>
int dh_launch(const DHParams& p, int nw, void* stream) {
  dim3 grid((p.N + 31) / 32, (p.M + 31) / 32, 1);
  if (nw >= 8) ODIN_LAUNCH((dense_h_kernel<8, A_KC, B_KC, SCA, SCB>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) ODIN_LAUNCH((dense_h_kernel<4, A_KC, B_KC, SCA, SCB>), grid, dim3(256), 0, stream, p);
  else if (nw == 2) ODIN_LAUNCH((dense_h_kernel<2, A_KC, B_KC, SCA, SCB>), grid, dim3(128), 0, stream, p);
  else ODIN_LAUNCH((dense_h_kernel<1, A_KC, B_KC, SCA, SCB>), grid, dim3(64), 0, stream, p);
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

}  // namespace

// both layer widths >= 256, every reduction length (K, N, B) a multiple of 8
bool odin_dense_h_ok(int B, int K, int N) {
  if (odin_exact_fp32() || ODIN_DIAG_ENV("ODIN_NODENSEH")) return false;
  return B >= 32 && K >= 256 && N >= 256 && (B & 7) == 0 && (K & 7) == 0 && (N & 7) == 0 &&
         (long)B * K < (1L << 29) && (long)B * N < (1L << 29) && (long)K * N < (1L << 29);
}

int odin_dense_h_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                     void* stream) {
  DHParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = w; p.C = y; p.bias = bias;
  p.M = B; p.N = N; p.K = K; p.lda = K; p.ldb = N; p.ldc = N; p.act = act;
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
  return dh_launch<true, true, true, false>(p, dh_waves(B, K, N, 1), stream);
}

// slab row 0: dW[K, N] = x^T dy, then db[N] = column sums of dy: reduction over the batch; ONE complete row
int odin_dense_h_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N, const uint32_t* dy_amax,
                       void* stream) {
  DHParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = dy; p.C = slab; p.colsum = slab + (size_t)K * N;
  p.M = K; p.N = N; p.K = B; p.lda = K; p.ldb = N; p.ldc = N;
  p.g_amax = odin_range_word_of(dy, (size_t)B * N, dy_amax, stream);
  if (p.g_amax == nullptr) return odin_fail(-3, "dense_h wgrad: no range word for dy");
  return dh_launch<false, false, false, true>(p, dh_waves(K, N, B, 2), stream);
}
