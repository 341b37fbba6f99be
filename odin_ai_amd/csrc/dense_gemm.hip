// dense_gemm.hip -- Dense layers (keras.layers.Dense, odin/networks/base_networks.py:1002-1014;
// the encoder / decoder projections, image_networks.py:470,497; FactorDiscriminator's 1000-unit
// stack, factor_vae.py:150-153) as small fp32 GEMMs on the matrix cores with operands straight
// from L2 -- no LDS staging.
//
// At batch 128-512 these products are 0.07-0.5 GFLOP: on the tiled convolution path they become
// a handful of workgroups that stage the whole [B, K] operand through LDS in several serial
// chunks (enc4: 21 + 18 + 18 us for 3 x 0.067 GFLOP).  Here one workgroup owns ONE 32 x 32
// output tile and its NW waves split the reduction; every wave streams its k-range with 8-deep
// batches of 16-byte (k-contiguous operand) or coalesced 4-byte (index-contiguous operand)
// loads, accumulates with v_mfma_f32_32x32x2_f32 and the partial tiles meet in LDS in a fixed
// order (bit-reproducible).  Three operand arrangements cover forward, data gradient and weight
// gradient without ever transposing a matrix in memory:
//
//   C[i][j] = sum_k A(i, k) * B(k, j)
//   forward :  A = x  [B, K]  (k contiguous)   B = w [K, N] (j contiguous)   C = y  [B, N]
//   dgrad   :  A = dy [B, N]  (k contiguous)   B = w [K, N] (k contiguous: row j) C = dx [B, K]
//   wgrad   :  A = x  [B, K]  (i contiguous)   B = dy [B, N] (j contiguous)  C = dW [K, N]
//
// MFMA 32x32x2: lane (l31, h) supplies A[i = l31][k = h] and B[k = h][j = l31].  A lane loads 4
// consecutive k of its row at once (k = kb + 4h + 0..3) and feeds 4 MFMAs: MFMA q of a group uses
// k = {kb + q (h = 0), kb + 4 + q (h = 1)} on BOTH operands, so the pairing is consistent.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct DGParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;   // forward: [N]
  const float* aux;    // dgrad: act'(aux) multiplier, same shape as C
  float* colsum;       // wgrad: db[j] = sum_k B(k, j) written by tile row 0 (may be null)
  int M, N, K;         // C is [M, N]; reduction length K
  int lda, ldb, ldc;
  int a_kc, b_kc;      // 1: operand's k index is the contiguous one
  int act, aux_act;
  unsigned* out_amax;  // dgrad: range word of C (max |C| folded in by every tile), may be null
};

constexpr int DG_U = 8;  // k-groups (of 8 k-values) in flight per wave

// Unconditional, branch-free operand loads: a masked element carries the out-of-range offset and
// reads zero through the buffer descriptor's range check.  (With run-time layout flags the
// compiler built exec-mask branches around every load and drained vmcnt between them: one L2
// round trip per load.)
template <bool KC, bool VEC>
__device__ __forceinline__ void dg_load4(const OdinRun& R, bool ok, int idx, int ld, int k, int K,
                                         float (&v)[4]) {
  if constexpr (KC && VEC) {  // K % 4 == 0 and k % 4 == 0: the 4 values are all valid or all beyond K
    const float4 t = odin_run_load4(R, (ok && k < K) ? (unsigned)((idx * ld + k) * 4) : ODIN_OOB);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else if constexpr (KC) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = odin_run_load1(R, (ok && k + q < K) ? (unsigned)((idx * ld + k + q) * 4) : ODIN_OOB);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      v[q] = odin_run_load1(R, (ok && k + q < K) ? (unsigned)(((k + q) * ld + idx) * 4) : ODIN_OOB);
  }
}

template <int NW, bool A_KC, bool B_KC, bool VEC>
__global__ __launch_bounds__(NW * 64) void dense_gemm_kernel(DGParams p) {
  __shared__ float red[NW > 1 ? (NW - 1) * 16 * 64 : 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int ia = i0 + l31, jb = j0 + l31;
  const bool a_ok = ia < p.M, b_ok = jb < p.N;
  // this wave's k range: groups of 8 k-values, dealt round-robin in blocks of DG_U groups
  const int ngroups = (p.K + 7) >> 3;
  const OdinRun RA = odin_run(p.A, (unsigned)((size_t)(A_KC ? p.M * p.lda : p.K * p.lda) * 4));
  const OdinRun RB = odin_run(p.B, (unsigned)((size_t)(B_KC ? p.N * p.ldb : p.K * p.ldb) * 4));
  f32x16 acc = f32x16_zero();
  float csum = 0.f;  // wgrad bias: column sum of B over this wave's k range (lane j = l31, half h)
  for (int g0 = wave * DG_U; g0 < ngroups; g0 += NW * DG_U) {
    float av[DG_U][4], bv[DG_U][4];
#pragma unroll
    for (int u = 0; u < DG_U; ++u) {
      const int k = (g0 + u) * 8 + 4 * h;  // groups beyond K read zeros
      dg_load4<A_KC, VEC>(RA, a_ok, ia, p.lda, k, p.K, av[u]);
      dg_load4<B_KC, VEC>(RB, b_ok, jb, p.ldb, k, p.K, bv[u]);
    }
#pragma unroll
    for (int u = 0; u < DG_U; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc = mfma32(av[u][q], bv[u][q], acc);
        csum += bv[u][q];
      }
    }
  }
  // ---- combine the NW partial tiles in wave order (fixed order: reproducible) ----
  if (NW > 1) {
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
      for (int w = 1; w < NW; ++w) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[((w - 1) * 16 + r) * 64 + lane];
      }
    }
  }
  if (p.colsum != nullptr && blockIdx.y == 0) {
    // db[j] = sum over all k of B(k, j): halves h = 0 / 1 and the NW waves
    __syncthreads();
    float t = csum + __shfl_xor(csum, 32);
    if (h == 0) red[wave * 32 + l31] = t;
    __syncthreads();
    if (wave == 0 && h == 0 && b_ok) {
      float s = 0.f;
      for (int w = 0; w < NW; ++w) s += red[w * 32 + l31];
      p.colsum[jb] = s;
    }
  }
  if (wave != 0) return;
  // ---- epilogue: lane holds column j = jb, rows i0 + (r & 3) + 8 (r >> 2) + 4 h ----
  const float bj = (b_ok && p.bias != nullptr) ? p.bias[jb] : 0.f;
  float amx = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (b_ok && i < p.M) {
      float v = odin_act(p.act, acc[r] + bj);
      if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, p.aux[(size_t)i * p.ldc + jb]);
      p.C[(size_t)i * p.ldc + jb] = v;
      amx = fmaxf(amx, fabsf(v));
    }
  }
  odin_amax_commit_wave(p.out_amax, amx, lane, blockIdx.y * gridDim.x + blockIdx.x);
}

template <bool A_KC, bool B_KC, bool VEC>
int dg_launch_t(DGParams& p, dim3 grid, int nw, void* stream) {
  if (nw >= 16) ODIN_LAUNCH((dense_gemm_kernel<16, A_KC, B_KC, VEC>), grid, dim3(1024), 0, stream, p);
  else if (nw == 8) ODIN_LAUNCH((dense_gemm_kernel<8, A_KC, B_KC, VEC>), grid, dim3(512), 0, stream, p);
  else if (nw == 4) ODIN_LAUNCH((dense_gemm_kernel<4, A_KC, B_KC, VEC>), grid, dim3(256), 0, stream, p);
  else if (nw == 2) ODIN_LAUNCH((dense_gemm_kernel<2, A_KC, B_KC, VEC>), grid, dim3(128), 0, stream, p);
  else ODIN_LAUNCH((dense_gemm_kernel<1, A_KC, B_KC, VEC>), grid, dim3(64), 0, stream, p);
  return odin_check_launch("dense_gemm");
}

int dg_launch(DGParams& p, void* stream) {
  dim3 grid((p.N + 31) / 32, (p.M + 31) / 32, 1);
  const long tiles = (long)grid.x * grid.y;
  const int ngroups = (p.K + 7) / 8;
  // waves per tile: enough k-groups per wave to amortise the launch, enough waves to fill the chip
  int nw = 1;
  while (nw < 16 && tiles * nw < 2 * 256 && ngroups / (nw * 2) >= DG_U) nw *= 2;
  // 16-byte loads along k: every k-contiguous operand has rows of a multiple of 4 floats
  const bool vec = (p.K % 4 == 0) && (!p.a_kc || ((p.lda & 3) == 0 && (((size_t)p.A) & 15) == 0)) &&
                   (!p.b_kc || ((p.ldb & 3) == 0 && (((size_t)p.B) & 15) == 0));
  if (p.a_kc && p.b_kc)
    return vec ? dg_launch_t<true, true, true>(p, grid, nw, stream)
               : dg_launch_t<true, true, false>(p, grid, nw, stream);
  if (p.a_kc)
    return vec ? dg_launch_t<true, false, true>(p, grid, nw, stream)
               : dg_launch_t<true, false, false>(p, grid, nw, stream);
  return dg_launch_t<false, false, false>(p, grid, nw, stream);
}

}  // namespace

// Small-GEMM regime: everything a few hundred 32x32 tiles cover (the tiled path wins once a
// layer alone fills the chip for long enough to amortise its LDS staging).
bool odin_dense_gemm_ok(int B, int K, int N) {
  static int off = -1;
  if (off < 0) off = ODIN_DIAG_ENV("ODIN_NODENSEGEMM") ? 1 : 0;
  if (off) return false;
  const double flop = 2.0 * B * K * N;
  return B >= 1 && B <= 4096 && K >= 1 && N >= 1 && flop <= 1.2e9 && (long)K * N < (1L << 28) &&
         (long)B * (K > N ? K : N) < (1L << 28);
}

int odin_dense_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K,
                        int N, int act, void* stream) {
  DGParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = w; p.C = y; p.bias = bias;
  p.M = B; p.N = N; p.K = K; p.lda = K; p.ldb = N; p.ldc = N;
  p.a_kc = 1; p.b_kc = 0; p.act = act;
  return dg_launch(p, stream);
}

int odin_dense_gemm_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          int B, int K, int N, uint32_t* dx_amax, void* stream) {
  DGParams p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = w; p.C = dx;
  p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act;
  p.M = B; p.N = K; p.K = N; p.lda = N; p.ldb = N; p.ldc = K;
  p.a_kc = 1; p.b_kc = 1;
  p.out_amax = dx_amax;
  return dg_launch(p, stream);
}

// slab: ONE row [dW (K, N) | db (N)]
int odin_dense_gemm_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N,
                          void* stream) {
  DGParams p;
  memset(&p, 0, sizeof(p));
  p.A = x; p.B = dy; p.C = slab; p.colsum = slab + (size_t)K * N;
  p.M = K; p.N = N; p.K = B; p.lda = K; p.ldb = N; p.ldc = N;
  p.a_kc = 0; p.b_kc = 0;
  return dg_launch(p, stream);
}
