// thin_dense.hip -- Dense layers with ONE thin side as streaming kernels on the vector ALU (round 6).
//
// FactorVAE's discriminator (factor_vae.py:150-176, factor_discriminator.py:16-95: Flatten -> 5 x Dense(1000, relu) ->
// Dense(1)) starts with a [B, zdim <= 32] x [zdim, 1000] layer and ends with a [B, 1000] x [1000, 1] layer.  Neither has
// a GEMM in it -- 0.8 and 0.1 MFLOP at B = 128 -- but round 5 ran them on the matrix-core families (dense_gemm / igemm:
// 8.5-11.8 us per launch, six such launches per iteration plus an absmax pass for the range word of the first layer's
// output: 75 us of a 0.81 ms iteration).  Here:
//   thin K (K <= 32, N >= 64): forward = thread owns 4 columns and its K x 4 weights, streams the rows; data gradient =
//     one wave per row, K running sums per lane, wave reduction; weight gradient = thread owns 4 columns, K x 4 sums over
//     its row chunk (slab rows = row chunks)
//   thin N (N <= 4, K >= 64):  forward = one wave per row; data gradient = elementwise; weight gradient = thread owns 4 k
// All sums in a fixed order (bit reproducible); every kernel that produces an activation / gradient tensor folds its
// max |value| into the range word it is handed (no absmax pass).
#include "odin_device.h"
#include "odin_internal.h"

namespace {

constexpr int TD_KMAX = 32;    // thin K: forward / data gradient
constexpr int TD_KWMAX = 16;   // thin K: weight gradient (K x 4 + 4 accumulators per thread)
constexpr int TD_NMAX = 4;     // thin N

struct TDParams {
  const float* x;     // [B, K]
  const float* w;     // [K, N]
  const float* bias;  // [N]
  float* y;           // [B, N]
  const float* dy;    // [B, N]
  const float* aux;   // [B, K] or null: dx *= act'(aux)
  float* dx;          // [B, K]
  float* slab;        // [rows][K * N + N]
  unsigned* amax;     // range word of the tensor the kernel writes (may be null)
  int B, K, N, act, aux_act, chunk;
};

// ---- thin K forward: grid (ceil(N / 1024), ceil(B / RB)), 256 threads = 256 x 4 columns ----
template <int RB, int KT>   // KT: K rounded up to 8 / 16 / 32 (register array, compile-time indices)
__global__ __launch_bounds__(256) void think_fwd_kernel(TDParams p) {
  __shared__ __attribute__((aligned(16))) float xs[RB * KT];   // rows padded to KT with zeros: no per-k branches below
  __shared__ float red[16];
  const int tid = threadIdx.x, n0 = (blockIdx.x * 256 + tid) * 4, b0 = blockIdx.y * RB;
  const int K = p.K, N = p.N;
  const int nb = (p.B - b0 < RB) ? p.B - b0 : RB;
  for (int e = tid; e < RB * KT; e += 256) {
    const int r = e / KT, k = e - r * KT;
    xs[e] = (r < nb && k < K) ? p.x[(size_t)(b0 + r) * K + k] : 0.f;
  }
  const bool on = n0 < N;            // (N is a multiple of 4: checked by the host)
  f32x4 wv[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)   // (unconditional loads of clamped rows / columns: nothing drains vmcnt between them)
    wv[k] = *reinterpret_cast<const f32x4*>(p.w + (size_t)(k < K ? k : K - 1) * N + (on ? n0 : 0));
  const float4 bv = (on && p.bias != nullptr) ? *reinterpret_cast<const float4*>(p.bias + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  float amx = 0.f;
  for (int r = 0; r < nb; ++r) {
    float4 a = bv;
#pragma unroll
    for (int k4 = 0; k4 < KT; k4 += 4) {
      const float4 xq = *reinterpret_cast<const float4*>(xs + r * KT + k4);   // (same address in every lane)
      const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a.x = fmaf(xv[i], wv[k4 + i].x, a.x); a.y = fmaf(xv[i], wv[k4 + i].y, a.y);
        a.z = fmaf(xv[i], wv[k4 + i].z, a.z); a.w = fmaf(xv[i], wv[k4 + i].w, a.w);
      }
    }
    a.x = odin_act(p.act, a.x); a.y = odin_act(p.act, a.y); a.z = odin_act(p.act, a.z); a.w = odin_act(p.act, a.w);
    if (on) {
      *reinterpret_cast<float4*>(p.y + (size_t)(b0 + r) * N + n0) = a;
      amx = odin_amax3(odin_amax3(amx, a.x, a.y), a.z, a.w);
    }
  }
  odin_amax_commit_wg(p.amax, amx, tid, 256, red, blockIdx.x + gridDim.x * blockIdx.y);
}

// ---- thin K data gradient: 4 rows per workgroup; thread = 4 columns of all 4 rows (ONE round of loads per 1024 columns:
// the first form -- a wave per row striding the columns -- was a chain of dependent L2 round trips: 10 us for 0.8 MFLOP),
// K x 4 running sums per thread, wave sums on the vector ALU, the 4 waves through LDS ----
template <int KT>   // K rounded up to 8 / 16 / 32 (register array)
__global__ __launch_bounds__(256) void think_dgrad_kernel(TDParams p) {
  __shared__ float red[16];
  __shared__ float part[4][4][TD_KMAX];   // [wave][row][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 4, K = p.K, N = p.N;
  const int nb = (p.B - b0 < 4) ? p.B - b0 : 4;
  float acc[4][KT];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < KT; ++k) acc[r][k] = 0.f;
  for (int n = 4 * tid; n < N; n += 1024) {
    f32x4 g[4], wv[KT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      g[r] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)(b0 + (r < nb ? r : nb - 1)) * N + n);
      if (r >= nb) g[r] = (f32x4)(0.f);
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.w + (size_t)(k < K ? k : K - 1) * N + n);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < KT; ++k) {   // (rows of w beyond K are clamped copies: their sums are never read)
        acc[r][k] = fmaf(g[r].x, wv[k].x, acc[r][k]); acc[r][k] = fmaf(g[r].y, wv[k].y, acc[r][k]);
        acc[r][k] = fmaf(g[r].z, wv[k].z, acc[r][k]); acc[r][k] = fmaf(g[r].w, wv[k].w, acc[r][k]);
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const float v = odin_wave_sum64_valu(acc[r][k]);
      if (lane == 0) part[wave][r][k] = v;
    }
  __syncthreads();
  float amx = 0.f;
  if (tid < 4 * K) {
    const int r = tid / K, k = tid - r * K;
    if (r < nb) {
      float v = (part[0][r][k] + part[1][r][k]) + (part[2][r][k] + part[3][r][k]);
      if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, p.aux[(size_t)(b0 + r) * K + k]);
      p.dx[(size_t)(b0 + r) * K + k] = v;
      amx = fabsf(v);
    }
  }
  odin_amax_commit_wg(p.amax, amx, tid, 256, red, blockIdx.x);
}

// ---- thin K weight gradient: grid (ceil(N / 1024), rows): dW[k][n] = sum over the chunk's rows of x[b][k] dy[b][n] ----
template <int KT>
__global__ __launch_bounds__(256) void think_wgrad_kernel(TDParams p) {
  ODIN_DYN_SMEM(float, xs);   // [chunk][KT], rows padded with zeros: no per-k branches in the loop
  const int tid = threadIdx.x, n0 = (blockIdx.x * 256 + tid) * 4, K = p.K, N = p.N;
  const int b0 = blockIdx.y * p.chunk;
  const int nb = (p.B - b0 < p.chunk) ? (p.B - b0 > 0 ? p.B - b0 : 0) : p.chunk;
  for (int e = tid; e < nb * KT; e += 256) {
    const int r = e / KT, k = e - r * KT;
    xs[e] = k < K ? p.x[(size_t)(b0 + r) * K + k] : 0.f;
  }
  __syncthreads();
  const bool on = n0 < N;
  f32x4 acc[KT];
  float4 accb = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < KT; ++k) acc[k] = (f32x4)(0.f);
  if (on) {
    for (int r0 = 0; r0 < nb; r0 += 8) {   // (8 rows of dy requested together: one round trip per 8 rows, rows ascending)
      f32x4 gq[8];
#pragma unroll
      // (unconditional loads of a clamped row: a conditional load becomes a branch that drains vmcnt -- 8 dependent
      // round trips, 15 us measured; rows beyond the chunk are zeroed after the load)
      for (int u = 0; u < 8; ++u) {
        const int rc = r0 + u < nb ? r0 + u : nb - 1;
        gq[u] = *reinterpret_cast<const f32x4*>(p.dy + (size_t)(b0 + rc) * N + n0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 g = r0 + u < nb ? gq[u] : (f32x4)(0.f);
        const int r = r0 + u < nb ? r0 + u : 0;
        accb.x += g.x; accb.y += g.y; accb.z += g.z; accb.w += g.w;
#pragma unroll
        for (int k4 = 0; k4 < KT; k4 += 4) {
          const float4 xq = *reinterpret_cast<const float4*>(xs + r * KT + k4);   // (same address in every lane)
          const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            acc[k4 + i].x = fmaf(xv[i], g.x, acc[k4 + i].x); acc[k4 + i].y = fmaf(xv[i], g.y, acc[k4 + i].y);
            acc[k4 + i].z = fmaf(xv[i], g.z, acc[k4 + i].z); acc[k4 + i].w = fmaf(xv[i], g.w, acc[k4 + i].w);
          }
        }
      }
    }
    float* row = p.slab + (size_t)blockIdx.y * ((size_t)K * N + N);
#pragma unroll
    for (int k = 0; k < KT; ++k)
      if (k < K) *reinterpret_cast<f32x4*>(row + (size_t)k * N + n0) = acc[k];
    *reinterpret_cast<float4*>(row + (size_t)K * N + n0) = accb;
  }
}

// ---- thin N forward: one wave per row, 4 rows per workgroup ----
__global__ __launch_bounds__(256) void thinn_fwd_kernel(TDParams p) {
  __shared__ float red[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x * 4 + wave, K = p.K, N = p.N;
  float acc[TD_NMAX] = {0.f, 0.f, 0.f, 0.f};
  float amx = 0.f;
  if (b < p.B) {
    const float* xr = p.x + (size_t)b * K;
    for (int k = 4 * lane; k < K; k += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
      for (int n = 0; n < TD_NMAX; ++n) {
        if (n < N) {
          acc[n] = fmaf(xv.x, p.w[(size_t)k * N + n], acc[n]);
          acc[n] = fmaf(xv.y, p.w[(size_t)(k + 1) * N + n], acc[n]);
          acc[n] = fmaf(xv.z, p.w[(size_t)(k + 2) * N + n], acc[n]);
          acc[n] = fmaf(xv.w, p.w[(size_t)(k + 3) * N + n], acc[n]);
        }
      }
    }
#pragma unroll
    for (int n = 0; n < TD_NMAX; ++n) {
      if (n < N) {
        float v = odin_wave_sum64_valu(acc[n]);
        if (lane == 0) {
          v = odin_act(p.act, v + (p.bias != nullptr ? p.bias[n] : 0.f));
          p.y[(size_t)b * N + n] = v;
          amx = fmaxf(amx, fabsf(v));
        }
      }
    }
  }
  odin_amax_commit_wg(p.amax, amx, tid, 256, red, blockIdx.x);
}

// ---- thin N data gradient: dx[b][k] = (sum_n dy[b][n] w[k][n]) act'(aux[b][k]): thread = 4 k of one row ----
__global__ __launch_bounds__(256) void thinn_dgrad_kernel(TDParams p) {
  __shared__ float red[16];
  const int tid = threadIdx.x, K = p.K, N = p.N, k4n = K >> 2;
  const long e = (long)blockIdx.x * 256 + tid;
  float amx = 0.f;
  if (e < (long)p.B * k4n) {
    const int b = (int)(e / k4n), k = 4 * (int)(e - (long)b * k4n);
    float g[TD_NMAX];
#pragma unroll
    for (int n = 0; n < TD_NMAX; ++n) g[n] = n < N ? p.dy[(size_t)b * N + n] : 0.f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < TD_NMAX; ++n)
        if (n < N) v[i] = fmaf(g[n], p.w[(size_t)(k + i) * N + n], v[i]);
    if (p.aux != nullptr) {
      const float4 a = *reinterpret_cast<const float4*>(p.aux + (size_t)b * K + k);
      v[0] *= odin_act_grad(p.aux_act, a.x); v[1] *= odin_act_grad(p.aux_act, a.y);
      v[2] *= odin_act_grad(p.aux_act, a.z); v[3] *= odin_act_grad(p.aux_act, a.w);
    }
    *reinterpret_cast<float4*>(p.dx + (size_t)b * K + k) = make_float4(v[0], v[1], v[2], v[3]);
    amx = odin_amax3(odin_amax3(0.f, v[0], v[1]), v[2], v[3]);
  }
  odin_amax_commit_wg(p.amax, amx, tid, 256, red, blockIdx.x);
}

// ---- thin N weight gradient: grid (ceil(K / 1024), rows): thread = 4 k, sums over the chunk's rows ----
__global__ __launch_bounds__(256) void thinn_wgrad_kernel(TDParams p) {
  ODIN_DYN_SMEM(float, gs);   // [chunk][N]
  const int tid = threadIdx.x, k0 = (blockIdx.x * 256 + tid) * 4, K = p.K, N = p.N;
  const int b0 = blockIdx.y * p.chunk;
  const int nb = (p.B - b0 < p.chunk) ? (p.B - b0 > 0 ? p.B - b0 : 0) : p.chunk;
  for (int e = tid; e < nb * N; e += 256) gs[e] = p.dy[(size_t)b0 * N + e];
  __syncthreads();
  float* row = p.slab + (size_t)blockIdx.y * ((size_t)K * N + N);
  if (k0 < K) {
    float acc[4][TD_NMAX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < TD_NMAX; ++n) acc[i][n] = 0.f;
    for (int r0 = 0; r0 < nb; r0 += 8) {   // (8 rows of x requested together)
      f32x4 xq[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int rc = r0 + u < nb ? r0 + u : nb - 1;
        xq[u] = *reinterpret_cast<const f32x4*>(p.x + (size_t)(b0 + rc) * K + k0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const f32x4 xv = r0 + u < nb ? xq[u] : (f32x4)(0.f);
        const int r = r0 + u < nb ? r0 + u : 0;
#pragma unroll
        for (int n = 0; n < TD_NMAX; ++n) {
          if (n < N) {
            const float g = gs[r * N + n];
            acc[0][n] = fmaf(xv.x, g, acc[0][n]); acc[1][n] = fmaf(xv.y, g, acc[1][n]);
            acc[2][n] = fmaf(xv.z, g, acc[2][n]); acc[3][n] = fmaf(xv.w, g, acc[3][n]);
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int n = 0; n < TD_NMAX; ++n)
        if (n < N) row[(size_t)(k0 + i) * N + n] = acc[i][n];
  }
  if (blockIdx.x == 0 && tid < N) {   // bias gradient of this chunk, rows ascending
    float s = 0.f;
    for (int r = 0; r < nb; ++r) s += gs[r * N + tid];
    row[(size_t)K * N + tid] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The discriminator's HEAD with its loss in one launch (round 6).  FactorVAE's discriminator ends in Dense(1000 -> 1)
// (factor_discriminator.py:60-95); around it the iteration ran thinn_fwd -> mean -> thinn_dgrad in the VAE step
// (total_correlation, factor_discriminator.py:169-198) and thinn_fwd -> dtc_loss -> thinn_wgrad -> thinn_dgrad in the
// discriminator step (dtc_loss, :200-235): seven launches of ~5 us each, every one a launch floor
// (profiles/r06_step_timeline_factorvae_shapes3d_b256.txt).  Nothing in them couples rows except the two means:
//   logit[b] = h[b, :] . w + bias
//   mode 0   out = mean_b logit[b],  dlogit[b] = dlogit_in[b]  (the caller's constant tc_coef / B)
//   mode 1   rows [0, n) are D(z), rows [n, 2 n) are D(z_perm):  out = (0.5 / n) sum softplus(-a_z) + softplus(a_perm),
//            dlogit = -sigmoid(-a_z) 0.5 / n | sigmoid(a_perm) 0.5 / n
//   dh[b, k] = dlogit[b] w[k] act'(h[b, k])         (+ max |dh| into its range word)
//   slab row of this workgroup: dW[k] = sum over its rows of h[b, k] dlogit[b] | db = sum dlogit[b]
// A workgroup owns 8 rows; a thread holds 4 k of each of them (all 8 loads in flight at once; K <= 2048).  The mean over
// the workgroups: each adds its partial sum as a FIXED-POINT number (2^-30 units: integer addition commutes, so the result
// does not depend on the order of arrival) together with a ticket to ONE 64-bit workspace word; the last arrival
// converts, writes out[0] and clears the word for the next launch.  A device-scope atomic performed at the coherence
// point -- no fence, no second launch.
struct DHeadParams {
  const float* h;
  const float* w;
  const float* bias;
  float* logit;
  const float* dlogit_in;
  float* dlogit_out;
  float* out;
  float* dh;
  unsigned* dh_amax;
  float* slab;
  unsigned long long* ws;   // [0]: fixed-point sum << 10 | tickets; zero between launches
  int B, K, mode, aux_act;
};

__device__ __forceinline__ float td_softplus(float x) { return fmaxf(x, 0.f) + log1pf(odin_exp(-fabsf(x))); }
__device__ __forceinline__ float td_sigmoid(float x) {
  const float e = odin_exp(-fabsf(x)), s = 1.f / (1.f + e);
  return x >= 0.f ? s : e * s;
}

template <int KV>   // float4 columns per thread: K <= 1024 KV
__global__ __launch_bounds__(256) void disc_head_kernel(DHeadParams p) {
  __shared__ float red[4 * 8];
  __shared__ float dl[8], lsum[8];
  __shared__ float ared[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * 8, K = p.K, k4n = K >> 2;
  const int nb = p.B - b0 < 8 ? p.B - b0 : 8;
  float4 wv[KV], hv[8][KV];
#pragma unroll
  for (int v = 0; v < KV; ++v) {
    const int k4 = tid + 256 * v;
    const bool ok = k4 < k4n;
    wv[v] = ok ? reinterpret_cast<const float4*>(p.w)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < 8; ++r)
      hv[r][v] = (ok && r < nb) ? reinterpret_cast<const float4*>(p.h + (size_t)(b0 + r) * K)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float a = 0.f;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      a = fmaf(hv[r][v].x, wv[v].x, a); a = fmaf(hv[r][v].y, wv[v].y, a);
      a = fmaf(hv[r][v].z, wv[v].z, a); a = fmaf(hv[r][v].w, wv[v].w, a);
    }
    a = odin_wave_sum64_valu(a);
    if (lane == 0) red[wave * 8 + r] = a;
  }
  __syncthreads();
  if (tid < 8) {
    const int b = b0 + tid;
    const bool valid = tid < nb;
    const float a = ((red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid])) + (p.bias != nullptr ? p.bias[0] : 0.f);
    float d = 0.f, ls = 0.f;
    if (valid) {
      p.logit[b] = a;
      if (p.mode == 0) {
        d = p.dlogit_in[b];
        ls = a;
      } else {
        const int n = p.B >> 1;
        const float inv = 0.5f / (float)n;
        if (b < n) { ls = td_softplus(-a); d = -td_sigmoid(-a) * inv; }
        else { ls = td_softplus(a); d = td_sigmoid(a) * inv; }
        if (p.dlogit_out != nullptr) p.dlogit_out[b] = d;
      }
    }
    dl[tid] = d;
    lsum[tid] = ls;
  }
  __syncthreads();
  if (tid == 0) {
    float sacc = 0.f;
    for (int r = 0; r < 8; ++r) sacc += lsum[r];
    // one atomic per workgroup: the sum in 2^-30 units above a 10-bit ticket count (54 bits: |sum| < 2^23; at most 1023
    // workgroups); the old value names the last arrival and, with its own share, the total
    const long long fx = (long long)((double)sacc * 1073741824.0);
    const unsigned long long mine = ((unsigned long long)fx << 10) + 1ull;
    const unsigned long long old = atomicAdd(&p.ws[0], mine);
    if ((old & 1023ull) == (unsigned long long)gridDim.x - 1ull) {
      (void)atomicExch(&p.ws[0], 0ull);
      const long long tot = (long long)(old + mine - (unsigned long long)gridDim.x) >> 10;
      const double sum = (double)tot / 1073741824.0;
      p.out[0] = p.mode == 0 ? (float)sum / (float)p.B : (float)sum * (0.5f / (float)(p.B >> 1));
    }
  }
  float amx = 0.f;
  if (p.dh != nullptr) {
    float* row = p.slab != nullptr ? p.slab + (size_t)blockIdx.x * ((size_t)K + 1) : nullptr;
#pragma unroll
    for (int v = 0; v < KV; ++v) {
      const int k4 = tid + 256 * v;
      if (k4 < k4n) {
        float4 sw = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (r < nb) {
            const float d = dl[r];
            const float4 x = hv[r][v];
            float4 g;
            g.x = d * wv[v].x * odin_act_grad(p.aux_act, x.x); g.y = d * wv[v].y * odin_act_grad(p.aux_act, x.y);
            g.z = d * wv[v].z * odin_act_grad(p.aux_act, x.z); g.w = d * wv[v].w * odin_act_grad(p.aux_act, x.w);
            reinterpret_cast<float4*>(p.dh + (size_t)(b0 + r) * K)[k4] = g;
            amx = odin_amax3(odin_amax3(amx, g.x, g.y), g.z, g.w);
            sw.x = fmaf(x.x, d, sw.x); sw.y = fmaf(x.y, d, sw.y); sw.z = fmaf(x.z, d, sw.z); sw.w = fmaf(x.w, d, sw.w);
          }
        }
        if (row != nullptr) { row[4 * k4] = sw.x; row[4 * k4 + 1] = sw.y; row[4 * k4 + 2] = sw.z; row[4 * k4 + 3] = sw.w; }
      }
    }
    if (row != nullptr && tid == 0) {
      float sb = 0.f;
      for (int r = 0; r < 8; ++r) sb += dl[r];
      row[K] = sb;
    }
  }
  odin_amax_commit_wg(p.dh_amax, amx, tid, 256, ared, blockIdx.x);
}

bool td_al16(const void* p) { return (((size_t)p) & 15) == 0; }

void td_fill(TDParams& p, int B, int K, int N) {
  memset(&p, 0, sizeof(p));
  p.B = B; p.K = K; p.N = N;
}

}  // namespace

// which thin family serves Dense [B, K] x [K, N]: 1 = thin K, 2 = thin N, 0 = neither
int odin_thin_dense_kind(int B, int K, int N) {
  if (ODIN_DIAG_ENV("ODIN_NOTHINDENSE")) return 0;
  if (B < 1 || B > (1 << 20)) return 0;
  if (K >= 1 && K <= TD_KMAX && N >= 64 && (N & 3) == 0 && (long)B * N < (1L << 29) && !odin_tiny_dense_ok(B, K, N)) return 1;
  if (N >= 1 && N <= TD_NMAX && K >= 64 && (K & 3) == 0 && (long)B * K < (1L << 29) && !odin_tiny_dense_ok(B, K, N)) return 2;
  return 0;
}
// weight-gradient slab rows (row chunks of the batch): 0 when the weight gradient is not served here
int odin_thin_dense_wgrad_rows(int B, int K, int N) {
  const int kind = odin_thin_dense_kind(B, K, N);
  if (kind == 0 || (kind == 1 && K > TD_KWMAX)) return 0;
  int rows = (B + 7) / 8;              // >= 8 rows per chunk (one round of loads), at most 32 chunks
  if (rows > 32) rows = 32;
  return rows < 1 ? 1 : rows;
}

int odin_thin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                        uint32_t* y_amax, void* stream) {
  const int kind = odin_thin_dense_kind(B, K, N);
  TDParams p;
  td_fill(p, B, K, N);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.act = act; p.amax = y_amax;
  if (kind == 1) {
    if (!td_al16(w) || !td_al16(y) || (bias != nullptr && !td_al16(bias))) return odin_fail(-2, "thin_dense_fwd: w / bias / y must be 16-byte aligned");
    const dim3 grid((N + 1023) / 1024, (B + 3) / 4);
    if (K <= 8) ODIN_LAUNCH((think_fwd_kernel<4, 8>), grid, dim3(256), 0, stream, p);
    else if (K <= 16) ODIN_LAUNCH((think_fwd_kernel<4, 16>), grid, dim3(256), 0, stream, p);
    else ODIN_LAUNCH((think_fwd_kernel<4, 32>), grid, dim3(256), 0, stream, p);
  } else {
    if (!td_al16(x)) return odin_fail(-2, "thin_dense_fwd: x must be 16-byte aligned");
    ODIN_LAUNCH(thinn_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, p);
  }
  return odin_check_launch("thin_dense_fwd");
}

int odin_thin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx, int B, int K, int N,
                          uint32_t* dx_amax, void* stream) {
  const int kind = odin_thin_dense_kind(B, K, N);
  TDParams p;
  td_fill(p, B, K, N);
  p.dy = dy; p.w = w; p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act; p.dx = dx;
  p.amax = dx_amax;
  if (kind == 1) {
    if (!td_al16(w) || !td_al16(dy)) return odin_fail(-2, "thin_dense_dgrad: w / dy must be 16-byte aligned");
    const dim3 grid((B + 3) / 4);
    if (K <= 8) ODIN_LAUNCH((think_dgrad_kernel<8>), grid, dim3(256), 0, stream, p);
    else if (K <= 16) ODIN_LAUNCH((think_dgrad_kernel<16>), grid, dim3(256), 0, stream, p);
    else ODIN_LAUNCH((think_dgrad_kernel<32>), grid, dim3(256), 0, stream, p);
  } else {
    if (!td_al16(dx) || (p.aux != nullptr && !td_al16(p.aux))) return odin_fail(-2, "thin_dense_dgrad: dx / aux must be 16-byte aligned");
    const long units = (long)B * (K >> 2);
    ODIN_LAUNCH(thinn_dgrad_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, stream, p);
  }
  return odin_check_launch("thin_dense_dgrad");
}

int odin_thin_dense_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N, void* stream) {
  const int kind = odin_thin_dense_kind(B, K, N);
  const int rows = odin_thin_dense_wgrad_rows(B, K, N);
  if (rows <= 0) return odin_fail(-2, "thin_dense_wgrad: shape not served");
  TDParams p;
  td_fill(p, B, K, N);
  p.x = x; p.dy = dy; p.slab = slab;
  p.chunk = (B + rows - 1) / rows;
  if (kind == 1) {
    if (!td_al16(dy) || !td_al16(slab) || (((size_t)K * N + N) & 3) != 0) return odin_fail(-2, "thin_dense_wgrad: dy / slab must be 16-byte aligned");
    const dim3 grid((N + 1023) / 1024, rows);
    if (K <= 8) ODIN_LAUNCH((think_wgrad_kernel<8>), grid, dim3(256), (size_t)p.chunk * 8 * 4, stream, p);
    else ODIN_LAUNCH((think_wgrad_kernel<16>), grid, dim3(256), (size_t)p.chunk * 16 * 4, stream, p);
  } else {
    if (!td_al16(x)) return odin_fail(-2, "thin_dense_wgrad: x must be 16-byte aligned");
    const dim3 grid((K + 1023) / 1024, rows);
    ODIN_LAUNCH(thinn_wgrad_kernel, grid, dim3(256), (size_t)p.chunk * N * 4, stream, p);
  }
  return odin_check_launch("thin_dense_wgrad");
}

// workgroups (= slab rows of the head's weight gradient) of odin_disc_head_fwd_bwd; 0: shapes outside its regime
extern "C" int odin_disc_head_rows(int B, int K) {
  if (B < 1 || K < 64 || (K & 3) != 0 || K > 2048) return 0;
  const int rows = (B + 7) / 8;
  return rows <= odin_max_slab_rows() ? rows : 0;
}

extern "C" int odin_disc_head_fwd_bwd(const float* h, const float* w, const float* bias, float* logit, int mode,
                                      const float* dlogit_in, float* dlogit_out, float* out, int aux_act, float* dh,
                                      uint32_t* dh_amax, float* wslab, int* rows_out, void* workspace, int B, int K,
                                      void* stream) {
  const int rows = odin_disc_head_rows(B, K);
  if (rows_out) *rows_out = rows;
  if (rows == 0 || rows > 1023) return odin_fail(-2, "disc_head: shapes outside the fused regime (K % 4 == 0, 64 <= K <= 2048)");
  if (mode != 0 && mode != 1) return odin_fail(-2, "disc_head: mode");
  if (mode == 1 && (B & 1)) return odin_fail(-2, "disc_head: dtc_loss wants an even number of rows ([z ; z_perm])");
  if (mode == 0 && dlogit_in == nullptr) return odin_fail(-2, "disc_head: mode 0 reads dlogit_in");
  if (!td_al16(h) || !td_al16(w) || (dh != nullptr && !td_al16(dh)) || (((size_t)workspace) & 7) != 0)
    return odin_fail(-2, "disc_head: h / w / dh must be 16-byte aligned");
  DHeadParams p;
  memset(&p, 0, sizeof(p));
  p.h = h; p.w = w; p.bias = bias; p.logit = logit; p.dlogit_in = dlogit_in; p.dlogit_out = dlogit_out; p.out = out;
  p.dh = dh; p.dh_amax = dh != nullptr ? dh_amax : nullptr; p.slab = dh != nullptr ? wslab : nullptr;
  p.ws = reinterpret_cast<unsigned long long*>(workspace);
  p.B = B; p.K = K; p.mode = mode; p.aux_act = aux_act;
  if (K <= 1024) ODIN_LAUNCH((disc_head_kernel<1>), dim3(rows), dim3(256), 0, stream, p);
  else ODIN_LAUNCH((disc_head_kernel<2>), dim3(rows), dim3(256), 0, stream, p);
  return odin_check_launch("disc_head");
}
