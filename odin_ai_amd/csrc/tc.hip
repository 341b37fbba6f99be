// tc.hip -- batch-coupled regularisers: beta-TCVAE total correlation (forward+backward,
// the [B,B,D] tensor lives only in LDS, one row at a time), FactorVAE permute_dims and the
// discriminator's dtc_loss.
//   total_correlation : odin/bay/vi/losses.py:101-157 (used by beta_vae.py:123-129)
//   permute_dims      : odin/bay/vi/utils.py:233-269
//   dtc_loss          : odin/bay/vi/autoencoder/factor_discriminator.py:200-235
#include "odin_device.h"
#include "odin_internal.h"

namespace {

constexpr float LOG2PI_F = 1.8378770664093453f;

__device__ __forceinline__ float softplus_t(float x) {
  return fmaxf(x, 0.f) + log1pf(odin_exp(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_t(float x) {
  float e = odin_exp(-fabsf(x));
  float s = 1.f / (1.f + e);
  return x >= 0.f ? s : e * s;
}
__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// ---- round 3: the estimator on precomputed, TRANSPOSED posterior parameters.  Round 2 evaluated
// softplus(raw), its logarithm and a division for every one of the B x B x D pairs, three times over, from
// [i][2D]-strided loads: 240 + 163 us at CelebA size (B = 512, D = 45), a fifth of the beta-TCVAE step, for
// ~36 M exponentials that fit in a few microseconds.  tc_prep_kernel evaluates them once per (i, l) and lays
// mu / 1/sigma / log sigma + log(2 pi)/2 out as [l][i] (and z, L as [l][j]) so that a wave's lanes read
// consecutive i (or j).
__global__ __launch_bounds__(256) void tc_prep_kernel(const float* z, const float* p, float* muT, float* isgT,
                                                      float* lsgT, float* zT, int Bj, int Bi, int D) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < Bi * D) {
    const int i = e / D, l = e - i * D;
    const float sg = softplus_t(p[(size_t)i * 2 * D + D + l]);
    muT[(size_t)l * Bi + i] = p[(size_t)i * 2 * D + l];
    isgT[(size_t)l * Bi + i] = 1.f / sg;
    lsgT[(size_t)l * Bi + i] = odin_log(sg) + 0.5f * LOG2PI_F;
  }
  if (e < Bj * D) {
    const int j = e / D, l = e - j * D;
    zT[(size_t)l * Bj + j] = z[e];
  }
}

// One workgroup per sample j (gridDim.x rows: the LOCAL shard under data parallelism); B = number
// of posteriors i the log-sum-exps run over (the GLOBAL batch).  LDS: lp[D][B] (log q(z_j | x_i)
// per latent), S[B], wS[B].  Outputs: logqz[j], LT[l][j] (log-sum-exp over i per latent), tc_part[j],
// dz[j][l].
// (1024 threads at B >= 512: the [D][B] row block in LDS -- 92 KB at CelebA size -- allows one workgroup per CU, so
// the waves that cover its latencies must come from inside it: 256 threads = one wave per SIMD took 79 us, 1024 take ~35)
__global__ __launch_bounds__(1024) void tc_rows_kernel(const float* zT, const float* muT, const float* isgT,
                                                      const float* lsgT, float* logqz, float* LT,
                                                      float* tc_part, float* dz, const float* coef, int B,
                                                      int D, int Bj) {
  ODIN_DYN_SMEM(float, smem);
  float* lp = smem;                 // [D][B]
  float* S = smem + (size_t)B * D;  // [B]
  float* wS = S + B;                // [B]
  float* Ll = wS + B;               // [D]
  float* zj = Ll + D;               // [D]
  float* misc = zj + D;             // [8]
  const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = (int)blockDim.x, NWV = NT >> 6;
  for (int l = tid; l < D; l += NT) zj[l] = zT[(size_t)l * Bj + j];
  __syncthreads();
  for (int i = tid; i < B; i += NT) {
    float s = 0.f;
    for (int l = 0; l < D; ++l) {
      const float d = (zj[l] - muT[(size_t)l * B + i]) * isgT[(size_t)l * B + i];
      const float v = -0.5f * d * d - lsgT[(size_t)l * B + i];
      lp[(size_t)l * B + i] = v;
      s += v;
    }
    S[i] = s;
  }
  __syncthreads();
  // per-latent log-sum-exp over i: wave w handles l = w, w+4, ...; index D = the joint S
  for (int l = wave; l <= D; l += NWV) {
    const float* row = l < D ? lp + (size_t)l * B : S;
    float mx = -3.0e38f;
    for (int i = lane; i < B; i += 64) mx = fmaxf(mx, row[i]);
    mx = wave_max64(mx);
    float sm = 0.f;
    for (int i = lane; i < B; i += 64) sm += odin_exp(row[i] - mx);
    sm = wave_sum64(sm);
    if (lane == 0) {
      float lse = mx + odin_log(sm);
      if (l < D) Ll[l] = lse; else misc[0] = lse;
    }
  }
  __syncthreads();
  const float lq = misc[0];
  if (tid == 0) {
    float t = 0.f;
    for (int l = 0; l < D; ++l) t += Ll[l];
    tc_part[j] = lq - t;
    logqz[j] = lq;
  }
  for (int l = tid; l < D; l += NT) LT[(size_t)l * Bj + j] = Ll[l];
  for (int i = tid; i < B; i += NT) wS[i] = odin_exp(S[i] - lq);
  __syncthreads();
  // dz[j,l] = coef/B * sum_i (wj[i] - wl[i,l]) * (-(z_j - mu_i)/sg_i^2)
  const float cf = coef[0] / (float)B;
  for (int l = wave; l < D; l += NWV) {
    float acc = 0.f;
    const float zz = zj[l], ll = Ll[l];
    for (int i = lane; i < B; i += 64) {
      const float is = isgT[(size_t)l * B + i];
      const float g = wS[i] - odin_exp(lp[(size_t)l * B + i] - ll);
      acc += g * (-(zz - muT[(size_t)l * B + i]) * is * is);
    }
    acc = wave_sum64(acc);
    if (lane == 0) dz[(size_t)j * D + l] = cf * acc;
  }
}

// One workgroup per posterior i (gridDim.x = global batch): sums over the Bj rows j this rank
// evaluated (all of them on one GPU; the local shard under data parallelism, where the partial
// sums of the ranks are then reduce-scattered).  Bn = global batch (the estimator's 1/B).
// LDS: Sw[Bj] (= w_joint[j, i]), the D parameters of posterior i.
__global__ __launch_bounds__(1024) void tc_cols_kernel(const float* zT, const float* muT, const float* isgT,
                                                      const float* lsgT, const float* logqz,
                                                      const float* LT, float* dloc, float* dscale,
                                                      const float* coef, int B, int D, int Bn) {
  ODIN_DYN_SMEM(float, smem);
  float* Sw = smem;          // [B]
  float* mu = Sw + B;        // [D]
  float* is = mu + D;        // [D]
  float* ls = is + D;        // [D]
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NT = (int)blockDim.x, NWV = NT >> 6;
  for (int l = tid; l < D; l += NT) {
    mu[l] = muT[(size_t)l * Bn + i];
    is[l] = isgT[(size_t)l * Bn + i];
    ls[l] = lsgT[(size_t)l * Bn + i];
  }
  __syncthreads();
  for (int j = tid; j < B; j += NT) {
    float s = 0.f;
    for (int l = 0; l < D; ++l) {
      const float d = (zT[(size_t)l * B + j] - mu[l]) * is[l];
      s += -0.5f * d * d - ls[l];
    }
    Sw[j] = odin_exp(s - logqz[j]);  // = wj[j, i]
  }
  __syncthreads();
  const float cf = coef[0] / (float)Bn;
  for (int l = wave; l < D; l += NWV) {
    const float m = mu[l], s1 = is[l], lsg = ls[l];
    float a1 = 0.f, a2 = 0.f;
    for (int j = lane; j < B; j += 64) {
      const float d = (zT[(size_t)l * B + j] - m) * s1;
      const float v = -0.5f * d * d - lsg;
      const float g = Sw[j] - odin_exp(v - LT[(size_t)l * B + j]);
      a1 += g * d * s1;
      a2 += g * (d * d - 1.f) * s1;
    }
    a1 = wave_sum64(a1);
    a2 = wave_sum64(a2);
    if (lane == 0) {
      dloc[(size_t)i * D + l] = cf * a1;
      dscale[(size_t)i * D + l] = cf * a2;
    }
  }
}

__global__ __launch_bounds__(256) void sum_div_kernel(const float* part, int n, float* out, float div) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / div;
}

__global__ __launch_bounds__(256) void mean_kernel(const float* part, int n, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += part[i];
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

__global__ __launch_bounds__(256) void permute_kernel(const float* z, const int* perm, float* out,
                                                      int B, int D) {
  int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * D) return;
  int l = e % D;
  out[e] = z[(size_t)perm[e] * D + l];
}

__device__ __forceinline__ unsigned hash3(unsigned a, unsigned b, unsigned c) {
  unsigned h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// one workgroup per latent column: random keys -> rank -> permutation (B <= 4096)
// (z != null: out[rank][l] = z[i][l] in the same pass -- permute_dims without a launch of its own)
__global__ __launch_bounds__(256) void random_perm_kernel(int* perm, int B, int D, unsigned k0,
                                                          unsigned k1, const int* step_dev, const float* z, float* out) {
  ODIN_DYN_SMEM(unsigned, keys);
  const int l = blockIdx.x;
  const unsigned step = step_dev ? (unsigned)step_dev[0] : 0u;
  for (int i = threadIdx.x; i < B; i += 256)
    keys[i] = hash3(hash3(k0, k1, step), (unsigned)l, (unsigned)i);
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += 256) {
    unsigned k = keys[i];
    int rank = 0;
    for (int j = 0; j < B; ++j) {
      unsigned kj = keys[j];
      rank += (kj < k || (kj == k && j < i)) ? 1 : 0;
    }
    perm[(size_t)rank * D + l] = i;
    if (z != nullptr) out[(size_t)rank * D + l] = z[(size_t)i * D + l];
  }
}

__global__ __launch_bounds__(256) void dtc_loss_kernel(const float* lz, const float* lp, float* out,
                                                       float* dlz, float* dlp, int n) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 0.5f / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    float a = lz[i], b = lp[i];
    acc += softplus_t(-a) + softplus_t(b);
    dlz[i] = -sigmoid_t(-a) * inv;
    dlp[i] = sigmoid_t(b) * inv;
  }
  acc = wave_sum64(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv;
}

}  // namespace

// workspace layout inside tc_out: [0] = TC | logqz [Bj] | LT [D][Bj] | tc_part [Bj] | muT, isgT, lsgT [D][Bi] each |
// zT [D][Bj]  =  odin_total_correlation_workspace(Bj, Bi, D) floats
extern "C" int odin_total_correlation_workspace(int B_local, int B_global, int D) {
  return 1 + B_local * (2 * D + 2) + 3 * B_global * D;
}

static int tc_launch(const float* z, const float* p, float* tc_out, float* dz, float* dloc,
                     float* dscale, const float* coef, int Bj, int Bi, int D, void* stream) {
  size_t lds = ((size_t)Bi * D + 2 * Bi + 2 * D + 8) * 4;
  if (lds > 158 * 1024) return odin_fail(-2, "total_correlation: B*D too large for LDS");
  if (Bj > 4096 || Bi > 4096) return odin_fail(-2, "total_correlation: B > 4096");
  float* logqz = tc_out + 1;
  float* LT = logqz + Bj;
  float* part = LT + (size_t)Bj * D;
  float* muT = part + Bj;
  float* isgT = muT + (size_t)Bi * D;
  float* lsgT = isgT + (size_t)Bi * D;
  float* zT = lsgT + (size_t)Bi * D;
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tc_rows_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  const int nmax = (Bi > Bj ? Bi : Bj) * D;
  ODIN_LAUNCH(tc_prep_kernel, dim3((nmax + 255) / 256), dim3(256), 0, stream, z, p, muT, isgT, lsgT, zT, Bj, Bi, D);
  // rows: coefficient coef/Bi inside (cf = coef[0] / B with B = Bi)
  const int nt_rows = Bi >= 512 ? 1024 : Bi >= 256 ? 512 : 256;   // threads ~ posteriors per row, <= 1024
  const int nt_cols = Bj >= 512 ? 1024 : Bj >= 256 ? 512 : 256;
  ODIN_LAUNCH(tc_rows_kernel, dim3(Bj), dim3(nt_rows), lds, stream, (const float*)zT, (const float*)muT,
              (const float*)isgT, (const float*)lsgT, logqz, LT, part, dz, coef, Bi, D, Bj);
  ODIN_LAUNCH(tc_cols_kernel, dim3(Bi), dim3(nt_cols), (size_t)(Bj + 3 * D) * 4, stream, (const float*)zT,
              (const float*)muT, (const float*)isgT, (const float*)lsgT, (const float*)logqz, (const float*)LT,
              dloc, dscale, coef, Bj, D, Bi);
  ODIN_LAUNCH(sum_div_kernel, dim3(1), dim3(256), 0, stream, (const float*)part, Bj, tc_out, (float)Bi);
  return odin_check_launch("total_correlation");
}

extern "C" int odin_total_correlation_fwd_bwd(const float* z, const float* p, float* tc_out,
                                              float* dz, float* dloc, float* dscale,
                                              const float* coef, int B, int D, void* stream) {
  return tc_launch(z, p, tc_out, dz, dloc, dscale, coef, B, B, D, stream);
}

extern "C" int odin_total_correlation_shard(const float* z_local, const float* p_global,
                                            float* tc_out, float* dz_local, float* dloc_part,
                                            float* dscale_part, const float* coef, int B_local,
                                            int B_global, int D, void* stream) {
  return tc_launch(z_local, p_global, tc_out, dz_local, dloc_part, dscale_part, coef, B_local,
                   B_global, D, stream);
}

extern "C" int odin_mean(const float* x, int n, float* out, void* stream) {
  ODIN_LAUNCH(mean_kernel, dim3(1), dim3(256), 0, stream, x, n, out);
  return odin_check_launch("mean");
}

extern "C" int odin_permute_dims(const float* z, const int32_t* perm, float* out, int B, int D,
                                 void* stream) {
  ODIN_LAUNCH(permute_kernel, dim3((B * D + 255) / 256), dim3(256), 0, stream, z, (const int*)perm,
              out, B, D);
  return odin_check_launch("permute_dims");
}

extern "C" int odin_random_perm(int32_t* perm, int B, int D, uint64_t seed,
                                const int32_t* step_dev, void* stream) {
  if (B > 16384) return odin_fail(-2, "random_perm: B too large");
  ODIN_LAUNCH(random_perm_kernel, dim3(D), dim3(256), (size_t)B * 4, stream, (int*)perm, B, D,
              (unsigned)seed, (unsigned)(seed >> 32), (const int*)step_dev, (const float*)nullptr, (float*)nullptr);
  return odin_check_launch("random_perm");
}

// odin_random_perm + odin_permute_dims(z, perm, out) in one launch (the rows of z are all local: one GPU)
extern "C" int odin_random_permute_dims(int32_t* perm, const float* z, float* out, int B, int D, uint64_t seed,
                                        const int32_t* step_dev, void* stream) {
  if (B > 16384) return odin_fail(-2, "random_permute_dims: B too large");
  if (z == nullptr || out == nullptr) return odin_fail(-2, "random_permute_dims: z / out");
  ODIN_LAUNCH(random_perm_kernel, dim3(D), dim3(256), (size_t)B * 4, stream, (int*)perm, B, D,
              (unsigned)seed, (unsigned)(seed >> 32), (const int*)step_dev, z, out);
  return odin_check_launch("random_permute_dims");
}

extern "C" int odin_dtc_loss_fwd_bwd(const float* logit_z, const float* logit_perm, float* out,
                                     float* dlogit_z, float* dlogit_perm, int n, void* stream) {
  ODIN_LAUNCH(dtc_loss_kernel, dim3(1), dim3(256), 0, stream, logit_z, logit_perm, out, dlogit_z,
              dlogit_perm, n);
  return odin_check_launch("dtc_loss");
}
