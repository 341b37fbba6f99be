// latent_block.hip -- the bottleneck of the VAE step as ONE launch per direction.
//
// Between the encoder's projection and the decoder's first Conv2DTranspose the reference executes
// DistributionDense (odin/bay/layers/dense_distribution.py: Dense(P -> 2D) + MVNDiag(loc, softplus(raw))),
// the reparameterised sample and its KL term (variational_autoencoder.py:515-542, helpers.py:236-286) and the
// decoder's projection Dense(D -> N0) (image_networks.py:494-497).  Each of them is ~0.1 MFLOP at batch 256:
// as separate launches (rng_normal, tiny_dense, latent_fwd, tiny_dense / tiny_dense, latent_bwd, 2 x tiny
// weight gradients, tiny_dense) they cost 4-13 us apiece -- launch floor plus one cold-L2 round trip per
// dependent load -- about 65 us of a 0.8 ms step.  Here a workgroup owns S samples, pulls both weight
// matrices and its S rows into LDS with one round of loads, and walks the whole chain in registers / LDS:
//
//   forward :  p = h Wl + bl;  eps ~ Philox (the stream odin_rng_normal writes);  z = loc + softplus(raw) eps;
//              kl_b, free-bits mask;  y0 = act(z W0 + b0)
//   backward:  dz = g0 W0^T;  dp = d(KL weight * kl + decoder + extra terms)/dp;  dh = (dp Wl^T) act'(h);
//              per-workgroup partial (dW0 | db0) and (dWl | dbl) rows for odin_slab_reduce
//
// Every sum runs in a fixed order: bit-reproducible like the rest of the step.
#include "odin_device.h"
#include "odin_internal.h"
#include "odin_latent_math.h"
#include <cstdlib>
#include <cstdint>

namespace {

struct LBFwd {
  const float* h;       // [B, P] encoder output
  const float* wl;      // [P, 2D]
  const float* bl;      // [2D]
  const float* eps_in;  // [B, D] or null: draw from Philox(seed, step)
  float* eps;           // [B, D] written when drawn here
  float* p;             // [B, 2D]
  float* z;             // [B, D]
  float* kl;            // [B]
  float* fbmask;        // [B]
  const float* w0;      // [D, N0]
  const float* b0;      // [N0]
  float* y0;            // [B, N0]
  const int* step_dev;
  unsigned k0, k1;
  int B, P, D, N0, act0, analytic, S;
  float free_bits;
};

// LDS (floats): wl [P * 2D] | w0 [D * N0] | hs [S * P] | ps [S * 2D] | zs [S * D] | red [256]
__global__ __launch_bounds__(256) void latent_block_fwd_kernel(LBFwd q) {
  ODIN_DYN_SMEM(float, sm);
  const int P = q.P, D = q.D, J = 2 * q.D, N0 = q.N0, S = q.S;
  float* wl = sm;
  float* w0 = wl + P * J;
  float* hs = w0 + D * N0;
  float* ps = hs + S * P;
  float* zs = ps + S * J;
  float* red = zs + S * D;
  const int tid = threadIdx.x, b0 = blockIdx.x * S;
  const int ns = (q.B - b0 < S) ? q.B - b0 : S;
  // ---- one round of loads: both weight matrices and this workgroup's rows ----
  for (int e = tid; e < P * J; e += 256) wl[e] = q.wl[e];
  for (int e = tid; e < D * N0; e += 256) w0[e] = q.w0[e];
  for (int e = tid; e < S * P; e += 256) hs[e] = (e < ns * P) ? q.h[(size_t)b0 * P + e] : 0.f;
  // (the noise: element f of the [B, D] stream is component f & 3 of counter f >> 2)
  float my_eps = 0.f;
  if (tid < S * D) {
    const int s = tid / D;
    const unsigned f = (unsigned)(b0 * D + tid);
    if (s < ns) {
      if (q.eps_in != nullptr) {
        my_eps = q.eps_in[f];
      } else {
        float v[4];
        odin_normal4(f >> 2, 0u, q.step_dev ? (unsigned)q.step_dev[0] : 0u, q.k0, q.k1, v);
        my_eps = v[f & 3];
        q.eps[f] = my_eps;
      }
    }
  }
  __syncthreads();
  // ---- p[s][j] = sum_k h[s][k] wl[k][j] + bl[j]: S * 2D outputs, the 256 threads split k ----
  {
    const int nout = S * J;
    const int ks = 256 / nout;  // >= 1 by the launch geometry
    const int kc = (P + ks - 1) / ks;
    const int o = tid % nout, kq = tid / nout;
    if (kq < ks) {
      const int s = o / J, j = o - s * J;
      const int klo = kq * kc, khi = (klo + kc < P) ? klo + kc : P;
      float a0 = 0.f, a1 = 0.f;
      int k = klo;
      for (; k + 1 < khi; k += 2) {
        a0 = fmaf(hs[s * P + k], wl[k * J + j], a0);
        a1 = fmaf(hs[s * P + k + 1], wl[(k + 1) * J + j], a1);
      }
      if (k < khi) a0 = fmaf(hs[s * P + k], wl[k * J + j], a0);
      red[kq * nout + o] = a0 + a1;
    }
    __syncthreads();
    if (tid < nout) {
      const int s = tid / J, j = tid - s * J;
      float a = red[tid];
      for (int w = 1; w < ks; ++w) a += red[w * nout + tid];
      a += q.bl[j];
      ps[tid] = a;
      if (s < ns) q.p[(size_t)(b0 + s) * J + j] = a;
    }
    __syncthreads();
  }
  // ---- reparameterise + KL (latent_fwd_kernel's arithmetic, summed over d in order) ----
  if (tid < S * D) {
    const int s = tid / D, d = tid - s * D;
    const float loc = ps[s * J + d], sc = softplus_f(ps[s * J + D + d]), e = my_eps;
    const float zz = loc + sc * e;
    zs[tid] = zz;
    if (s < ns) q.z[(size_t)(b0 + s) * D + d] = zz;
    const float ls = odin_log(sc);
    float t;
    if (q.analytic == 2) t = ls + 0.5f * (1.f + loc * loc) / (sc * sc) - 0.5f;
    else if (q.analytic) t = 0.5f * (sc * sc + loc * loc - 1.f) - ls;
    else t = 0.5f * (zz * zz - e * e) - ls;
    red[tid] = t;
  }
  __syncthreads();
  if (tid < ns) {
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc += red[tid * D + d];
    float m = 1.f;
    if (q.free_bits >= 0.f) {
      const float thr = q.free_bits * (float)D;
      if (!(acc > thr)) { acc = thr; m = 0.f; }
    }
    q.kl[b0 + tid] = acc;
    q.fbmask[b0 + tid] = m;
  }
  // ---- y0[s][n] = act(sum_d z[s][d] w0[d][n] + b0[n]) ----
  for (int o = tid; o < ns * N0; o += 256) {
    const int s = o / N0, n = o - s * N0;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) acc = fmaf(zs[s * D + d], w0[d * N0 + n], acc);
    q.y0[(size_t)(b0 + s) * N0 + n] = odin_act(q.act0, acc + q.b0[n]);
  }
}

struct LBBwd {
  const float* g0;       // [B, N0] dL/d(pre-activation of the decoder's first Dense)
  const float* w0;       // [D, N0]
  const float* z; const float* p; const float* eps; const float* fbmask; const float* klw;
  const float* dz2; const float* dloc_x; const float* dscale_x;   // optional extra terms (FactorVAE, TC)
  const float* wl;       // [P, 2D]
  const float* h;        // [B, P] encoder output (its activation's derivative is taken from it)
  float* dz; float* dp; float* dh;
  float* slab0;          // [gridDim.x][D * N0 + N0]
  float* slabl;          // [gridDim.x][P * 2D + 2D]
  int B, P, D, N0, h_act, analytic, S;
};

// LDS (floats): wl [P * 2D] | w0 [D * N0] | hs [S * P] | gs [S * N0] | zs [S * D] | dps [S * 2D] | red [256]
__global__ __launch_bounds__(256) void latent_block_bwd_kernel(LBBwd q) {
  ODIN_DYN_SMEM(float, sm);
  const int P = q.P, D = q.D, J = 2 * q.D, N0 = q.N0, S = q.S;
  float* wl = sm;
  float* w0 = wl + P * J;
  float* hs = w0 + D * N0;
  float* gs = hs + S * P;
  float* zs = gs + S * N0;
  float* dps = zs + S * D;
  float* red = dps + S * J;
  const int tid = threadIdx.x, b0 = blockIdx.x * S;
  const int ns = (q.B - b0 < S) ? q.B - b0 : S;
  for (int e = tid; e < P * J; e += 256) wl[e] = q.wl[e];
  for (int e = tid; e < D * N0; e += 256) w0[e] = q.w0[e];
  for (int e = tid; e < S * P; e += 256) hs[e] = (e < ns * P) ? q.h[(size_t)b0 * P + e] : 0.f;
  for (int e = tid; e < S * N0; e += 256) gs[e] = (e < ns * N0) ? q.g0[(size_t)b0 * N0 + e] : 0.f;
  for (int e = tid; e < S * D; e += 256) zs[e] = (e < ns * D) ? q.z[(size_t)b0 * D + e] : 0.f;
  // per-(s, d) operands of the latent backward, fetched in the same round
  float loc = 0.f, raw = 0.f, e_ = 0.f, fbm = 0.f, x2 = 0.f, xl = 0.f, xs = 0.f;
  const bool mine = tid < S * D && tid / D < ns;
  if (mine) {
    const int s = tid / D, d = tid - s * D;
    const size_t i = (size_t)(b0 + s) * D + d;
    loc = q.p[(size_t)(b0 + s) * J + d];
    raw = q.p[(size_t)(b0 + s) * J + D + d];
    e_ = q.eps[i];
    fbm = q.fbmask[b0 + s];
    if (q.dz2 != nullptr) x2 = q.dz2[i];
    if (q.dloc_x != nullptr) xl = q.dloc_x[i];
    if (q.dscale_x != nullptr) xs = q.dscale_x[i];
  }
  const float klw = q.klw[0];
  __syncthreads();
  // ---- dz[s][d] = sum_n g0[s][n] w0[d][n]: S * D outputs, the threads split n ----
  {
    const int nout = S * D;
    const int ks = 256 / nout;
    const int kc = (N0 + ks - 1) / ks;
    const int o = tid % nout, kq = tid / nout;
    if (kq < ks) {
      const int s = o / D, d = o - s * D;
      const int nlo = kq * kc, nhi = (nlo + kc < N0) ? nlo + kc : N0;
      float a0 = 0.f, a1 = 0.f;
      int n = nlo;
      for (; n + 1 < nhi; n += 2) {
        a0 = fmaf(gs[s * N0 + n], w0[d * N0 + n], a0);
        a1 = fmaf(gs[s * N0 + n + 1], w0[d * N0 + n + 1], a1);
      }
      if (n < nhi) a0 = fmaf(gs[s * N0 + n], w0[d * N0 + n], a0);
      red[kq * nout + o] = a0 + a1;
    }
    __syncthreads();
    float g = 0.f;
    if (tid < nout) {
      g = red[tid];
      for (int w = 1; w < ks; ++w) g += red[w * nout + tid];
    }
    // ---- latent backward (latent_bwd_kernel's arithmetic) ----
    if (tid < nout) {
      const int s = tid / D, d = tid - s * D;
      const float sc = softplus_f(raw), zz = zs[tid];
      const float w = klw * fbm;
      float dloc, dsc;
      if (q.analytic == 2) {
        const float i2 = 1.f / (sc * sc);
        dloc = w * loc * i2;
        dsc = w * (1.f / sc - (1.f + loc * loc) * i2 / sc);
      } else if (q.analytic) { dloc = w * loc; dsc = w * (sc - 1.f / sc); }
      else { dloc = w * zz; dsc = w * (zz * e_ - 1.f / sc); }
      dloc += g; dsc += g * e_;
      if (q.dz2 != nullptr) { dloc += x2; dsc += x2 * e_; }
      if (q.dloc_x != nullptr) dloc += xl;
      if (q.dscale_x != nullptr) dsc += xs;
      const float draw = dsc * sigmoid_f(raw);
      const bool live = s < ns;
      dps[s * J + d] = live ? dloc : 0.f;
      dps[s * J + D + d] = live ? draw : 0.f;
      if (live) {
        q.dz[(size_t)(b0 + s) * D + d] = g;
        q.dp[(size_t)(b0 + s) * J + d] = dloc;
        q.dp[(size_t)(b0 + s) * J + D + d] = draw;
      }
    }
    __syncthreads();
  }
  // ---- dh[s][k] = (sum_j dp[s][j] wl[k][j]) act'(h[s][k]) ----
  for (int o = tid; o < ns * P; o += 256) {
    const int s = o / P, k = o - s * P;
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc = fmaf(dps[s * J + j], wl[k * J + j], acc);
    q.dh[(size_t)(b0 + s) * P + k] = acc * odin_act_grad(q.h_act, hs[o]);
  }
  // ---- this workgroup's partial weight gradients: sums over its S samples, s ascending ----
  {
    float* row = q.slab0 + (size_t)blockIdx.x * (D * N0 + N0);
    for (int o = tid; o < D * N0; o += 256) {
      const int d = o / N0, n = o - d * N0;
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc = fmaf(zs[s * D + d], gs[s * N0 + n], acc);
      row[o] = acc;
    }
    for (int n = tid; n < N0; n += 256) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += gs[s * N0 + n];
      row[D * N0 + n] = acc;
    }
  }
  {
    float* row = q.slabl + (size_t)blockIdx.x * (P * J + J);
    for (int o = tid; o < P * J; o += 256) {
      const int k = o / J, j = o - k * J;
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc = fmaf(hs[s * P + k], dps[s * J + j], acc);
      row[o] = acc;
    }
    for (int j = tid; j < J; j += 256) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += dps[s * J + j];
      row[P * J + j] = acc;
    }
  }
}

// samples per workgroup: the S * 2D outputs of the first product fit the 256 threads; 4-8 samples per
// workgroup keep the weight re-reads (one copy per workgroup) and the slab rows small
int lb_samples(int D) {
  int S = 256 / (2 * D);
  if (S > 8) S = 8;
  return S;
}

size_t lb_lds_floats(int P, int D, int N0, int S) {
  return (size_t)P * 2 * D + (size_t)D * N0 + (size_t)S * P + (size_t)S * N0 + (size_t)S * D + (size_t)S * 2 * D + 256;
}

template <typename K>
int lb_set_lds(K kern, size_t bytes) {
#ifndef ODIN_SIM
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess)
    return odin_fail(-4, "latent_block: cannot raise the dynamic LDS limit");
#else
  (void)kern; (void)bytes;
#endif
  return 0;
}

}  // namespace

// Number of workgroups (= slab rows of the backward launch); 0 when the shapes are outside the fused regime
// (both weight matrices + S rows must fit in LDS).
extern "C" int odin_latent_block_rows(int B, int P, int D, int N0) {
  if (getenv("ODIN_NOLATBLOCK")) return 0;
  if (B < 1 || P < 1 || D < 1 || N0 < 1 || 2 * D > 128) return 0;
  const int S = lb_samples(D);
  if (lb_lds_floats(P, D, N0, S) * 4 > 120 * 1024) return 0;
  const int rows = (B + S - 1) / S;
  return rows <= ODIN_MAX_COLSUM_BLOCKS ? rows : 0;
}

extern "C" int odin_latent_block_fwd(const float* h, const float* wl, const float* bl, const float* eps_in,
                                     float* eps_out, uint64_t seed, const int32_t* step_dev, float* p,
                                     float* z, float* kl, float* fbmask, const float* w0, const float* b0,
                                     float* y0, int B, int P, int D, int N0, int act0, int analytic,
                                     float free_bits, void* stream) {
  const int rows = odin_latent_block_rows(B, P, D, N0);
  if (rows == 0) return odin_fail(-2, "latent_block_fwd: shapes outside the fused regime");
  LBFwd q;
  memset(&q, 0, sizeof(q));
  q.h = h; q.wl = wl; q.bl = bl; q.eps_in = eps_in; q.eps = eps_out; q.p = p; q.z = z; q.kl = kl;
  q.fbmask = fbmask; q.w0 = w0; q.b0 = b0; q.y0 = y0; q.step_dev = (const int*)step_dev;
  q.k0 = (unsigned)seed; q.k1 = (unsigned)(seed >> 32);
  q.B = B; q.P = P; q.D = D; q.N0 = N0; q.act0 = act0; q.analytic = analytic; q.S = lb_samples(D);
  q.free_bits = free_bits;
  const size_t lds = lb_lds_floats(P, D, N0, q.S) * 4;
  if (int rc = lb_set_lds(&latent_block_fwd_kernel, lds)) return rc;
  ODIN_LAUNCH(latent_block_fwd_kernel, dim3(rows), dim3(256), lds, stream, q);
  return odin_check_launch("latent_block_fwd");
}

extern "C" int odin_latent_block_bwd(const float* g0, const float* w0, const float* z, const float* p,
                                     const float* eps, const float* fbmask, const float* klw,
                                     const float* dz_extra, const float* dloc_x, const float* dscale_x,
                                     const float* wl, const float* h, int h_act, float* dz, float* dp,
                                     float* dh, float* slab0, float* slabl, int B, int P, int D, int N0,
                                     int analytic, void* stream) {
  const int rows = odin_latent_block_rows(B, P, D, N0);
  if (rows == 0) return odin_fail(-2, "latent_block_bwd: shapes outside the fused regime");
  LBBwd q;
  memset(&q, 0, sizeof(q));
  q.g0 = g0; q.w0 = w0; q.z = z; q.p = p; q.eps = eps; q.fbmask = fbmask; q.klw = klw;
  q.dz2 = dz_extra; q.dloc_x = dloc_x; q.dscale_x = dscale_x; q.wl = wl; q.h = h; q.h_act = h_act;
  q.dz = dz; q.dp = dp; q.dh = dh; q.slab0 = slab0; q.slabl = slabl;
  q.B = B; q.P = P; q.D = D; q.N0 = N0; q.analytic = analytic; q.S = lb_samples(D);
  const size_t lds = lb_lds_floats(P, D, N0, q.S) * 4;
  if (int rc = lb_set_lds(&latent_block_bwd_kernel, lds)) return rc;
  ODIN_LAUNCH(latent_block_bwd_kernel, dim3(rows), dim3(256), lds, stream, q);
  return odin_check_launch("latent_block_bwd");
}
