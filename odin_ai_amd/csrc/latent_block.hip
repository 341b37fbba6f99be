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

// One round of loads for everything a workgroup needs: every array is issued into registers first (masked
// elements load element 0 instead -- conditional loads make the compiler drain vmcnt), the LDS stores follow.
// (One copy loop per array waits a cold-L2 round trip per array AND per iteration: 13-19 us for these kernels.)
// A per-sample segment: `n` floats valid in the source, `cap` <= 256 * NL floats written (zeros beyond n).
template <int NL>
struct LBSmall {
  float r[NL];
  __device__ __forceinline__ void issue(const float* src, int n, int tid) {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = u * 256 + tid;
      r[u] = src[e < n ? e : 0];
    }
  }
  __device__ __forceinline__ void commit(float* dst, int n, int cap, int tid) const {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = u * 256 + tid;
      if (e < cap) dst[e] = e < n ? r[u] : 0.f;
    }
  }
};

// The two weight matrices: float4 (when 16-byte aligned) or scalar loads, LB_DEPTH per thread in flight.  issue()
// of the first batch of both matrices precedes everything else, commit() follows the small segments' stores:
// one exposed round trip for the whole working set (longer matrices take further batches).
constexpr int LB_DEPTH = 8;
template <bool V4>
struct LBBig {
  float4 r[LB_DEPTH];
  __device__ __forceinline__ void issue(const float* src, int n, int e0, int tid) {
    const int units = V4 ? n >> 2 : n;
#pragma unroll
    for (int u = 0; u < LB_DEPTH; ++u) {
      const int e = e0 + u * 256 + tid;
      const int ec = e < units ? e : 0;
      if (V4) r[u] = reinterpret_cast<const float4*>(src)[ec];
      else r[u].x = src[ec];
    }
  }
  __device__ __forceinline__ void commit(float* dst, int n, int e0, int tid) const {
    const int units = V4 ? n >> 2 : n;
#pragma unroll
    for (int u = 0; u < LB_DEPTH; ++u) {
      const int e = e0 + u * 256 + tid;
      if (e < units) {
        if (V4) reinterpret_cast<float4*>(dst)[e] = r[u];
        else dst[e] = r[u].x;
      }
    }
  }
};
// the remaining batches of a long matrix
template <bool V4>
__device__ __forceinline__ void lb_big_rest(float* dst, const float* src, int n, int tid) {
  const int units = V4 ? n >> 2 : n;
  for (int e0 = 256 * LB_DEPTH; e0 < units; e0 += 256 * LB_DEPTH) {
    LBBig<V4> t;
    t.issue(src, n, e0, tid);
    t.commit(dst, n, e0, tid);
  }
}

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
  const float* cap;     // BetaCapacityVAE: device scalar C(step), kl <- |kl - C| (null: off)
};

// LDS (floats): wl [P * 2D] | w0 [D * N0] | hs [S * P] | bls [2D] | b0s [N0] | es [S * D] | ps [S * 2D] |
//               zs [S * D] | red [256]
template <int S, bool V4>
__global__ __launch_bounds__(256) void latent_block_fwd_kernel(LBFwd q) {
  ODIN_DYN_SMEM(float, sm);
  const int P = q.P, D = q.D, J = 2 * q.D, N0 = q.N0;
  const int o_wl = 0, o_w0 = o_wl + P * J, o_hs = o_w0 + D * N0, o_bl = o_hs + S * P, o_b0 = o_bl + J;
  const int o_es = o_b0 + N0, o_ps = o_es + S * D, o_zs = o_ps + S * J, o_red = o_zs + S * D;
  float *wl = sm + o_wl, *w0 = sm + o_w0, *hs = sm + o_hs, *bls = sm + o_bl, *b0s = sm + o_b0;
  float *es = sm + o_es, *ps = sm + o_ps, *zs = sm + o_zs, *red = sm + o_red;
  const int tid = threadIdx.x, b0 = blockIdx.x * S;
  const int ns = (q.B - b0 < S) ? q.B - b0 : S;
  const unsigned step = q.step_dev ? (unsigned)q.step_dev[0] : 0u;
  {
    LBBig<V4> rwl, rw0;
    LBSmall<8> rh, rb0;   // (N0 <= 2048)
    LBSmall<1> rbl, re;
    const bool have_eps = q.eps_in != nullptr;
    rwl.issue(q.wl, P * J, 0, tid);
    rw0.issue(q.w0, D * N0, 0, tid);
    rh.issue(q.h + (size_t)b0 * P, ns * P, tid);
    rbl.issue(q.bl, J, tid);
    rb0.issue(q.b0, N0, tid);
    re.issue(have_eps ? q.eps_in + (size_t)b0 * D : q.wl, have_eps ? ns * D : 0, tid);
    rh.commit(hs, ns * P, S * P, tid);
    rbl.commit(bls, J, J, tid);
    rb0.commit(b0s, N0, N0, tid);
    if (have_eps) re.commit(es, ns * D, S * D, tid);
    rwl.commit(wl, P * J, 0, tid);
    rw0.commit(w0, D * N0, 0, tid);
    lb_big_rest<V4>(wl, q.wl, P * J, tid);
    lb_big_rest<V4>(w0, q.w0, D * N0, tid);
  }
  // (drawn here: element f of the [B, D] stream is component f & 3 of counter f >> 2)
  if (q.eps_in == nullptr && tid < S * D) {
    const unsigned f = (unsigned)(b0 * D + tid);
    float v[4];
    odin_normal4(f >> 2, 0u, step, q.k0, q.k1, v);
    const bool live = tid < ns * D;
    const float e = live ? v[f & 3] : 0.f;
    es[tid] = e;
    if (live) q.eps[f] = e;
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
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = klo;
      for (; k + 3 < khi; k += 4) {
        a0 = fmaf(hs[s * P + k], wl[k * J + j], a0);
        a1 = fmaf(hs[s * P + k + 1], wl[(k + 1) * J + j], a1);
        a2 = fmaf(hs[s * P + k + 2], wl[(k + 2) * J + j], a2);
        a3 = fmaf(hs[s * P + k + 3], wl[(k + 3) * J + j], a3);
      }
      for (; k < khi; ++k) a0 = fmaf(hs[s * P + k], wl[k * J + j], a0);
      red[kq * nout + o] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (tid < nout) {
      const int s = tid / J, j = tid - s * J;
      float a = red[tid];
      for (int w = 1; w < ks; ++w) a += red[w * nout + tid];
      a += bls[j];
      ps[tid] = a;
      if (s < ns) q.p[(size_t)(b0 + s) * J + j] = a;
    }
    __syncthreads();
  }
  // ---- reparameterise + KL (latent_fwd_kernel's arithmetic, summed over d in order) ----
  if (tid < S * D) {
    const int s = tid / D, d = tid - s * D;
    const float loc = ps[s * J + d], sc = softplus_f(ps[s * J + D + d]), e = es[tid];
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
    if (q.cap != nullptr) {  // beta_vae.py:171-177: |kl - C(step)|, gradient sign(kl - C)
      const float d = acc - q.cap[0];
      m *= d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      acc = fabsf(d);
    }
    q.kl[b0 + tid] = acc;
    q.fbmask[b0 + tid] = m;
  }
  // ---- y0[s][n] = act(sum_d z[s][d] w0[d][n] + b0[n]) ----
  for (int o = tid; o < ns * N0; o += 256) {
    const int s = o / N0, n = o - s * N0;
    float acc = 0.f;
#pragma unroll 4
    for (int d = 0; d < D; ++d) acc = fmaf(zs[s * D + d], w0[d * N0 + n], acc);
    q.y0[(size_t)(b0 + s) * N0 + n] = odin_act(q.act0, acc + b0s[n]);
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
  unsigned* dh_amax;     // range word of dh (max |dh| folded in by every workgroup), may be null
};

// LDS (floats): wl [P * 2D] | w0 [D * N0] | hs [S * P] | gs [S * N0] | zs [S * D] | pls [S * 2D] | es [S * D] |
//               x2 [S * D] | xl [S * D] | xs [S * D] | fb [S] | dps [S * 2D] | red [256]
template <int S, bool V4>
__global__ __launch_bounds__(256) void latent_block_bwd_kernel(LBBwd q) {
  ODIN_DYN_SMEM(float, sm);
  const int P = q.P, D = q.D, J = 2 * q.D, N0 = q.N0;
  const int o_wl = 0, o_w0 = o_wl + P * J, o_hs = o_w0 + D * N0, o_gs = o_hs + S * P, o_zs = o_gs + S * N0;
  const int o_pl = o_zs + S * D, o_es = o_pl + S * J, o_x2 = o_es + S * D, o_xl = o_x2 + S * D;
  const int o_xs = o_xl + S * D, o_fb = o_xs + S * D, o_dp = o_fb + S, o_red = o_dp + S * J;
  float *wl = sm + o_wl, *w0 = sm + o_w0, *hs = sm + o_hs, *gs = sm + o_gs, *zs = sm + o_zs;
  float *pls = sm + o_pl, *es = sm + o_es, *x2 = sm + o_x2, *xl = sm + o_xl, *xs = sm + o_xs;
  float *fb = sm + o_fb, *dps = sm + o_dp, *red = sm + o_red;
  const int tid = threadIdx.x, b0 = blockIdx.x * S;
  const int ns = (q.B - b0 < S) ? q.B - b0 : S;
  const float klw = q.klw[0];
  {
    const size_t bd = (size_t)b0 * D;
    LBBig<V4> rwl, rw0;
    LBSmall<8> rh, rg;
    LBSmall<1> rz, rp, re, r2, rl, rs, rf;
    const bool h2 = q.dz2 != nullptr, hl = q.dloc_x != nullptr, hs_ = q.dscale_x != nullptr;
    rwl.issue(q.wl, P * J, 0, tid);
    rw0.issue(q.w0, D * N0, 0, tid);
    rh.issue(q.h + (size_t)b0 * P, ns * P, tid);
    rg.issue(q.g0 + (size_t)b0 * N0, ns * N0, tid);
    rz.issue(q.z + bd, ns * D, tid);
    rp.issue(q.p + (size_t)b0 * J, ns * J, tid);
    re.issue(q.eps + bd, ns * D, tid);
    r2.issue(h2 ? q.dz2 + bd : q.wl, h2 ? ns * D : 0, tid);
    rl.issue(hl ? q.dloc_x + bd : q.wl, hl ? ns * D : 0, tid);
    rs.issue(hs_ ? q.dscale_x + bd : q.wl, hs_ ? ns * D : 0, tid);
    rf.issue(q.fbmask + b0, ns, tid);
    rh.commit(hs, ns * P, S * P, tid);
    rg.commit(gs, ns * N0, S * N0, tid);
    rz.commit(zs, ns * D, S * D, tid);
    rp.commit(pls, ns * J, S * J, tid);
    re.commit(es, ns * D, S * D, tid);
    r2.commit(x2, h2 ? ns * D : 0, S * D, tid);
    rl.commit(xl, hl ? ns * D : 0, S * D, tid);
    rs.commit(xs, hs_ ? ns * D : 0, S * D, tid);
    rf.commit(fb, ns, S, tid);
    rwl.commit(wl, P * J, 0, tid);
    rw0.commit(w0, D * N0, 0, tid);
    lb_big_rest<V4>(wl, q.wl, P * J, tid);
    lb_big_rest<V4>(w0, q.w0, D * N0, tid);
  }
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
    // ---- latent backward (latent_bwd_kernel's arithmetic) ----
    if (tid < nout) {
      float g = red[tid];
      for (int w = 1; w < ks; ++w) g += red[w * nout + tid];
      const int s = tid / D, d = tid - s * D;
      const float loc = pls[s * J + d], raw = pls[s * J + D + d], e_ = es[tid];
      const float sc = softplus_f(raw), zz = zs[tid];
      const float w = klw * fb[s];
      float dloc, dsc;
      if (q.analytic == 2) {
        const float i2 = 1.f / (sc * sc);
        dloc = w * loc * i2;
        dsc = w * (1.f / sc - (1.f + loc * loc) * i2 / sc);
      } else if (q.analytic) { dloc = w * loc; dsc = w * (sc - 1.f / sc); }
      else { dloc = w * zz; dsc = w * (zz * e_ - 1.f / sc); }
      dloc += g; dsc += g * e_;
      if (q.dz2 != nullptr) { dloc += x2[tid]; dsc += x2[tid] * e_; }
      if (q.dloc_x != nullptr) dloc += xl[tid];
      if (q.dscale_x != nullptr) dsc += xs[tid];
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
  float amx = 0.f;
  for (int o = tid; o < ns * P; o += 256) {
    const int s = o / P, k = o - s * P;
    float c0 = 0.f, c1 = 0.f;
    for (int j = 0; j < J; j += 2) {   // (J = 2 D is even)
      c0 = fmaf(dps[s * J + j], wl[k * J + j], c0);
      c1 = fmaf(dps[s * J + j + 1], wl[k * J + j + 1], c1);
    }
    const float v = (c0 + c1) * odin_act_grad(q.h_act, hs[o]);
    q.dh[(size_t)(b0 + s) * P + k] = v;
    amx = fmaxf(amx, fabsf(v));
  }
  odin_amax_commit_wg(q.dh_amax, amx, tid, 256, red, blockIdx.x);  // (red: free since the dz reduction)
  // ---- this workgroup's partial weight gradients: sums over its S samples, s ascending ----
  {
    float* row = q.slab0 + (size_t)blockIdx.x * (D * N0 + N0);
    for (int o = tid; o < D * N0; o += 256) {
      const int d = o / N0, n = o - d * N0;
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) acc = fmaf(zs[s * D + d], gs[s * N0 + n], acc);
      row[o] = acc;
    }
    for (int n = tid; n < N0; n += 256) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) acc += gs[s * N0 + n];
      row[D * N0 + n] = acc;
    }
  }
  {
    float* row = q.slabl + (size_t)blockIdx.x * (P * J + J);
    for (int o = tid; o < P * J; o += 256) {
      const int k = o / J, j = o - k * J;
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) acc = fmaf(hs[s * P + k], dps[s * J + j], acc);
      row[o] = acc;
    }
    for (int j = tid; j < J; j += 256) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) acc += dps[s * J + j];
      row[P * J + j] = acc;
    }
  }
}

// samples per workgroup (1, 2, 4 or 8): at least ~128 workgroups -- every phase of a workgroup is a short
// dependent chain, so the launch lasts as long as ONE workgroup does -- and S * 2D outputs within 256 threads
// (per-sample rows are staged with at most 8 loads per thread: S * P, S * N0 <= 2048)
int lb_samples(int B, int P, int D, int N0) {
  int S = 8;
  while (S > 1 && (S * 2 * D > 256 || B / S < 128 || S * P > 2048 || S * N0 > 2048)) S >>= 1;
  return S;
}

// (the larger of the two kernels' layouts: the backward one)
size_t lb_lds_floats(int P, int D, int N0, int S) {
  return (size_t)P * 2 * D + (size_t)D * N0 + (size_t)S * P + (size_t)S * N0 + (size_t)S * (9 * D + 1) +
         (size_t)2 * D + N0 + 256 + 8;
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

// 16-byte loads / LDS stores of the two weight matrices: aligned sources, lengths in whole float4s
bool lb_vec_ok(const float* wl, const float* w0, int P, int D, int N0) {
  return (((size_t)wl | (size_t)w0) & 15) == 0 && ((P * 2 * D) & 3) == 0 && ((D * N0) & 3) == 0;
}

template <int S, bool V4>
int lb_launch_fwd(const LBFwd& q, int rows, size_t lds, void* stream) {
  if (int rc = lb_set_lds(&latent_block_fwd_kernel<S, V4>, lds)) return rc;
  ODIN_LAUNCH((latent_block_fwd_kernel<S, V4>), dim3(rows), dim3(256), lds, stream, q);
  return odin_check_launch("latent_block_fwd");
}
template <int S, bool V4>
int lb_launch_bwd(const LBBwd& q, int rows, size_t lds, void* stream) {
  if (int rc = lb_set_lds(&latent_block_bwd_kernel<S, V4>, lds)) return rc;
  ODIN_LAUNCH((latent_block_bwd_kernel<S, V4>), dim3(rows), dim3(256), lds, stream, q);
  return odin_check_launch("latent_block_bwd");
}

}  // namespace

// Number of workgroups (= slab rows of the backward launch); 0 when the shapes are outside the fused regime
// (both weight matrices + S rows must fit in LDS).
extern "C" int odin_latent_block_rows(int B, int P, int D, int N0) {
  if (ODIN_DIAG_ENV("ODIN_NOLATBLOCK")) return 0;
  if (B < 1 || P < 1 || D < 1 || N0 < 1 || 2 * D > 128 || P > 2048 || N0 > 2048) return 0;
  const int S = lb_samples(B, P, D, N0);
  if (lb_lds_floats(P, D, N0, S) * 4 > 120 * 1024) return 0;
  const int rows = (B + S - 1) / S;
  return rows <= ODIN_MAX_COLSUM_BLOCKS ? rows : 0;
}

extern "C" int odin_latent_block_fwd(const float* h, const float* wl, const float* bl, const float* eps_in,
                                     float* eps_out, uint64_t seed, const int32_t* step_dev, float* p,
                                     float* z, float* kl, float* fbmask, const float* w0, const float* b0,
                                     float* y0, int B, int P, int D, int N0, int act0, int analytic,
                                     float free_bits, const float* capacity, void* stream) {
  const int rows = odin_latent_block_rows(B, P, D, N0);
  if (rows == 0) return odin_fail(-2, "latent_block_fwd: shapes outside the fused regime");
  LBFwd q;
  memset(&q, 0, sizeof(q));
  q.h = h; q.wl = wl; q.bl = bl; q.eps_in = eps_in; q.eps = eps_out; q.p = p; q.z = z; q.kl = kl;
  q.fbmask = fbmask; q.w0 = w0; q.b0 = b0; q.y0 = y0; q.step_dev = (const int*)step_dev;
  q.k0 = (unsigned)seed; q.k1 = (unsigned)(seed >> 32);
  q.B = B; q.P = P; q.D = D; q.N0 = N0; q.act0 = act0; q.analytic = analytic; q.S = lb_samples(B, P, D, N0);
  q.free_bits = free_bits;
  q.cap = capacity;
  const size_t lds = lb_lds_floats(P, D, N0, q.S) * 4;
  const bool v4 = lb_vec_ok(wl, w0, P, D, N0);
  switch (q.S) {
    case 8: return v4 ? lb_launch_fwd<8, true>(q, rows, lds, stream) : lb_launch_fwd<8, false>(q, rows, lds, stream);
    case 4: return v4 ? lb_launch_fwd<4, true>(q, rows, lds, stream) : lb_launch_fwd<4, false>(q, rows, lds, stream);
    case 2: return v4 ? lb_launch_fwd<2, true>(q, rows, lds, stream) : lb_launch_fwd<2, false>(q, rows, lds, stream);
    default: return v4 ? lb_launch_fwd<1, true>(q, rows, lds, stream) : lb_launch_fwd<1, false>(q, rows, lds, stream);
  }
}

extern "C" int odin_latent_block_bwd(const float* g0, const float* w0, const float* z, const float* p,
                                     const float* eps, const float* fbmask, const float* klw,
                                     const float* dz_extra, const float* dloc_x, const float* dscale_x,
                                     const float* wl, const float* h, int h_act, float* dz, float* dp,
                                     float* dh, float* slab0, float* slabl, int B, int P, int D, int N0,
                                     int analytic, uint32_t* dh_amax, void* stream) {
  const int rows = odin_latent_block_rows(B, P, D, N0);
  if (rows == 0) return odin_fail(-2, "latent_block_bwd: shapes outside the fused regime");
  LBBwd q;
  memset(&q, 0, sizeof(q));
  q.g0 = g0; q.w0 = w0; q.z = z; q.p = p; q.eps = eps; q.fbmask = fbmask; q.klw = klw;
  q.dz2 = dz_extra; q.dloc_x = dloc_x; q.dscale_x = dscale_x; q.wl = wl; q.h = h; q.h_act = h_act;
  q.dz = dz; q.dp = dp; q.dh = dh; q.slab0 = slab0; q.slabl = slabl;
  q.B = B; q.P = P; q.D = D; q.N0 = N0; q.analytic = analytic; q.S = lb_samples(B, P, D, N0);
  q.dh_amax = dh_amax;
  const size_t lds = lb_lds_floats(P, D, N0, q.S) * 4;
  const bool v4 = lb_vec_ok(wl, w0, P, D, N0);
  switch (q.S) {
    case 8: return v4 ? lb_launch_bwd<8, true>(q, rows, lds, stream) : lb_launch_bwd<8, false>(q, rows, lds, stream);
    case 4: return v4 ? lb_launch_bwd<4, true>(q, rows, lds, stream) : lb_launch_bwd<4, false>(q, rows, lds, stream);
    case 2: return v4 ? lb_launch_bwd<2, true>(q, rows, lds, stream) : lb_launch_bwd<2, false>(q, rows, lds, stream);
    default: return v4 ? lb_launch_bwd<1, true>(q, rows, lds, stream) : lb_launch_bwd<1, false>(q, rows, lds, stream);
  }
}
