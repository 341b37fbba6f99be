// neck.hip -- the NECK of the 64x64 VAEs as one launch per direction: the encoder's last convolution, its projection,
// the latent block, the decoder's projection and its first Conv2DTranspose (round 6).
//
//   image_networks.py:466-471  Conv2D(64, 4, 2, 'same', act) on [8, 8, 64] -> [4, 4, 64]; Flatten; Dense(P, linear)
//   dense_distribution.py:339-380, continuous.py:443-483  DistributionDense(P -> 2D), MVNDiag(loc, softplus(raw)), z
//   variational_autoencoder.py:515-542, helpers.py:236-286  KL term (Monte Carlo / analytic / reverse), free bits
//   image_networks.py:494-502  Dense(D -> 16 C0, linear); Reshape(4, 4, C0); Conv2DTranspose(64, 4, 2, 'same', act)
//
// Round 5 ran this chain as igemm (conv) | igemm / dense_h (projection) | latent_block | smalldeconv: four launches of
// 6-12 us forward (37 us of a 480 us step for 3 % of its FLOPs) and four more backward (58 us): each a launch floor
// plus a few dependent L2 round trips, 128-1000 short workgroups that each fetch their own operands.  None of the
// layers couples samples, so here a workgroup owns S = 2 samples and walks the whole chain with everything but the
// two big weight matrices (W3: 256 KB, W4: 0.5-1 MB -- streamed once per workgroup from L2) in LDS / registers:
//
//   phase 1  conv3 as an implicit GEMM on the f16 matrix pipe (two planes per operand, odin_device.h): rows = the 32
//            output pixels of the 2 samples, K = 16 taps x 64 channels split over 8 wave pairs (a wave = 2 taps x one
//            32-column block: 24 v_mfma_f32_32x32x16_f16), the padded input images as planes in LDS, the weight
//            fragments straight from L2 (8 dwords per lane and k-step: the B-operand layout needs 8 consecutive k of
//            ONE column); partial tiles meet in LDS in wave order (bit reproducible)
//   phase 2  the projection on the vector ALU: thread = (4 columns, K slice), W4 rows as coalesced 16-byte loads, 16 in
//            flight per thread; partials through LDS in slice order
//   phase 3  latent_block.hip's forward arithmetic (same Philox stream, same sums)
//   phase 4  smalldeconv.hip's forward arithmetic (thread = (channel, stride class, pixel quarter), weights in
//            registers, zero-bordered image in LDS), range word of y1 kept
//
// The backward kernel runs the chain in reverse for the DATA gradients (dec1 <- ... <- conv3, the conv's data gradient
// on the matrix pipe by stride class) and leaves the per-sample-pair partial weight gradients of the three small
// matrices (W1, W0, Wl) as slab rows; the two big weight gradients (dW3: reduction over 4096 pixels, dW4: over the
// batch) couple ALL samples and stay on the weight-gradient kernels (odin_conv2d_wgrad / odin_dense_wgrad).
#include "odin_device.h"
#include "odin_internal.h"
#include "odin_latent_math.h"
#include <cstdint>
#include <cstdlib>
#include <utility>

namespace {

constexpr int NK_NT = 1024;    // threads per workgroup (16 waves: 4 per SIMD, <= 128 registers)
constexpr int NK_S = 2;        // samples per workgroup: 2 x 16 output pixels = the 32 rows of one MFMA tile
constexpr int NK_C = 64;       // channels of conv3's input and output, channels of deconv1's output
constexpr int NK_HW = 8;       // conv3 input / deconv1 output rows and columns
constexpr int NK_OP = 16;      // conv3 output / deconv1 input pixels (4 x 4)
constexpr int NK_K3 = NK_OP * NK_C;   // 1024: conv3's flattened output = the projection's reduction length
constexpr int NK_PW = NK_HW + 2;      // padded row / column count of a staged image
constexpr int NK_PITCH = 144;  // bytes per padded pixel and plane (64 f16 + 16: consecutive pixels on different banks)
constexpr int NK_PLANE = NK_S * NK_PW * NK_PW * NK_PITCH;   // one plane of the two padded images: 28 800 bytes

struct NeckFwd {
  odin_neck_args a;
  unsigned k0, k1;
  int N0, w1_al;
  long long* stamps;   // diagnostics (odin_debug_set_neck_stamps): 100 MHz wall-clock stamps of workgroup 0 at the phase boundaries
};

#ifdef ODIN_SIM
#define NK_STAMP(i) ((void)0)
#else
#define NK_STAMP(i)                                                                              \
  do {                                                                                           \
    if (q.stamps != nullptr && threadIdx.x == 0 && blockIdx.x == 0) q.stamps[i] = (long long)wall_clock64(); \
  } while (0)
#endif

// small arrays: all loads first, the LDS stores afterwards (latent_block.hip: one exposed round trip for the lot)
template <int NL>
struct NKSmall {
  float r[NL];
  __device__ __forceinline__ void issue(const float* src, int n, int tid) {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = u * NK_NT + tid;
      r[u] = src[e < n ? e : 0];
    }
  }
  __device__ __forceinline__ void commit(float* dst, int n, int tid) const {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int e = u * NK_NT + tid;
      if (e < n) dst[e] = r[u];
    }
  }
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop the optimiser cannot leave rolled (a rolled
// loop over a register array that holds loads in flight sends the array through scratch memory)
template <int... Is, class F>
__device__ __forceinline__ void nk_static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nk_static_for(F&& f) {
  nk_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

__device__ __forceinline__ float4 nk_ld4(const float* q, int al) {
  if (al) return *reinterpret_cast<const float4*>(q);
  return make_float4(q[0], q[1], q[2], q[3]);
}

// LDS map of the forward kernel (bytes).  Region A holds the input planes during phase 1's MFMAs, then the partial tiles
// (64 KB), then phase 2's partials (<= 32 KB) and phase 3's (4 KB).
struct NKFwdLds {
  int a, h3, h4, wl, w0, b3, b4, bl, b0, b1, es, ps, zs, x1, xp, red, end;
  __host__ __device__ NKFwdLds(int P, int D, int C0) {
    const int J = 2 * D, N0 = NK_OP * C0;
    int o = 0;
    a = o; o += 16 * 1024 * 4;                       // max(2 planes = 57 600, 16 partial tiles = 65 536)
    h3 = o; o += NK_S * NK_K3 * 4;
    h4 = o; o += NK_S * P * 4;
    wl = o; o += P * J * 4;
    w0 = o; o += D * N0 * 4;
    b3 = o; o += NK_C * 4;
    b4 = o; o += P * 4;
    bl = o; o += J * 4;
    b0 = o; o += N0 * 4;
    b1 = o; o += NK_C * 4;
    es = o; o += NK_S * D * 4;
    ps = o; o += NK_S * J * 4;
    zs = o; o += NK_S * D * 4;
    x1 = o; o += NK_S * N0 * 4;
    xp = o; o += NK_S * 36 * C0 * 4;
    red = o; o += 64 * 4;
    end = (o + 15) & ~15;
  }
};

template <int C0>
__global__ __launch_bounds__(NK_NT) void neck_fwd_kernel(NeckFwd q) {
  ODIN_DYN_SMEM(char, sm);
  const odin_neck_args& a = q.a;
  const int P = a.P, D = a.D, J = 2 * a.D, N0 = NK_OP * C0;
  const NKFwdLds L(P, D, C0);
  char* planes = sm + L.a;
  float* part = reinterpret_cast<float*>(sm + L.a);
  float* h3s = reinterpret_cast<float*>(sm + L.h3);
  float* h4s = reinterpret_cast<float*>(sm + L.h4);
  float* wls = reinterpret_cast<float*>(sm + L.wl);
  float* w0s = reinterpret_cast<float*>(sm + L.w0);
  float* b3s = reinterpret_cast<float*>(sm + L.b3);
  float* b4s = reinterpret_cast<float*>(sm + L.b4);
  float* bls = reinterpret_cast<float*>(sm + L.bl);
  float* b0s = reinterpret_cast<float*>(sm + L.b0);
  float* b1s = reinterpret_cast<float*>(sm + L.b1);
  float* es = reinterpret_cast<float*>(sm + L.es);
  float* ps = reinterpret_cast<float*>(sm + L.ps);
  float* zs = reinterpret_cast<float*>(sm + L.zs);
  float* x1s = reinterpret_cast<float*>(sm + L.x1);
  float* xp = reinterpret_cast<float*>(sm + L.xp);
  float* red = reinterpret_cast<float*>(sm + L.red);
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int b0 = blockIdx.x * NK_S;
  const int ns = (a.B - b0 < NK_S) ? a.B - b0 : NK_S;
  const unsigned step = a.step_dev ? (unsigned)a.step_dev[0] : 0u;

  // ================= phase 0: every load that depends on nothing goes out first =================
  NK_STAMP(0);
  const OdinRangeReq x_rq = odin_range_issue(a.x_amax, lane);
  // conv3 weight fragments of this wave: taps 2 tp, 2 tp + 1, columns 32 nt .. 32 nt + 31; lane (n = lane & 31, kg = lane >> 5)
  // of k-step j needs W3[tap][16 j + 8 kg + i][32 nt + n], i = 0..7 (Keras layout (kh, kw, Cin, Cout): one dword per i)
  const int tp = wave >> 1, nt = wave & 1, l31 = lane & 31, kg = lane >> 5;
  float wrawA[4][8], wrawB[4][8];   // (tap 2 tp now, tap 2 tp + 1 behind the staging barrier)
  auto w3_issue = [&](float (&wraw)[4][8], int tap) {
    const float* base = a.w3 + ((size_t)tap * NK_C + 8 * kg) * NK_C + 32 * nt + l31;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) wraw[j][i] = base[(16 * j + i) * NK_C];
  };
  w3_issue(wrawA, 2 * tp);
  // the two input images: 2 x 16-byte loads per thread (range-checked: a missing second sample reads zeros)
  const OdinRun XR = odin_run(a.x, (unsigned)((size_t)a.B * NK_HW * NK_HW * NK_C * 4));
  float4 xv[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + NK_NT * u;   // float4 index inside the workgroup's S images
    xv[u] = odin_run_load4(XR, (unsigned)(((size_t)b0 * NK_HW * NK_HW * NK_C + 4 * (size_t)e) * 4));
  }
  // the small arrays (scalar loads: the flat parameter buffer packs tensors without padding)
  NKSmall<4> rwl;    // P * 2D <= 4096
  NKSmall<2> rw0;    // D * N0 <= 2048
  NKSmall<1> rb3, rb4, rbl, rb0, rb1, re;
  const bool have_eps = a.eps_in != nullptr;
  rwl.issue(a.wl, P * J, tid);
  rw0.issue(a.w0, D * N0, tid);
  rb3.issue(a.b3, NK_C, tid);
  rb4.issue(a.b4, P, tid);
  rbl.issue(a.bl, J, tid);
  rb0.issue(a.b0, N0, tid);
  rb1.issue(a.b1, NK_C, tid);
  re.issue(have_eps ? a.eps_in + (size_t)b0 * D : a.wl, have_eps ? ns * D : 0, tid);

  // zero border of the padded images (both planes): 36 pixels per image, 8 x 16 bytes per pixel and plane
  for (int e = tid; e < NK_S * 36 * 2 * 8; e += NK_NT) {
    const int piece = e & 7, pl = (e >> 3) & 1, bp = e >> 4;
    const int s = bp / 36, idx = bp - s * 36;
    int py, px;
    if (idx < 10) { py = 0; px = idx; }
    else if (idx < 20) { py = NK_PW - 1; px = idx - 10; }
    else { py = 1 + ((idx - 20) >> 1); px = ((idx - 20) & 1) * (NK_PW - 1); }
    *reinterpret_cast<float4*>(planes + pl * NK_PLANE + (s * NK_PW * NK_PW + py * NK_PW + px) * NK_PITCH + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the input's range word: carried times 2^gk only when its bound leaves the f16 window (odin_act_needs_scale)
  const unsigned x_mb = odin_range_finish(x_rq);
  const bool x_scl = a.x_amax != nullptr && odin_act_needs_scale(x_mb);
  const int gk = x_scl ? odin_range_shift(x_mb) : 0;
  const float in_s = odin_pow2(gk), in_s2k = odin_pow2(gk + 11);
  const float out_s = odin_pow2(-gk), out_sx = odin_pow2(-gk - 11);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int e = tid + NK_NT * u;
    const int s = e >> 10, pix = (e >> 4) & 63, c4 = e & 15;
    const int py = (pix >> 3) + 1, px = (pix & 7) + 1;
    u32x2 h, l;
    odin_split_h4<true>(xv[u], in_s, in_s2k, h, l);
    char* d = planes + (s * NK_PW * NK_PW + py * NK_PW + px) * NK_PITCH + c4 * 8;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + NK_PLANE) = l;
  }
  rwl.commit(wls, P * J, tid);
  rw0.commit(w0s, D * N0, tid);
  rb3.commit(b3s, NK_C, tid);
  rb4.commit(b4s, P, tid);
  rbl.commit(bls, J, tid);
  rb0.commit(b0s, N0, tid);
  rb1.commit(b1s, NK_C, tid);
  if (have_eps && tid < NK_S * D) es[tid] = tid < ns * D ? re.r[0] : 0.f;
  // (noise drawn here: element f of the [B, D] stream is component f & 3 of counter f >> 2 -- odin_rng_normal's stream)
  if (!have_eps && tid < NK_S * D) {
    const unsigned f = (unsigned)(b0 * D + tid);
    float v[4];
    odin_normal4(f >> 2, 0u, step, q.k0, q.k1, v);
    const bool live = tid < ns * D;
    const float e = live ? v[f & 3] : 0.f;
    es[tid] = e;
    if (live) a.eps[f] = e;
  }
  __syncthreads();
  NK_STAMP(1);

  // ================= phase 1: conv3 on the matrix pipe =================
  {
    // this lane's output pixel (row m of the tile) and the byte offset of its padded input pixel for tap (0, 0)
    const int m = l31, s = m >> 4, oy = (m >> 2) & 3, ox = m & 3;
    const int pix0 = (s * NK_PW * NK_PW + 2 * oy * NK_PW + 2 * ox) * NK_PITCH + 16 * kg;
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
    auto run_tap = [&](const float (&wraw)[4][8], int tap) {
      const int kh = tap >> 2, kw = tap & 3;
      const char* ap = planes + pix0 + (kh * NK_PW + kw) * NK_PITCH;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap + 32 * j);
        const u32x4 al = *reinterpret_cast<const u32x4*>(ap + NK_PLANE + 32 * j);
        u32x2 h0, l0, h1, l1;
        odin_split_h4<false>(make_float4(wraw[j][0], wraw[j][1], wraw[j][2], wraw[j][3]), 1.f, ODIN_LO_SCALE, h0, l0);
        odin_split_h4<false>(make_float4(wraw[j][4], wraw[j][5], wraw[j][6], wraw[j][7]), 1.f, ODIN_LO_SCALE, h1, l1);
        u32x4 bh, bl;
        bh.x = h0.x; bh.y = h0.y; bh.z = h1.x; bh.w = h1.y;
        bl.x = l0.x; bl.y = l0.y; bl.z = l1.x; bl.w = l1.y;
        acx = mfma32_f16(ah, bl, acx);
        acc = mfma32_f16(ah, bh, acc);
        acx = mfma32_f16(al, bh, acx);
      }
    };
    w3_issue(wrawB, 2 * tp + 1);
    run_tap(wrawA, 2 * tp);
    run_tap(wrawB, 2 * tp + 1);
    __syncthreads();   // every wave is done with the planes: the partial tiles take their place
    NK_STAMP(2);
    float* mine = part + wave * 1024;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = fmaf(acx[r], out_sx, acc[r] * out_s);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + NK_NT * u, co = e & 63, m2 = e >> 6;   // output (row m2 = 16 s + pixel, channel co)
      const int ntile = co >> 5, col = co & 31, hh = (m2 >> 2) & 1, r = (m2 & 3) + 4 * (m2 >> 3);
      const float* src = part + ntile * 1024 + r * 64 + col + 32 * hh;
      float t = src[0];
#pragma unroll
      for (int w = 1; w < 8; ++w) t += src[2 * w * 1024];
      const float y = odin_act(a.act3, t + b3s[co]);
      h3s[e] = y;
      if ((m2 >> 4) < ns) a.y3[(size_t)b0 * NK_K3 + e] = y;
    }
    __syncthreads();
  }

  // ================= phase 2: the projection h4 = act4(h3 W4 + b4) on the vector ALU =================
  NK_STAMP(3);
  {
    const int nq = P >> 2, KS = NK_NT / nq, RK = NK_K3 / KS;   // float4 columns, K slices, rows per slice
    const int qc = tid % nq, ks = tid / nq;
    float acc0[4] = {0.f, 0.f, 0.f, 0.f}, acc1[4] = {0.f, 0.f, 0.f, 0.f};
    const float4* wp = reinterpret_cast<const float4*>(a.w4) + (size_t)ks * RK * nq + qc;
    const float* hx = h3s + ks * RK;
    for (int k0 = 0; k0 < RK; k0 += 16) {
      float4 wv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) wv[i] = wp[(size_t)(k0 + i) * nq];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float x0 = hx[k0 + i], x1 = hx[NK_K3 + k0 + i];
        acc0[0] = fmaf(x0, wv[i].x, acc0[0]); acc0[1] = fmaf(x0, wv[i].y, acc0[1]);
        acc0[2] = fmaf(x0, wv[i].z, acc0[2]); acc0[3] = fmaf(x0, wv[i].w, acc0[3]);
        acc1[0] = fmaf(x1, wv[i].x, acc1[0]); acc1[1] = fmaf(x1, wv[i].y, acc1[1]);
        acc1[2] = fmaf(x1, wv[i].z, acc1[2]); acc1[3] = fmaf(x1, wv[i].w, acc1[3]);
      }
    }
    // partials [ks][s][P] over region A (the partial tiles of phase 1 were consumed before the last barrier)
    float* p2 = part + (size_t)ks * NK_S * P + 4 * qc;
    *reinterpret_cast<float4*>(p2) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    *reinterpret_cast<float4*>(p2 + P) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    __syncthreads();
    if (tid < NK_S * P) {
      const int s = tid / P, n = tid - s * P;
      float t = part[tid];
      for (int w = 1; w < KS; ++w) t += part[w * NK_S * P + tid];
      const float y = odin_act(a.act4, t + b4s[n]);
      h4s[tid] = y;
      if (s < ns) a.y4[(size_t)(b0 + s) * P + n] = y;
    }
    __syncthreads();
  }

  NK_STAMP(4);
  // deconv1's weights of this thread (4 taps x C0 values): requested here, used after the latent block
  const int co1 = tid & 63, cr = (tid >> 7) & 1, cc = (tid >> 6) & 1, qq = tid >> 8;
  float wr[4][C0];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kh = (t >> 1) ? 3 - cr : 1 - cr, kw = (t & 1) ? 3 - cc : 1 - cc;
    const float* src = a.w1 + ((size_t)((kh * 4 + kw) * NK_C + co1)) * C0;
#pragma unroll
    for (int v = 0; v < C0 / 4; ++v) {
      const float4 t4 = nk_ld4(src + 4 * v, q.w1_al);
      wr[t][4 * v] = t4.x; wr[t][4 * v + 1] = t4.y; wr[t][4 * v + 2] = t4.z; wr[t][4 * v + 3] = t4.w;
    }
  }

  // ================= phase 3: the latent block (latent_block.hip's forward arithmetic) =================
  {
    // p[s][j] = sum_k h4[s][k] wl[k][j] + bl[j]: S * 2D outputs, the threads split k
    const int nout = NK_S * J;
    const int ksl = NK_NT / nout;
    const int kc = (P + ksl - 1) / ksl;
    const int o = tid % nout, kq = tid / nout;
    float* red3 = part;   // [ksl][nout] <= 4 KB
    if (kq < ksl) {
      const int s = o / J, j = o - s * J;
      const int klo = kq * kc, khi = (klo + kc < P) ? klo + kc : P;
      float a0 = 0.f, a1 = 0.f;
      int k = klo;
      for (; k + 1 < khi; k += 2) {
        a0 = fmaf(h4s[s * P + k], wls[k * J + j], a0);
        a1 = fmaf(h4s[s * P + k + 1], wls[(k + 1) * J + j], a1);
      }
      if (k < khi) a0 = fmaf(h4s[s * P + k], wls[k * J + j], a0);
      red3[kq * nout + o] = a0 + a1;
    }
    __syncthreads();
    if (tid < nout) {
      const int s = tid / J, j = tid - s * J;
      float t = red3[tid];
      for (int w = 1; w < ksl; ++w) t += red3[w * nout + tid];
      t += bls[j];
      ps[tid] = t;
      if (s < ns) a.p[(size_t)(b0 + s) * J + j] = t;
    }
    __syncthreads();
    // reparameterise + KL (summed over d in order)
    if (tid < NK_S * D) {
      const int s = tid / D, d = tid - s * D;
      const float loc = ps[s * J + d], sc = softplus_f(ps[s * J + D + d]), e = es[tid];
      const float zz = loc + sc * e;
      zs[tid] = zz;
      if (s < ns) a.z[(size_t)(b0 + s) * D + d] = zz;
      const float ls = odin_log(sc);
      float t;
      if (a.analytic == 2) t = ls + 0.5f * (1.f + loc * loc) / (sc * sc) - 0.5f;
      else if (a.analytic) t = 0.5f * (sc * sc + loc * loc - 1.f) - ls;
      else t = 0.5f * (zz * zz - e * e) - ls;
      red3[2048 + tid] = t;
    }
    __syncthreads();
    if (tid < ns) {
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc += red3[2048 + tid * D + d];
      float m = 1.f;
      if (a.free_bits >= 0.f) {
        const float thr = a.free_bits * (float)D;
        if (!(acc > thr)) { acc = thr; m = 0.f; }
      }
      if (a.capacity != nullptr) {  // beta_vae.py:171-177: |kl - C(step)|, gradient sign(kl - C)
        const float dd = acc - a.capacity[0];
        m *= dd > 0.f ? 1.f : (dd < 0.f ? -1.f : 0.f);
        acc = fabsf(dd);
      }
      a.kl[b0 + tid] = acc;
      a.fbmask[b0 + tid] = m;
    }
    // (y1 == NULL: the encoder's half only -- FactorVAE's second pass wants z alone)
    if (a.y1 == nullptr) return;
    // y0[s][n] = act0(sum_d z[s][d] w0[d][n] + b0[n]) = the decoder's first image [4, 4, C0]
    for (int o2 = tid; o2 < NK_S * N0; o2 += NK_NT) {
      const int s = o2 / N0, n = o2 - s * N0;
      float acc = 0.f;
      for (int d = 0; d < D; ++d) acc = fmaf(zs[s * D + d], w0s[d * N0 + n], acc);
      const float y = odin_act(a.act0, acc + b0s[n]);
      x1s[o2] = y;
      if (s < ns) a.y0[(size_t)(b0 + s) * N0 + n] = y;
    }
    __syncthreads();
  }

  // ================= phase 4: deconv1 (smalldeconv.hip's forward arithmetic) =================
  NK_STAMP(5);
  {
    // zero-bordered input images [S][6][6][C0]
    for (int e = tid; e < NK_S * 36 * C0; e += NK_NT) {
      const int s = e / (36 * C0), r = e - s * 36 * C0;
      const int pp = r / C0, ci = r - pp * C0;
      const int pr = pp / 6, pc = pp - pr * 6;
      const bool in = pr >= 1 && pr <= 4 && pc >= 1 && pc <= 4;
      xp[e] = in ? x1s[s * N0 + ((pr - 1) * 4 + pc - 1) * C0 + ci] : 0.f;
    }
    __syncthreads();
    const float bv = b1s[co1];
    float amx = 0.f;
    for (int s = 0; s < ns; ++s) {
      const float* img = xp + s * 36 * C0;
      float* out = a.y1 + (size_t)(b0 + s) * (NK_HW * NK_HW) * NK_C + co1;
      for (int pix = qq; pix < NK_OP; pix += 4) {
        const int i = pix >> 2, j = pix & 3;
        // tap a (kh = 1 - r) reads padded row i + 1 + r, tap b (kh = 3 - r) padded row i + r (TF SAME, pads (1, 1))
        const float* ra = img + ((i + 1 + cr) * 6) * C0;
        const float* rb = img + ((i + cr) * 6) * C0;
        const int ca = (j + 1 + cc) * C0, cb = (j + cc) * C0;
        float acc = bv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float* src = ((t >> 1) ? rb : ra) + ((t & 1) ? cb : ca);
#pragma unroll
          for (int v = 0; v < C0 / 4; ++v) {
            const float4 x4 = *reinterpret_cast<const float4*>(src + 4 * v);   // (same address in all lanes of the wave)
            acc = fmaf(x4.x, wr[t][4 * v], acc);
            acc = fmaf(x4.y, wr[t][4 * v + 1], acc);
            acc = fmaf(x4.z, wr[t][4 * v + 2], acc);
            acc = fmaf(x4.w, wr[t][4 * v + 3], acc);
          }
        }
        const float o = odin_act(a.act1, acc);
        amx = fmaxf(amx, fabsf(o));
        out[(size_t)((2 * i + cr) * NK_HW + 2 * j + cc) * NK_C] = o;
      }
    }
    odin_amax_commit_wg(a.y1_amax, amx, tid, NK_NT, red, blockIdx.x);
  }
  NK_STAMP(6);
}


// ------------------------------------------------------------------------------------------------------------------
// backward: dy1 -> (dW1) -> dx1 -> g0 -> (dW0) -> dz -> dp -> (dWl) -> dh4 -> dy3 -> dx, two samples per workgroup
// ------------------------------------------------------------------------------------------------------------------
constexpr int NK_GP = 68;      // pixel pitch (floats) of the zero-bordered gradient image (smalldeconv.hip: SD_GP)
constexpr int NK_QW = 6;       // padded rows / columns of a 4 x 4 image
constexpr int NK_QPLANE = NK_S * NK_QW * NK_QW * NK_PITCH;   // one plane of the padded dy3 images: 10 368 bytes

// LDS map (bytes).  `big` = [g1p: dy1 as zero-bordered images | w1d: deconv1's weights regrouped for the data gradient];
// both are dead after the deconv's backward: conv3's partial tiles (64 KB) then lie over the start of it, the dy3
// planes behind them at 64 KB, y3 / dy3 (fp32) behind those.
struct NKBwdLds {
  int g1p, w1d, part, q3, y3, d3, xp, x1, g0, w0, wl, h4, d4, zs, pl, es, x2, xl, xs, fb, dps, red, end;
  __host__ __device__ NKBwdLds(int P, int D, int C0) {
    const int J = 2 * D, N0 = NK_OP * C0;
    g1p = 0;
    w1d = NK_S * NK_PW * NK_PW * NK_GP * 4;                 // 54 400
    part = 0;
    q3 = 16 * 1024 * 4;                                     // 65 536
    y3 = q3 + 2 * NK_QPLANE;                                // 86 272
    d3 = y3 + NK_S * NK_K3 * 4;
    int o = d3 + NK_S * NK_K3 * 4;                          // 102 656
    const int big_end = w1d + 16 * NK_C * C0 * 4;           // 87 168 (C0 = 8) / 119 936 (C0 = 16)
    if (o < big_end) o = big_end;
    xp = o; o += NK_S * 36 * C0 * 4;
    x1 = o; o += NK_S * N0 * 4;
    g0 = o; o += NK_S * N0 * 4;
    w0 = o; o += D * N0 * 4;
    wl = o; o += P * J * 4;
    h4 = o; o += NK_S * P * 4;
    d4 = o; o += NK_S * P * 4;
    zs = o; o += NK_S * D * 4;
    pl = o; o += NK_S * J * 4;
    es = o; o += NK_S * D * 4;
    x2 = o; o += NK_S * D * 4;
    xl = o; o += NK_S * D * 4;
    xs = o; o += NK_S * D * 4;
    fb = o; o += 16 * 4;
    dps = o; o += NK_S * J * 4;
    red = o; o += NK_NT * 4;
    end = (o + 15) & ~15;
  }
};

template <int C0, int NQ>   // NQ = P / 32: float4 columns of a W4 row per lane of its 8-lane group (4: P = 128, 8: P = 256)
__global__ __launch_bounds__(NK_NT) void neck_bwd_kernel(NeckFwd q) {
  ODIN_DYN_SMEM(char, sm);
  const odin_neck_args& a = q.a;
  const int P = a.P, D = a.D, J = 2 * a.D, N0 = NK_OP * C0;
  const NKBwdLds L(P, D, C0);
  float* g1p = reinterpret_cast<float*>(sm + L.g1p);
  float* w1d = reinterpret_cast<float*>(sm + L.w1d);
  float* part = reinterpret_cast<float*>(sm + L.part);
  char* q3 = sm + L.q3;
  float* y3s = reinterpret_cast<float*>(sm + L.y3);
  float* d3s = reinterpret_cast<float*>(sm + L.d3);
  float* xp = reinterpret_cast<float*>(sm + L.xp);
  float* x1s = reinterpret_cast<float*>(sm + L.x1);
  float* g0s = reinterpret_cast<float*>(sm + L.g0);
  float* w0s = reinterpret_cast<float*>(sm + L.w0);
  float* wls = reinterpret_cast<float*>(sm + L.wl);
  float* h4s = reinterpret_cast<float*>(sm + L.h4);
  float* d4s = reinterpret_cast<float*>(sm + L.d4);
  float* zs = reinterpret_cast<float*>(sm + L.zs);
  float* pls = reinterpret_cast<float*>(sm + L.pl);
  float* es = reinterpret_cast<float*>(sm + L.es);
  float* x2 = reinterpret_cast<float*>(sm + L.x2);
  float* xl = reinterpret_cast<float*>(sm + L.xl);
  float* xs = reinterpret_cast<float*>(sm + L.xs);
  float* fb = reinterpret_cast<float*>(sm + L.fb);
  float* dps = reinterpret_cast<float*>(sm + L.dps);
  float* red = reinterpret_cast<float*>(sm + L.red);
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int b0 = blockIdx.x * NK_S;
  const int ns = (a.B - b0 < NK_S) ? a.B - b0 : NK_S;
  const float klw = a.klw[0];

  // ================= loads =================
  NK_STAMP(8);
  // dy1 as zero-bordered images [S][10][10][NK_GP]: S * 100 * 16 float4 units, 4 rounds of one 16-byte load per thread
  {
    const OdinRun GR = odin_run(a.dy1, (unsigned)((size_t)a.B * NK_HW * NK_HW * NK_C * 4));
    constexpr int units = NK_S * NK_PW * NK_PW * 16;   // 3200
    float4 r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = u * NK_NT + tid;
      const int pp = e >> 4, c4 = e & 15;
      const int s = pp / (NK_PW * NK_PW), rem = pp - s * (NK_PW * NK_PW);
      const int pr = rem / NK_PW, pc = rem - pr * NK_PW;
      const bool in = e < units && pr >= 1 && pr <= NK_HW && pc >= 1 && pc <= NK_HW;
      r[u] = odin_run_load4(GR, in ? (unsigned)((((size_t)(b0 + s) * 64 + (pr - 1) * NK_HW + pc - 1) * 16 + c4) * 16) : ODIN_OOB);
    }
    // the small arrays
    NKSmall<4> rwl;
    NKSmall<2> rw0;
    NKSmall<1> rx1, rh4, rz, rp, re, r2, rl, rs, rf;
    const bool h2 = a.dz_extra != nullptr, hl = a.dloc_x != nullptr, hs_ = a.dscale_x != nullptr;
    const size_t bd = (size_t)b0 * D;
    rwl.issue(a.wl, P * J, tid);
    rw0.issue(a.w0, D * N0, tid);
    rx1.issue(a.y0 + (size_t)b0 * N0, ns * N0, tid);
    rh4.issue(a.y4 + (size_t)b0 * P, ns * P, tid);
    rz.issue(a.z + bd, ns * D, tid);
    rp.issue(a.p + (size_t)b0 * J, ns * J, tid);
    re.issue(a.eps + bd, ns * D, tid);
    r2.issue(h2 ? a.dz_extra + bd : a.wl, h2 ? ns * D : 0, tid);
    rl.issue(hl ? a.dloc_x + bd : a.wl, hl ? ns * D : 0, tid);
    rs.issue(hs_ ? a.dscale_x + bd : a.wl, hs_ ? ns * D : 0, tid);
    rf.issue(a.fbmask + b0, ns, tid);
    // deconv1's weights [tap][co][ci] -> w1d [tap][lane = (cs, ci)][k] with co = cs C0 + k: in the data gradient a lane
    // (cs, ci) reads its C0 weights of a tap as C0 / 4 16-byte LDS reads (the scalar reads of smalldeconv.hip's layout
    // made this phase LDS-issue bound: 7 us of the launch); the 16-byte chunks of a lane are rotated by (lane >> SH) so
    // that the 16 lanes of a read pass fall on 16 different bank groups
    constexpr int NCH = C0 / 4, SH = NCH == 2 ? 3 : 2;
    constexpr int WU = 16 * NK_C * C0 / 4 / NK_NT;   // float4 units of W1 per thread: 2 (C0 = 8) / 4 (C0 = 16)
    float4 w1v[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) w1v[u] = nk_ld4(a.w1 + 4 * (tid + NK_NT * u), q.w1_al);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = u * NK_NT + tid;
      if (e < units) *reinterpret_cast<float4*>(g1p + (e >> 4) * NK_GP + 4 * (e & 15)) = r[u];
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int e = tid + NK_NT * u;
      const int ci4 = e % NCH, tc = e / NCH, co = tc & 63, tap = tc >> 6;
      const int cs = co / C0, k = co - cs * C0;
      const float vv[4] = {w1v[u].x, w1v[u].y, w1v[u].z, w1v[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ln = cs * C0 + 4 * ci4 + i;
        w1d[((tap * 64 + ln) * NCH + (((k >> 2) + (ln >> SH)) & (NCH - 1))) * 4 + (k & 3)] = vv[i];
      }
    }
    rwl.commit(wls, P * J, tid);
    rw0.commit(w0s, D * N0, tid);
    // (per-sample rows: zeros for a missing second sample)
    auto put = [&](float* dst, const NKSmall<1>& rr, int n, int cap) { if (tid < cap) dst[tid] = tid < n ? rr.r[0] : 0.f; };
    put(x1s, rx1, ns * N0, NK_S * N0);
    put(h4s, rh4, ns * P, NK_S * P);
    put(zs, rz, ns * D, NK_S * D);
    put(pls, rp, ns * J, NK_S * J);
    put(es, re, ns * D, NK_S * D);
    put(x2, r2, h2 ? ns * D : 0, NK_S * D);
    put(xl, rl, hl ? ns * D : 0, NK_S * D);
    put(xs, rs, hs_ ? ns * D : 0, NK_S * D);
    put(fb, rf, ns, NK_S);
  }
  __syncthreads();
  // zero-bordered x1 images [S][6][6][C0] (the weight gradient's other operand)
  for (int e = tid; e < NK_S * 36 * C0; e += NK_NT) {
    const int s = e / (36 * C0), r = e - s * 36 * C0;
    const int pp = r / C0, ci = r - pp * C0;
    const int pr = pp / 6, pc = pp - pr * 6;
    const bool in = pr >= 1 && pr <= 4 && pc >= 1 && pc <= 4;
    xp[e] = in ? x1s[s * N0 + ((pr - 1) * 4 + pc - 1) * C0 + ci] : 0.f;
  }
  __syncthreads();

  // ================= deconv1 backward (smalldeconv.hip's arithmetic, sums in the same order) =================
  NK_STAMP(9);
  constexpr int PW = NK_PW, PH = NK_PW;
  {
    // weight gradient: thread (channel co, parity class (r, c), tap t of the class's four) owns dW1[kh][kw][co][0..C0)
    const int co = tid & 63, cr = (tid >> 7) & 1, cc = (tid >> 6) & 1, t = tid >> 8;
    const int kh = (t >> 1) ? 3 - cr : 1 - cr, kw = (t & 1) ? 3 - cc : 1 - cc;
    const int roff = (t >> 1) ? cr : 1 + cr, coff = (t & 1) ? cc : 1 + cc;
    float acc[C0];
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[c] = 0.f;
    for (int s = 0; s < ns; ++s) {
      const float* img = xp + s * 36 * C0;
      const float* gimg = g1p + (size_t)s * PH * PW * NK_GP + co;
      for (int i = 0; i < 4; ++i) {
        const float* row = img + ((i + roff) * 6 + coff) * C0;
        const float* grow = gimg + ((2 * i + cr + 1) * PW + cc + 1) * NK_GP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = grow[2 * j * NK_GP];
#pragma unroll
          for (int v = 0; v < C0 / 4; ++v) {
            const float4 x4 = *reinterpret_cast<const float4*>(row + j * C0 + 4 * v);
            acc[4 * v] = fmaf(x4.x, g, acc[4 * v]);
            acc[4 * v + 1] = fmaf(x4.y, g, acc[4 * v + 1]);
            acc[4 * v + 2] = fmaf(x4.z, g, acc[4 * v + 2]);
            acc[4 * v + 3] = fmaf(x4.w, g, acc[4 * v + 3]);
          }
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(a.slab1 + (size_t)blockIdx.x * (16 * NK_C * C0) +
                                            (size_t)((kh * 4 + kw) * NK_C + co) * C0);
#pragma unroll
    for (int v = 0; v < C0 / 4; ++v) dst[v] = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
  }
  NK_STAMP(10);
  {
    // data gradient: wave = input pixel, lane (cs, ci) sums its C0 channels co = cs C0 + k of all 16 taps, the 64 / C0
    // lanes of a ci meet by shuffles; x act0'(y0) -> g0 = dL/d(pre-activation of the decoder's projection)
    const int cs = lane / C0, ci = lane - cs * C0;
    constexpr int NCH = C0 / 4, SH = NCH == 2 ? 3 : 2;
    const int pix = wave;   // 16 waves = the 16 input pixels; both samples share the weight reads
    const int i = pix >> 2, j = pix & 3;
    const float* gp0 = g1p + (2 * i * PW + 2 * j) * NK_GP + cs * C0;
    const float* wl0 = w1d + lane * C0;
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 1
    for (int kh = 0; kh < 4; ++kh) {
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const float* gp = gp0 + (kh * PW + kw) * NK_GP;
        const float* wp = wl0 + (kh * 4 + kw) * 64 * C0;
#pragma unroll
        for (int v = 0; v < NCH; ++v) {
          const float4 w4 = *reinterpret_cast<const float4*>(wp + 4 * ((v + (lane >> SH)) & (NCH - 1)));
          const float4 ga = *reinterpret_cast<const float4*>(gp + 4 * v);
          const float4 gb = *reinterpret_cast<const float4*>(gp + PH * PW * NK_GP + 4 * v);
          acc0 = fmaf(ga.x, w4.x, acc0); acc0 = fmaf(ga.y, w4.y, acc0);
          acc0 = fmaf(ga.z, w4.z, acc0); acc0 = fmaf(ga.w, w4.w, acc0);
          acc1 = fmaf(gb.x, w4.x, acc1); acc1 = fmaf(gb.y, w4.y, acc1);
          acc1 = fmaf(gb.z, w4.z, acc1); acc1 = fmaf(gb.w, w4.w, acc1);
        }
      }
    }
#pragma unroll
    for (int m = C0; m < 64; m <<= 1) { acc0 += __shfl_xor(acc0, m); acc1 += __shfl_xor(acc1, m); }
    if (cs == 0) {
      const int o = pix * C0 + ci;
      g0s[o] = acc0 * odin_act_grad(a.act0, x1s[o]);
      g0s[N0 + o] = ns > 1 ? acc1 * odin_act_grad(a.act0, x1s[N0 + o]) : 0.f;
    }
  }
  __syncthreads();   // g0 complete; g1p / w1d are dead from here on
  NK_STAMP(11);

  // y3 of the two samples -> LDS (over the dead w1d area): needed for act3' below; requested now, stored before use
  float4 y3v = make_float4(0.f, 0.f, 0.f, 0.f);   // float4 `tid` of [S][1024 / 4] (512 of them)
  if (tid < NK_S * NK_K3 / 4 && (tid >> 8) < ns) y3v = reinterpret_cast<const float4*>(a.y3 + (size_t)b0 * NK_K3)[tid];

  // ================= latent block backward (latent_block.hip's arithmetic) =================
  {
    // dz[s][d] = sum_n g0[s][n] w0[d][n]: S * D outputs, the threads split n
    const int nout = NK_S * D;
    const int ksl = NK_NT / nout;
    const int kc = (N0 + ksl - 1) / ksl;
    const int o = tid % nout, kq = tid / nout;
    if (kq < ksl) {
      const int s = o / D, d = o - s * D;
      const int nlo = kq * kc, nhi = (nlo + kc < N0) ? nlo + kc : N0;
      float a0 = 0.f;
      for (int n = nlo; n < nhi; ++n) a0 = fmaf(g0s[s * N0 + n], w0s[d * N0 + n], a0);
      red[kq * nout + o] = a0;
    }
    __syncthreads();
    if (tid < nout) {
      float g = red[tid];
      for (int w = 1; w < ksl; ++w) g += red[w * nout + tid];
      const int s = tid / D, d = tid - s * D;
      const float loc = pls[s * J + d], raw = pls[s * J + D + d], e_ = es[tid];
      const float sc = softplus_f(raw), zz = zs[tid];
      const float w = klw * fb[s];
      float dloc, dsc;
      if (a.analytic == 2) {
        const float i2 = 1.f / (sc * sc);
        dloc = w * loc * i2;
        dsc = w * (1.f / sc - (1.f + loc * loc) * i2 / sc);
      } else if (a.analytic) { dloc = w * loc; dsc = w * (sc - 1.f / sc); }
      else { dloc = w * zz; dsc = w * (zz * e_ - 1.f / sc); }
      dloc += g; dsc += g * e_;
      if (a.dz_extra != nullptr) { dloc += x2[tid]; dsc += x2[tid] * e_; }
      if (a.dloc_x != nullptr) dloc += xl[tid];
      if (a.dscale_x != nullptr) dsc += xs[tid];
      const float draw = dsc * sigmoid_f(raw);
      const bool live = s < ns;
      dps[s * J + d] = live ? dloc : 0.f;
      dps[s * J + D + d] = live ? draw : 0.f;
      if (live) {
        a.dz[(size_t)(b0 + s) * D + d] = g;
        a.dp[(size_t)(b0 + s) * J + d] = dloc;
        a.dp[(size_t)(b0 + s) * J + D + d] = draw;
      }
    }
    __syncthreads();
  }
  NK_STAMP(12);
  float amx4 = 0.f;
  // dh4[s][k] = (sum_j dp[s][j] wl[k][j]) act4'(h4[s][k])
  if (tid < NK_S * P) {
    const int s = tid / P, k = tid - s * P;
    float c0 = 0.f, c1 = 0.f;
    for (int j = 0; j < J; j += 2) {   // (J = 2 D is even)
      c0 = fmaf(dps[s * J + j], wls[k * J + j], c0);
      c1 = fmaf(dps[s * J + j + 1], wls[k * J + j + 1], c1);
    }
    const float v = (c0 + c1) * odin_act_grad(a.act4, h4s[tid]);
    d4s[tid] = s < ns ? v : 0.f;
    if (s < ns) {
      a.dh4[(size_t)(b0 + s) * P + k] = v;
      amx4 = fabsf(v);
    }
  }
  // this workgroup's partial weight gradients of the two small Dense layers (sums over its samples, s ascending)
  {
    float* row = a.slab0 + (size_t)blockIdx.x * (D * N0 + N0);
    for (int o = tid; o < D * N0; o += NK_NT) {
      const int d = o / N0, n = o - d * N0;
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < NK_S; ++s) acc = fmaf(zs[s * D + d], g0s[s * N0 + n], acc);
      row[o] = acc;
    }
    for (int n = tid; n < N0; n += NK_NT) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < NK_S; ++s) acc += g0s[s * N0 + n];
      row[D * N0 + n] = acc;
    }
  }
  {
    float* row = a.slabl + (size_t)blockIdx.x * (P * J + J);
    for (int o = tid; o < P * J; o += NK_NT) {
      const int k = o / J, j = o - k * J;
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < NK_S; ++s) acc = fmaf(h4s[s * P + k], dps[s * J + j], acc);
      row[o] = acc;
    }
    for (int j = tid; j < J; j += NK_NT) {
      float acc = 0.f;
#pragma unroll
      for (int s = 0; s < NK_S; ++s) acc += dps[s * J + j];
      row[P * J + j] = acc;
    }
  }
  if (tid < NK_S * NK_K3 / 4) reinterpret_cast<float4*>(y3s)[tid] = y3v;
  __syncthreads();   // dh4 (d4s) and y3 (y3s) are in LDS
  odin_amax_commit_wg(a.dh4_amax, amx4, tid, NK_NT, red, blockIdx.x);
  NK_STAMP(13);

  // ================= the projection's data gradient: dy3[s][k] = (sum_n dh4[s][n] W4[k][n]) act3'(y3[s][k]) =================
  // a group of 8 lanes owns a row k of W4 (P floats, contiguous): lane i of the group reads the float4 columns i, i + 8, ...
  // (128 contiguous bytes per group and load), 128 rows per pass, 8 passes
  float amx3 = 0.f;
  {
    const int gi = tid & 7, grp = tid >> 3;   // 128 groups
    constexpr int PB = NQ == 4 ? 2 : 1;       // passes whose loads are in flight together (32 registers of W4 rows)
    float dv0[4 * NQ], dv1[4 * NQ];           // this lane's columns of dh4, both samples
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
      const float4 t0 = *reinterpret_cast<const float4*>(d4s + 4 * (gi + 8 * c));
      const float4 t1 = *reinterpret_cast<const float4*>(d4s + P + 4 * (gi + 8 * c));
      dv0[4 * c] = t0.x; dv0[4 * c + 1] = t0.y; dv0[4 * c + 2] = t0.z; dv0[4 * c + 3] = t0.w;
      dv1[4 * c] = t1.x; dv1[4 * c + 1] = t1.y; dv1[4 * c + 2] = t1.z; dv1[4 * c + 3] = t1.w;
    }
    const float4* w4 = reinterpret_cast<const float4*>(a.w4);
    for (int pass = 0; pass < 8; pass += PB) {
      float4 wv[PB][NQ];
#pragma unroll
      for (int h = 0; h < PB; ++h) {
        const int k = (pass + h) * 128 + grp;
#pragma unroll
        for (int c = 0; c < NQ; ++c) wv[h][c] = w4[(size_t)k * (8 * NQ) + gi + 8 * c];
      }
#pragma unroll
      for (int h = 0; h < PB; ++h) {
        const int k = (pass + h) * 128 + grp;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
          s0 = fmaf(wv[h][c].x, dv0[4 * c], s0); s0 = fmaf(wv[h][c].y, dv0[4 * c + 1], s0);
          s0 = fmaf(wv[h][c].z, dv0[4 * c + 2], s0); s0 = fmaf(wv[h][c].w, dv0[4 * c + 3], s0);
          s1 = fmaf(wv[h][c].x, dv1[4 * c], s1); s1 = fmaf(wv[h][c].y, dv1[4 * c + 1], s1);
          s1 = fmaf(wv[h][c].z, dv1[4 * c + 2], s1); s1 = fmaf(wv[h][c].w, dv1[4 * c + 3], s1);
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { s0 += __shfl_xor(s0, m); s1 += __shfl_xor(s1, m); }
        if (gi < NK_S) {
          const float v = (gi == 0 ? s0 : s1) * odin_act_grad(a.act3, y3s[gi * NK_K3 + k]);
          d3s[gi * NK_K3 + k] = v;
          if (gi < ns) {
            a.dy3[(size_t)(b0 + gi) * NK_K3 + k] = v;
            amx3 = fmaxf(amx3, fabsf(v));
          }
        }
      }
    }
  }
  NK_STAMP(14);
  // the workgroup's max |dy3|: its planes are carried times the power of two that brings it to [2^14, 2^15)
  {
    const float m = odin_wave_max64(amx3);
    if (lane == 0) red[wave] = m;
  }
  __syncthreads();
  float wgmax = red[0];
#pragma unroll
  for (int w = 1; w < NK_NT / 64; ++w) wgmax = fmaxf(wgmax, red[w]);
  __syncthreads();   // (red is reused by the commits below)
  if (a.dy3_amax != nullptr && tid == 0)
    atomicMax(a.dy3_amax + (blockIdx.x & (ODIN_RANGE_SLOTS - 1)) * ODIN_RANGE_STRIDE, odin_fbits(wgmax));

  // ================= conv3's data gradient on the matrix pipe, by stride class =================
  // class (ry, rx): input pixels (2 a + ry, 2 b + rx), a, b in 0..3: rows m = (s, a, b) of a 32-row tile; taps
  // kh in {1 - ry, 3 - ry} read dy3 row a + ry (kh = 1 - ry) / a + ry - 1 (kh = 3 - ry), columns alike: the images are
  // staged zero-bordered ([6][6]) as planes.  wave = (class, column block nt, kh choice): 2 taps x 4 k-steps x 3 MFMAs
  const int gk3 = odin_range_shift(odin_fbits(wgmax));
  {
    const float s3 = odin_pow2(gk3), s3k = odin_pow2(gk3 + 11);
    // planes [S][6][6][64 f16]: border pixels zero, interior from d3s (fp32 [S][16][64])
    for (int e = tid; e < NK_S * 36 * 16; e += NK_NT) {
      const int c4 = e & 15, pp = e >> 4;
      const int s = pp / 36, rem = pp - s * 36;
      const int pr = rem / 6, pc = rem - pr * 6;
      const bool in = pr >= 1 && pr <= 4 && pc >= 1 && pc <= 4;
      const float4 v = in ? *reinterpret_cast<const float4*>(d3s + s * NK_K3 + ((pr - 1) * 4 + pc - 1) * NK_C + 4 * c4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
      u32x2 h, l;
      odin_split_h4<true>(v, s3, s3k, h, l);
      char* d = q3 + pp * NK_PITCH + c4 * 8;
      *reinterpret_cast<u32x2*>(d) = h;
      *reinterpret_cast<u32x2*>(d + NK_QPLANE) = l;
    }
  }
  const int cls = wave >> 2, nt = (wave >> 1) & 1, kh_sel = wave & 1;
  const int ry = cls >> 1, rx = cls & 1, l31 = lane & 31, kg = lane >> 5;
  // weight fragments: B[k = co][n = ci] = W3[tap][ci][co].  The operand layout wants 8 consecutive co of ONE ci per lane:
  // read that way from memory every load instruction touches 32 different rows (8 us for 256 KB per workgroup, measured).
  // Instead a half tap [32 ci][32 co] (4 KB) is loaded row-contiguous -- 8 rows x 128 bytes per instruction -- and
  // transposed through this wave's own 4 KB of LDS (the slot its partial tile takes afterwards): 16-byte slot s of row
  // r at r * 8 + (s ^ (r & 7))
  f32x4 wq[16];    // [tap t][half hq][u]: filled and consumed with compile-time indices only (native vectors: a
                   // float4 array went through scratch memory)
  const int kh = kh_sel ? 3 - ry : 1 - ry;
  nk_static_for<16>([&](auto II) __attribute__((always_inline)) {
    constexpr int i = decltype(II)::value, t = i >> 3, hq = (i >> 2) & 1, u = i & 3;
    const int kw = t ? 3 - rx : 1 - rx;
    const f32x4* base = reinterpret_cast<const f32x4*>(a.w3 + ((size_t)(kh * 4 + kw) * NK_C + 32 * nt) * NK_C);
    const int f = lane + 64 * u, r = f >> 3, sl = f & 7;   // row (ci) and 16-byte slot of the half tap
    wq[i] = base[r * 16 + 8 * hq + sl];
  });
  // act2'(x) factors of this thread's 8 outputs (x = the layer below's output), requested ahead of the MFMAs
  float xa[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + NK_NT * u, ci = e & 63, mm = (e >> 6) & 31, c2 = e >> 11;
    const int s = mm >> 4, aa = (mm >> 2) & 3, bb = mm & 3;
    const int iy = 2 * aa + (c2 >> 1), ix = 2 * bb + (c2 & 1);
    xa[u] = (a.act2 != 0 && s < ns) ? a.x[((size_t)(b0 + s) * 64 + iy * NK_HW + ix) * NK_C + ci] : 1.f;
  }
  __syncthreads();   // planes complete
  NK_STAMP(15);
  {
    const int m = l31, s = m >> 4, aa = (m >> 2) & 3, bb = m & 3;
    const int prow = aa + ry + (kh_sel ? 0 : 1);   // padded dy3 row of this wave's kh
    f32x16 acc = f32x16_zero(), acx = f32x16_zero();
    float* mine = part + wave * 1024;   // (over the dead g1p area: disjoint from the planes at 64 KB)
    char* stg = reinterpret_cast<char*>(mine);
    nk_static_for<4>([&](auto RR) __attribute__((always_inline)) {
      constexpr int t = decltype(RR)::value >> 1, hq = decltype(RR)::value & 1;
      const int pcol = bb + rx + (t ? 0 : 1);
      const char* ap = q3 + (s * 36 + prow * 6 + pcol) * NK_PITCH + 16 * kg;
      odin_wave_sync();   // the previous round's fragment reads are done
      nk_static_for<4>([&](auto UU) __attribute__((always_inline)) {
        constexpr int u = decltype(UU)::value;
        const int f = lane + 64 * u, r = f >> 3, sl = f & 7;
        *reinterpret_cast<f32x4*>(stg + (r * 8 + (sl ^ (r & 7))) * 16) = wq[(t * 2 + hq) * 4 + u];
      });
      odin_wave_sync();
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = 2 * hq + jj, s0 = 4 * jj + 2 * kg;
        const float4 b0v = *reinterpret_cast<const float4*>(stg + (l31 * 8 + (s0 ^ (l31 & 7))) * 16);
        const float4 b1v = *reinterpret_cast<const float4*>(stg + (l31 * 8 + ((s0 + 1) ^ (l31 & 7))) * 16);
        const u32x4 ah = *reinterpret_cast<const u32x4*>(ap + 32 * j);
        const u32x4 al = *reinterpret_cast<const u32x4*>(ap + NK_QPLANE + 32 * j);
        u32x2 h0, l0, h1, l1;
        odin_split_h4<false>(b0v, 1.f, ODIN_LO_SCALE, h0, l0);
        odin_split_h4<false>(b1v, 1.f, ODIN_LO_SCALE, h1, l1);
        u32x4 bh, bl;
        bh.x = h0.x; bh.y = h0.y; bh.z = h1.x; bh.w = h1.y;
        bl.x = l0.x; bl.y = l0.y; bl.z = l1.x; bl.w = l1.y;
        acx = mfma32_f16(ah, bl, acx);
        acc = mfma32_f16(ah, bh, acc);
        acx = mfma32_f16(al, bh, acx);
      }
    });
    odin_wave_sync();   // the last fragments are read: the partial tile takes the slot
    const float o_s = odin_pow2(-gk3), o_sx = odin_pow2(-gk3 - 11);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = fmaf(acx[r], o_sx, acc[r] * o_s);
  }
  __syncthreads();
  NK_STAMP(16);
  float amxx = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = tid + NK_NT * u, ci = e & 63, mm = (e >> 6) & 31, c2 = e >> 11;
    const int ntile = ci >> 5, col = ci & 31, hh = (mm >> 2) & 1, r = (mm & 3) + 4 * (mm >> 3);
    const float* src = part + ((c2 * 2 + ntile) * 2) * 1024 + r * 64 + col + 32 * hh;
    float v = src[0] + src[1024];
    v *= odin_act_grad(a.act2, xa[u]);
    const int s = mm >> 4, aa = (mm >> 2) & 3, bb = mm & 3;
    const int iy = 2 * aa + (c2 >> 1), ix = 2 * bb + (c2 & 1);
    if (s < ns) {
      a.dx[((size_t)(b0 + s) * 64 + iy * NK_HW + ix) * NK_C + ci] = v;
      amxx = fmaxf(amxx, fabsf(v));
    }
  }
  odin_amax_commit_wg(a.dx_amax, amxx, tid, NK_NT, red, blockIdx.x);
  NK_STAMP(17);
}

template <typename K>
int nk_set_lds(K kern, size_t bytes) {
#ifndef ODIN_SIM
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess)
    return odin_fail(-4, "neck: cannot raise the dynamic LDS limit");
#else
  (void)kern; (void)bytes;
#endif
  return 0;
}

}  // namespace

// Workgroups of a neck launch (= slab rows of the backward launch's small weight gradients); 0 when the shapes are
// outside the fused regime.  Conv2D 64 -> 64 (k4, s2, SAME) on 8 x 8, Dense 1024 -> P with P in {128, 256}, D <= 32,
// Dense D -> 16 C0 with C0 in {8, 16}, Conv2DTranspose C0 -> 64 (k4, s2, SAME) on 4 x 4.
extern "C" int odin_neck_rows(int B, int P, int D, int C0) {
  if (ODIN_DIAG_ENV("ODIN_NONECK")) return 0;
  if (B < 1 || (P != 128 && P != 256) || D < 1 || D > 32 || (C0 != 8 && C0 != 16)) return 0;
  if (P * 2 * D > 4 * NK_NT || D * NK_OP * C0 > 2 * NK_NT) return 0;
  const int rows = (B + NK_S - 1) / NK_S;
  return rows <= ODIN_MAX_COLSUM_BLOCKS ? rows : 0;
}

static long long* g_nk_stamps = nullptr;
// diagnostics: workgroup 0 of the neck launches records wall-clock stamps (100 MHz) at its phase boundaries into
// buf[0..6] (forward) / buf[8..17] (backward); NULL switches it off
extern "C" int odin_debug_set_neck_stamps(void* buf) {
  g_nk_stamps = (long long*)buf;
  return 0;
}

extern "C" int odin_neck_bwd(const odin_neck_args* args, void* stream) {
  const odin_neck_args& a = *args;
  const int rows = odin_neck_rows(a.B, a.P, a.D, a.C0);
  if (rows == 0) return odin_fail(-2, "neck_bwd: shapes outside the fused regime");
  if ((((size_t)a.w4 | (size_t)a.w3 | (size_t)a.dy1 | (size_t)a.x | (size_t)a.slab1) & 15) != 0)
    return odin_fail(-2, "neck_bwd: dy1 / x / w3 / w4 / slab1 must be 16-byte aligned");
  NeckFwd q;
  memset(&q, 0, sizeof(q));
  q.a = a;
  q.N0 = NK_OP * a.C0;
  q.w1_al = (((size_t)a.w1) & 15) == 0;
  q.stamps = g_nk_stamps;
  const NKBwdLds L(a.P, a.D, a.C0);
  const size_t lds = (size_t)L.end;
#define NK_BWD(C0_, NQ_)                                                                  \
  do {                                                                                    \
    if (int rc = nk_set_lds(&neck_bwd_kernel<C0_, NQ_>, lds)) return rc;                  \
    ODIN_LAUNCH((neck_bwd_kernel<C0_, NQ_>), dim3(rows), dim3(NK_NT), lds, stream, q);    \
  } while (0)
  if (a.C0 == 8 && a.P == 128) NK_BWD(8, 4);
  else if (a.C0 == 8) NK_BWD(8, 8);
  else if (a.P == 128) NK_BWD(16, 4);
  else NK_BWD(16, 8);
#undef NK_BWD
  return odin_check_launch("neck_bwd");
}

extern "C" int odin_neck_fwd(const odin_neck_args* args, void* stream) {
  const odin_neck_args& a = *args;
  const int rows = odin_neck_rows(a.B, a.P, a.D, a.C0);
  if (rows == 0) return odin_fail(-2, "neck_fwd: shapes outside the fused regime");
  if ((((size_t)a.w4 | (size_t)a.x) & 15) != 0) return odin_fail(-2, "neck_fwd: x / w4 must be 16-byte aligned");
  NeckFwd q;
  memset(&q, 0, sizeof(q));
  q.a = a;
  q.k0 = (unsigned)a.seed; q.k1 = (unsigned)(a.seed >> 32);
  q.N0 = NK_OP * a.C0;
  q.w1_al = (((size_t)a.w1) & 15) == 0;
  q.stamps = g_nk_stamps;
  const NKFwdLds L(a.P, a.D, a.C0);
  const size_t lds = (size_t)L.end;
  if (a.C0 == 8) {
    if (int rc = nk_set_lds(&neck_fwd_kernel<8>, lds)) return rc;
    ODIN_LAUNCH((neck_fwd_kernel<8>), dim3(rows), dim3(NK_NT), lds, stream, q);
  } else {
    if (int rc = nk_set_lds(&neck_fwd_kernel<16>, lds)) return rc;
    ODIN_LAUNCH((neck_fwd_kernel<16>), dim3(rows), dim3(NK_NT), lds, stream, q);
  }
  return odin_check_launch("neck_fwd");
}
