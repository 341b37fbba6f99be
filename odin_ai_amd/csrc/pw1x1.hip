// pw1x1.hip -- 1x1 convolutions with a handful of output maps (the decoders' last layer: Conv2D(
// n_channels * n_params, 1, linear), image_networks.py:505-511 / 697-703 -- 1 to 6 maps over 32
// channels).  On the matrix-core path such a layer fills 2 of 32 output columns and a [B*H*W, 32]
// x [32, 2] product takes milliseconds; it is a pure streaming problem: every byte of the
// [pixels, Cin] operand is read once (forward, weight gradient) or written once (data gradient).
//
// Lane <-> (pixel, channel quad): Q = Cin/4 consecutive lanes share a pixel and each owns 4 of its
// channels, so every 16-byte access of a wave is contiguous (64 lanes = 64/Q whole pixels).
//   forward :  y[p][o] = sum_c x[p][c] w[c][o] + b[o]          (partial dots reduced over the Q
//                                                                lanes of the pixel by shuffles)
//   dgrad   :  dx[p][c] = (sum_o dy[p][o] w[c][o]) * act'(aux[p][c]), + per-workgroup column sums
//   wgrad   :  dW[c][o] = sum_p x[p][c] dy[p][o], db[o] = sum_p dy[p][o]: per-lane accumulators,
//              fixed-order reduction over the workgroup, one slab row per workgroup.
// U independent pixel groups per loop iteration keep U x 16 bytes in flight per lane.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

constexpr int PW_MAXCO = 8;
constexpr int PW_U = 4;

template <int CO>
__global__ __launch_bounds__(256) void pw1x1_fwd_kernel(const float4* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, long npix, int Q,
                                                        int act) {
  __shared__ float wl[64 * PW_MAXCO];
  const int CI = 4 * Q;
  for (int e = threadIdx.x; e < CI * CO; e += 256) wl[e] = w[e];
  __syncthreads();
  const int q = threadIdx.x % Q;
  float wr[4][CO];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < CO; ++o) wr[k][o] = wl[(4 * q + k) * CO + o];
  float br[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) br[o] = bias != nullptr ? bias[o] : 0.f;
  const long ppb = 256 / Q;  // pixels per workgroup per group
  const long stride = (long)gridDim.x * ppb * PW_U;
  // (the loop bound is workgroup-uniform: every lane takes part in the shuffles below)
  for (long base = (long)blockIdx.x * ppb * PW_U; base < npix; base += stride) {
    const long p0 = base + threadIdx.x / Q;
    float4 v[PW_U];
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      const long p = p0 + u * ppb;
      v[u] = p < npix ? x[p * Q + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      const long p = p0 + u * ppb;
      float acc[CO];
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        float a = v[u].x * wr[0][o];
        a = fmaf(v[u].y, wr[1][o], a);
        a = fmaf(v[u].z, wr[2][o], a);
        a = fmaf(v[u].w, wr[3][o], a);
        // fixed-order butterfly over the Q lanes of the pixel (wave-uniform trip count)
        for (int m = 1; m < Q; m <<= 1) a += __shfl_xor(a, m);
        acc[o] = a;
      }
      if (q == 0 && p < npix) {
#pragma unroll
        for (int o = 0; o < CO; ++o) y[p * CO + o] = odin_act(act, acc[o] + br[o]);
      }
    }
  }
}

template <int CO>
__global__ __launch_bounds__(256) void pw1x1_dgrad_kernel(const float* __restrict__ dy,
                                                          const float* __restrict__ w,
                                                          const float4* __restrict__ aux, int aux_act,
                                                          float4* __restrict__ dx,
                                                          float* __restrict__ colsum, long npix,
                                                          int Q) {
  __shared__ float wl[64 * PW_MAXCO];
  __shared__ float4 red[256];
  const int CI = 4 * Q;
  for (int e = threadIdx.x; e < CI * CO; e += 256) wl[e] = w[e];
  __syncthreads();
  const int q = threadIdx.x % Q;
  float wr[4][CO];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < CO; ++o) wr[k][o] = wl[(4 * q + k) * CO + o];
  const long ppb = 256 / Q;
  const long stride = (long)gridDim.x * ppb * PW_U;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long base = (long)blockIdx.x * ppb * PW_U; base < npix; base += stride) {
    const long p0 = base + threadIdx.x / Q;
    float g[PW_U][CO];
    float4 a[PW_U];
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      const long p = p0 + u * ppb;
      const bool ok = p < npix;
#pragma unroll
      for (int o = 0; o < CO; ++o) g[u][o] = ok ? dy[p * CO + o] : 0.f;
      a[u] = (ok && aux != nullptr) ? aux[p * Q + q] : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      const long p = p0 + u * ppb;
      if (p >= npix) continue;
      float r[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = 0.f;
#pragma unroll
        for (int o = 0; o < CO; ++o) t = fmaf(g[u][o], wr[k][o], t);
        r[k] = t;
      }
      if (aux != nullptr) {
        r[0] *= odin_act_grad(aux_act, a[u].x);
        r[1] *= odin_act_grad(aux_act, a[u].y);
        r[2] *= odin_act_grad(aux_act, a[u].z);
        r[3] *= odin_act_grad(aux_act, a[u].w);
      }
      dx[p * Q + q] = make_float4(r[0], r[1], r[2], r[3]);
      cs.x += r[0]; cs.y += r[1]; cs.z += r[2]; cs.w += r[3];
    }
  }
  if (colsum != nullptr) {
    red[threadIdx.x] = cs;
    __syncthreads();
    if (threadIdx.x < Q) {  // thread q sums the 256/Q lanes that own quad q, in lane order
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = threadIdx.x; i < 256; i += Q) {
        t.x += red[i].x; t.y += red[i].y; t.z += red[i].z; t.w += red[i].w;
      }
      float* row = colsum + (size_t)blockIdx.x * CI + 4 * threadIdx.x;
      row[0] = t.x; row[1] = t.y; row[2] = t.z; row[3] = t.w;
    }
  }
}

template <int CO>
__global__ __launch_bounds__(256) void pw1x1_wgrad_kernel(const float4* __restrict__ x,
                                                          const float* __restrict__ dy,
                                                          float* __restrict__ slab, long npix,
                                                          int Q) {
  ODIN_DYN_SMEM(float, red);  // [256][4 * CO + CO]
  const int CI = 4 * Q;
  const int q = threadIdx.x % Q;
  const long ppb = 256 / Q;
  const long stride = (long)gridDim.x * ppb * PW_U;
  float acc[4][CO], db[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    db[o] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k][o] = 0.f;
  }
  for (long base = (long)blockIdx.x * ppb * PW_U; base < npix; base += stride) {
    const long p0 = base + threadIdx.x / Q;
    float4 v[PW_U];
    float g[PW_U][CO];
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
      const long p = p0 + u * ppb;
      const bool ok = p < npix;
      v[u] = ok ? x[p * Q + q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int o = 0; o < CO; ++o) g[u][o] = ok ? dy[p * CO + o] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < PW_U; ++u) {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        acc[0][o] = fmaf(v[u].x, g[u][o], acc[0][o]);
        acc[1][o] = fmaf(v[u].y, g[u][o], acc[1][o]);
        acc[2][o] = fmaf(v[u].z, g[u][o], acc[2][o]);
        acc[3][o] = fmaf(v[u].w, g[u][o], acc[3][o]);
        db[o] += g[u][o];  // (every lane of the pixel carries the same sum; lane q == 0's is used)
      }
    }
  }
  constexpr int RW = 5 * CO;
  float* mine = red + threadIdx.x * RW;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int o = 0; o < CO; ++o) mine[k * CO + o] = acc[k][o];
#pragma unroll
  for (int o = 0; o < CO; ++o) mine[4 * CO + o] = db[o];
  __syncthreads();
  // slab row: [dW (Cin, Cout) | db (Cout)]; entry e of dW = (c = 4q + k, o)
  float* row = slab + (size_t)blockIdx.x * (CI * CO + CO);
  for (int e = threadIdx.x; e < CI * CO + CO; e += 256) {
    float t = 0.f;
    if (e < CI * CO) {
      const int c = e / CO, o = e - c * CO;
      const int qq = c >> 2, k = c & 3;
      for (int i = qq; i < 256; i += Q) t += red[i * RW + k * CO + o];
    } else {
      const int o = e - CI * CO;
      for (int i = 0; i < 256; i += Q) t += red[i * RW + 4 * CO + o];
    }
    row[e] = t;
  }
}

int pw_grid(long npix, int Q, int cap) {
  const long groups = (npix * Q + 256L * PW_U - 1) / (256L * PW_U);
  long g = groups < cap ? groups : cap;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

bool odin_pw1x1_applicable(const odin_conv_desc* d) {
  const int ci = d->Cin;
  return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->Cout >= 1 && d->Cout <= PW_MAXCO &&
         (ci == 8 || ci == 16 || ci == 32 || ci == 64) && !d->center && d->H == d->OH &&
         d->W == d->OW && !ODIN_DIAG_ENV("ODIN_NOPW1X1");
}

#define ODIN_PW_SWITCH(CALL)                                                         \
  switch (d->Cout) {                                                                 \
    case 1: CALL(1); break;                                                          \
    case 2: CALL(2); break;                                                          \
    case 3: CALL(3); break;                                                          \
    case 4: CALL(4); break;                                                          \
    case 5: CALL(5); break;                                                          \
    case 6: CALL(6); break;                                                          \
    case 7: CALL(7); break;                                                          \
    default: CALL(8); break;                                                         \
  }

int odin_pw1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                   const odin_conv_desc* d, void* stream) {
  const long npix = (long)d->B * d->H * d->W;
  const int Q = d->Cin / 4;
  const int grid = pw_grid(npix, Q, 4 * odin_num_cus());
#define CALL(CO_)                                                                              \
  ODIN_LAUNCH((pw1x1_fwd_kernel<CO_>), dim3(grid), dim3(256), 0, stream, (const float4*)x, w, bias, \
              y, npix, Q, d->act)
  ODIN_PW_SWITCH(CALL)
#undef CALL
  return odin_check_launch("pw1x1_fwd");
}

int odin_pw1x1_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                     float* colsum_slab, int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  const long npix = (long)d->B * d->H * d->W;
  const int Q = d->Cin / 4;
  int cap = 4 * odin_num_cus();
  if (colsum_slab != nullptr || slab_rows_out != nullptr)
    cap = cap < ODIN_MAX_COLSUM_BLOCKS ? cap : ODIN_MAX_COLSUM_BLOCKS;
  const int grid = pw_grid(npix, Q, cap);
  if (slab_rows_out) *slab_rows_out = grid;
  if (dx == nullptr) return 0;  // dry run
  const float4* auxp = (aux != nullptr && aux_act != 0) ? (const float4*)aux : (const float4*)nullptr;
#define CALL(CO_)                                                                              \
  ODIN_LAUNCH((pw1x1_dgrad_kernel<CO_>), dim3(grid), dim3(256), 0, stream, dy, w, auxp, aux_act,   \
              (float4*)dx, colsum_slab, npix, Q)
  ODIN_PW_SWITCH(CALL)
#undef CALL
  return odin_check_launch("pw1x1_dgrad");
}

int odin_pw1x1_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out,
                     const odin_conv_desc* d, void* stream) {
  const long npix = (long)d->B * d->H * d->W;
  const int Q = d->Cin / 4;
  const int grid = pw_grid(npix, Q, ODIN_MAX_SLAB_BLOCKS);
  if (slab_rows_out) *slab_rows_out = grid;
  if (slab == nullptr) return 0;  // dry run
  const size_t lds = (size_t)256 * 5 * d->Cout * 4;
#define CALL(CO_)                                                                              \
  ODIN_LAUNCH((pw1x1_wgrad_kernel<CO_>), dim3(grid), dim3(256), lds, stream, (const float4*)x, dy, \
              slab, npix, Q)
  ODIN_PW_SWITCH(CALL)
#undef CALL
  return odin_check_launch("pw1x1_wgrad");
}
