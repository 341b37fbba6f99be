// odin_latent_math.h -- device helpers shared by pointwise.hip and latent_block.hip: softplus / sigmoid as the
// latent kernels evaluate them, the Philox4x32-10 counter RNG and its Box-Muller normals (one counter = 4 values).
#pragma once
#include "odin_device.h"

namespace {

__device__ __forceinline__ float softplus_f(float x) {
  return fmaxf(x, 0.f) + log1pf(odin_exp(-fabsf(x)));
}
__device__ __forceinline__ float sigmoid_f(float x) {
  float e = odin_exp(-fabsf(x));
  float s = 1.f / (1.f + e);
  return x >= 0.f ? s : e * s;
}
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                              unsigned k0, unsigned k1, unsigned out[4]) {
  const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    unsigned n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// the 4 standard normals of counter (c0, c1) at `step`: elements 4*c .. 4*c+3 of the stream odin_rng_normal writes
__device__ __forceinline__ void odin_normal4(unsigned c0, unsigned c1, unsigned step, unsigned k0, unsigned k1,
                                             float v[4]) {
  unsigned r[4];
  philox4x32_10(c0, c1, step, 0u, k0, k1, r);
  float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.f / 16777216.f);
  float u1 = ((float)(r[1] >> 8) + 0.5f) * (1.f / 16777216.f);
  float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.f / 16777216.f);
  float u3 = ((float)(r[3] >> 8) + 0.5f) * (1.f / 16777216.f);
  float ra = sqrtf(-2.f * odin_log(u0)), rb = sqrtf(-2.f * odin_log(u2));
  v[0] = ra * cosf(6.2831853071795865f * u1);
  v[1] = ra * sinf(6.2831853071795865f * u1);
  v[2] = rb * cosf(6.2831853071795865f * u3);
  v[3] = rb * sinf(6.2831853071795865f * u3);
}

}  // namespace
