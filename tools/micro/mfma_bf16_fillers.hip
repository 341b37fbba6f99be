// Microbenchmark: VALU fillers between bf16 MFMAs (one wave per SIMD), and a VALU-only partner wave
// beside a bf16 MFMA stream (two waves per SIMD) -- the bf16 counterpart of mfma_fillers.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));

template <int SHAPE, int NV, int KIND, int PARTNER>  // SHAPE 0: 32x32x16 bf16, 1: 16x16x32 bf16
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f16v b0 = {0}, b1 = {0};
  f4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  s8v x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); y[i] = (short)(3 * threadIdx.x + i); }
  float fx = threadIdx.x * 1e-3f, fy = 1.0f + fx;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = fx + i;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (SHAPE == 0) {
          if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
        } else {
          if (u & 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(x), "v"(y));
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int j = (u * NV + q) & 7;
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(fx), "v"(fy));
          if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        }
      }
    }
  } else if (PARTNER) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16 * PARTNER; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[u & 7]) : "v"(fx), "v"(fy));
    }
  }
  long long t1 = clock64();
  float s = b0[0] + b1[1] + c0[0] + c1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int NV, int KIND, int PARTNER>
void run() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(cyc, 0, 64);
  const int iters = 1000;
  const int threads = PARTNER ? 512 : 256;
  for (int r = 0; r < 2; ++r) { k<SHAPE, NV, KIND, PARTNER><<<256, threads>>>(out, cyc, iters); (void)hipDeviceSynchronize(); }
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%s + %2d x %-9s partner %3d FMA/16 MFMA: %.1f ticks per MFMA (MFMA wave), partner wave %.1f ticks per 16 MFMA\n",
         SHAPE ? "16x16x32_bf16" : "32x32x16_bf16", NV, KIND == 3 ? "v_exp_f32" : "v_fma_f32", 16 * PARTNER,
         (double)h[0] / (iters * 16.0), (double)h[4] / iters);
  (void)hipFree(out); (void)hipFree(cyc);
}

template <int SHAPE>
void sweep() {
  run<SHAPE, 0, 0, 0>(); run<SHAPE, 1, 0, 0>(); run<SHAPE, 2, 0, 0>(); run<SHAPE, 3, 0, 0>(); run<SHAPE, 4, 0, 0>();
  run<SHAPE, 5, 0, 0>(); run<SHAPE, 6, 0, 0>(); run<SHAPE, 8, 0, 0>(); run<SHAPE, 12, 0, 0>();
  run<SHAPE, 1, 3, 0>(); run<SHAPE, 2, 3, 0>(); run<SHAPE, 4, 3, 0>();
  run<SHAPE, 0, 0, 1>(); run<SHAPE, 0, 0, 2>(); run<SHAPE, 0, 0, 4>(); run<SHAPE, 0, 0, 6>(); run<SHAPE, 0, 0, 8>();
}

int main() { sweep<0>(); sweep<1>(); return 0; }
