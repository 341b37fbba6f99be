// Microbenchmark: independent VALU "fillers" between the MFMAs of ONE wave per SIMD -- how many are
// free, per MFMA shape?  (v_mfma_f32_16x16x4_f32: 8 passes; v_mfma_f32_32x32x2_f32: 16 passes.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int SHAPE, int NV, int KIND>  // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: ds_read_b128 (LDS), 3: v_exp_f32
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[4096];
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  f16v b0 = {0}, b1 = {0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  float v[8];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pv[8];
  f4 lv[8];
  for (int i = 0; i < 8; ++i) { v[i] = x + i; pv[i] = f2{x, y}; lv[i] = f4{0, 0, 0, 0}; }
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
  __syncthreads();
  const unsigned lp = (threadIdx.x & 63) * 16;  // byte address inside the (only) LDS array
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (SHAPE == 0) {
        if (u & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
      } else {
        if (u & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
      }
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int j = (u * NV + q) & 7;
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(x), "v"(y));
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[j]) : "v"(pv[(j + 1) & 7]));
        if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[j]) : "v"(lp));
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
      }
    }
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = clock64();
  float s = a0[0] + a1[1] + b0[0] + b1[1];
  for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0] + lv[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int SHAPE, int NV, int KIND>
void run() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) { k<SHAPE, NV, KIND><<<256, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize(); }
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  static const char* kn[] = {"v_fma_f32", "v_pk_fma_f32", "ds_read_b128", "v_exp_f32"};
  printf("%s + %d x %-13s: %.1f ticks per MFMA\n", SHAPE ? "32x32x2_f32 (64 cyc)" : "16x16x4_f32 (32 cyc)", NV, kn[KIND],
         (double)h[0] / (iters * 16.0));
  (void)hipFree(out); (void)hipFree(cyc);
}

template <int SHAPE, int KIND>
void sweep() {
  run<SHAPE, 0, KIND>(); run<SHAPE, 1, KIND>(); run<SHAPE, 2, KIND>(); run<SHAPE, 3, KIND>(); run<SHAPE, 4, KIND>();
  run<SHAPE, 6, KIND>(); run<SHAPE, 8, KIND>(); run<SHAPE, 12, KIND>();
}

int main() {
  sweep<0, 0>(); sweep<1, 0>();
  sweep<0, 1>(); sweep<1, 1>();
  sweep<0, 3>(); sweep<1, 3>();
  sweep<0, 2>(); sweep<1, 2>();
  return 0;
}
