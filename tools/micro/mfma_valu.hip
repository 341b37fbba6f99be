// Microbenchmark: can one wave overlap its own VALU / transcendental / LDS instructions with a
// stream of v_mfma_f32_16x16x4_f32?  One wave per SIMD (256 threads per workgroup, 1 workgroup per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NT, int PARTNER>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[(u + q) & 7] = fmaf(v[(u + q) & 7], 1.0001f, 0.5f);
#pragma unroll
        for (int q = 0; q < NT; ++q) v[(u + q + 3) & 7] = __builtin_amdgcn_exp2f(v[(u + q + 3) & 7]);
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x2, NV, 0);
        __builtin_amdgcn_sched_group_barrier(0x400, NT, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if (PARTNER) {
    // partner wave on the same SIMD: VALU only
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16 * PARTNER; ++u) v[u & 7] = fmaf(v[u & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = clock64();
  float s = a0[0] + a1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int NV, int NT, int PARTNER>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  hipMemset(cyc, 0, 64);
  const int iters = 2000;
  k<NV, NT, PARTNER><<<256, threads>>>(out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<NV, NT, PARTNER><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-34s NV=%d NT=%d partner=%d: %.1f ticks/MFMA (wave0), partner wave4 %.1f ticks per 16 FMA-groups, %.3f ns/MFMA\n", name, NV, NT,
         PARTNER, (double)h[0] / (iters * 16), (double)h[4] / iters, ms * 1e6 / (iters * 16));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>("bare MFMA", 256);
  run<2, 0, 0>("MFMA + 2 FMA", 256);
  run<4, 0, 0>("MFMA + 4 FMA", 256);
  run<6, 0, 0>("MFMA + 6 FMA", 256);
  run<8, 0, 0>("MFMA + 8 FMA", 256);
  run<0, 1, 0>("MFMA + 1 exp", 256);
  run<0, 2, 0>("MFMA + 2 exp", 256);
  run<2, 1, 0>("MFMA + 2 FMA + 1 exp", 256);
  run<4, 1, 0>("MFMA + 4 FMA + 1 exp", 256);
  run<0, 0, 1>("bare MFMA, partner 16 FMA/iter", 512);
  run<0, 0, 4>("bare MFMA, partner 64 FMA/iter", 512);
  run<0, 0, 8>("bare MFMA, partner 128 FMA/iter", 512);
  return 0;
}
