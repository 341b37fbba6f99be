// Microbenchmark: two waves per SIMD (512 threads per workgroup, one workgroup per CU), each
// alternating a stream of NM v_mfma_f32_16x16x4_f32 with a block of NVB VALU instructions (an
// "epilogue").  Reports ticks per MFMA seen by the SIMD (both waves together).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NM, int NVB, int NVI, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (PRIO == 1) { if (wave < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
    if (PRIO == 2) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int u = 0; u < NM; ++u) {
      if (u & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NVI; ++q) v[(u + q) & 7] = fmaf(v[(u + q) & 7], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x2, NVI, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int u = 0; u < NVB; ++u) v[u & 7] = fmaf(v[u & 7], 1.0001f, 0.5f);
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  float s = a0[0] + a1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int NM, int NVB, int NVI, int PRIO>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(cyc, 0, 64);
  const int iters = 500;
  k<NM, NVB, NVI, PRIO><<<256, threads>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  k<NM, NVB, NVI, PRIO><<<256, threads>>>(out, cyc, iters);
  (void)hipDeviceSynchronize();
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const int waves_per_simd = threads / 256;
  long long worst = h[0] > h[4] ? h[0] : h[4];
  printf("%-44s NM=%d NVB=%d NVI=%d prio=%d waves/SIMD=%d: %.1f ticks per SIMD-MFMA (wave0 %.0f, wave4 %.0f ticks/iter; ideal %d)\n",
         name, NM, NVB, NVI, PRIO, waves_per_simd, (double)worst / (iters * NM * waves_per_simd), (double)h[0] / iters,
         (double)h[4] / iters, 32 * NM * waves_per_simd);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<64, 0, 0, 0>("1 wave: bare", 256);
  run<64, 0, 0, 0>("2 waves: bare", 512);
  run<64, 200, 0, 0>("1 wave: 64 MFMA then 200 VALU", 256);
  run<64, 200, 0, 0>("2 waves: 64 MFMA then 200 VALU", 512);
  run<64, 200, 0, 1>("2 waves: same, MFMA prio 2/1, VALU prio 0", 512);
  run<64, 200, 0, 2>("2 waves: same, MFMA prio 3, VALU prio 0", 512);
  run<64, 0, 3, 0>("2 waves: 3 VALU after every MFMA", 512);
  run<64, 100, 0, 0>("2 waves: 64 MFMA then 100 VALU", 512);
  run<64, 400, 0, 0>("2 waves: 64 MFMA then 400 VALU", 512);
  run<64, 400, 0, 2>("2 waves: 64 MFMA then 400 VALU, prio", 512);
  run<16, 50, 0, 0>("2 waves: 16 MFMA then 50 VALU", 512);
  run<256, 800, 0, 0>("2 waves: 256 MFMA then 800 VALU", 512);
  return 0;
}
