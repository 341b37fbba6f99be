// Microbenchmark: one MFMA wave per SIMD (waves 0-3, the oldest) streaming v_mfma_f32_16x16x4_f32,
// plus K younger VALU-only waves per SIMD: how much vector-ALU work fits in the MFMA stream's shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: scalar FMAs, 1: exp2 (transcendental), 2: 3 FMA + 1 exp2 mix
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, int viters) {
  const int wave = threadIdx.x >> 6;
  f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
  float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = x + i;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        if (u & 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    for (int it = 0; it < viters; ++it) {
#pragma unroll
      for (int u = 0; u < 64; ++u) {
        float& r = v[u & 7];
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
        if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
        if (MODE == 2) {
          if ((u & 3) == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(r));
          else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(x), "v"(y));
        }
      }
    }
  }
  long long t1 = clock64();
  float s = a0[0] + a1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int kwaves, int iters) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&cyc, 128);
  (void)hipMemset(cyc, 0, 128);
  const int threads = 256 * (1 + kwaves);
  const int viters = 200;
  for (int r = 0; r < 2; ++r) { k<MODE><<<256, threads>>>(out, cyc, iters, viters); (void)hipDeviceSynchronize(); }
  long long h[16]; (void)hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
  double mf = iters ? (double)h[0] / (iters * 64.0) : 0;
  printf("%-28s K=%d: MFMA wave %.1f ticks/MFMA;", name, kwaves, mf);
  for (int w = 1; w <= kwaves; ++w) printf(" wave%d %.2f ticks/op", 4 * w, (double)h[4 * w] / (viters * 64.0));
  // ops per MFMA-tick across the K waves while the MFMA wave is running (valid when VALU waves finish first)
  printf("\n");
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  run<0>("FMA, no MFMA wave", 1, 0);
  run<0>("FMA, no MFMA wave", 2, 0);
  run<0>("FMA", 1, 2000);
  run<0>("FMA", 2, 3000);
  run<0>("FMA", 3, 4000);
  run<1>("exp, no MFMA wave", 1, 0);
  run<1>("exp", 1, 4000);
  run<1>("exp", 2, 6000);
  run<2>("3 FMA + 1 exp", 1, 3000);
  run<2>("3 FMA + 1 exp", 2, 4000);
  run<2>("3 FMA + 1 exp", 3, 5000);
  return 0;
}
