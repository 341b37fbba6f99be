// Microbenchmark: two symmetric waves per SIMD, each streaming v_mfma_f32_32x32x16_bf16 with NV
// fillers of a given kind behind every MFMA (the shape of tconv_planes' steady state).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

template <int NV, int KIND, int WAVES>  // KIND 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_min_f32, 3 v_exp_f32, 4 ds_read_b128, 5 mix (4 pk_fma, 1 exp, 2 min, 1 ds_read)
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  f16v b0 = {0}, b1 = {0};
  s8v x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); y[i] = (short)(3 * threadIdx.x + i); }
  float fx = threadIdx.x * 1e-3f, fy = 1.0f + fx;
  float v[8]; f2 pv[8]; f4 lv[4];
  for (int i = 0; i < 8; ++i) { v[i] = fx + i; pv[i] = f2{fx, fy}; }
  for (int i = 0; i < 4; ++i) lv[i] = f4{0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const unsigned lp = (threadIdx.x & 63) * 16 + wave * 1024;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int j = (u * NV + q) & 7;
        int kind = KIND;
        if (KIND == 5) kind = (q & 7) < 4 ? 1 : ((q & 7) == 4 ? 3 : ((q & 7) < 7 ? 2 : 4));
        if (kind == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(fx), "v"(fy));
        if (kind == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pv[j]) : "v"(pv[(j + 1) & 7]));
        if (kind == 2) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[j]) : "v"(fy));
        if (kind == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if (kind == 4) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[j & 3]) : "v"(lp));
      }
    }
    if (KIND >= 4) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  long long t1 = clock64();
  float s = b0[0] + b1[1];
  for (int i = 0; i < 8; ++i) s += v[i] + pv[i][0];
  for (int i = 0; i < 4; ++i) s += lv[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int NV, int KIND, int WAVES>
void run() {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(cyc, 0, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) { k<NV, KIND, WAVES><<<256, 256 * WAVES>>>(out, cyc, iters); (void)hipDeviceSynchronize(); }
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  static const char* kn[] = {"v_fma_f32", "v_pk_fma_f32", "v_min_f32", "v_exp_f32", "ds_read_b128", "mix"};
  long long worst = h[0] > h[4] ? h[0] : h[4];
  printf("%d wave(s)/SIMD, %2d x %-13s per MFMA: %.1f ticks per SIMD-MFMA (wave0 %.1f, wave4 %.1f ticks per own MFMA)\n", WAVES, NV, kn[KIND],
         (double)worst / (iters * 16.0 * WAVES), (double)h[0] / (iters * 16.0), (double)h[4] / (iters * 16.0));
  (void)hipFree(out); (void)hipFree(cyc);
}

template <int KIND>
void sweep() {
  run<2, KIND, 1>(); run<4, KIND, 1>(); run<6, KIND, 1>(); run<8, KIND, 1>();
  run<2, KIND, 2>(); run<4, KIND, 2>(); run<5, KIND, 2>(); run<6, KIND, 2>(); run<8, KIND, 2>(); run<12, KIND, 2>();
}

int main() {
  run<0, 0, 1>(); run<0, 0, 2>();
  sweep<0>(); sweep<1>(); sweep<2>(); sweep<3>();
  run<1, 4, 2>(); run<2, 4, 2>(); run<3, 4, 2>();
  run<8, 5, 1>(); run<8, 5, 2>(); run<16, 5, 2>();
  return 0;
}
