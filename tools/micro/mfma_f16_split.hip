// Microbenchmark: the instruction kinds of the fp32 -> 2 x fp16 plane split (hi = RNE f16, lo = f16 of the remainder
// x 2^11) between v_mfma_f32_32x32x16_f16, one or two MFMA waves per SIMD, NV fillers behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(unsigned& a, unsigned b, unsigned sc) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == 2) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(a) : "v"(b));
  if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(b), "s"(sc));
  if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a) : "v"(b), "s"(sc));
  if (KIND == 5) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a) : "s"(sc));
  if (KIND == 6) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == 7) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a) : "v"(b));
  if (KIND == 8) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a) : "v"(b));
}

template <int NV, int KIND, int WAVES, int BF>  // WAVES: MFMA waves per SIMD (1 or 2); BF: bf16 MFMA instead
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f16v b0 = {0}, b1 = {0};
  h8v x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (_Float16)(0.001f * (threadIdx.x + i)); y[i] = (_Float16)(0.002f * (3 * threadIdx.x + i)); }
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = 0x3c003c00u + threadIdx.x * 16u + i;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4 * WAVES) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (BF) {
          if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
        } else {
          if (u & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
        }
#pragma unroll
        for (int q = 0; q < NV; ++q) filler<KIND>(v[(u * NV + q) & 7], v[(u * NV + q + 3) & 7], 0x45000000u);
      }
    }
  }
  long long t1 = clock64();
  float s = b0[0] + b1[1];
  for (int i = 0; i < 8; ++i) s += (float)v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int NV, int KIND, int WAVES, int BF = 0>
void run(const char* name) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(cyc, 0, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) { k<NV, KIND, WAVES, BF><<<256, 256 * WAVES>>>(out, cyc, iters); (void)hipDeviceSynchronize(); }
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%s %d MFMA wave(s)/SIMD, %2d x %-22s per MFMA: %.1f ticks per MFMA per SIMD (wave 0 %.1f)\n", BF ? "bf16" : "f16 ", WAVES, NV, name,
         (double)(WAVES == 2 ? (h[0] > h[4] ? h[0] : h[4]) : h[0]) / (iters * 16.0 * WAVES), (double)h[0] / (iters * 16.0));
  (void)hipFree(out); (void)hipFree(cyc);
}

#define KINDS(NV, W)                                                                                          \
  run<NV, 0, W>("v_fma_f32"); run<NV, 1, W>("v_cvt_pk_f16_f32"); run<NV, 2, W>("v_cvt_f32_f16");            \
  run<NV, 3, W>("v_fma_mix_f32 lo"); run<NV, 4, W>("v_fma_mix_f32 hi"); run<NV, 5, W>("v_mul_f32 s");        \
  run<NV, 6, W>("v_cvt_pkrtz_f16_f32"); run<NV, 7, W>("v_cvt_f32_f16_sdwa"); run<NV, 8, W>("v_pk_mul_f16");

int main() {
  run<0, 0, 1>("(none)"); run<0, 0, 2>("(none)"); run<0, 0, 1, 1>("(none)"); run<0, 0, 2, 1>("(none)");
  KINDS(2, 1) KINDS(4, 1) KINDS(6, 1) KINDS(8, 1) KINDS(4, 2) KINDS(6, 2) KINDS(8, 2)
  return 0;
}
