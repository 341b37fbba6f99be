// Microbenchmark: the instruction kinds of the fp32 -> 3 x bf16 plane split between bf16 MFMAs (32x32x16), one or two
// MFMA waves per SIMD, fillers spread evenly (NV per MFMA) or in one burst of 4 NV behind every fourth MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef short s8v __attribute__((ext_vector_type(8)));

template <int KIND>
__device__ __forceinline__ void filler(unsigned& a, unsigned b, unsigned sc, char* lds) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
  if (KIND == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(a));
  if (KIND == 2) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if (KIND == 3) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a) : "v"(b));
  if (KIND == 4) asm volatile("v_and_or_b32 %0, %0, %2, %1" : "+v"(a) : "v"(b), "s"(sc));
  if (KIND == 5) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(a));
  if (KIND == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(sc));
  if (KIND == 7) asm volatile("ds_write_b64 %0, %1" ::"v"((unsigned)(size_t)lds), "v"((unsigned long long)a) : "memory");
  if (KIND == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
  if (KIND == 9) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
}

template <int NV, int KIND, int BURST, int WAVES>  // WAVES: MFMA waves per SIMD (1 or 2)
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  __shared__ char lds[64 * 8 * 8];
  const int wave = threadIdx.x >> 6;
  f16v b0 = {0}, b1 = {0};
  s8v x, y;
  for (int i = 0; i < 8; ++i) { x[i] = (short)(threadIdx.x + i); y[i] = (short)(3 * threadIdx.x + i); }
  unsigned v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 977u + i;
  char* my = lds + (threadIdx.x & 511) * 8;
  __syncthreads();
  long long t0 = clock64();
  if (wave < 4 * WAVES) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b1) : "v"(x), "v"(y));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(b0) : "v"(x), "v"(y));
        if (!BURST) {
#pragma unroll
          for (int q = 0; q < NV; ++q) filler<KIND>(v[(u * NV + q) & 7], v[(u * NV + q + 3) & 7], 0xffff0000u, my);
        } else if ((u & 3) == 1) {
#pragma unroll
          for (int q = 0; q < 4 * NV; ++q) filler<KIND>(v[q & 7], v[(q + 3) & 7], 0xffff0000u, my);
        }
      }
    }
  }
  long long t1 = clock64();
  float s = b0[0] + b1[1];
  for (int i = 0; i < 8; ++i) s += (float)v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

template <int NV, int KIND, int BURST, int WAVES>
void run(const char* name) {
  float* out; long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
  (void)hipMemset(cyc, 0, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) { k<NV, KIND, BURST, WAVES><<<256, 256 * WAVES>>>(out, cyc, iters); (void)hipDeviceSynchronize(); }
  long long h[8]; (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%d MFMA wave(s)/SIMD, %2d x %-14s per MFMA, %s: %.1f ticks per MFMA per SIMD (wave 0 %.1f%s)\n", WAVES, NV, name,
         BURST ? "bursts of 4x behind every 4th" : "spread", (double)(WAVES == 2 ? (h[0] > h[4] ? h[0] : h[4]) : h[0]) / (iters * 16.0 * WAVES),
         (double)h[0] / (iters * 16.0), WAVES == 2 ? ", both waves run the same stream" : "");
  (void)hipFree(out); (void)hipFree(cyc);
}

#define KINDS(NV, B, W)                                                                                      \
  run<NV, 0, B, W>("v_fma_f32"); run<NV, 1, B, W>("v_and_b32 lit"); run<NV, 2, B, W>("v_sub_f32");          \
  run<NV, 3, B, W>("v_or_b32_sdwa"); run<NV, 4, B, W>("v_and_or_b32 s"); run<NV, 5, B, W>("v_lshrrev_b32"); \
  run<NV, 6, B, W>("v_perm_b32 s"); run<NV, 8, B, W>("v_cndmask_b32"); run<NV, 9, B, W>("v_add_u32");

int main() {
  run<0, 0, 0, 1>("(none)"); run<0, 0, 0, 2>("(none)");
  KINDS(4, 0, 1) KINDS(8, 0, 1) KINDS(8, 1, 1) KINDS(4, 0, 2) KINDS(8, 0, 2) KINDS(8, 1, 2)
  run<1, 7, 0, 1>("ds_write_b64"); run<1, 7, 0, 2>("ds_write_b64"); run<1, 7, 1, 2>("ds_write_b64");
  return 0;
}
