// odin_device.h -- shared device-side helpers for the gfx950 kernels.
// With -DODIN_SIM the same sources compile as host C++ against tests/sim/hipsim.h
// (a debugging emulator, test infrastructure only); the product build is hipcc/gfx950.
#pragma once
#include "../../include/odin_hip.h"
#ifdef ODIN_SIM
#include "hipsim.h"
#define ODIN_DYN_SMEM(T, name) T* name = (T*)sim::S().dyn_smem
#define ODIN_LAUNCH(kern, grid, block, shmem, stream, ...) \
  sim::launch(grid, block, shmem, [&]() { kern(__VA_ARGS__); })
#else
#include <hip/hip_runtime.h>
#define ODIN_DYN_SMEM(T, name)                                                     \
  extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw_lds[];   \
  T* name = (T*)name##_raw_lds
#define ODIN_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__)
#endif

#ifdef ODIN_SIM
static inline float odin_exp(float x) { return expf(x); }
static inline float odin_log(float x) { return logf(x); }
#else
__device__ __forceinline__ float odin_exp(float x) { return __expf(x); }
__device__ __forceinline__ float odin_log(float x) { return __logf(x); }
#endif

#ifdef ODIN_SIM
#define ODIN_SCHED_GROUP(mask, n) ((void)0)
#define ODIN_SCHED_FENCE() ((void)0)
#else
#define ODIN_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define ODIN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define ODIN_SG_MFMA 0x8
#define ODIN_SG_DSREAD 0x100
#define ODIN_SG_VALU 0x402  // VALU | TRANS (everything on the vector ALU except MFMA)

typedef float f32x16 __attribute__((ext_vector_type(16)));
// pairs of floats: arithmetic on them compiles to the packed v_pk_{add,mul,fma}_f32 instructions
// (two results per issue slot -- epilogues are VALU-issue bound)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 odin_f2(float a, float b) {
  f32x2 r;
  r.x = a;
  r.y = b;
  return r;
}
__device__ __forceinline__ float odin_exp2(float x) {
#ifdef ODIN_SIM
  return exp2f(x);
#else
  return __builtin_amdgcn_exp2f(x);
#endif
}
__device__ __forceinline__ float odin_log2(float x) {
#ifdef ODIN_SIM
  return log2f(x);
#else
  return __builtin_amdgcn_logf(x);  // v_log_f32
#endif
}
__device__ __forceinline__ float odin_rcp(float x) {
#ifdef ODIN_SIM
  return 1.f / x;
#else
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32 (1 ulp)
#endif
}
typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf chain.  Lane l supplies
// A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; D: col = l&31,
// row = (r&3) + 8*(r>>2) + 4*(l>>5) for accumulator register r.
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
#ifdef ODIN_SIM
  return sim::mfma_32x32x2(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// ---- fp32 through the bf16 matrix pipe -------------------------------------------------
// An fp32 value splits EXACTLY into three bf16 pieces by truncation (8 + 8 + 8 mantissa bits):
// x = x0 + x1 + x2.  A product x*w is then the sum of 9 exact bf16 x bf16 products; the six with
// piece indices i + j <= 2 carry everything above 2^-24 relative, and v_mfma_f32_32x32x16_bf16
// (16 k-values in 8 passes: 16x the fp32 MFMA rate) accumulates them in fp32.  6 bf16 MFMAs per
// 16 k-values replace 8 fp32 MFMAs: 2.7x less matrix-pipe time at fp32-class accuracy
// (error <= 3 * 2^-24 per product; the dropped terms are x1*w2, x2*w1, x2*w2).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // 8 bf16: element j in bits 16*(j&1) of word j>>1
__device__ __forceinline__ unsigned odin_fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float odin_bitsf(unsigned u) { return __builtin_bit_cast(float, u); }
// pack the bf16 (upper) halves of two floats: a -> low half, b -> high half
__device__ __forceinline__ unsigned odin_pack_bf16(float a, float b) {
  return (odin_fbits(a) >> 16) | (odin_fbits(b) & 0xFFFF0000u);
}
__device__ __forceinline__ u32x2 odin_u2(unsigned a, unsigned b) {
  u32x2 r;
  r.x = a;
  r.y = b;
  return r;
}
// remainder after removing the leading bf16 piece (exact)
__device__ __forceinline__ float odin_bf16_rest(float x) {
  return x - odin_bitsf(odin_fbits(x) & 0xFFFF0000u);
}
// D = A(32 x 16) * B(16 x 32) + C.  Lane l supplies A[row l&31][k = 8*(l>>5) + j] and
// B[k = 8*(l>>5) + j][col l&31] in element j; the C/D layout is that of mfma32 above.
__device__ __forceinline__ f32x16 mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
#ifdef ODIN_SIM
  for (int j = 0; j < 8; ++j) {
    const float af = odin_bitsf((a[j >> 1] >> (16 * (j & 1))) << 16);
    const float bf = odin_bitsf((b[j >> 1] >> (16 * (j & 1))) << 16);
    c = sim::mfma_32x32x2(af, bf, c);
  }
  return c;
#else
  typedef __bf16 odin_bf16x8 __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(odin_bf16x8, a),
                                                 __builtin_bit_cast(odin_bf16x8, b), c, 0, 0, 0);
#endif
}

// D(16 x 16) += A(16 x 32) * B(32 x 16), v_mfma_f32_16x16x32_bf16 (8 passes = 16 cycles: the same FLOP per cycle as
// the 32x32x16 form).  Lane l supplies A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15] in
// element j; D: col = l & 15, row = 4 (l >> 4) + r for accumulator register r.
__device__ __forceinline__ f32x4 mfma16_bf16(u32x4 a, u32x4 b, f32x4 c) {
#ifdef ODIN_SIM
  for (int j = 0; j < 8; ++j) {
    const float af = odin_bitsf((a[j >> 1] >> (16 * (j & 1))) << 16);
    const float bf = odin_bitsf((b[j >> 1] >> (16 * (j & 1))) << 16);
    c = sim::mfma_16x16x4(af, bf, c);
  }
  return c;
#else
  typedef __bf16 odin_bf16x8b __attribute__((ext_vector_type(8)));
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(odin_bf16x8b, a),
                                                 __builtin_bit_cast(odin_bf16x8b, b), c, 0, 0, 0);
#endif
}

// ---- fp32 through the f16 matrix pipe: two planes -----------------------------------------------
// x = h + 2^-11 l with h = RNE_f16(x) and l = RNE_f16(2^11 (x - h)): |x - h| <= 2^-11 |x| and the second
// rounding leaves <= 2^-22 |x| (the low plane is carried scaled by 2^11 so that it stays a NORMAL f16 down
// to |x| = 2^-25; below that its absolute error is 2^-36).  A product x w is then
//   h_x h_w  +  2^-11 (h_x l_w + l_x h_w)   (+ the dropped l_x l_w <= 2^-22 |x w|):
// three v_mfma_f32_32x32x16_f16 per 16 k-values into TWO fp32 accumulators (main, cross), combined once as
// main + 2^-11 cross.  Error <= 3 * 2^-22 per product -- below the rounding noise of an fp32 dot product of the
// layers' reduction lengths (512-1024 terms, 2^-24 per addition) -- at half the matrix work and about half the
// split work of the three-plane bf16 form.  Range: |x| <= 65504 for activations and weights (larger values become
// inf and trip the optimiser's NaN policy); gradient operands are carried times a per-tensor power of two
// (odin_range_shift below) through the split.
#define ODIN_LO_SCALE 2048.f
#define ODIN_LO_UNSCALE 4.8828125e-4f  // 2^-11
// (a, b) -> packed f16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned odin_pack_f16(float a, float b) {
  typedef _Float16 odin_h2 __attribute__((ext_vector_type(2)));
#ifdef ODIN_SIM
  odin_h2 h;
  h.x = (_Float16)a;
  h.y = (_Float16)b;
  return __builtin_bit_cast(unsigned, h);
#else
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
#endif
}
// m - 2048 * (f16 half `HI` of hp): one v_fma_mix_f32 (the f16 operand is widened inside the instruction)
template <int HI>
__device__ __forceinline__ float odin_lo_rest(unsigned hp, float m) {
#ifdef ODIN_SIM
  typedef _Float16 odin_h2 __attribute__((ext_vector_type(2)));
  const odin_h2 h = __builtin_bit_cast(odin_h2, hp);
  return fmaf((float)(HI ? h.y : h.x), -ODIN_LO_SCALE, m);
#else
  float r;
  if (HI) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "s"(-ODIN_LO_SCALE), "v"(m));
  else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hp), "s"(-ODIN_LO_SCALE), "v"(m));
  return r;
#endif
}
// four consecutive fp32 values (times s when SC) -> their two f16 planes (4 f16 = 8 bytes each); s2k = 2048 s
template <bool SC>
__device__ __forceinline__ void odin_split_h4(const float4& v, float s, float s2k, u32x2& h, u32x2& l) {
  const float x0 = SC ? v.x * s : v.x, x1 = SC ? v.y * s : v.y, x2 = SC ? v.z * s : v.z, x3 = SC ? v.w * s : v.w;
  const float k = SC ? s2k : ODIN_LO_SCALE;
  const unsigned h01 = odin_pack_f16(x0, x1), h23 = odin_pack_f16(x2, x3);
  const float r0 = odin_lo_rest<0>(h01, v.x * k), r1 = odin_lo_rest<1>(h01, v.y * k);
  const float r2 = odin_lo_rest<0>(h23, v.z * k), r3 = odin_lo_rest<1>(h23, v.w * k);
  h = odin_u2(h01, h23);
  l = odin_u2(odin_pack_f16(r0, r1), odin_pack_f16(r2, r3));
}
// D = A(32 x 16) * B(16 x 32) + C on f16 operands (8 f16 per lane and operand; layouts of mfma32_bf16)
__device__ __forceinline__ f32x16 mfma32_f16(u32x4 a, u32x4 b, f32x16 c) {
  typedef _Float16 odin_h8 __attribute__((ext_vector_type(8)));
#ifdef ODIN_SIM
  const odin_h8 ah = __builtin_bit_cast(odin_h8, a), bh = __builtin_bit_cast(odin_h8, b);
  for (int j = 0; j < 8; ++j) c = sim::mfma_32x32x2((float)ah[j], (float)bh[j], c);
  return c;
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(odin_h8, a), __builtin_bit_cast(odin_h8, b), c, 0, 0, 0);
#endif
}
// Dynamic range of gradient operands.  f16 planes hold |x| <= 65504, and gradient tensors range from 1e-8 (a mean
// over a large batch, deep layers) to 1e5 (reverse KL at random initialisation): every gradient tensor that feeds
// a plane kernel travels with a device word holding the fp32 bit pattern of (an upper bound of) max |x| -- its
// producer kernel keeps a running maximum of what it stores and issues one atomicMax per workgroup; a tensor without a
// tracked producer gets the word from odin_absmax (runtime.hip) -- and the consumer scales the tensor by the power
// of two that brings that maximum into [2^14, 2^15) on its way into the planes, and its result back.  Exact (powers
// of two), never overflows, and the planes' absolute error floor sits 2^-50 below the tensor's maximum.
// k = odin_range_shift(bits): the tensor is carried times 2^k, -113 <= k <= 115
__device__ __forceinline__ int odin_range_shift(unsigned mb) {
  int e = (int)((mb >> 23) & 0xFFu);
  if (e == 255) e = 141;  // an inf / NaN maximum: no scaling, the non-finite values propagate
  if (e < 26) e = 26;     // a zero (or < 2^-101) tensor
  return 141 - e;
}
// ACTIVATION operands of the plane kernels (round 5): carried unscaled while their bound lies in [2^-8, 2^15) -- every
// benchmark tensor does, and the unscaled split is 4 VALU instructions per 4 values cheaper -- and through the scaled
// split of the gradient operands otherwise (|x| up to 3e38 and down to 1e-38 keep their 22 bits).  Wave-uniform; a
// kernel holds both bodies and branches once.  No word, a zero tensor or a non-finite bound: unscaled.
__device__ __forceinline__ bool odin_act_needs_scale(unsigned mb) {
  const int e = (int)((mb >> 23) & 0xFFu);
  return e != 0 && e != 255 && (e >= 127 + 15 || e < 127 - 8);
}
__device__ __forceinline__ float odin_pow2(int k) { return odin_bitsf((unsigned)(127 + k) << 23); }  // -126 <= k <= 127
// max over the 64 lanes of NON-NEGATIVE values (every lane receives it).  On the vector ALU (quad / row DPP moves and
// the gfx950 row swaps): the six ds_bpermute round trips of the shuffle form sat at the very end of every producing
// kernel (the range-word commit), ~0.5 us per launch
__device__ __forceinline__ float odin_wave_max64(float v) {
#ifdef ODIN_SIM
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
#else
  // (bit patterns of non-negative floats order like unsigned integers; a NaN bound stays the largest pattern)
  unsigned m = __float_as_uint(v), o;
  o = __builtin_amdgcn_update_dpp(0u, m, 0xB1, 0xF, 0xF, false); m = o > m ? o : m;    // quad_perm [1,0,3,2]
  o = __builtin_amdgcn_update_dpp(0u, m, 0x4E, 0xF, 0xF, false); m = o > m ? o : m;    // quad_perm [2,3,0,1]
  o = __builtin_amdgcn_update_dpp(0u, m, 0x141, 0xF, 0xF, false); m = o > m ? o : m;   // row_half_mirror
  o = __builtin_amdgcn_update_dpp(0u, m, 0x140, 0xF, 0xF, false); m = o > m ? o : m;   // row_mirror
  {
    const auto q = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    m = q[0] > q[1] ? q[0] : q[1];
  }
  {
    const auto q = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    m = q[0] > q[1] ? q[0] : q[1];
  }
  return __uint_as_float(m);
#endif
}
// A range "word" is a BLOCK of ODIN_RANGE_SLOTS sub-words, ODIN_RANGE_STRIDE words apart (one per 256-byte
// line): same-address atomics of a whole launch serialise at the memory side (~8 ns each measured: one atomicMax
// per wave of a 256-workgroup launch added 16 us to the kernel), so a workgroup reduces its waves through LDS and
// issues ONE atomicMax into the slot blockIdx selects; the consumer takes the maximum of the slots (scalar loads).
#define ODIN_RANGE_SLOTS 32
#define ODIN_RANGE_STRIDE 64
static_assert(ODIN_RANGE_SLOTS * ODIN_RANGE_STRIDE == ODIN_RANGE_WORDS, "include/odin_hip.h: ODIN_RANGE_WORDS");
// consumer side: the bound (fp32 bit pattern) of a tensor, wave-uniform
__device__ __forceinline__ unsigned odin_range_load(const unsigned* block) {
  unsigned m = 0u;
#pragma unroll
  for (int s = 0; s < ODIN_RANGE_SLOTS; ++s) {
    const unsigned v = block[s * ODIN_RANGE_STRIDE];
    m = v > m ? v : m;
  }
#ifdef ODIN_SIM
  return m;
#else
  return __builtin_amdgcn_readfirstlane(m);
#endif
}
// The same read in two halves (round 5): ONE vector load at the top of a kernel -- lane s < 32 fetches sub-word s --
// and the wave maximum taken where the bound is first needed (the first split, behind the prologue's loads).  The 32
// scalar loads of odin_range_load stall the wave at their s_waitcnt for an L2 round trip (~1 us of a launch) wherever
// the compiler puts it, and it puts it in front of the next vector load; a vector load waits in vmcnt order with the
// weight / row loads behind it.
struct OdinRangeReq { unsigned v; };
__device__ __forceinline__ OdinRangeReq odin_range_issue(const unsigned* block, int lane) {
  OdinRangeReq r;
  r.v = 0u;
  if (block != nullptr && lane < ODIN_RANGE_SLOTS) r.v = block[lane * ODIN_RANGE_STRIDE];
  return r;
}
__device__ __forceinline__ unsigned odin_range_finish(const OdinRangeReq& r) {
#ifdef ODIN_SIM
  unsigned m = r.v;
  for (int k = 32; k >= 1; k >>= 1) { const unsigned o = __shfl_xor(m, k); m = o > m ? o : m; }
  return m;
#else
  // max over the 64 lanes on the vector ALU (quad / row DPP moves, then the gfx950 row swaps): no LDS round trip
  unsigned m = r.v;
  unsigned o;
  o = __builtin_amdgcn_update_dpp(0u, m, 0xB1, 0xF, 0xF, false); m = o > m ? o : m;    // quad_perm [1,0,3,2]
  o = __builtin_amdgcn_update_dpp(0u, m, 0x4E, 0xF, 0xF, false); m = o > m ? o : m;    // quad_perm [2,3,0,1]
  o = __builtin_amdgcn_update_dpp(0u, m, 0x141, 0xF, 0xF, false); m = o > m ? o : m;   // row_half_mirror
  o = __builtin_amdgcn_update_dpp(0u, m, 0x140, 0xF, 0xF, false); m = o > m ? o : m;   // row_mirror
  {
    const auto q = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    m = q[0] > q[1] ? q[0] : q[1];
  }
  {
    const auto q = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    m = q[0] > q[1] ? q[0] : q[1];
  }
  return __builtin_amdgcn_readfirstlane(m);
#endif
}
// running maximum of |values|: max(m, |a|, |b|) as ONE v_max3_f32 with source modifiers (the generic fmaxf / fabsf form
// compiles to four instructions per pair: IEEE canonicalisation of each |x|)
__device__ __forceinline__ float odin_amax3(float m, float a, float b) {
#ifdef ODIN_SIM
  return fmaxf(m, fmaxf(fabsf(a), fabsf(b)));
#else
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
  return r;
#endif
}
// producer side, one wave speaking for its workgroup (fp32 bit patterns of non-negative values order like
// unsigned integers)
__device__ __forceinline__ void odin_amax_commit_wave(unsigned* block, float amx, int lane, unsigned wg) {
  if (block == nullptr) return;
  const float m = odin_wave_max64(amx);
  if (lane == 0) atomicMax(block + (wg & (ODIN_RANGE_SLOTS - 1)) * ODIN_RANGE_STRIDE, odin_fbits(m));
}
// producer side, all waves of a workgroup (call from uniform control flow; `red` = LDS scratch of >= 16 floats)
__device__ __forceinline__ void odin_amax_commit_wg(unsigned* block, float amx, int tid, int nthreads, float* red,
                                                    unsigned wg) {
  if (block == nullptr) return;
  const float m = odin_wave_max64(amx);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    float t = red[0];
    for (int w = 1; w < (nthreads >> 6); ++w) t = fmaxf(t, red[w]);
    atomicMax(block + (wg & (ODIN_RANGE_SLOTS - 1)) * ODIN_RANGE_STRIDE, odin_fbits(t));
  }
}

__device__ __forceinline__ f32x16 f32x16_zero() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Range-checked staging loads.  A run of HBM (one patch row, one DY tile, a weight slice) is
// described by a wave-uniform (base, valid bytes) pair; a lane whose byte offset falls outside
// [0, bytes) reads zeros (give masked lanes the offset ODIN_OOB).  On gfx950 this is a buffer
// descriptor + buffer_load with the hardware range check: no exec-mask branch and no zero
// pre-initialisation around the load, so every load of a tile issues back to back and nothing
// makes the compiler drain vmcnt early (branchy `if (ok) v = *p` staging serialised one memory
// latency per patch row).
#define ODIN_OOB 0x7FFFFFF0u
struct OdinRun {
#ifdef ODIN_SIM
  const char* base;
  unsigned bytes;
#else
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ OdinRun odin_run(const void* base, unsigned bytes) {
  OdinRun R;
#ifdef ODIN_SIM
  R.base = (const char*)base;
  R.bytes = bytes;
#else
  const unsigned long long a = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  void* pu = (void*)(((unsigned long long)hi << 32) | lo);
  R.r = __builtin_amdgcn_make_buffer_rsrc(pu, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
#endif
  return R;
}
__device__ __forceinline__ float4 odin_run_load4(const OdinRun& R, unsigned off) {
#ifdef ODIN_SIM
  if ((unsigned long long)off + 16 > R.bytes) return make_float4(0.f, 0.f, 0.f, 0.f);
  return *reinterpret_cast<const float4*>(R.base + off);
#else
  typedef unsigned int odin_u32x4 __attribute__((ext_vector_type(4)));
  const odin_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(R.r, off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                     __uint_as_float(v.w));
#endif
}
__device__ __forceinline__ float odin_run_load1(const OdinRun& R, unsigned off) {
#ifdef ODIN_SIM
  if ((unsigned long long)off + 4 > R.bytes) return 0.f;
  return *reinterpret_cast<const float*>(R.base + off);
#else
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R.r, off, 0, 0));
#endif
}

// The same with a wave-uniform byte offset added by the scalar unit (buffer `soffset`): the per-lane part
// of an address is computed ONCE per kernel, the per-tile part is one scalar.  Only the per-lane offset
// takes part in the range check (as on the hardware): give masked lanes ODIN_OOB_V.
#define ODIN_OOB_V 0xFFFF0000u
__device__ __forceinline__ float4 odin_run_load4s(const OdinRun& R, unsigned voff, unsigned soff) {
#ifdef ODIN_SIM
  if ((unsigned long long)voff + 16 > R.bytes) return make_float4(0.f, 0.f, 0.f, 0.f);
  return *reinterpret_cast<const float4*>(R.base + voff + soff);
#else
  typedef unsigned int odin_u32x4 __attribute__((ext_vector_type(4)));
  const odin_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(R.r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                     __uint_as_float(v.w));
#endif
}
__device__ __forceinline__ float odin_run_load1s(const OdinRun& R, unsigned voff, unsigned soff) {
#ifdef ODIN_SIM
  if ((unsigned long long)voff + 4 > R.bytes) return 0.f;
  return *reinterpret_cast<const float*>(R.base + voff + soff);
#else
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(R.r, voff, soff, 0));
#endif
}
__device__ __forceinline__ void odin_run_store4s(const OdinRun& R, unsigned voff, unsigned soff, float4 v) {
#ifdef ODIN_SIM
  if ((unsigned long long)voff + 16 <= R.bytes)
    *reinterpret_cast<float4*>(const_cast<char*>(R.base) + voff + soff) = v;
#else
  typedef unsigned int odin_u32x4 __attribute__((ext_vector_type(4)));
  odin_u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, R.r, voff, soff, 0);
#endif
}
__device__ __forceinline__ void odin_run_store1s(const OdinRun& R, unsigned voff, unsigned soff, float v) {
#ifdef ODIN_SIM
  if ((unsigned long long)voff + 4 <= R.bytes)
    *reinterpret_cast<float*>(const_cast<char*>(R.base) + voff + soff) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), R.r, voff, soff, 0);
#endif
}

// range-checked store: lanes with an offset outside [0, bytes) (ODIN_OOB) write nothing -- a
// predicated store without an exec-mask branch, so it stays inside one scheduling region
__device__ __forceinline__ void odin_run_store1(const OdinRun& R, unsigned off, float v) {
#ifdef ODIN_SIM
  if ((unsigned long long)off + 4 <= R.bytes)
    *reinterpret_cast<float*>(const_cast<char*>(R.base) + off) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), R.r, off, 0, 0);
#endif
}

// the same for two floats (8-byte aligned offset) and a range-checked two-float load
__device__ __forceinline__ void odin_run_store2(const OdinRun& R, unsigned off, float2 v) {
#ifdef ODIN_SIM
  if ((unsigned long long)off + 8 <= R.bytes)
    *reinterpret_cast<float2*>(const_cast<char*>(R.base) + off) = v;
#else
  typedef unsigned int odin_u32x2 __attribute__((ext_vector_type(2)));
  odin_u32x2 d;
  d.x = __float_as_uint(v.x);
  d.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(d, R.r, off, 0, 0);
#endif
}
__device__ __forceinline__ float2 odin_run_load2(const OdinRun& R, unsigned off) {
#ifdef ODIN_SIM
  if ((unsigned long long)off + 8 > R.bytes) return make_float2(0.f, 0.f);
  return *reinterpret_cast<const float2*>(R.base + off);
#else
  typedef unsigned int odin_u32x2 __attribute__((ext_vector_type(2)));
  const odin_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(R.r, off, 0, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
#endif
}

// LDS-DMA: every active lane moves 16 bytes from the run straight into LDS at
// lds_base + lane*16 (lds_base wave-uniform, 16-byte aligned); no VGPR destination, no ds_write.
// Completion is tracked by the issuing wave's vmcnt.  Never issued with out-of-range offsets
// (mask those lanes off instead).
__device__ __forceinline__ void odin_run_dma16(const OdinRun& R, float* lds_base, unsigned off,
                                               int lane) {
#ifdef ODIN_SIM
  *reinterpret_cast<float4*>(lds_base + 4 * lane) = odin_run_load4(R, off);
#else
  (void)lane;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      R.r, (__attribute__((address_space(3))) void*)lds_base, 16, off, 0, 0, 0);
#endif
}
__device__ __forceinline__ void odin_wait_vmem() {
#ifndef ODIN_SIM
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// 16-byte store with the streaming (non-temporal) hint: for outputs of HBM-bound kernels
__device__ __forceinline__ void odin_store4_stream(float4* p, const float4& v) {
#ifdef ODIN_SIM
  *p = v;
#else
  f32x4 r;
  r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w;
  __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(p));
#endif
}

enum { ODIN_ACT_LINEAR = 0, ODIN_ACT_ELU = 1, ODIN_ACT_RELU = 2 };

__device__ __forceinline__ float odin_act(int act, float v) {
  if (act == ODIN_ACT_ELU) return v > 0.f ? v : odin_exp(v) - 1.f;
  if (act == ODIN_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
// derivative of the activation expressed from its OUTPUT y
__device__ __forceinline__ float odin_act_grad(int act, float y) {
  if (act == ODIN_ACT_ELU) return y > 0.f ? 1.f : y + 1.f;
  if (act == ODIN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  return 1.f;
}

// a / b for 0 <= a < 2^22, 1 <= b < 2^22 through the float reciprocal and one correction step: ~8 instructions
// against the ~40 of the integer division sequence (kernel prologues that build index tables)
__device__ __forceinline__ int odin_div_small(int a, int b) {
#ifdef ODIN_SIM
  return a / b;
#else
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += r >= b ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
#endif
}

// rendezvous of the lanes of ONE wave around wave-private LDS traffic.  On the hardware the DS
// pipe executes a wave's LDS instructions in order, so only the compiler has to be kept from
// moving accesses across; the simulator runs lanes as fibers and needs a real rendezvous.
__device__ __forceinline__ void odin_wave_sync() {
#ifdef ODIN_SIM
  (void)sim::wave_deposit(0.f, 0.f);
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Sum over the 4 rows of 16 lanes (lanes l, l ^ 16, l ^ 32, l ^ 48); every lane receives the
// total.  gfx950 row swaps (v_permlane32_swap / v_permlane16_swap) keep it on the vector ALU: no
// LDS round trip (ds_bpermute + lgkmcnt wait) in the middle of an MFMA stream.
__device__ __forceinline__ float odin_rowsum4(float x) {
#ifdef ODIN_SIM
  const float y = x + __shfl_xor(x, 32);
  return y + __shfl_xor(y, 16);
#else
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const unsigned w = __float_as_uint(y);
  const auto q = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
#endif
}

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// The same sum (every lane receives it) without touching LDS: quad / row DPP moves and the gfx950
// row swaps -- ~12 VALU instructions instead of six ds_bpermute round trips.
__device__ __forceinline__ float odin_wave_sum64_valu(float v) {
#ifdef ODIN_SIM
  return wave_sum64(v);
#else
  // quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror: after each step the value
  // is uniform over 2, 4, 8, 16 lanes
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xB1, 0xF, 0xF, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4E, 0xF, 0xF, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x141, 0xF, 0xF, false));
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x140, 0xF, 0xF, false));
  const unsigned w = __float_as_uint(v);
  const auto q = __builtin_amdgcn_permlane16_swap(w, w, false, false);
  v = __uint_as_float(q[0]) + __uint_as_float(q[1]);
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

static inline int odin_floordiv(int a, int b) {
  int q = a / b;
  if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
  return q;
}
