// blk_common.h -- device helpers shared by the block-window plane kernels (blk_planes.hip: 4x4 / stride-2 layers;
// blk5_planes.hip: 5x5 / stride-1 layers)
#pragma once
#include "odin_device.h"

namespace {

// bk_mfma16 note -- a hardware / compiler hazard found with tools/race_layer.py and tools/race_speech.py: two
// v_mfma_f32_16x16x32_f16 into the SAME accumulator with ONE independent MFMA between them (acx += wh xl; acc += wh xh;
// acx += wl xh -- the natural order of the three plane products) produced wrong values in lanes 12-15 of every 16-lane row
// of one accumulator register, intermittently (7 % of the launches of the audio decoder2 forward, always in a
// workgroup's FIRST tile, i.e. while the SIMD's other wave is not yet issuing MFMAs and the two instructions go out back
// to back: the compiler inserts no wait state there, the second instruction reads the accumulator before the first has
// written all of it).  An extra barrier or any other delay hides it.  Every kernel of the block-window family therefore
// issues MFMAs into the same accumulator at least four instructions apart: sweeps over the pixel blocks, or three
// accumulators (main, high x low, low x high) per block and tap parity.
// D(16 x 16) += A(16 x 32) * B(32 x 16) on f16 operands.  Lane l supplies A[row l & 15][k = 8 (l >> 4) + j] and
// B[k = 8 (l >> 4) + j][col l & 15] in element j; D: col = l & 15, row = 4 (l >> 4) + r for accumulator register r.
__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
  typedef _Float16 bk_h8 __attribute__((ext_vector_type(8)));
#ifdef ODIN_SIM
  const bk_h8 ah = __builtin_bit_cast(bk_h8, a), bh = __builtin_bit_cast(bk_h8, b);
  for (int j = 0; j < 8; ++j) c = sim::mfma_16x16x4((float)ah[j], (float)bh[j], c);
  return c;
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bk_h8, a), __builtin_bit_cast(bk_h8, b), c, 0, 0, 0);
#endif
}

__device__ __forceinline__ f32x4 bk_zero4() {
  f32x4 z;
  z[0] = 0.f; z[1] = 0.f; z[2] = 0.f; z[3] = 0.f;
  return z;
}

__device__ __forceinline__ int bk_uniform(int v) {
#ifdef ODIN_SIM
  return v;
#else
  return __builtin_amdgcn_readfirstlane(v);
#endif
}

// eight consecutive fp32 values (two float4) -> one MFMA operand per plane
__device__ __forceinline__ void bk_split8(const float4& a, const float4& b, float s, float s2k, u32x4& h, u32x4& l) {
  u32x2 h0, l0, h1, l1;
  odin_split_h4<true>(a, s, s2k, h0, l0);
  odin_split_h4<true>(b, s, s2k, h1, l1);
  h[0] = h0[0]; h[1] = h0[1]; h[2] = h1[0]; h[3] = h1[1];
  l[0] = l0[0]; l[1] = l0[1]; l[2] = l1[0]; l[3] = l1[1];
}

// activation and its derivative (from the OUTPUT) with the function a compile-time constant -- ACT = ODIN_ACT_* -- or, ACT < 0,
// the run-time switch of odin_act (two scalar branches per ELEMENT inside the epilogues: measured on the first build,
// 140 branches per tile)
template <int ACT>
__device__ __forceinline__ float bk_act(int rt, float v) {
  if (ACT == ODIN_ACT_ELU) {
    const float em1 = odin_exp2(v * 1.44269504088896341f) - 1.f;
    return v > 0.f ? v : em1;
  }
  if (ACT == ODIN_ACT_RELU) return fmaxf(v, 0.f);
  if (ACT == ODIN_ACT_LINEAR) return v;
  return odin_act(rt, v);
}
template <int ACT>
__device__ __forceinline__ float bk_act_grad(int rt, float y) {
  if (ACT == ODIN_ACT_ELU) return 1.f + fminf(y, 0.f);
  if (ACT == ODIN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (ACT == ODIN_ACT_LINEAR) return 1.f;
  return odin_act_grad(rt, y);
}
// x + the value of lane ^ 16 / lane ^ 32 without an LDS round trip (gfx950 row swaps)
__device__ __forceinline__ float bk_add_xor16(float x) {
#ifdef ODIN_SIM
  return x + __shfl_xor(x, 16);
#else
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}
__device__ __forceinline__ float bk_add_xor32(float x) {
#ifdef ODIN_SIM
  return x + __shfl_xor(x, 32);
#else
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
#endif
}

// the scale of a plane operand from its range word: gradients always (odin_range_shift), activations only when their
// bound leaves the f16 window (odin_act_needs_scale); gk = 0: carried as it is
__device__ __forceinline__ int bk_shift(unsigned mb, int is_grad) {
  return (is_grad || odin_act_needs_scale(mb)) ? odin_range_shift(mb) : 0;
}

// tile index -> (sample, tile row, tile column)
__device__ __forceinline__ void bk_decode(int T, int nty, int ntx, int& b, int& ty, int& tx) {
  const int per = nty * ntx;
  b = odin_div_small(T, per);
  const int r = T - b * per;
  ty = odin_div_small(r, ntx);
  tx = r - ty * ntx;
}

// swizzle of the 16-byte k-pieces of a pixel in a window whose rows are 16 pixel slots apart: the 16 lanes of a
// ds_read_b128 group (two window rows x 8 columns, one k-piece) hit 16 distinct slots of the bank row
__host__ __device__ constexpr int tb_swz(int row, int col) { return ((col >> 2) + 2 * (row & 1)) & 3; }

// ds_read_b64_tr_b16: the 16 lanes of a group hand in the addresses of 4 pixels x 4 channel quads (lane 4 q + p: pixel q,
// channels 4 p .. 4 p + 3 of the group's 16) and lane i receives channel i of the 4 pixels -- four consecutive k of an MFMA
// operand whose reduction index is the pixel.  Simulator: base of pixel q = 0 with the channel group's byte offset,
// pixels `pix_stride` bytes apart (an unswizzled window)
__device__ __forceinline__ u32x2 bk_tr(const char* addr_hw, const char* base_sim, int pix_stride, int l16) {
#ifdef ODIN_SIM
  (void)addr_hw;
  unsigned short e[4];
  for (int q = 0; q < 4; ++q) e[q] = *reinterpret_cast<const unsigned short*>(base_sim + q * pix_stride + 2 * l16);
  return odin_u2((unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16));
#else
  (void)base_sim; (void)pix_stride; (void)l16;
  typedef short bk_s4 __attribute__((ext_vector_type(4)));
  const bk_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bk_s4*)addr_hw);
  return __builtin_bit_cast(u32x2, v);
#endif
}

}  // namespace
