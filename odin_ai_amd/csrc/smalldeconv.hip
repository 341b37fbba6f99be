// smalldeconv.hip -- the decoders' FIRST Conv2DTranspose: 4x4 / stride 2 / SAME from a tiny image with few channels
// ([hh, ww, C0], hh ww <= 64, C0 in {8, 16}: the reshaped output of the decoder's projection) to 64 channels
// (image_networks.py:494-500: Dense -> Reshape(4, 4, proj / 16) -> Conv2DTranspose(64, 4, 2, 'same', activation); the
// dSprites / Shapes3D / audio decoders, examples/vae/vae_audio.py:84-110).  67 MFLOP at batch 256: on the implicit-GEMM
// kernel (igemm.hip) the forward launch took 7 us and the paired backward launch 16 us -- launch floors and dependent
// L2 round trips, no arithmetic to speak of.  Here a workgroup owns S samples and keeps everything it touches in LDS /
// registers; exact fp32 FMAs on the vector ALU (131 k MACs per sample and direction), 16 waves per workgroup so that
// the LDS round trips of one wave hide behind the others (the same stages inside the 256-thread bottleneck launch --
// one wave per SIMD -- were latency chains: profiles/r05_latent_block2_experiment.txt).
//
//   forward : thread = (output channel co, output parity class (r, c), pixel quarter): its 4 taps x C0 weights in
//             registers, the input image zero-bordered in LDS (broadcast reads), y = act(b + sum) stored 64 channels
//             (256 bytes) per wave-instruction; max |y| folded into the layer's range word (the plane kernel above
//             reads it: odin_conv_desc.y_amax)
//   backward: the output gradient as a zero-bordered image in LDS (no bounds tests in the loops);
//             weight gradient: thread = (co, class, tap): C0 accumulators over the S samples' pixels -> one slab row per
//             workgroup (fixed order); data gradient: wave = input pixel, lane = (channel slice cs, ci): C0 channels of
//             16 taps, the 64 / C0 lanes of a ci meet by shuffles; x act'(aux); max |dx| folded into dx_amax
#include "odin_device.h"
#include "odin_internal.h"

namespace {

constexpr int SD_C1 = 64;   // output channels: one lane per channel
constexpr int SD_GP = 68;   // pixel pitch (floats) of the staged gradient image (68 mod 64 = 4: 16-byte reads of lanes on
                            // different pixels fall on different bank groups)
constexpr int SD_NT = 1024;

struct SDParams {
  const float* x;      // [B, hh, ww, C0]
  const float* w;      // [4, 4, 64, C0] (Keras Conv2DTranspose: kh, kw, out, in)
  const float* bias;   // [64]
  float* y;            // forward: [B, 2 hh, 2 ww, 64]
  const float* dy;     // backward: [B, 2 hh, 2 ww, 64]
  const float* aux;    // backward: dx *= act'(aux), aux [B, hh, ww, C0] (= x: the layer below's output) or null
  float* dx;           // backward: [B, hh, ww, C0] (null: no data gradient)
  float* slab;         // backward: [gridDim.x][16 * 64 * C0] (null: no weight gradient)
  unsigned* y_amax;    // forward: range word of y (may be null)
  unsigned* dx_amax;   // backward: range word of dx (may be null)
  int B, hh, ww, act, aux_act, S;
  int w_al;            // w is 16-byte aligned (the flat parameter buffer packs tensors without padding: an odd latent
                       // width puts every decoder weight at an 8-byte offset -- then four scalar loads per quad)
};

__device__ __forceinline__ float4 sd_ld4(const float* q, int al) {
  if (al) return *reinterpret_cast<const float4*>(q);
  return make_float4(q[0], q[1], q[2], q[3]);
}

// padded input image: [hh + 2][ww + 2][C0], zero border
template <int C0>
__device__ __forceinline__ void sd_stage_x(const SDParams& p, float* xp, int b0, int ns, int tid) {
  const int HH = p.hh, WW = p.ww, N0 = HH * WW * C0;
  const int npad = p.S * (HH + 2) * (WW + 2) * C0;
  for (int e = tid; e < npad; e += SD_NT) {
    const int s = e / ((HH + 2) * (WW + 2) * C0), r = e - s * ((HH + 2) * (WW + 2) * C0);
    const int pp = r / C0, ci = r - pp * C0;
    const int pr = pp / (WW + 2), pc = pp - pr * (WW + 2);
    const bool in = s < ns && pr >= 1 && pr <= HH && pc >= 1 && pc <= WW;
    xp[e] = in ? p.x[(size_t)(b0 + s) * N0 + ((pr - 1) * WW + pc - 1) * C0 + ci] : 0.f;
  }
}

template <int C0>
__global__ __launch_bounds__(SD_NT) void smalldeconv_fwd_kernel(SDParams p) {
  ODIN_DYN_SMEM(float, xp);   // [S][hh + 2][ww + 2][C0]
  __shared__ float ared[16];
  const int tid = threadIdx.x, b0 = blockIdx.x * p.S;
  const int ns = (p.B - b0 < p.S) ? p.B - b0 : p.S;
  const int HH = p.hh, WW = p.ww, W2 = WW + 2;
  // this thread's role: output channel co, output parity class (r, c) -- taps kh in {1 - r, 3 - r}, kw in {1 - c, 3 - c}
  // (TF SAME, pads (1, 1): oh = 2 ih - 1 + kh) -- and one quarter of the class's hh x ww pixels
  const int co = tid & 63, cr = (tid >> 7) & 1, cc = (tid >> 6) & 1, q = tid >> 8;
  float wr[4][C0];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int kh = (t >> 1) ? 3 - cr : 1 - cr, kw = (t & 1) ? 3 - cc : 1 - cc;
    const float* src = p.w + ((size_t)((kh * 4 + kw) * SD_C1 + co)) * C0;
#pragma unroll
    for (int v = 0; v < C0 / 4; ++v) {
      const float4 t4 = sd_ld4(src + 4 * v, p.w_al);
      wr[t][4 * v] = t4.x; wr[t][4 * v + 1] = t4.y; wr[t][4 * v + 2] = t4.z; wr[t][4 * v + 3] = t4.w;
    }
  }
  const float bv = p.bias != nullptr ? p.bias[co] : 0.f;
  sd_stage_x<C0>(p, xp, b0, ns, tid);
  __syncthreads();
  float amx = 0.f;
  for (int s = 0; s < ns; ++s) {
    const float* img = xp + s * (HH + 2) * W2 * C0;
    float* out = p.y + (size_t)(b0 + s) * (4 * HH * WW) * SD_C1 + co;
    for (int pix = q; pix < HH * WW; pix += 4) {
      const int i = pix / WW, j = pix - i * WW;
      // padded input rows / columns of the two row / column taps: tap a (kh = 1 - r) reads row i + r, b row i - 1 + r
      const float* ra = img + ((i + 1 + cr) * W2) * C0;
      const float* rb = img + ((i + cr) * W2) * C0;
      const int ca = (j + 1 + cc) * C0, cb = (j + cc) * C0;
      float acc = bv;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float* src = ((t >> 1) ? rb : ra) + ((t & 1) ? cb : ca);
#pragma unroll
        for (int v = 0; v < C0 / 4; ++v) {
          const float4 xv = *reinterpret_cast<const float4*>(src + 4 * v);   // (same address in all lanes of the wave)
          acc = fmaf(xv.x, wr[t][4 * v], acc);
          acc = fmaf(xv.y, wr[t][4 * v + 1], acc);
          acc = fmaf(xv.z, wr[t][4 * v + 2], acc);
          acc = fmaf(xv.w, wr[t][4 * v + 3], acc);
        }
      }
      const float o = odin_act(p.act, acc);
      amx = fmaxf(amx, fabsf(o));
      out[(size_t)((2 * i + cr) * (2 * WW) + 2 * j + cc) * SD_C1] = o;
    }
  }
  odin_amax_commit_wg(p.y_amax, amx, tid, SD_NT, ared, blockIdx.x);
}

template <int C0>
__global__ __launch_bounds__(SD_NT) void smalldeconv_bwd_kernel(SDParams p) {
  ODIN_DYN_SMEM(float, sm);
  __shared__ float ared[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b0 = blockIdx.x * p.S;
  const int ns = (p.B - b0 < p.S) ? p.B - b0 : p.S;
  const int HH = p.hh, WW = p.ww, W2 = WW + 2, npix = 4 * HH * WW, PH = 2 * HH + 2, PW = 2 * WW + 2;
  const int N0 = HH * WW * C0;
  // LDS: xp [S][hh + 2][ww + 2][C0] | w1d [16 taps][C0 (k)][64 (cs C0 + ci)] | g1p [S][PH][PW][SD_GP]
  float* xp = sm;
  float* w1d = sm + ((p.S * (HH + 2) * W2 * C0 + 3) & ~3);
  float* g1p = w1d + 16 * SD_C1 * C0;
  // ---- the output gradient as a zero-bordered image: batches of 4 x 16-byte loads per thread in flight ----
  {
    const float4* src = reinterpret_cast<const float4*>(p.dy + (size_t)b0 * npix * SD_C1);
    const int units = p.S * PH * PW * (SD_C1 / 4);
    for (int e0 = 0; e0 < units; e0 += SD_NT * 4) {
      float4 r[4];
      bool in[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * SD_NT + tid;
        const int pp = e >> 4, c4 = e & 15;
        const int s = pp / (PH * PW), rem = pp - s * (PH * PW);
        const int pr = rem / PW, pc = rem - pr * PW;
        in[u] = e < units && s < ns && pr >= 1 && pr <= 2 * HH && pc >= 1 && pc <= 2 * WW;
        r[u] = src[in[u] ? ((s * npix + (pr - 1) * (2 * WW) + pc - 1) << 4) + c4 : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * SD_NT + tid;
        if (e < units)
          *reinterpret_cast<float4*>(g1p + (e >> 4) * SD_GP + 4 * (e & 15)) = in[u] ? r[u] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
  // w [tap][co][ci] -> w1d [tap][k][cs][ci] with co = cs C0 + k: in the data gradient a lane (cs, ci) reads its C0 weights
  // of a tap at stride 64 floats, the 64 lanes of a wave consecutive floats
  if (p.dx != nullptr) {
    constexpr int units = 16 * SD_C1 * C0 / 4;
    for (int e = tid; e < units; e += SD_NT) {
      const int ci4 = e % (C0 / 4), tc = e / (C0 / 4), co = tc & 63, tap = tc >> 6;
      const int cs = co / C0, k = co - cs * C0;
      *reinterpret_cast<float4*>(w1d + ((tap * C0 + k) * (SD_C1 / C0) + cs) * C0 + 4 * ci4) = sd_ld4(p.w + 4 * e, p.w_al);
    }
  }
  if (p.slab != nullptr) sd_stage_x<C0>(p, xp, b0, ns, tid);   // (the weight gradient's other operand)
  __syncthreads();
  // ---- weight gradient: thread (channel co, parity class (r, c), tap t of the class's four) owns dW[kh][kw][co][0..C0);
  // sums run over the samples, then the hh x ww output pixels of the class, in order ----
  if (p.slab != nullptr) {
    const int co = tid & 63, cr = (tid >> 7) & 1, cc = (tid >> 6) & 1, t = tid >> 8;
    const int kh = (t >> 1) ? 3 - cr : 1 - cr, kw = (t & 1) ? 3 - cc : 1 - cc;
    const int roff = (t >> 1) ? cr : 1 + cr, coff = (t & 1) ? cc : 1 + cc;   // padded input (row, column) = (i, j) + these
    float acc[C0];
#pragma unroll
    for (int c = 0; c < C0; ++c) acc[c] = 0.f;
    for (int s = 0; s < ns; ++s) {
      const float* img = xp + s * (HH + 2) * W2 * C0;
      const float* gimg = g1p + (size_t)s * PH * PW * SD_GP + co;
      for (int i = 0; i < HH; ++i) {
        const float* row = img + ((i + roff) * W2 + coff) * C0;
        const float* grow = gimg + ((2 * i + cr + 1) * PW + cc + 1) * SD_GP;
        for (int j = 0; j < WW; ++j) {
          const float g = grow[2 * j * SD_GP];
#pragma unroll
          for (int v = 0; v < C0 / 4; ++v) {
            const float4 xv = *reinterpret_cast<const float4*>(row + j * C0 + 4 * v);
            acc[4 * v] = fmaf(xv.x, g, acc[4 * v]);
            acc[4 * v + 1] = fmaf(xv.y, g, acc[4 * v + 1]);
            acc[4 * v + 2] = fmaf(xv.z, g, acc[4 * v + 2]);
            acc[4 * v + 3] = fmaf(xv.w, g, acc[4 * v + 3]);
          }
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(p.slab + (size_t)blockIdx.x * (16 * SD_C1 * C0) +
                                            (size_t)((kh * 4 + kw) * SD_C1 + co) * C0);
#pragma unroll
    for (int v = 0; v < C0 / 4; ++v) dst[v] = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
  }
  // ---- data gradient: dx[s][ih][iw][ci] = act'(aux) * sum over (kh, kw, co) of dy[2 ih - 1 + kh][2 iw - 1 + kw][co]
  // * W[kh][kw][co][ci]: wave = input pixels wave, wave + 16, ...; lane (cs, ci) sums its C0 channels co = cs C0 + k of
  // all 16 taps (their gradient pixels lie inside the zero-bordered image), the 64 / C0 lanes of a ci meet by shuffles ----
  float amx = 0.f;
  if (p.dx != nullptr) {
    const int cs = lane / C0, ci = lane - cs * C0;
    const OdinRun RX = odin_run(p.aux != nullptr ? p.aux : p.x,
                                p.aux != nullptr ? (unsigned)((size_t)p.B * N0 * 4) : 0u);
    for (int s = 0; s < ns; ++s) {
      const float* gimg = g1p + (size_t)s * PH * PW * SD_GP + cs * C0;
      for (int pix = wave; pix < HH * WW; pix += SD_NT / 64) {
        const int i = pix / WW, j = pix - i * WW;
        const float* gp0 = gimg + (2 * i * PW + 2 * j) * SD_GP;   // padded pixel (2 i + kh, 2 j + kw) is tap (kh, kw)'s
        float acc = 0.f;
#pragma unroll 1   // (16 taps unrolled at once: 161 spilled registers at C0 = 16)
        for (int kh = 0; kh < 4; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 4; ++kw) {
            const float* gp = gp0 + (kh * PW + kw) * SD_GP;
            const float* wp = w1d + ((kh * 4 + kw) * C0) * SD_C1 + lane;
#pragma unroll
            for (int v = 0; v < C0 / 4; ++v) {
              const float4 g = *reinterpret_cast<const float4*>(gp + 4 * v);
              acc = fmaf(g.x, wp[(4 * v) * SD_C1], acc);
              acc = fmaf(g.y, wp[(4 * v + 1) * SD_C1], acc);
              acc = fmaf(g.z, wp[(4 * v + 2) * SD_C1], acc);
              acc = fmaf(g.w, wp[(4 * v + 3) * SD_C1], acc);
            }
          }
        }
#pragma unroll
        for (int m = C0; m < 64; m <<= 1) acc += __shfl_xor(acc, m);
        if (cs == 0) {
          const size_t o = (size_t)(b0 + s) * N0 + pix * C0 + ci;
          float v = acc;
          if (p.aux != nullptr) v *= odin_act_grad(p.aux_act, odin_run_load1(RX, (unsigned)(o * 4)));
          p.dx[o] = v;
          amx = fmaxf(amx, fabsf(v));
        }
      }
    }
  }
  odin_amax_commit_wg(p.dx != nullptr ? p.dx_amax : nullptr, amx, tid, SD_NT, ared, blockIdx.x);
}

// ---- the generic form of the forward pass (round 6): any kernel size <= 5, stride 2, C0 in {4, 8, 12, 16}, any SAME pads.
// MNIST's first Conv2DTranspose (7 x 7 x 4 -> 14 x 14 x 64, k5 s2; image_networks.py:255-258) has 4 input channels: the
// implicit-GEMM families need multiples of 8, so it ran on the generic gather kernel followed by an absmax pass for its
// range word: 62 + 5 us for 0.08 GFLOP (profiles/r05_bench_mnist_conv_b128_per_op.txt).  One sample per workgroup:
// the input image and the whole weight tensor (K K 64 C0 floats, <= 100 KB) in LDS, a wave owns output pixels
// wave, wave + 16, ... (the valid taps of a pixel are wave-uniform: no divergence), lane = output channel.
struct SDGParams {
  const float* x; const float* w; const float* bias; float* y; unsigned* y_amax;
  int B, H, W, C0, K, pt, pl, act, w_al;
};

__global__ __launch_bounds__(SD_NT) void smalldeconv_gen_fwd_kernel(SDGParams p) {
  ODIN_DYN_SMEM(float, sm);   // ws [K K][64][C0] | xs [H W][C0]
  __shared__ float ared[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int C0 = p.C0, K = p.K, HW = p.H * p.W, OW = 2 * p.W, NW = K * K * SD_C1 * C0;
  float* ws = sm;
  float* xs = sm + NW;
  for (int e = 4 * tid; e < NW; e += 4 * SD_NT) *reinterpret_cast<float4*>(ws + e) = sd_ld4(p.w + e, p.w_al);
  for (int e = tid; e < HW * C0; e += SD_NT) xs[e] = p.x[(size_t)b * HW * C0 + e];
  const float bv = p.bias != nullptr ? p.bias[lane] : 0.f;
  __syncthreads();
  float amx = 0.f;
  for (int pix = wave; pix < 4 * HW; pix += SD_NT / 64) {
    const int oh = pix / OW, ow = pix - oh * OW;
    float acc = bv;
    // oh = 2 ih - pt + kh: kh runs over the taps of oh's parity whose ih lands inside the image
    for (int kh = (oh + p.pt) & 1; kh < K; kh += 2) {
      const int ih = (oh + p.pt - kh) >> 1;
      if (ih < 0 || ih >= p.H) continue;
      for (int kw = (ow + p.pl) & 1; kw < K; kw += 2) {
        const int iw = (ow + p.pl - kw) >> 1;
        if (iw < 0 || iw >= p.W) continue;
        const float* wp = ws + ((kh * K + kw) * SD_C1 + lane) * C0;
        const float* xp = xs + (ih * p.W + iw) * C0;
        for (int c = 0; c < C0; c += 4) {
          const float4 wv = *reinterpret_cast<const float4*>(wp + c);
          const float4 xv = *reinterpret_cast<const float4*>(xp + c);   // (same address in all lanes)
          acc = fmaf(xv.x, wv.x, acc); acc = fmaf(xv.y, wv.y, acc);
          acc = fmaf(xv.z, wv.z, acc); acc = fmaf(xv.w, wv.w, acc);
        }
      }
    }
    const float o = odin_act(p.act, acc);
    amx = fmaxf(amx, fabsf(o));
    p.y[((size_t)b * 4 * HW + pix) * SD_C1 + lane] = o;
  }
  odin_amax_commit_wg(p.y_amax, amx, tid, SD_NT, ared, blockIdx.x);
}

// samples per workgroup: one up to batch 256 (every CU busy, 256 slab rows), more beyond to stay within
// ODIN_MAX_SLAB_BLOCKS rows.  Same-box sweep at batch 256 (dSprites / Shapes3D step, ms): S = 1: 0.5066 / 0.568,
// S = 2: 0.5101 / 0.5911, S = 4: 0.5248 / 0.5807 (tools/r05_sdprobe.sh)
int sd_samples(int B) { return (B + ODIN_MAX_SLAB_BLOCKS - 1) / ODIN_MAX_SLAB_BLOCKS; }

size_t sd_bwd_lds(int S, int hh, int ww, int C0) {
  return ((size_t)((S * (hh + 2) * (ww + 2) * C0 + 3) & ~3) + (size_t)16 * SD_C1 * C0 +
          (size_t)S * (2 * hh + 2) * (2 * ww + 2) * SD_GP) * 4;
}

template <typename K>
void sd_set_lds(K kern, size_t bytes) {
#ifndef ODIN_SIM
  if (bytes > 48 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess)
    (void)hipGetLastError();
#else
  (void)kern; (void)bytes;
#endif
}

void sd_fill(SDParams& p, const odin_conv_desc* d) {
  memset(&p, 0, sizeof(p));
  p.B = d->B; p.hh = d->H; p.ww = d->W; p.act = d->act;
  p.S = sd_samples(d->B);
}

}  // namespace

// Conv2DTranspose 4x4 / stride 2 / SAME pads (1, 1), Cin in {8, 16}, Cout = 64, at most 64 input pixels; the exact-fp32
// switch keeps it (it IS exact fp32)
bool odin_smalldeconv_applicable(const odin_conv_desc* d) {
  if (ODIN_DIAG_ENV("ODIN_NOSMALLDECONV")) return false;
  if (d->KH != 4 || d->KW != 4 || d->stride != 2 || d->pad_t != 1 || d->pad_l != 1 || d->center) return false;
  if (d->Cout != SD_C1 || (d->Cin != 8 && d->Cin != 16) || d->OH != 2 * d->H || d->OW != 2 * d->W) return false;
  if (d->H < 1 || d->W < 1 || d->H * d->W > 64) return false;
  if ((long)d->B * d->OH * d->OW * d->Cout >= (1L << 29)) return false;
  return sd_bwd_lds(sd_samples(d->B), d->H, d->W, d->Cin) <= 150 * 1024;
}

// the generic forward: Conv2DTranspose(64, k <= 5, stride 2, SAME) from a thin small image
bool odin_smalldeconv_gen_applicable(const odin_conv_desc* d) {
  if (ODIN_DIAG_ENV("ODIN_NOSMALLDECONV")) return false;
  if (d->KH != d->KW || d->KH < 2 || d->KH > 5 || d->stride != 2 || d->center) return false;
  if (d->Cout != SD_C1 || (d->Cin & 3) != 0 || d->Cin < 4 || d->Cin > 16 || d->OH != 2 * d->H || d->OW != 2 * d->W) return false;
  if (d->H < 1 || d->W < 1 || d->H * d->W > 64 || d->B > 65535) return false;
  if ((long)d->B * d->OH * d->OW * d->Cout >= (1L << 29)) return false;
  return (size_t)(d->KH * d->KW * SD_C1 * d->Cin + d->H * d->W * d->Cin) * 4 <= 120 * 1024;
}
int odin_smalldeconv_gen_fwd(const float* x, const float* w, const float* bias, float* y, const odin_conv_desc* d,
                             void* stream) {
  SDGParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.y_amax = d->y_amax;
  p.B = d->B; p.H = d->H; p.W = d->W; p.C0 = d->Cin; p.K = d->KH; p.pt = d->pad_t; p.pl = d->pad_l; p.act = d->act;
  p.w_al = (((size_t)w) & 15) == 0;
  const size_t lds = (size_t)(d->KH * d->KW * SD_C1 * d->Cin + d->H * d->W * d->Cin) * 4;
  sd_set_lds(&smalldeconv_gen_fwd_kernel, lds);
  ODIN_LAUNCH(smalldeconv_gen_fwd_kernel, dim3(d->B), dim3(SD_NT), lds, stream, p);
  return odin_check_launch("smalldeconv_gen_fwd");
}

int odin_smalldeconv_rows(const odin_conv_desc* d) {
  const int S = sd_samples(d->B);
  return (d->B + S - 1) / S;
}

int odin_smalldeconv_fwd(const float* x, const float* w, const float* bias, float* y, const odin_conv_desc* d,
                         void* stream) {
  SDParams p;
  sd_fill(p, d);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.y_amax = d->y_amax;
  p.w_al = (((size_t)w) & 15) == 0;
  const int rows = odin_smalldeconv_rows(d);
  const size_t lds = (size_t)p.S * (d->H + 2) * (d->W + 2) * d->Cin * 4;
  if (d->Cin == 8) ODIN_LAUNCH((smalldeconv_fwd_kernel<8>), dim3(rows), dim3(SD_NT), lds, stream, p);
  else ODIN_LAUNCH((smalldeconv_fwd_kernel<16>), dim3(rows), dim3(SD_NT), lds, stream, p);
  return odin_check_launch("smalldeconv_fwd");
}

// either half may be left out (dx == NULL / slab == NULL); a dry run (both NULL) only reports the slab rows
int odin_smalldeconv_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                         float* slab, int* rows_out, const odin_conv_desc* d, void* stream) {
  const int rows = odin_smalldeconv_rows(d);
  if (rows_out) *rows_out = rows;
  if (dx == nullptr && slab == nullptr) return 0;
  if ((((size_t)dy | (size_t)slab) & 15) != 0) return odin_fail(-2, "smalldeconv: dy / slab must be 16-byte aligned");
  SDParams p;
  sd_fill(p, d);
  p.x = x; p.w = w; p.dy = dy; p.dx = dx; p.slab = slab; p.dx_amax = d->dx_amax;
  p.w_al = (((size_t)w) & 15) == 0;
  p.aux = (aux != nullptr && aux_act != 0) ? aux : nullptr; p.aux_act = aux_act;
  const size_t lds = sd_bwd_lds(p.S, d->H, d->W, d->Cin);
  if (d->Cin == 8) {
    sd_set_lds(&smalldeconv_bwd_kernel<8>, lds);
    ODIN_LAUNCH((smalldeconv_bwd_kernel<8>), dim3(rows), dim3(SD_NT), lds, stream, p);
  } else {
    sd_set_lds(&smalldeconv_bwd_kernel<16>, lds);
    ODIN_LAUNCH((smalldeconv_bwd_kernel<16>), dim3(rows), dim3(SD_NT), lds, stream, p);
  }
  return odin_check_launch("smalldeconv_bwd");
}
