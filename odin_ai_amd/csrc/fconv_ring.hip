// fconv_ring.hip -- strided gather convolution 4x4 / stride 2 over 32 reduction channels with a
// ROLLING ROW WINDOW in LDS, LDS-DMA producers and matrix-core consumers.
//
// Serves (all TF `SAME`, pads (1, 1)):
//   Conv2D(k4, s2) forward over 32 input channels          (image_networks.py:464-466, encoder1/2)
//   Conv2DTranspose(k4, s2) DATA GRADIENT, 32 output maps  (tape.gradient of decoder3 / decoder4,
//                                                           image_networks.py:503-506)
// i.e.  out[b, oh, ow, n] = sum_{kh, kw, c < 32} in[b, 2 oh - 1 + kh, 2 ow - 1 + kw, c] W[kh, kw, c, n]
// -- the dominant launches of the training step (decoder4's data gradient alone is 8.6 GFLOP).
//
// Why a second kernel beside gather_conv.hip: there a workgroup stages a [10 rows x 66 pixels] patch
// through registers (ds_write) between two barriers, one wave per SIMD, so patch commit + issue +
// epilogue (8 k of 26 k cycles per tile, in-kernel stamps profiles/r02_*) sit in series with the
// MFMAs, and consecutive tiles re-fetch their two halo rows.  Here:
//   * the input rows live in a RING of LDS row slots (slot = global padded row index mod NSLOT); a
//     tile of TRO output rows needs 2 TRO + 2 input rows, the next tile shares two of them, so
//     every input row is fetched ONCE per workgroup (images follow each other through a shared
//     zero row);
//   * waves 4-7 (one per SIMD, beside the consumers) do nothing but issue LDS-DMA
//     (buffer_load ... lds, no VGPR data path) for the rows of tile t + 1 while waves 0-3 multiply
//     tile t: ONE barrier per tile;
//   * a row is stored as two parity planes (even / odd padded column), so the stride-2 taps read
//     CONSECUTIVE pixel slots; a slot is 128 B = 8 pieces of 4 channels, piece p of slot j sits at
//     position p ^ ((j >> 1) & 7): the 16-byte operand reads of every 16-lane group hit 16 distinct
//     bank quads (conflict-free), and the image is DMA-compatible (lane-linear LDS, swizzle on the
//     SOURCE address);
//   * v_mfma_f32_16x16x4_f32 (exact fp32, same FLOP rate as 32x32x2): a wave owns 16 output pixels
//     x 32 output channels as two 16 x 16 accumulators that alternate (40-cycle dependency hidden),
//     one 16-byte read feeds 4 MFMAs per operand: 3 LDS reads per 8 MFMAs, no cross-wave reduction;
//   * the weight slice is re-laid once per workgroup as [tap][4-channel piece][out channel][4].
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

__device__ float odin_fr_zero_row[2304];  // 9 KB of zeros: the DMA source of SAME-padding rows

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifdef ODIN_SIM
#define ODIN_UNIFORM_FR(x) (x)
#else
#define ODIN_UNIFORM_FR(x) __builtin_amdgcn_readfirstlane(x)
#endif

__device__ __forceinline__ f32x4v mfma16(float a, float b, f32x4v c) {
#ifdef ODIN_SIM
  return sim::mfma_16x16x4(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

struct FRParams {
  const float* in;     // [B, H, W, 32]
  const float* w;      // [16 taps][32][CO]
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, OH, OW, CO], out *= ELU'(aux)
  float* out;          // [B, OH, OW, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] partial column sums of out (may be null)
  int B, H, W, OH, OW, CO;
  int CS, ci_off;      // channels per input pixel in memory (32 or 64) and the first of this pass's 32
  int TRO;             // output rows per tile (TRO * OW == 64)
  int NSLOT, RB;       // ring slots, bytes per ring row
  int tiles_per_img, n_tiles, tiles_per_wg;
  long long* stamps;   // diagnostics: s_memtime stamps of workgroup 0 (consumer wave 0: [0,32), producer wave 4: [32,64))
};

constexpr int FR_WBYTES = 16 * 8 * 32 * 16;  // weight image: 64 KB

#if defined(ODIN_SIM) || !defined(ODIN_DIAG)  // in-kernel stamps: diagnostics build only (make diag)
#define FR_STAMP(base, k) ((void)0)
#else
#define FR_STAMP(base, k)                                                                        \
  do {                                                                                           \
    if (p.stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && stamp_i < 31)  \
      p.stamps[(base) + stamp_i++] = ((long long)(k) << 56) | (long long)(clock64() & 0xFFFFFFFFFFFFFFll); \
  } while (0)
#endif

// EPI 1: bias + ELU (Conv2D forward); EPI 2: linear, x ELU'(aux), column sums (deconv data-gradient);
// EPI 0: raw partial sums (first of two reduction passes over 64 input channels).  ACC: add the partial
// sums the previous pass left in `out` before the epilogue.
template <int EPI, bool ACC>
__global__ __launch_bounds__(512) void fconv_ring_kernel(FRParams p) {
  ODIN_DYN_SMEM(char, smem);
  char* wl = smem;
  char* ring = smem + FR_WBYTES;
  __shared__ float cred[4 * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = ODIN_UNIFORM_FR(tid >> 6);
  const int n0 = blockIdx.y * 32;
  const int HP = p.H + 1;              // period of the global padded row index: [zero row][H rows]
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  const int hs = p.W / 2 + 1;          // slots per parity plane
  int stamp_i = 0;
  (void)stamp_i;
  if (wave == 0) FR_STAMP(0, 1);
  // global padded row index of the first input row of tile T: g0(T) = HP * b + 2 TRO t
  auto g0_of = [&](int T) {
    const int b = T / p.tiles_per_img, t = T - b * p.tiles_per_img;
    return HP * b + 2 * p.TRO * t;
  };
  const int nlive = 2 * p.TRO + 2;
  if (T0 >= T1) return;  // (workgroup-uniform; the launcher never creates such a workgroup)

  if (wave >= 4) {
    // ---------------------------- producers: LDS-DMA ----------------------------
    // This wave shares its SIMD with a consumer that issues MFMAs back to back: every VALU
    // instruction here waits for a gap in that stream (an item built with integer divisions took
    // ~1400 cycles beside the consumer: 12.8 k cycles per tile, the consumers waited at the barrier).
    // So everything per-lane is computed ONCE (source offset and LDS piece of each (parity, chunk)
    // item this wave owns), the row walk is scalar counters, and an item is a few SALU instructions
    // plus the DMA.
    const int pw = wave - 4;
    const int cpr = (p.W / 2) / 8;          // 1 KB DMA chunks per parity plane of a row: 4 (W 64) or 2 (W 32)
    const int ipr = 2 * cpr;                // DMA instructions per row: 8 or 4
    const int ipw = (ipr - pw + 3) / 4;     // ... of this producer wave: 2, 1 or (16-pixel rows, waves 2, 3) 0
    unsigned gofs[2];                       // byte offset inside an input row of this lane's 16 bytes
    int dofs[2];                            // byte offset inside a ring row of the item's 1 KB
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int item = pw + 4 * k;
      const int par = item / cpr, ch = item - par * cpr;
      // LDS pieces [64 ch, 64 ch + 64) of this plane's real-pixel span.  Parity plane 1 holds padded
      // columns 1, 3, .. (input columns 0, 2, ..) in slots 0 .. W/2 - 1; parity plane 0 holds padded
      // columns 2, 4, .. (input columns 1, 3, ..) in slots 1 .. W/2
      const int q = ch * 64 + lane;
      const int jrel = q >> 3, pos = q & 7;
      const int j = par ? jrel : jrel + 1;
      const int iw = par ? 2 * jrel : 2 * jrel + 1;
      const int c4 = pos ^ ((j >> 1) & 7);
      gofs[k] = (unsigned)((iw * p.CS + 4 * c4) * 4);
      dofs[k] = (par * hs + (par ? 0 : 1)) * 128 + ch * 1024;
    }
    const OdinRun ZR = odin_run(odin_fr_zero_row, (unsigned)sizeof(odin_fr_zero_row));
    // scalar row walk: g = global padded row, gi = g mod HP (0: zero row), b = image, slot = g mod NSLOT
    int g_hi = ODIN_UNIFORM_FR(g0_of(T0));
    int gi = ODIN_UNIFORM_FR(g_hi % HP), bimg = ODIN_UNIFORM_FR(g_hi / HP);
    int slot = ODIN_UNIFORM_FR(g_hi % p.NSLOT);
    int g_next0 = g_hi;                     // g0 of the tile whose rows are issued next
    int t_in_img = ODIN_UNIFORM_FR(T0 % p.tiles_per_img);
    // iteration T issues the rows of tile T that are not resident yet (T0: all 2 TRO + 2 of them;
    // afterwards 2 TRO, or 2 TRO + 1 across an image boundary) and then meets the consumers at
    // barrier T.  After barrier T the consumers multiply tile T while iteration T + 1 streams the
    // next rows into slots whose rows belong to tiles < T.
    for (int T = T0; T <= T1; ++T) {
      if (T < T1) {
        const int g_need = g_next0 + nlive;
        for (int g = g_hi; g < g_need; ++g) {   // wave-uniform
          char* rowl = ring + (size_t)slot * p.RB;
          if (gi == 0) {
            for (int k = 0; k < ipw; ++k)
              odin_run_dma16(ZR, reinterpret_cast<float*>(rowl + dofs[k]), (unsigned)(lane * 16), lane);
          } else {
            const OdinRun R = odin_run(p.in + (size_t)(bimg * p.H + gi - 1) * p.W * p.CS + p.ci_off,
                                       (unsigned)((p.W * p.CS - p.ci_off) * 4));
            for (int k = 0; k < ipw; ++k)
              odin_run_dma16(R, reinterpret_cast<float*>(rowl + dofs[k]), gofs[k], lane);
          }
          if (++gi == HP) { gi = 0; ++bimg; }
          if (++slot == p.NSLOT) slot = 0;
        }
        g_hi = g_need;
        // first input row of the next tile: + 2 TRO, + 1 more across an image boundary
        g_next0 += 2 * p.TRO;
        if (++t_in_img == p.tiles_per_img) { t_in_img = 0; g_next0 += 1; }
      }
      if (pw == 0) FR_STAMP(32, 21);
      odin_wait_vmem();
      if (pw == 0) FR_STAMP(32, 22);
      __syncthreads();  // barrier T
      if (pw == 0) FR_STAMP(32, 20);
    }
  } else {
  // ---------------------------- consumers: LDS reads + MFMA ----------------------------
  // once per workgroup, beside the producers' first DMAs: zero the two SAME-padding slots of every
  // ring row (no DMA ever writes them) and re-lay the weight slice as [tap][piece][out channel][4]
  for (int e = tid; e < p.NSLOT * 16; e += 256) {
    const int sl = e >> 4, q = e & 15;
    char* rowl = ring + (size_t)sl * p.RB + ((q & 8) ? (2 * hs - 1) * 128 : 0);  // plane 1 slot W/2 | plane 0 slot 0
    reinterpret_cast<float4*>(rowl)[q & 7] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {
    const OdinRun WR = odin_run(p.w, (unsigned)((size_t)16 * p.CS * p.CO * 4));
#pragma unroll 1
    for (int e0 = tid; e0 < 16 * 8 * 32; e0 += 256 * 8) {
      float v[8][4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 32 loads in flight per lane; item = (tap, piece c4, out channel)
        const int e = e0 + 256 * u;
        const int co = e & 31, c4 = (e >> 5) & 7, tap = e >> 8;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          v[u][k] = odin_run_load1(WR, n0 + co < p.CO
                                           ? (unsigned)(((tap * p.CS + p.ci_off + 4 * c4 + k) * p.CO + n0 + co) * 4)
                                           : ODIN_OOB);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        reinterpret_cast<float4*>(wl)[e0 + 256 * u] = make_float4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
  if (wave == 0) FR_STAMP(0, 2);
  const int l15 = lane & 15, kq = lane >> 4;
  // this lane's output pixel inside the tile
  // (8-pixel output rows: a wave's 16 pixels are two rows -> the row is a per-lane quantity)
  const int orow = (p.OW == 32) ? (wave >> 1) : (p.OW == 16) ? wave : 2 * wave + (l15 >> 3);
  const int ocol = (p.OW == 32) ? 16 * (wave & 1) + l15 : (p.OW == 16) ? l15 : (l15 & 7);
  const int orow_w = (p.OW == 32) ? (wave >> 1) : (p.OW == 16) ? wave : 2 * wave;  // wave-uniform part
  // lane-constant byte offsets inside a ring row: [column shift 0 / 1][channel group 0 / 1]
  int lo[2][2];
#pragma unroll
  for (int sh = 0; sh < 2; ++sh)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int j = ocol + sh;
      lo[sh][g] = j * 128 + (((4 * g + kq) ^ ((j >> 1) & 7)) << 4);
    }
  const char* wlane = wl + ((kq * 32 + l15) << 4);
  float csum[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) csum[i] = 0.f;
  float4 bias4[2];
  if (EPI == 1) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = n0 + cb * 16 + 4 * kq;
      bias4[cb] = (p.bias != nullptr && n + 3 < p.CO) ? *reinterpret_cast<const float4*>(p.bias + n)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();  // barrier T0: the first tile's rows are in
  for (int T = T0; T < T1; ++T) {
    if (wave == 0) FR_STAMP(0, 10);
    const int b = T / p.tiles_per_img, t = T - b * p.tiles_per_img;
    const int g0 = HP * b + 2 * p.TRO * t;
    const int oh = p.TRO * t + orow;
    const size_t opix = ((size_t)b * p.OH + oh) * p.OW + ocol;
    float4 ax[2], pv[2];
    if (EPI == 2) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        ax[cb] = *reinterpret_cast<const float4*>(p.aux + opix * p.CO + n0 + cb * 16 + 4 * kq);
    }
    if (ACC) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
        pv[cb] = *reinterpret_cast<const float4*>(p.out + opix * p.CO + n0 + cb * 16 + 4 * kq);
    }
    f32x4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    // 32 steps = 16 taps x 2 channel groups; operands of step s + 1 are read under the MFMAs of step s
    float4 bq[2], a0[2], a1[2];
    // ring rows of this lane's 4 tap rows (one wave-uniform modulo per tile; the lane's own two rows
    // further down when a wave spans two 8-pixel output rows)
    const char* rowb[4];
    {
      int sl = (g0 + 2 * orow_w) % p.NSLOT + 2 * (orow - orow_w);
      if (sl >= p.NSLOT) sl -= p.NSLOT;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        rowb[kh] = ring + (size_t)sl * p.RB;
        if (++sl == p.NSLOT) sl = 0;
      }
    }
    auto loads = [&](int s, float4& vb, float4& va0, float4& va1) {
      const int tap = s >> 1, g = s & 1;
      const int kh = tap >> 2, kw = tap & 3;
      const char* rowp = rowb[kh] + ((kw & 1) * hs) * 128;
      vb = *reinterpret_cast<const float4*>(rowp + lo[kw >> 1][g]);
      const char* wp = wlane + ((tap * 8 + 4 * g) * 32 << 4);
      va0 = *reinterpret_cast<const float4*>(wp);
      va1 = *reinterpret_cast<const float4*>(wp + 256);
    };
    loads(0, bq[0], a0[0], a1[0]);
    ODIN_SCHED_FENCE();
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s + 1 < 32) loads(s + 1, bq[nxt], a0[nxt], a1[nxt]);
      acc0 = mfma16(a0[cur].x, bq[cur].x, acc0);
      acc1 = mfma16(a1[cur].x, bq[cur].x, acc1);
      acc0 = mfma16(a0[cur].y, bq[cur].y, acc0);
      acc1 = mfma16(a1[cur].y, bq[cur].y, acc1);
      acc0 = mfma16(a0[cur].z, bq[cur].z, acc0);
      acc1 = mfma16(a1[cur].z, bq[cur].z, acc1);
      acc0 = mfma16(a0[cur].w, bq[cur].w, acc0);
      acc1 = mfma16(a1[cur].w, bq[cur].w, acc1);
      // the three 16-byte reads of step s + 1 go out in the shadow of this step's first MFMAs: a
      // full step (256 cycles) lies between a read and its first use
      if (s + 1 < 32) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          ODIN_SCHED_GROUP(ODIN_SG_MFMA, 1);
          ODIN_SCHED_GROUP(ODIN_SG_DSREAD, 1);
        }
        ODIN_SCHED_GROUP(ODIN_SG_MFMA, 5);
      }
      ODIN_SCHED_FENCE();
    }
    if (wave == 0) FR_STAMP(0, 11);
    // ---- epilogue: lane = pixel l15 x channels n0 + cb * 16 + 4 kq + 0..3 ----
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const f32x4v a = cb == 0 ? acc0 : acc1;
      float v[4] = {a[0], a[1], a[2], a[3]};
      if (ACC) {
        v[0] += pv[cb].x; v[1] += pv[cb].y; v[2] += pv[cb].z; v[3] += pv[cb].w;
      }
      if (EPI == 1) {
        const float bb[4] = {bias4[cb].x, bias4[cb].y, bias4[cb].z, bias4[cb].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float tt = v[k] + bb[k];
          v[k] = fmaxf(tt, 0.f) + (odin_exp2(fminf(tt, 0.f) * 1.44269504088896341f) - 1.f);
        }
      } else if (EPI == 2) {
        const float aa[4] = {ax[cb].x, ax[cb].y, ax[cb].z, ax[cb].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaf(v[k], fminf(aa[k], 0.f), v[k]);  // x (1 + min(y, 0))
#pragma unroll
        for (int k = 0; k < 4; ++k) csum[4 * cb + k] += v[k];
      }
      *reinterpret_cast<float4*>(p.out + opix * p.CO + n0 + cb * 16 + 4 * kq) =
          make_float4(v[0], v[1], v[2], v[3]);
    }
    if (wave == 0) FR_STAMP(0, 12);
    __syncthreads();  // barrier T + 1
  }
  if (EPI == 2 && p.colsum != nullptr) {
    // column sums of this workgroup's outputs: 16 pixel lanes by shuffles, 4 waves through LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = csum[i];
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (l15 == 0) cred[wave * 32 + (i >> 2) * 16 + 4 * kq + (i & 3)] = v;
    }
  }
  }  // consumers
  if (EPI == 2 && p.colsum != nullptr) {
    __syncthreads();
    if (tid < 32 && n0 + tid < p.CO)
      p.colsum[(size_t)blockIdx.x * p.CO + n0 + tid] =
          (cred[tid] + cred[32 + tid]) + (cred[64 + tid] + cred[96 + tid]);
  }
}

}  // namespace

static long long* g_fr_stamps = nullptr;
void odin_fconv_ring_set_stamps(void* buf) { g_fr_stamps = (long long*)buf; }

// consumers and producers meet at __syncthreads(): both roles execute the same number of them
// (1 after the weights, then T1 - T0 + 1, then 1 for the column sums)

static int fr_launch_pass(FRParams p, int epi, bool acc, dim3 grid, size_t lds, void* stream) {
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    const void* fns[5] = {reinterpret_cast<const void*>(&fconv_ring_kernel<1, false>),
                          reinterpret_cast<const void*>(&fconv_ring_kernel<2, false>),
                          reinterpret_cast<const void*>(&fconv_ring_kernel<0, false>),
                          reinterpret_cast<const void*>(&fconv_ring_kernel<1, true>),
                          reinterpret_cast<const void*>(&fconv_ring_kernel<2, true>)};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess)
        (void)hipGetLastError();
    attr_done = true;
  }
#endif
  if (epi == 0) ODIN_LAUNCH((fconv_ring_kernel<0, false>), grid, dim3(512), lds, stream, p);
  else if (epi == 1 && !acc) ODIN_LAUNCH((fconv_ring_kernel<1, false>), grid, dim3(512), lds, stream, p);
  else if (epi == 1) ODIN_LAUNCH((fconv_ring_kernel<1, true>), grid, dim3(512), lds, stream, p);
  else if (!acc) ODIN_LAUNCH((fconv_ring_kernel<2, false>), grid, dim3(512), lds, stream, p);
  else ODIN_LAUNCH((fconv_ring_kernel<2, true>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("fconv_ring");
}

// CI = 32: one launch.  CI = 64: two reduction passes over 32 channels each (the weight image of 64
// channels, 128 KB, does not fit beside the row window): the first leaves raw partial sums in `out`,
// the second adds them and runs the epilogue.
int odin_fconv_ring_launch(const float* in, const float* w, const float* bias, const float* aux,
                           float* out, float* colsum, int* rows_out, int B, int H, int W, int CI,
                           int OH, int OW, int CO, int epi, void* stream) {
  FRParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.CO = CO;
  p.CS = CI; p.ci_off = 0;
  p.TRO = 64 / OW;
  p.NSLOT = 4 * p.TRO + 3;
  p.RB = (W + 2) * 128;
  p.tiles_per_img = OH / p.TRO;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  p.tiles_per_wg = (p.n_tiles + cap - 1) / cap;
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  p.stamps = g_fr_stamps;
  const size_t lds = (size_t)FR_WBYTES + (size_t)p.NSLOT * p.RB;
  if (lds > 159 * 1024) return odin_fail(-2, "fconv_ring: ring does not fit the LDS");
  dim3 grid(gx, gy, 1);
  if (CI == 32) return fr_launch_pass(p, epi, false, grid, lds, stream);
  FRParams q = p;
  q.colsum = nullptr;
  int rc = fr_launch_pass(q, 0, false, grid, lds, stream);
  if (rc != 0) return rc;
  p.ci_off = 32;
  return fr_launch_pass(p, epi, true, grid, lds, stream);
}

bool odin_fconv_ring_applicable(int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                                int pt, int pl, int center) {
  static int off = -1;
  if (off < 0) off = ODIN_DIAG_ENV("ODIN_NOFRING") ? 1 : 0;
  return !off && KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && (CI == 32 || CI == 64) &&
         (CO % 32) == 0 && !center && H == 2 * OH && W == 2 * OW && (OW == 8 || OW == 16 || OW == 32) &&
         (OH % (64 / OW)) == 0;
}
