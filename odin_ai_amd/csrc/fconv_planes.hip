// fconv_planes.hip -- STRIDED gather convolution 4x4 / stride 2 over 32 reduction channels (TF `SAME`,
// pads (1, 1)) with fp32 operands carried through the bf16 matrix pipe as three exact bf16 planes:
//   Conv2D(k4, s2) forward over 32 input channels                    (image_networks.py:462-468)
//   Conv2DTranspose(k4, s2) DATA GRADIENT over 32 output channels     (tape.gradient of decoder3/4)
//       out[b, oh, ow, n] = sum over (kh, kw, c < 32) of in[b, 2 oh - 1 + kh, 2 ow - 1 + kw, c] * W[kh, kw, c, n]
//
// The strided gather reads an input area 4x the output area, so the LDS cannot hold the weight
// planes (96 KB) beside a useful row window.  Instead the REDUCTION is split over the 8 waves of a
// workgroup: wave v owns the taps (kh = v >> 1, kw = 2 (v & 1) + {0, 1}) and keeps their weight
// fragments -- 2 taps x 2 k-halves x 3 planes = 48 registers -- for the whole kernel; all waves
// multiply the same 32-pixel x 32-channel output tile (24 MFMAs each), leave their partial tiles in
// LDS and meet at the tile's barrier; behind it wave v sums the eight partials of accumulator
// registers 2 v, 2 v + 1 and runs their epilogue while the next tile's MFMAs are already going (two
// scratch buffers).  The row window is the one of wgrad_planes.hip (fine rows as two column-parity
// planes of [slot][32 bf16], split once on the way in), here with the 16-byte k-pieces XOR-swizzled by
// (slot >> 2) for conflict-free ds_read_b128.  One workgroup barrier per tile.
#include "odin_device.h"
#include "odin_internal.h"
#include <cstdlib>

namespace {

struct FPParams {
  const float* in;     // [B, 2 OH, 2 OW, 32]
  const float* w;      // [16 taps][32][CO]
  const float* bias;   // EPI 1: [CO]
  const float* aux;    // EPI 2: [B, OH, OW, CO], out *= ELU'(aux)
  float* out;          // [B, OH, OW, CO]
  float* colsum;       // EPI 2: [gridDim.x][CO] partial column sums of out (may be null)
  int B, OH, CO;
  int tiles_per_img, n_tiles, tiles_per_wg;
};

__device__ __forceinline__ void fp_split4(const float4& v, u32x2& h, u32x2& m, u32x2& l) {
  h = odin_u2(odin_pack_bf16(v.x, v.y), odin_pack_bf16(v.z, v.w));
  const float r0 = odin_bf16_rest(v.x), r1 = odin_bf16_rest(v.y), r2 = odin_bf16_rest(v.z),
              r3 = odin_bf16_rest(v.w);
  m = odin_u2(odin_pack_bf16(r0, r1), odin_pack_bf16(r2, r3));
  l = odin_u2(odin_pack_bf16(odin_bf16_rest(r0), odin_bf16_rest(r1)),
              odin_pack_bf16(odin_bf16_rest(r2), odin_bf16_rest(r3)));
}

struct FpItem {
  float4 v;
  int dst;  // byte offset of the hi-plane store inside the ring; < 0: no item (wave-uniform)
};

constexpr int FP_MAXU = 4;

// EPI 1: bias + ELU (Conv2D forward); EPI 2: x ELU'(aux), column sums (deconv data gradient)
template <int EPI, int OW>
__global__ __launch_bounds__(512) void fconv_planes_kernel(FPParams p) {
  constexpr int TC = 32 / OW;            // output rows per tile
  constexpr int WU = 2 * OW;             // input row length
  constexpr int SU = OW + 1;             // slots per column-parity plane
  // one parity plane + one spare slot: the two parity planes then sit 128 bytes apart modulo the 256-byte bank
  // row, so the 8 pixels of a row-fill item (alternating parity) spread over all banks -- with SU * 64 the pixel
  // pairs (1, 2), (3, 4), (5, 6) shared their banks (15-19 % of the LDS cycles of fconv_planes / wgrad_planes)
  constexpr int PARB = (SU + 1) * 64;
  constexpr int PBU = 2 * PARB;
  constexpr int RBU = 3 * PBU;
  constexpr int NSU = 4 * TC + 3;        // live input rows (2 TC + 2) + the next tile's (2 TC + 1 at an image seam)
  constexpr int IPU = WU / 8;            // 1 KB load items per input row
  constexpr int RJ = 8 / IPU > 0 ? 8 / IPU : 1;
  constexpr int RED = 8 * 8 * 64 * 8;    // one reduction buffer: [register pair][wave][lane][8 B]
  ODIN_DYN_SMEM(char, smem);
  char* ring = smem;
  char* red = smem + NSU * RBU;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef ODIN_SIM
  const int wave = tid >> 6;
#else
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int l31 = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const int HU = 2 * p.OH, HPU = HU + 1;
  const int T0 = blockIdx.x * p.tiles_per_wg;
  int T1 = T0 + p.tiles_per_wg;
  if (T1 > p.n_tiles) T1 = p.n_tiles;
  if (T0 >= T1) return;

  // ---- SAME-padding slots (parity plane 0 slot 0, parity plane 1 slot OW) of every ring row and plane ----
  for (int e = tid; e < NSU * 24; e += 512) {
    const int sl = e / 24, rem = e - sl * 24;
    const int pl = rem >> 3, side = (rem >> 2) & 1, piece = rem & 3;
    *reinterpret_cast<float4*>(ring + sl * RBU + pl * PBU + (side ? PARB + OW * 64 : 0) + piece * 16) =
        make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- this wave's weight fragments: taps (kh, kw0), (kh, kw0 + 1); lane = output channel l31, k = 8 half + e ----
  const int kh = wave >> 1, kw0 = 2 * (wave & 1);
  // ---- row fills (as wgrad_planes.hip; the k-pieces of a pixel slot are swizzled by the slot) ----
  const OdinRun RU = odin_run(p.in, (unsigned)((size_t)p.B * HU * WU * 32 * 4));
  int fu_g, fu_gi, fu_b, fu_slot, need_gu0, ft_t;
  {
    const int b0 = T0 / p.tiles_per_img, t0 = T0 - b0 * p.tiles_per_img;
    fu_g = HPU * b0 + 2 * TC * t0;
    fu_gi = 2 * TC * t0;
    fu_b = b0;
    fu_slot = fu_g % NSU;
    need_gu0 = fu_g;
    ft_t = t0;
  }
  const int ch4 = lane & 7, pxl = lane >> 3;
  const int r0w = wave / IPU, cblk = wave - r0w * IPU;
  const int pcw = 8 * cblk + pxl + 1;  // padded column of this lane's pixel: parity pcw & 1, slot pcw >> 1
  const int u_lds = (pcw & 1) * PARB + (pcw >> 1) * 64 + (((ch4 >> 1) ^ (((pcw >> 1) >> 2) & 3)) << 4) + (ch4 & 1) * 8;
  const unsigned u_g = (unsigned)(((8 * cblk + pxl) * 32 + 4 * ch4) * 4);
  const unsigned u_rowbytes = (unsigned)(WU * 32 * 4);
  auto load_fill = [&](FpItem (&iu)[FP_MAXU], bool live) {
    const int nrows = live ? need_gu0 + 2 * TC + 2 - fu_g : 0;
#pragma unroll
    for (int j = 0; j < FP_MAXU; ++j) {
      const int r = r0w + RJ * j;
      const bool valid = r < nrows;
      int gi = fu_gi + r, b = fu_b;
      if (gi >= HPU) { gi -= HPU; ++b; }
      int slot = fu_slot + r;
      if (slot >= NSU) slot -= NSU;
      iu[j].dst = valid ? slot * RBU + u_lds : -1;
      const bool real = valid && gi != 0 && b < p.B;  // gi == 0: the zero row between images
      iu[j].v = odin_run_load4(RU, real ? (unsigned)(b * HU + gi - 1) * u_rowbytes + u_g : ODIN_OOB);
    }
    if (live) {
      fu_g += nrows;
      fu_gi += nrows;
      if (fu_gi >= HPU) { fu_gi -= HPU; ++fu_b; }
      fu_slot += nrows;
      if (fu_slot >= NSU) fu_slot -= NSU;
      need_gu0 += 2 * TC;
      if (++ft_t == p.tiles_per_img) { ft_t = 0; need_gu0 += 1; }
    }
  };
  auto store_item = [&](const FpItem& it) {
#ifdef ODIN_SIM
    if (it.dst < 0) return;
#else
    if (__builtin_amdgcn_readfirstlane(it.dst) < 0) return;  // wave-uniform: a scalar branch
#endif
    u32x2 h, m, l;
    fp_split4(it.v, h, m, l);
    char* d = ring + it.dst;
    *reinterpret_cast<u32x2*>(d) = h;
    *reinterpret_cast<u32x2*>(d + PBU) = m;
    *reinterpret_cast<u32x2*>(d + 2 * PBU) = l;
  };

  // ---- this lane's output pixel inside the tile and its read offsets ----
  const int orow = (OW == 32) ? 0 : (OW == 16) ? (l31 >> 4) : (l31 >> 3);
  const int ocol = (OW == 32) ? l31 : (OW == 16) ? (l31 & 15) : (l31 & 7);
  // B fragment of tap t, k-half kk: slot ocol + (kw >> 1) of parity kw & 1, piece (2 kk + half) ^ ((slot >> 2) & 3)
  int boff[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int kw = kw0 + t, slot = ocol + (kw >> 1);
      boff[t][kk] = (kw & 1) * PARB + slot * 64 + (((2 * kk + half) ^ ((slot >> 2) & 3)) << 4);
    }
  // the accumulator registers this wave finishes: r = 2 wave, 2 wave + 1 -> channels c0, c0 + 1
  const int c0 = n0 + ((2 * wave) & 3) + 8 * ((2 * wave) >> 2) + 4 * half;
  float bias2[2] = {0.f, 0.f};
  if (EPI == 1) { bias2[0] = p.bias[c0]; bias2[1] = p.bias[c0 + 1]; }
  float csum[2] = {0.f, 0.f};

  FpItem iuA[FP_MAXU], iuB[FP_MAXU], iuC[FP_MAXU];
  load_fill(iuA, true);  // (in flight while the weight fragments are fetched and split)
  u32x4 wf[2][2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int tap = kh * 4 + kw0 + t;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = p.w[((size_t)(tap * 32 + 16 * kk + 8 * half + e)) * p.CO + n0 + l31];
      u32x2 h0, m0, l0, h1, m1, l1;
      fp_split4(make_float4(v[0], v[1], v[2], v[3]), h0, m0, l0);
      fp_split4(make_float4(v[4], v[5], v[6], v[7]), h1, m1, l1);
      wf[t][kk][0][0] = h0.x; wf[t][kk][0][1] = h0.y; wf[t][kk][0][2] = h1.x; wf[t][kk][0][3] = h1.y;
      wf[t][kk][1][0] = m0.x; wf[t][kk][1][1] = m0.y; wf[t][kk][1][2] = m1.x; wf[t][kk][1][3] = m1.y;
      wf[t][kk][2][0] = l0.x; wf[t][kk][2][1] = l0.y; wf[t][kk][2][2] = l1.x; wf[t][kk][2][3] = l1.y;
    }

#pragma unroll
  for (int j = 0; j < FP_MAXU; ++j) store_item(iuA[j]);
  load_fill(iuA, T0 + 1 < T1);
  load_fill(iuB, T0 + 2 < T1);
  __syncthreads();

  int b_cur = T0 / p.tiles_per_img, t_cur = T0 - b_cur * p.tiles_per_img;
  int su0 = (HPU * b_cur + 2 * TC * t_cur) % NSU;
  size_t opixP = 0;
  float2 auxP = make_float2(0.f, 0.f);

  // sums the eight partial tiles of registers 2 wave, 2 wave + 1 of tile T - 1 and finishes them
  auto finish = [&](int buf) {
    // scratch layout [register pair][source wave][lane][8 B]: this wave reads pair `wave` of all eight sources --
    // 512 contiguous bytes per read (the round-2 layout [wave][r4][lane][16 B] made these reads 8-byte pieces at
    // a 16-byte stride: 29 % of the kernel's LDS cycles were bank conflicts, profiles/r03_kpmc_planes_8wave.txt)
    const char* q = red + buf * RED + ((wave * 8 * 64 + lane) << 3);
    float2 s = *reinterpret_cast<const float2*>(q);
#pragma unroll
    for (int wv = 1; wv < 8; ++wv) {
      const float2 t = *reinterpret_cast<const float2*>(q + wv * (64 * 8));
      s.x += t.x; s.y += t.y;
    }
    float v[2] = {s.x, s.y};
    if (EPI == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float tt = v[k] + bias2[k];
        v[k] = fmaxf(tt, 0.f) + (odin_exp2(fminf(tt, 0.f) * 1.44269504088896341f) - 1.f);
      }
    } else {
      v[0] = fmaf(v[0], fminf(auxP.x, 0.f), v[0]);  // x ELU'(aux) = 1 + min(aux, 0)
      v[1] = fmaf(v[1], fminf(auxP.y, 0.f), v[1]);
      csum[0] += v[0];
      csum[1] += v[1];
    }
    *reinterpret_cast<float2*>(p.out + opixP * p.CO + c0) = make_float2(v[0], v[1]);
  };

  auto run_tile = [&](int T, FpItem (&ldu)[FP_MAXU], const FpItem (&stu)[FP_MAXU]) {
    // row slots of this lane's two tap rows... one tap row: kh is the wave's
    int su = su0 + 2 * orow + kh;
    if (su >= NSU) su -= NSU;
    const char* rowp = ring + su * RBU;
    u32x4 fb[2][2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[t][kk][pl] = *reinterpret_cast<const u32x4*>(rowp + boff[t][kk] + pl * PBU);
    ODIN_SCHED_FENCE();
    load_fill(ldu, T + 3 < T1);
    const int oh = TC * t_cur + orow;
    const size_t opix = ((size_t)b_cur * p.OH + oh) * OW + ocol;
    float2 auxN = make_float2(0.f, 0.f);
    if (EPI == 2) auxN = *reinterpret_cast<const float2*>(p.aux + opix * p.CO + c0);
    if (T > T0) finish((T - 1) & 1);  // tile T - 1: its partials are complete behind the last barrier
    ODIN_SCHED_FENCE();
    f32x16 acc = f32x16_zero();
#pragma unroll
    for (int m = 0; m < 24; ++m) {
      const int t = (m >> 1) & 1, kk = m & 1, pp = m >> 2;
      // plane products, smallest first: 0*2, 2*0, 1*1, 0*1, 1*0, 0*0 (weights x pixels)
      const int ia = (pp == 0) ? 0 : (pp == 1) ? 2 : (pp == 2) ? 1 : (pp == 3) ? 0 : (pp == 4) ? 1 : 0;
      const int ib = (pp == 0) ? 2 : (pp == 1) ? 0 : (pp == 2) ? 1 : (pp == 3) ? 1 : (pp == 4) ? 0 : 0;
      acc = mfma32_bf16(wf[t][kk][ia], fb[t][kk][ib], acc);
      if ((m & 3) == 1 && (m >> 2) < FP_MAXU) store_item(stu[m >> 2]);  // rows of tile T + 1
      ODIN_SCHED_FENCE();
    }
    // this wave's partial tile -> scratch [T & 1][register pair][wave][lane]
    char* d = red + (T & 1) * RED + ((wave * 64 + lane) << 3);
#pragma unroll
    for (int pr = 0; pr < 8; ++pr)
      *reinterpret_cast<float2*>(d + pr * (8 * 64 * 8)) = make_float2(acc[2 * pr], acc[2 * pr + 1]);
    opixP = opix;
    auxP = auxN;
    su0 += 2 * TC;
    if (++t_cur == p.tiles_per_img) { t_cur = 0; ++b_cur; ++su0; }
    if (su0 >= NSU) su0 -= NSU;
    __syncthreads();  // partial tiles complete; every wave is past tile T's rows; tile T + 1's rows are stored
  };
#pragma unroll 1
  for (int T = T0; T < T1; T += 3) {
    run_tile(T, iuC, iuA);
    if (T + 1 < T1) run_tile(T + 1, iuA, iuB);
    if (T + 2 < T1) run_tile(T + 2, iuB, iuC);
  }
  finish((T1 - 1) & 1);

  if (EPI == 2 && p.colsum != nullptr) {
    // column sums of this workgroup's outputs: the 32 pixel lanes of each half by shuffles; every
    // (wave, half) owns its own two channels
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float v = csum[k];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      if (l31 == 0) p.colsum[(size_t)blockIdx.x * p.CO + c0 + k] = v;
    }
  }
}

template <int EPI, int OW>
int fp_launch(const FPParams& p, dim3 grid, void* stream) {
  constexpr int TC = 32 / OW;
  const size_t lds = (size_t)(4 * TC + 3) * 3 * 2 * (OW + 2) * 64 + 2 * (8 * 4 * 64 * 16);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&fconv_planes_kernel<EPI, OW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024) != hipSuccess)
      (void)hipGetLastError();
    attr_done = true;
  }
#endif
  ODIN_LAUNCH((fconv_planes_kernel<EPI, OW>), grid, dim3(512), lds, stream, p);
  return odin_check_launch("fconv_planes(bf16x3)");
}

}  // namespace

bool odin_fconv_planes_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                                  int pt, int pl, int center) {
  // (read per call: the A/B tests switch paths inside one process; a captured graph never comes here)
  if (getenv("ODIN_NOPLANES") || getenv("ODIN_SPLIT") || getenv("ODIN_NOFPLANES")) return false;
  return KH == 4 && KW == 4 && S == 2 && pt == 1 && pl == 1 && CI == 32 && (CO % 32) == 0 && !center &&
         H == 2 * OH && W == 2 * OW && (OW == 8 || OW == 16 || OW == 32) && (OH % (32 / OW)) == 0 &&
         (size_t)B * H * W * CI * 4 < (1ull << 31);
}

// epi 1: Conv2D forward (bias + ELU); epi 2: Conv2DTranspose data gradient (x ELU'(aux), column sums)
int odin_fconv_planes_launch(const float* in, const float* w, const float* bias, const float* aux,
                             float* out, float* colsum, int* rows_out, int B, int OH, int OW, int CO,
                             int epi, void* stream) {
  FPParams p;
  memset(&p, 0, sizeof(p));
  p.in = in; p.w = w; p.bias = bias; p.aux = aux; p.out = out; p.colsum = colsum;
  p.B = B; p.OH = OH; p.CO = CO;
  const int TC = 32 / OW;
  p.tiles_per_img = OH / TC;
  p.n_tiles = B * p.tiles_per_img;
  const int gy = CO / 32;
  int cap = odin_num_cus() / gy;
  if (cap < 1) cap = 1;
  if (cap > ODIN_MAX_COLSUM_BLOCKS) cap = ODIN_MAX_COLSUM_BLOCKS;
  p.tiles_per_wg = (p.n_tiles + cap - 1) / cap;
  const int gx = (p.n_tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  if (rows_out) *rows_out = gx;
  if (out == nullptr) return 0;  // dry run
  dim3 grid(gx, gy, 1);
  if (epi == 1) {
    if (OW == 32) return fp_launch<1, 32>(p, grid, stream);
    if (OW == 16) return fp_launch<1, 16>(p, grid, stream);
    return fp_launch<1, 8>(p, grid, stream);
  }
  if (OW == 32) return fp_launch<2, 32>(p, grid, stream);
  if (OW == 16) return fp_launch<2, 16>(p, grid, stream);
  return fp_launch<2, 8>(p, grid, stream);
}
