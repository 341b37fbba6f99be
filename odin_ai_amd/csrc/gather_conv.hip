// gather_conv.hip -- NHWC implicit-GEMM convolution on the gfx950 f32 matrix cores.
//
// One kernel, two gather modes, covers six reference operations (all TF `SAME`):
//   MODE_F (gather conv):   out[b,oh,ow,n] = sum_{kh,kw,c} in[b,oh*S-pt+kh,ow*S-pl+kw,c] * W
//        = Conv2D forward              (odin/networks/image_networks.py:166-169)
//        = Conv2DTranspose data-grad   (tape.gradient, odin/networks/base_networks.py:518)
//        = Dense forward (1x1 image)   (odin/networks/base_networks.py:1002-1014)
//   MODE_T (transposed gather): out[b,oh,ow,n] = sum over (kh,kw,c) with
//        (oh+pt-kh)%S==0: in[b,(oh+pt-kh)/S,(ow+pl-kw)/S,c] * W
//        = Conv2DTranspose forward     (odin/networks/image_networks.py:170-173)
//        = Conv2D data-grad, Dense data-grad
// Weight layouts in HBM (wmode): 0 = [kh][kw][reduce][out]  1 = [kh][kw][out][reduce],
// which are exactly Keras' Conv2D (kh,kw,Cin,Cout) and Conv2DTranspose (kh,kw,Cout,Cin)
// layouts seen from the forward (0 / 1) or the data-grad (1 / 0) side: no weight is ever
// transposed in memory.
//
// Tiling (MI355X): a workgroup owns TR full-width output rows (~128 output pixels) x 32
// output channels.  The input patch those pixels touch (with halo, zero-filled SAME
// padding, optional CenterAt0 fold-in) is staged ONCE into LDS with an odd pixel pitch,
// the weight slice [taps][CIC][32] is staged into LDS (kept resident across the
// workgroup's persistent tile loop when it fits), and every wave runs
// v_mfma_f32_32x32x2_f32 with A = weights (rows = output channels) and B = pixels
// (columns), so each lane ends up with 4 consecutive output channels of one pixel ->
// float4 NHWC stores with the bias / activation / activation-gradient epilogue fused.
// Reduction order per output is the fixed k-ordered fmaf chain of the MFMA: results are
// bit-reproducible run to run.
#include "odin_device.h"
#include "odin_internal.h"

namespace {

enum { MODE_F = 0, MODE_T = 1 };

struct GParams {
  const float* in;
  const float* w;
  const float* bias;
  const float* aux;   // epilogue: out *= act'(aux) (aux_act), same shape as out
  float* out;
  float* colsum_slab;  // optional [gridDim.x][CO] partial column sums of `out`
  int B, H, W, CI, OH, OW, CO;
  int KH, KW, S, pt, pl;
  int wmode, act, aux_act, center;
  // plan
  int TR, RPI, NIMG, n_tiles;  // rows per tile, rows per image in a tile, images per tile
  int NRI, PW, P;              // patch rows per image, patch width (pixels), pixel pitch
  int ih_off, iw_lo;           // F: ih_lo = oh0*S + ih_off ; T: ih_lo = oh0/S + ih_off
  int CIC, n_chunks, WP, w_resident;
  int patch_floats, MT, MTP, SPP;  // M-tiles total / per phase, slots per phase
};

__device__ __forceinline__ void stage_weights(const GParams& p, float* wl, int c0, int n0,
                                              int tid, int nthreads) {
  const int ntaps = p.KH * p.KW;
  const int total = ntaps * p.CIC * 32;
  if (p.wmode == 0) {
    for (int e = tid; e < total; e += nthreads) {
      int co = e & 31, t2 = e >> 5;
      int ci = t2 % p.CIC, tap = t2 / p.CIC;
      int c = c0 + ci, n = n0 + co;
      float v = 0.f;
      if (c < p.CI && n < p.CO) v = p.w[((size_t)tap * p.CI + c) * p.CO + n];
      wl[(tap * p.CIC + ci) * p.WP + co] = v;
    }
  } else {
    for (int e = tid; e < total; e += nthreads) {
      int ci = e % p.CIC, t2 = e / p.CIC;
      int co = t2 & 31, tap = t2 >> 5;
      int c = c0 + ci, n = n0 + co;
      float v = 0.f;
      if (c < p.CI && n < p.CO) v = p.w[((size_t)tap * p.CO + n) * p.CI + c];
      wl[(tap * p.CIC + ci) * p.WP + co] = v;
    }
  }
}

__device__ __forceinline__ void stage_patch(const GParams& p, float* patch, int b0, int ih_lo,
                                            int c0, int tid, int nthreads) {
  const bool vec = ((p.CI & 3) == 0) && ((p.CIC & 3) == 0);
  if (vec) {
    const int c4n = p.CIC >> 2;
    const int total = p.NIMG * p.NRI * p.PW * c4n;
    for (int e = tid; e < total; e += nthreads) {
      int c4 = e % c4n, q = e / c4n;
      int pcol = q % p.PW, q2 = q / p.PW;
      int prow = q2 % p.NRI, img = q2 / p.NRI;
      int b = b0 + img, ih = ih_lo + prow, iw = p.iw_lo + pcol, c = c0 + c4 * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b < p.B && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && c < p.CI) {
        v = *reinterpret_cast<const float4*>(p.in + (((size_t)b * p.H + ih) * p.W + iw) * p.CI + c);
        if (p.center) {
          v.x = 2.f * v.x - 1.f; v.y = 2.f * v.y - 1.f;
          v.z = 2.f * v.z - 1.f; v.w = 2.f * v.w - 1.f;
        }
      }
      float* d = patch + ((img * p.NRI + prow) * p.PW + pcol) * p.P + c4 * 4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
  } else {
    const int total = p.NIMG * p.NRI * p.PW * p.CIC;
    for (int e = tid; e < total; e += nthreads) {
      int ch = e % p.CIC, q = e / p.CIC;
      int pcol = q % p.PW, q2 = q / p.PW;
      int prow = q2 % p.NRI, img = q2 / p.NRI;
      int b = b0 + img, ih = ih_lo + prow, iw = p.iw_lo + pcol, c = c0 + ch;
      float v = 0.f;
      if (b < p.B && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && c < p.CI) {
        v = p.in[(((size_t)b * p.H + ih) * p.W + iw) * p.CI + c];
        if (p.center) v = 2.f * v - 1.f;
      }
      patch[((img * p.NRI + prow) * p.PW + pcol) * p.P + ch] = v;
    }
  }
}

// geometry of the output pixel a lane owns inside M-tile `mt`
struct Slot {
  int base;    // patch float index of tap (0,0) channel 0
  int opix;    // linear output pixel index (b*OH+oh)*OW+ow, or -1 if masked
  int kh0, kw0, njh, njw;  // MODE_T tap set
};

template <int MODE>
__device__ __forceinline__ Slot slot_geometry(const GParams& p, int mt, int l31, int gr0) {
  Slot s;
  s.kh0 = s.kw0 = 0;
  s.njh = p.KH;
  s.njw = p.KW;
  const int total_rows = p.B * p.OH;
  if (MODE == MODE_F) {
    int sl = mt * 32 + l31;
    int r = sl / p.OW, c = sl - r * p.OW;
    bool valid = (r < p.TR) && (gr0 + r < total_rows);
    int img = r / p.RPI, rl = r - img * p.RPI;
    s.base = valid ? ((img * p.NRI + rl * p.S) * p.PW + c * p.S) * p.P : 0;
    s.opix = valid ? (gr0 + r) * p.OW + c : -1;
  } else {
    const int S = p.S;
    int phase = mt / p.MTP, mtl = mt - phase * p.MTP;
    int ph = phase / S, pw = phase - ph * S;
    int sl = mtl * 32 + l31;
    const int IWs = p.OW / S, RPS = p.RPI / S;
    int img = sl / (RPS * IWs), rem = sl - img * (RPS * IWs);
    int rq = rem / IWs, cq = rem - rq * IWs;
    int row_in_tile = img * p.RPI + ph + S * rq;
    bool valid = (sl < p.SPP) && (gr0 + row_in_tile < total_rows);
    s.kh0 = (ph + p.pt) % S;
    s.kw0 = (pw + p.pl) % S;
    s.njh = (p.KH - s.kh0 + S - 1) / S;
    s.njw = (p.KW - s.kw0 + S - 1) / S;
    int dh = (ph + p.pt - s.kh0) / S, dw = (pw + p.pl - s.kw0) / S;
    // patch row of tap jh=0: rq + dh - lo_h, where lo_h == p.ih_off ; same for columns
    int prow = rq + dh - p.ih_off, pcol = cq + dw - p.iw_lo;
    // masked lanes read the last patch pixel (tap offsets are negative in this mode)
    s.base = valid ? ((img * p.NRI + prow) * p.PW + pcol) * p.P
                   : ((p.NIMG * p.NRI - 1) * p.PW + (p.PW - 1)) * p.P;
    s.opix = valid ? (gr0 + row_in_tile) * p.OW + (pw + S * cq) : -1;
  }
  return s;
}

template <int MODE>
__device__ __forceinline__ f32x16 mtile_compute(const GParams& p, const float* patch,
                                                const float* wl, const Slot& s, int l31, int h,
                                                f32x16 acc) {
  // NOTE: tap loops must be wave-uniform: in MODE_T every lane of an M-tile shares the
  // phase, hence (kh0,kw0,njh,njw); masked lanes were given njh=0 individually, so use
  // the M-tile-wide maximum via the unmasked formula instead.
  const int njh = (MODE == MODE_F) ? p.KH : (p.KH - s.kh0 + p.S - 1) / p.S;
  const int njw = (MODE == MODE_F) ? p.KW : (p.KW - s.kw0 + p.S - 1) / p.S;
  for (int jh = 0; jh < njh; ++jh) {
    for (int jw = 0; jw < njw; ++jw) {
      int tapoff, wt;
      if (MODE == MODE_F) {
        tapoff = (jh * p.PW + jw) * p.P;
        wt = jh * p.KW + jw;
      } else {
        tapoff = -(jh * p.PW + jw) * p.P;
        wt = (s.kh0 + p.S * jh) * p.KW + (s.kw0 + p.S * jw);
      }
      const float* ap = patch + s.base + tapoff + h;
      const float* wp = wl + (wt * p.CIC + h) * p.WP + l31;
      const int wstep = 2 * p.WP;
#pragma unroll 8
      for (int c = 0; c < p.CIC; c += 2) {
        acc = mfma32(wp[0], ap[c], acc);
        wp += wstep;
      }
    }
  }
  return acc;
}

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void gather_conv_kernel(GParams p) {
  ODIN_DYN_SMEM(float, smem);
  float* patch = smem;
  float* wl = smem + p.patch_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 32;
  constexpr int NT = NW * 64;

  if (p.w_resident) stage_weights(p, wl, 0, n0, tid, NT);

  float bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = 0.f;

  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int gr0 = tile * p.TR;
    const int b0 = gr0 / p.OH, oh0 = gr0 - b0 * p.OH;
    const int ih_lo = (MODE == MODE_F) ? oh0 * p.S + p.ih_off : oh0 / p.S + p.ih_off;
    f32x16 acc0 = f32x16_zero(), acc1 = f32x16_zero();
    const int mt0 = wave, mt1 = wave + NW;
    Slot s0 = slot_geometry<MODE>(p, mt0 < p.MT ? mt0 : 0, l31, gr0);
    Slot s1 = slot_geometry<MODE>(p, mt1 < p.MT ? mt1 : 0, l31, gr0);
    for (int ch = 0; ch < p.n_chunks; ++ch) {
      const int c0 = ch * p.CIC;
      __syncthreads();
      stage_patch(p, patch, b0, ih_lo, c0, tid, NT);
      if (!p.w_resident) stage_weights(p, wl, c0, n0, tid, NT);
      __syncthreads();
      if (mt0 < p.MT) acc0 = mtile_compute<MODE>(p, patch, wl, s0, l31, h, acc0);
      if (mt1 < p.MT) acc1 = mtile_compute<MODE>(p, patch, wl, s1, l31, h, acc1);
    }
    // ---- epilogue: bias + activation (+ activation-gradient multiply) + NHWC store ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int mt = mi == 0 ? mt0 : mt1;
      const Slot& s = mi == 0 ? s0 : s1;
      const f32x16& acc = mi == 0 ? acc0 : acc1;
      if (mt >= p.MT || s.opix < 0) continue;
      const size_t obase = (size_t)s.opix * p.CO;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + 8 * q + 4 * h;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float t = acc[4 * q + j];
          if (p.bias != nullptr && n + j < p.CO) t += p.bias[n + j];
          v[j] = odin_act(p.act, t);
        }
        if (p.aux != nullptr) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < p.CO) v[j] *= odin_act_grad(p.aux_act, p.aux[obase + n + j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) bsum[4 * q + j] += (n + j < p.CO) ? v[j] : 0.f;
        if (((p.CO & 3) == 0) && n + 3 < p.CO) {
          *reinterpret_cast<float4*>(p.out + obase + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (n + j < p.CO) p.out[obase + n + j] = v[j];
        }
      }
    }
  }

  if (p.colsum_slab != nullptr) {
    // per-workgroup partial column sums (bias gradient of Conv2DTranspose layers):
    // reduce the 32 pixel lanes by shuffles, the NW waves through LDS.
    __syncthreads();
    float* red = smem;  // [NW][32]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float v = bsum[i];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      bsum[i] = v;
    }
    if (l31 == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[wave * 32 + 8 * (i >> 2) + 4 * h + (i & 3)] = bsum[i];
    }
    __syncthreads();
    if (tid < 32) {
      float t = 0.f;
      for (int w2 = 0; w2 < NW; ++w2) t += red[w2 * 32 + tid];
      if (n0 + tid < p.CO) p.colsum_slab[(size_t)blockIdx.x * p.CO + n0 + tid] = t;
    }
  }
}

// --------------------------------------------------------------------------------------
// host-side planner
// --------------------------------------------------------------------------------------
constexpr int LDS_BUDGET_FLOATS = (160 * 1024 - 2048) / 4;
constexpr int NW_G = 4;

bool plan_gather(GParams& p, int mode, int max_blocks, int* grid_x, size_t* lds_bytes) {
  const int S = p.S;
  if (mode == MODE_T && (p.OW % S != 0 || p.OH % S != 0)) return false;
  const int img_pix = p.OH * p.OW;
  const int TARGET = 32 * NW_G;  // one M-tile per wave
  if (img_pix <= TARGET) {
    p.NIMG = TARGET / img_pix;
    if (p.NIMG > p.B) p.NIMG = p.B;
    if (p.NIMG < 1) p.NIMG = 1;
    p.RPI = p.OH;
    p.TR = p.NIMG * p.OH;
  } else {
    p.NIMG = 1;
    // prefer the largest tile <= TARGET, otherwise the smallest legal one
    int pick = 0;
    for (int tr = 1; tr <= p.OH; ++tr) {
      if (p.OH % tr) continue;
      if (mode == MODE_T && tr % S) continue;
      if (tr * p.OW <= TARGET) pick = tr;
    }
    if (pick == 0) {
      for (int tr = 1; tr <= p.OH && pick == 0; ++tr) {
        if (p.OH % tr) continue;
        if (mode == MODE_T && tr % S) continue;
        pick = tr;
      }
    }
    p.TR = p.RPI = pick;
  }
  const int total_rows = p.B * p.OH;
  p.n_tiles = (total_rows + p.TR - 1) / p.TR;
  if (mode == MODE_F) {
    p.NRI = (p.RPI - 1) * S + p.KH;
    p.PW = (p.OW - 1) * S + p.KW;
    p.ih_off = -p.pt;
    p.iw_lo = -p.pl;
    int slots = p.TR * p.OW;
    p.MT = (slots + 31) / 32;
    p.MTP = p.MT;
    p.SPP = slots;
  } else {
    int lo_h = odin_floordiv(p.pt - p.KH + 1, S), lo_w = odin_floordiv(p.pl - p.KW + 1, S);
    p.ih_off = lo_h;
    p.iw_lo = lo_w;
    p.NRI = odin_floordiv(p.RPI - 1 + p.pt, S) - lo_h + 1;
    p.PW = odin_floordiv(p.OW - 1 + p.pl, S) - lo_w + 1;
    p.SPP = p.NIMG * (p.RPI / S) * (p.OW / S);
    p.MTP = (p.SPP + 31) / 32;
    p.MT = p.MTP * S * S;
  }
  if (p.MT > 2 * NW_G) return false;
  const int CIp = (p.CI + 1) & ~1;
  p.WP = (p.wmode == 0) ? 32 : 33;
  const int ntaps = p.KH * p.KW;
  int cic = CIp;
  // keep float4 staging possible when CI % 4 == 0
  const int gran = ((p.CI & 3) == 0) ? 4 : 2;
  while (true) {
    int P = cic + 1;
    long pf = (long)p.NIMG * p.NRI * p.PW * P + 8;
    long wf = (long)ntaps * cic * p.WP;
    if (pf + wf <= LDS_BUDGET_FLOATS) break;
    if (cic <= gran) return false;
    // next smaller chunk: halve, rounded up to the granularity
    int nc = (CIp + cic - 1) / cic + 1;
    int ncic = ((CIp + nc - 1) / nc + gran - 1) / gran * gran;
    if (ncic >= cic) ncic = cic - gran;
    cic = ncic;
  }
  p.CIC = cic;
  p.P = cic + 1;
  p.n_chunks = (CIp + cic - 1) / cic;
  p.w_resident = (p.n_chunks == 1) ? 1 : 0;
  p.patch_floats = (int)(((long)p.NIMG * p.NRI * p.PW * p.P + 8 + 3) & ~3L);
  long wf = (long)ntaps * cic * p.WP;
  long total = p.patch_floats + wf;
  if (total < NW_G * 32) total = NW_G * 32;
  *lds_bytes = (size_t)total * 4;
  if (max_blocks < 0) {  // slab-producing launch: rows are bounded
    int cap = -max_blocks;
    *grid_x = p.n_tiles < cap ? p.n_tiles : cap;
  } else {
    int per_cu = (int)((160 * 1024) / (*lds_bytes));
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 4) per_cu = 4;
    int cap = max_blocks * per_cu;
    int gx = p.n_tiles < cap ? p.n_tiles : cap;
    *grid_x = gx < 1 ? 1 : gx;
  }
  return true;
}

int launch_gather(int mode, GParams& p, void* stream, int max_blocks, int* rows_out = nullptr) {
  int gx;
  size_t lds;
  if (!plan_gather(p, mode, max_blocks, &gx, &lds)) return odin_fail(-2, "gather_conv: no tiling plan");
  if (rows_out) *rows_out = gx;
  if (p.out == nullptr) return 0;  // dry run: planning only
  dim3 grid(gx, (p.CO + 31) / 32, 1), block(NW_G * 64);
#ifndef ODIN_SIM
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_kernel<MODE_F, NW_G>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_conv_kernel<MODE_T, NW_G>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_done = true;
  }
#endif
  if (mode == MODE_F)
    ODIN_LAUNCH((gather_conv_kernel<MODE_F, NW_G>), grid, block, lds, stream, p);
  else
    ODIN_LAUNCH((gather_conv_kernel<MODE_T, NW_G>), grid, block, lds, stream, p);
  return odin_check_launch("gather_conv");
}

void fill_common(GParams& p, const odin_conv_desc* d) {
  memset(&p, 0, sizeof(p));
  p.B = d->B;
  p.KH = d->KH;
  p.KW = d->KW;
  p.S = d->stride;
  p.pt = d->pad_t;
  p.pl = d->pad_l;
}

}  // namespace

extern "C" int odin_max_slab_rows(void) { return ODIN_MAX_SLAB_BLOCKS; }

// ---- Conv2D -------------------------------------------------------------------------
extern "C" int odin_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                               const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 0; p.act = d->act; p.center = d->center;
  return launch_gather(MODE_F, p, stream, odin_num_cus());
}

// dx[b,ih,iw,ci] = sum_{kh,kw,co} dy[b,(ih+pt-kh)/S,(iw+pl-kw)/S,co] * W[kh,kw,ci,co];
// optionally multiplied by act'(aux) (aux = this layer's input = previous layer's output)
extern "C" int odin_conv2d_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                 float* dx, float* colsum_slab, int* slab_rows_out,
                                 const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 1;
  return launch_gather(MODE_T, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}

// ---- Conv2DTranspose (desc: H,W,Cin = input; OH=H*S, OW=W*S, Cout = output; pads = the
// SAME pads of the forward conv on the OUTPUT size) ------------------------------------
extern "C" int odin_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                                 const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.H = d->H; p.W = d->W; p.CI = d->Cin; p.OH = d->OH; p.OW = d->OW; p.CO = d->Cout;
  p.wmode = 1; p.act = d->act; p.center = d->center;
  return launch_gather(MODE_T, p, stream, odin_num_cus());
}

extern "C" int odin_deconv2d_dgrad(const float* dy, const float* w, const float* aux,
                                   int aux_act, float* dx, float* colsum_slab,
                                   int* slab_rows_out, const odin_conv_desc* d, void* stream) {
  GParams p;
  fill_common(p, d);
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.H = d->OH; p.W = d->OW; p.CI = d->Cout; p.OH = d->H; p.OW = d->W; p.CO = d->Cin;
  p.wmode = 0;
  return launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}

// ---- Dense: y[B,N] = act(x[B,K] @ w[K,N] + b) ----------------------------------------
extern "C" int odin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B,
                              int K, int N, int act, void* stream) {
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.w = w; p.bias = bias; p.out = y;
  p.B = B; p.H = 1; p.W = 1; p.CI = K; p.OH = 1; p.OW = 1; p.CO = N;
  p.KH = p.KW = 1; p.S = 1; p.act = act; p.wmode = 0;
  return launch_gather(MODE_F, p, stream, odin_num_cus());
}

// dx[B,K] = (dy[B,N] @ w[K,N]^T) * act'(aux)
extern "C" int odin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                                float* dx, float* colsum_slab, int* slab_rows_out, int B, int K,
                                int N, void* stream) {
  GParams p;
  memset(&p, 0, sizeof(p));
  p.in = dy; p.w = w; p.out = dx; p.aux = aux; p.aux_act = aux_act; p.colsum_slab = colsum_slab;
  p.B = B; p.H = 1; p.W = 1; p.CI = N; p.OH = 1; p.OW = 1; p.CO = K;
  p.KH = p.KW = 1; p.S = 1; p.wmode = 1;
  return launch_gather(MODE_F, p, stream, colsum_slab ? -ODIN_MAX_SLAB_BLOCKS : odin_num_cus(),
                       slab_rows_out);
}
