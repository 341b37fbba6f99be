// Microbenchmark (round 6): HBM write rate of a 16-byte-per-lane store instruction by the SHAPE of the bytes one wave
// instruction covers.  Every variant writes the same tensor [pixels][32 channels] fp32 (128 B per pixel) exactly once,
// 512 workgroups x 512 threads, grid-stride over "tiles" of 256 pixels, rotating over three 268 MB buffers so that the
// lines written are not resident in the Infinity Cache:
//   seg 16  : a lane's 16 B at a 128-B stride (64 pixels, one chunk each)            -- worst case
//   seg 32  : 32-B segments (the row-window plane kernels' epilogue: 2 lanes per pixel chunk pair)
//   seg 64  : 64-B segments at a 256-B stride (tconv_blk: 16 pixels of one parity x 4 chunks)
//   seg 128 : 128-B segments at a 256-B stride (8 pixels of one parity, whole pixels)
//   seg 1024: 1 KB contiguous (8 adjacent pixels)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int SEG>
__global__ __launch_bounds__(512) void k(float4* out, size_t n_pix) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float4 v = make_float4(1.f, 2.f, 3.f, (float)lane);
  // a tile = 256 consecutive pixels = 32 KB; a wave writes 32 pixels of it = 4 KB in 4 instructions
  for (size_t t = blockIdx.x; t * 256 < n_pix; t += gridDim.x) {
    float4* base = out + t * 256 * 8;  // 8 float4 per pixel
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      size_t pix, chunk;
      if (SEG == 1024) { pix = wave * 32 + i * 8 + (lane >> 3); chunk = lane & 7; }
      if (SEG == 128) { pix = 2 * (wave * 16 + (i >> 1) * 8 + (lane >> 3)) + (i & 1); chunk = lane & 7; pix = pix % 256; }
      if (SEG == 64) { const int nb = i & 1, par = i >> 1; pix = 2 * (wave * 16 + (lane & 15)) + par; chunk = 4 * nb + (lane >> 4); pix = pix % 256; }
      if (SEG == 32) { const int q = i; pix = wave * 32 + (lane & 31); chunk = 2 * q + (lane >> 5); }
      if (SEG == 16) { pix = (wave * 32 + i * 8) + (lane & 7) * 0 + (lane >> 3) ; chunk = (lane & 7); pix = (pix * 37 + lane) % 256; chunk = (i * 2 + (lane & 1)) & 7; }
      base[pix * 8 + chunk] = v;
    }
  }
}

template <int SEG>
float run(float4** bufs, size_t n_pix) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k<SEG><<<512, 512>>>(bufs[i % 3], n_pix);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int N = 12;
  for (int i = 0; i < N; ++i) k<SEG><<<512, 512>>>(bufs[i % 3], n_pix);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / N * 1e3f;
}

// the same bytes as 16 x 16-pixel tiles of a [B, 96, 80, 32] image (16 rows of 2 KB, 10 KB apart): what a block-window tile writes
template <int RD, int ORDER = 0>
__global__ __launch_bounds__(512) void k2d(float4* out, const float4* in, size_t n_tiles, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t per = 6 * 5, tpw = (n_tiles + gridDim.x - 1) / gridDim.x;
  for (size_t i0 = 0; i0 < tpw; ++i0) {
    // ORDER 0: a workgroup walks consecutive tiles (one image each: every workgroup at the same place of its image at the
    // same time); 1: grid-stride; 2: consecutive, but workgroup w starts (7 w) tiles into its run
    size_t t = ORDER == 1 ? i0 * gridDim.x + blockIdx.x : blockIdx.x * tpw + (ORDER == 2 ? (i0 + 7 * blockIdx.x) % tpw : i0);
    if (t >= n_tiles) continue;
    const size_t b = t / per, r = t % per, ty = r / 5, tx = r % 5;
    if (RD) {   // the coarse window of the tile: 10 x 10 pixels of a [B, 48, 40, 32] tensor, two float4 per thread
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = threadIdx.x + 512 * j;
        const int px = e >> 3, ch4 = e & 7, wr = px / 10, wc = px % 10;
        const long gr = (long)8 * ty - 1 + wr, gc = (long)8 * tx - 1 + wc;
        if (e < 800 && gr >= 0 && gr < 48 && gc >= 0 && gc < 40) {
          const float4 v = in[((b * 48 + gr) * 40 + gc) * 8 + ch4];
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
      }
    }
    const float4 v = make_float4(1.f + acc.x, 2.f + acc.y, 3.f + acc.z, (float)lane + acc.w);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // wave: parity class (wave & 3) and 16-channel block (wave >> 2); i: pixel block (tconv_blk's roles)
      const int cls = wave & 3, nb = wave >> 2, cpw = cls & 1, rpar = cls >> 1, l15 = lane & 15, lq = lane >> 4;
      const int ri = 2 * i + (l15 >> 3), cj = l15 & 7;
      const size_t oh = 16 * ty + 2 * ri + rpar, ow = 16 * tx + 2 * cj + cpw;
      out[((b * 96 + oh) * 80 + ow) * 8 + 4 * nb + lq] = v;
    }
  }
  if (acc.x == 12345.f) sink[0] = acc.y;
}
template <int RD, int ORDER>
float run2d(float4** bufs, float4* in, size_t n_tiles, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) k2d<RD, ORDER><<<256, 512>>>(bufs[i % 3], in, n_tiles, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int N = 12;
  for (int i = 0; i < N; ++i) k2d<RD, ORDER><<<256, 512>>>(bufs[i % 3], in, n_tiles, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / N * 1e3f;
}

// stores only: 256-pixel tiles of TR rows x TC pixels of a [B, 96, 80 or 128, 32] image, 256 workgroups, consecutive tiles
template <int TR, int TC, int IW>
__global__ __launch_bounds__(512) void kshape(float4* out, size_t n_img) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  constexpr int NTY = 96 / TR, NTX = (IW + TC - 1) / TC;
  const size_t n_tiles = n_img * NTY * NTX, tpw = (n_tiles + gridDim.x - 1) / gridDim.x;
  for (size_t t = blockIdx.x * tpw; t < (blockIdx.x + 1) * tpw && t < n_tiles; ++t) {
    const size_t b = t / (NTY * NTX), r = t % (NTY * NTX), ty = r / NTX, tx = r % NTX;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + 512 * i;           // 2048 float4 of the tile
      const int px = e >> 3, ch = e & 7, rr = px / TC, cc = px % TC;
      const size_t ow = tx * TC + cc;
      if (ow < IW) out[((b * 96 + ty * TR + rr) * IW + ow) * 8 + ch] = v;
    }
  }
}
template <int TR, int TC, int IW>
float runshape(float4** bufs, size_t n_img) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kshape<TR, TC, IW><<<256, 512>>>(bufs[i % 3], n_img);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int N = 12;
  for (int i = 0; i < N; ++i) kshape<TR, TC, IW><<<256, 512>>>(bufs[i % 3], n_img);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / N * 1e3f;
}

// the 64 x 64 decoders' last layer: [256, 64, 64, 32] fp32, a tile = 4 output rows = 32 KB contiguous, 8192 tiles.
// ORDER 0: workgroup w walks tiles 32 w .. 32 w + 31 (its own image: the row-window kernels' assignment); 1: grid-stride;
// 2: workgroup w walks tiles of 8 different images (4 each)
template <int ORDER, int RD>
__global__ __launch_bounds__(512) void kimg(float4* out, const float4* in, float* sink) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < 32; ++i) {
    size_t t;
    if (ORDER == 0) t = (size_t)blockIdx.x * 32 + i;
    else if (ORDER == 1) t = (size_t)i * 256 + blockIdx.x;
    else t = ((size_t)((blockIdx.x & 31) * 8 + (i >> 2))) * 32 + (blockIdx.x >> 5) * 4 + (i & 3);
    if (RD) {   // the tile's gradient input of the same size (dec4 backward reads dy like this)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float4 q = in[t * 2048 + threadIdx.x + 512 * j]; acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w; }
    }
    if (RD != 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) out[t * 2048 + threadIdx.x + 512 * j] = v;
    }
  }
  if (acc.x == 12345.f) sink[0] = acc.y;
}
template <int ORDER, int RD>
float runimg(float4** bufs, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) kimg<ORDER, RD><<<256, 512>>>(bufs[i % 3], bufs[(i + 1) % 3], sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int N = 12;
  for (int i = 0; i < N; ++i) kimg<ORDER, RD><<<256, 512>>>(bufs[i % 3], bufs[(i + 1) % 3], sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / N * 1e3f;
}

int main() {
  const size_t n_pix = (size_t)256 * 96 * 80;   // the audio decoder4 output: 251 MB
  float4* bufs[3];
  for (int i = 0; i < 3; ++i) hipMalloc(&bufs[i], (size_t)8192 * 32768 + (1 << 20));   // (the largest tensor of the variants below: 268 MB)
  const double mb = n_pix * 128 / 1e6;
  float t;
  t = run<1024>(bufs, n_pix); printf("seg 1024 B: %7.1f us  %5.2f TB/s\n", t, mb / t * 1e-3 * 1e3 / 1e3);
  t = run<128>(bufs, n_pix);  printf("seg  128 B: %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run<64>(bufs, n_pix);   printf("seg   64 B: %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run<32>(bufs, n_pix);   printf("seg   32 B: %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run<1024>(bufs, n_pix); printf("seg 1024 B: %7.1f us  %5.2f TB/s (again)\n", t, mb / t);
  t = runshape<16, 16, 80>(bufs, 256); printf("tile 16 rows x 16 px (2 KB runs, 10 KB apart):   %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = runshape<8, 32, 80>(bufs, 256);  printf("tile  8 rows x 32 px (4 KB runs; ragged 80 = 32+32+16): %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = runshape<4, 64, 80>(bufs, 256);  printf("tile  4 rows x 64 px (8 KB runs; ragged 80 = 64+16):    %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = runshape<16, 16, 64>(bufs, 300); printf("64-px rows: tile 16 x 16 (2 KB runs, 8 KB apart): %7.1f us  %5.2f TB/s\n", t, 300.0 * 96 * 64 * 128 / 1e6 / t);
  t = runshape<4, 64, 64>(bufs, 300);  printf("64-px rows: tile 4 x 64 (32 KB contiguous):       %7.1f us  %5.2f TB/s\n", t, 300.0 * 96 * 64 * 128 / 1e6 / t);
  float* sink0; hipMalloc(&sink0, 16);
  const double mbi = 8192.0 * 32768 / 1e6;
  t = runimg<0, 0>(bufs, sink0); printf("[256,64,64,32] stores, one image per workgroup:        %7.1f us  %5.2f TB/s\n", t, mbi / t);
  t = runimg<1, 0>(bufs, sink0); printf("[256,64,64,32] stores, grid-stride tiles:              %7.1f us  %5.2f TB/s\n", t, mbi / t);
  t = runimg<2, 0>(bufs, sink0); printf("[256,64,64,32] stores, 8 images x 4 tiles per workgroup: %7.1f us  %5.2f TB/s\n", t, mbi / t);
  t = runimg<0, 2>(bufs, sink0); printf("[256,64,64,32] loads only, one image per workgroup:    %7.1f us  %5.2f TB/s\n", t, mbi / t);
  t = runimg<1, 2>(bufs, sink0); printf("[256,64,64,32] loads only, grid-stride tiles:          %7.1f us  %5.2f TB/s\n", t, mbi / t);
  t = runimg<0, 1>(bufs, sink0); printf("[256,64,64,32] load + store, one image per workgroup:  %7.1f us  %5.2f TB/s\n", t, 2 * mbi / t);
  t = runimg<1, 1>(bufs, sink0); printf("[256,64,64,32] load + store, grid-stride tiles:        %7.1f us  %5.2f TB/s\n", t, 2 * mbi / t);
  float4* in; float* sink;
  hipMalloc(&in, (size_t)256 * 48 * 40 * 128); hipMemset(in, 0, (size_t)256 * 48 * 40 * 128); hipMalloc(&sink, 16);
  const size_t n_tiles = (size_t)256 * 30;
  t = run2d<0, 0>(bufs, in, n_tiles, sink); printf("2-D tiles, 256 WGs x 30 consecutive tiles (one image each), stores only: %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run2d<0, 1>(bufs, in, n_tiles, sink); printf("2-D tiles, grid-stride tile order, stores only:                          %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run2d<0, 2>(bufs, in, n_tiles, sink); printf("2-D tiles, consecutive with a per-workgroup start skew, stores only:     %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = run2d<1, 0>(bufs, in, n_tiles, sink); printf("consecutive + the 10 x 10 window loads (63 MB):                          %7.1f us  %5.2f TB/s\n", t, (mb + 62.9) / t);
  t = run2d<1, 1>(bufs, in, n_tiles, sink); printf("grid-stride + window loads:                                              %7.1f us  %5.2f TB/s\n", t, (mb + 62.9) / t);
  t = run2d<1, 2>(bufs, in, n_tiles, sink); printf("skewed + window loads:                                                   %7.1f us  %5.2f TB/s\n", t, (mb + 62.9) / t);
  return 0;
}
