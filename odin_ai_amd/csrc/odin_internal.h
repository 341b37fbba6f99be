// odin_internal.h -- host-side helpers shared by the translation units of libodin_hip.so
#pragma once
#include <cstdio>
#include <cstring>
#include "../../include/odin_hip.h"

#define ODIN_MAX_SLAB_BLOCKS 256     // weight-gradient slabs: rows per (ci, co) block set
#define ODIN_MAX_COLSUM_BLOCKS 512   // column-sum / fused-tail slabs (two workgroups per CU)

// Environment switches.  The product library reads TWO: ODIN_EXACT_FP32 (any value: every convolution on the exact
// fp32 matrix-core kernels -- no f16 / bf16 plane kernels) and ODIN_RCCL_LIB (comm.hip).  Everything else is an A/B
// switch of the diagnostics build (`make diag`, -DODIN_DIAG) and compiles to "not set" here.
#include <cstdlib>
#ifdef ODIN_DIAG
#define ODIN_DIAG_ENV(name) getenv(name)
#else
#define ODIN_DIAG_ENV(name) ((const char*)nullptr)
#endif
// (read per call: the tests switch it inside one process; a captured graph never comes here)
static inline bool odin_exact_fp32() { return getenv("ODIN_EXACT_FP32") != nullptr; }

int odin_fail(int code, const char* msg);
int odin_check_launch(const char* what);
int odin_wgrad_planes_flush(void* stream);  // issue the calling thread's deferred plane weight gradients (wgrad_planes.hip)
int odin_num_cus();
// range word of a gradient tensor (include/odin_hip.h: odin_conv_desc.dy_amax): the caller's word, or a scratch
// word filled by one pass over the tensor; nullptr on failure
const uint32_t* odin_range_word_of(const float* t, size_t n, const uint32_t* given, void* stream);
// fold max|t| into `word` with one pass (a producer whose kernel family does not track its outputs); counted by
// odin_debug_absmax_fallbacks
int odin_absmax_fold(const float* t, size_t n, uint32_t* word, void* stream);

// first-layer (Cin <= 4) convolutions on the vector ALUs (smallc_conv.hip)
bool odin_smallc_applicable(const odin_conv_desc* d);
int odin_smallc_fwd(const float* x, const float* w, const float* bias, float* y,
                    const odin_conv_desc* d, void* stream);
int odin_smallc_wgrad(const float* x, const float* dy, float* slab, int* rows_out,
                      const odin_conv_desc* d, void* stream);

// tiny Dense layers on the vector ALUs (pointwise.hip)
bool odin_tiny_dense_ok(int B, int K, int N);
int odin_tiny_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K,
                        int N, int act, void* stream);
int odin_tiny_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          float* colsum_slab, int* slab_rows_out, int B, int K, int N, void* stream);

// 1x1 convolutions with <= 8 output maps as streaming kernels (pw1x1.hip)
bool odin_pw1x1_applicable(const odin_conv_desc* d);
int odin_pw1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                   const odin_conv_desc* d, void* stream);
int odin_pw1x1_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                     float* colsum_slab, int* slab_rows_out, const odin_conv_desc* d, void* stream);
int odin_pw1x1_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out,
                     const odin_conv_desc* d, void* stream);

// Dense layers as small matrix-core GEMMs with operands straight from L2 (dense_gemm.hip)
bool odin_dense_gemm_ok(int B, int K, int N);
int odin_dense_gemm_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K,
                        int N, int act, void* stream);
int odin_dense_gemm_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          int B, int K, int N, uint32_t* dx_amax, void* stream);
// odin_dense_dgrad with the range words of dy (read by the plane GEMM) and dx (written when odin_dense_dgrad_tracks)
int odin_dense_dgrad_ranged(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                            float* colsum_slab, int* slab_rows_out, int B, int K, int N, const uint32_t* dy_amax,
                            uint32_t* dx_amax, void* stream);
bool odin_dense_dgrad_tracks(int B, int K, int N);
int odin_zero_u32(uint32_t* p, size_t n, void* stream);  // zero n words with a kernel (runtime.hip: why not a memset)
int odin_dense_gemm_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N,
                          void* stream);

// Dense layers with one thin side (K <= 32 or N <= 4) as streaming kernels on the vector ALU (thin_dense.hip): kind 1 =
// thin K, 2 = thin N, 0 = not served; pointers must be 16-byte aligned (the dispatchers check)
int odin_thin_dense_kind(int B, int K, int N);
int odin_thin_dense_wgrad_rows(int B, int K, int N);
int odin_thin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                        uint32_t* y_amax, void* stream);
int odin_thin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx, int B, int K, int N,
                          uint32_t* dx_amax, void* stream);
int odin_thin_dense_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N, void* stream);

// Dense layers with both widths >= 256 on the f16 matrix pipe as two planes (dense_h.hip)
bool odin_dense_h_ok(int B, int K, int N);
int odin_dense_h_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                     const uint32_t* x_amax, uint32_t* y_amax, void* stream);
int odin_dense_h_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx, int B, int K, int N,
                       const uint32_t* dy_amax, uint32_t* dx_amax, void* stream);
int odin_dense_h_wgrad(const float* x, const float* dy, float* slab, int B, int K, int N, const uint32_t* dy_amax,
                       const uint32_t* x_amax, void* stream);
int odin_dense_h_bwd_pair(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          float* slab, int B, int K, int N, const uint32_t* dy_amax, uint32_t* dx_amax,
                          const uint32_t* x_amax, void* stream);

// the decoders' first Conv2DTranspose (tiny image, 8 / 16 -> 64 channels) on the vector ALUs, one workgroup per
// sample pair (smalldeconv.hip)
// bwd_planes.hip: weight + data gradient of a Conv2DTranspose(k4, s2) over 32 output channels in one launch
bool odin_bwd_planes_applicable(int B, int H, int W, int Cin, int Cout);
int odin_bwd_planes_rows(int B, int H, int W, int Cin);
int odin_bwd_planes_launch(const float* x, const float* dy, const float* w, const float* aux, float* dx, float* colsum,
                           float* wslab, int B, int H, int W, int Cin, int Cout, const uint32_t* dy_amax,
                           const uint32_t* x_amax, uint32_t* dx_amax, void* stream);
bool odin_smalldeconv_applicable(const odin_conv_desc* d);
int odin_smalldeconv_rows(const odin_conv_desc* d);
bool odin_smalldeconv_gen_applicable(const odin_conv_desc* d);   // forward only: k <= 5, stride 2, Cin in {4, 8, 12, 16}
int odin_smalldeconv_gen_fwd(const float* x, const float* w, const float* bias, float* y, const odin_conv_desc* d,
                             void* stream);
int odin_smalldeconv_fwd(const float* x, const float* w, const float* bias, float* y, const odin_conv_desc* d,
                         void* stream);
int odin_smalldeconv_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                         float* slab, int* rows_out, const odin_conv_desc* d, void* stream);

// 4x4 / stride-2 gather convolution over 32 channels with a rolling LDS row window (fconv_ring.hip)
bool odin_fconv_ring_applicable(int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                                int pt, int pl, int center);
int odin_fconv_ring_launch(const float* in, const float* w, const float* bias, const float* aux,
                           float* out, float* colsum, int* rows_out, int B, int H, int W, int CI,
                           int OH, int OW, int CO, int epi, void* stream);
void odin_fconv_ring_set_stamps(void* buf);

// transposed 4x4 / stride-2 gather over 32 channels, rolling LDS row window (tconv_ring.hip)
void odin_tconv_ring_set_stamps(void* buf);
bool odin_tconv_ring_applicable(int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl,
                                int center);
int odin_tconv_ring_launch(const float* in, const float* w, const float* bias, const float* aux,
                           float* out, float* colsum, int* rows_out, const float* w1, const float* b1,
                           const float* target, float* logits, float* llk_part, int* n_part_out,
                           float* slab, const float* scale, int C1, int B, int H, int W, int CO,
                           int epi, void* stream);

// the same transposed gathers through the bf16 matrix pipe: fp32 operands as three exact bf16 planes,
// split once on the way into LDS (tconv_planes.hip)
// block-window plane kernels for the 4x4 / stride-2 layers of any image size (blk_planes.hip)
bool odin_tconv_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center);
int odin_tconv_blk_rows(int B, int H, int W, int CO);
int odin_tconv_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int H, int W, int CI, int CO, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream);
bool odin_fconv_blk_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S, int pt, int pl,
                               int center);
int odin_fconv_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int OH, int OW, int CI, int CO, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream);
bool odin_wgrad_blk_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S, int pt, int pl,
                               int center);
int odin_wgrad_blk_launch(const float* U, const float* V, float* slab, int* rows_out, int B, int OH, int OW, int CI,
                          int CO, int want_bias, int grad_u, const uint32_t* g_amax, const uint32_t* a_amax,
                          void* stream);
// 5x5 / stride-1 layers over block windows (blk5_planes.hip)
bool odin_conv5_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center);
int odin_conv5_blk_launch(const float* in, const float* w, const float* bias, const float* aux, float* out,
                          float* colsum, int* rows_out, int B, int H, int W, int CI, int CO, int K, int epi, int act,
                          const uint32_t* in_amax, uint32_t* out_amax, void* stream);
bool odin_wgrad5_blk_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt, int pl, int center);
int odin_wgrad5_blk_launch(const float* x, const float* dy, float* slab, int* rows_out, int B, int H, int W, int CI,
                           int CO, int K, int want_bias, const uint32_t* g_amax, const uint32_t* a_amax, void* stream);
bool odin_blk_enabled(double flop);
bool odin_blk_first();                 // diagnostics: the block-window kernels precede the row-window ones   // the block-window families are on and take a launch of this many FLOP
bool odin_bwd_blk_applicable(int B, int H, int W, int Cin, int Cout);
int odin_bwd_blk_rows(int B, int H, int W, int Cin);
int odin_bwd_blk_launch(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                        float* colsum, float* wslab, int B, int H, int W, int Cin, int Cout, const uint32_t* dy_amax,
                        const uint32_t* x_amax, uint32_t* dx_amax, void* stream);
void odin_tconv_planes_set_stamps(void* buf);
bool odin_tconv_planes_applicable(int B, int H, int W, int CI, int CO, int KH, int KW, int S, int pt,
                                  int pl, int center, int epi, int C1);
int odin_tconv_planes_launch(const float* in, const float* w, const float* bias, const float* aux,
                             float* out, float* colsum, int* rows_out, const float* w1, const float* b1,
                             const float* target, float* logits, float* llk_part, int* n_part_out,
                             float* slab, const float* scale, int C1, int B, int H, int W, int CI,
                             int CO, int epi, const uint32_t* in_amax, uint32_t* out_amax, void* stream);

// weight gradients of the 4x4 / stride-2 layers with both operands as bf16 planes, transposing LDS reads
// (wgrad_planes.hip)
bool odin_wgrad_planes_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                                  int S, int pt, int pl, int center);
int odin_wgrad_planes_launch(const float* U, const float* V, float* slab, int* rows_out, int B, int OH,
                             int OW, int CI, int CO, int want_bias, int grad_u, const uint32_t* g_amax,
                             const uint32_t* a_amax, void* stream);

// the same strided gathers through the bf16 matrix pipe, reduction split over the waves (fconv_planes.hip)
bool odin_fconv_planes_applicable(int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                                  int pt, int pl, int center);
void odin_fconv_planes_set_stamps(void* buf);
int odin_fconv_planes_launch(const float* in, const float* w, const float* bias, const float* aux,
                             float* out, float* colsum, int* rows_out, int B, int OH, int OW, int CI, int CO,
                             int epi, const uint32_t* in_amax, uint32_t* out_amax, void* stream);

// small-spatial layers as implicit GEMMs with both operands straight from L2 (igemm.hip)
bool odin_igemm_applicable(int tmode, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                           int S, int center);
void odin_igemm_set_stamps(void* buf);
int odin_igemm_tiles(int tmode, int B, int OH, int OW, int S);
int odin_igemm_launch(int tmode, const float* in, const float* w, const float* bias, const float* aux,
                      int aux_act, float* out, float* colsum, int B, int H, int W, int CI, int OH, int OW,
                      int CO, int KH, int KW, int S, int pt, int pl, int act, uint32_t* out_amax, void* stream);
// between begin and end an igemm weight-gradient launch waits for the next igemm data-gradient launch on the same
// stream and shares its launch (igemm_pair_kernel); end flushes a weight gradient that found no partner
void odin_igemm_pair_begin();
int odin_igemm_pair_end();
// igemm_h.hip: the same implicit GEMMs on the f16 matrix pipe (two planes per operand), any spatial size
bool odin_igemm_h_applicable(int tmode, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW, int S,
                             int center);
int odin_igemm_h_rows(int tmode, int B, int OH, int OW, int S);
int odin_igemm_h_launch(int tmode, const float* in, const float* w, const float* bias, const float* aux, int aux_act,
                        float* out, float* colsum, int B, int H, int W, int CI, int OH, int OW, int CO, int KH, int KW,
                        int S, int pt, int pl, int act, const uint32_t* in_amax, int in_is_grad, uint32_t* out_amax,
                        void* stream);
bool odin_igemm_h_wgrad_applicable(int B, int FH, int FW, int CU, int h, int w, int CV, int KH, int KW, int S,
                                   int center);
int odin_igemm_h_wgrad_rows(int B, int h, int w, int KH, int KW, int CU, int CV);
int odin_igemm_h_wgrad_launch(const float* u, const float* v, float* slab, int slab_stride, int B, int FH, int FW,
                              int CU, int h, int w, int CV, int KH, int KW, int S, int pt, int pl, int want_bias,
                              int grad_u, const uint32_t* g_amax, const uint32_t* a_amax, void* stream);
bool odin_igemm_wgrad_applicable(int B, int FH, int FW, int CU, int h, int w, int CV, int KH, int KW, int S,
                                 int center);
int odin_igemm_wgrad_rows(int B, int h, int w, int KH, int KW, int CU, int CV);
int odin_igemm_wgrad_launch(const float* u, const float* v, float* slab, int slab_stride, int B, int FH,
                            int FW, int CU, int h, int w, int CV, int KH, int KW, int S, int pt, int pl,
                            int want_bias, void* stream);
