/* odin_hip.h -- C ABI of libodin_hip.so: the MI355X (gfx950) VAE training-step path.
 *
 * The reference (trungnt13/odin-ai) is pure Python: it has no FFI of its own, its device
 * arithmetic is executed by TensorFlow 2.5 / TFP 0.13 ops called from the Python files
 * cited below.  Each entry point here replaces one such call site; a maintainer binds
 * them with ctypes.  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 (int32 where noted), borrowed for the
 *     duration of the call; nothing is allocated, freed or retained by the library;
 *   - activations are NHWC, Conv2D kernels (kh,kw,Cin,Cout), Conv2DTranspose kernels
 *     (kh,kw,Cout,Cin), Dense kernels (in,out) -- the Keras layouts, never transposed;
 *   - `stream` is a hipStream_t passed as void*; all work is asynchronous on it;
 *   - return value 0 = OK, negative = error (odin_last_error() has the text).
 * File:line citations are relative to the reference repository root.
 */
#ifndef ODIN_HIP_H
#define ODIN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ODIN_LINEAR = 0, ODIN_ELU = 1, ODIN_RELU = 2 };

/* Geometry of one Conv2D / Conv2DTranspose layer (TF `SAME` padding resolved by the
 * caller: pad_t/pad_l are the SAME "before" pads of the strided conv; for a transposed
 * conv they are the pads of the forward conv on the OUTPUT size). */
typedef struct odin_conv_desc {
  int B, H, W, Cin;   /* layer input  [B,H,W,Cin]   */
  int OH, OW, Cout;   /* layer output [B,OH,OW,Cout] */
  int KH, KW, stride;
  int pad_t, pad_l;
  int act;            /* fused epilogue activation (forward only) */
  int center;         /* fold CenterAt0 (2x-1) into the input load (image_networks.py:121-126) */
  /* Backward pass only, both optional (NULL): the dynamic-range side channel of the 4x4/stride-2 layers whose
   * fp32 operands travel through the f16 matrix pipe as two planes.  Each range word (a block of
   * ODIN_RANGE_WORDS uint32: 32 sub-words on separate memory lines, the bound is their maximum) holds the fp32
   * BIT PATTERN of an upper bound of max|t| of a gradient tensor t; the caller zeroes the words once per step
   * (odin_range_reset), producers fold their outputs in with one atomicMax per workgroup, consumers scale t by an
   * exact power of two on its way into the planes.  dy_amax: word of dy (this layer's pre-activation gradient) -- read by the data / weight gradient,
   * written by the fused tail that produces dy.  dx_amax: word of dx -- CONTRACT (round 5): a data gradient (and the
   * fused tail for dy_amax) that is handed a word leaves a valid bound in it whatever kernel family ran: folded in from
   * the kernel's epilogue where the family tracks its outputs (odin_*_dgrad_keeps_range = 1: free), by one extra pass
   * over the tensor otherwise.  A consumer without a word computes the bound itself (odin_absmax: one extra pass over
   * the tensor), and only if it is a plane kernel. */
  uint32_t* dy_amax;
  uint32_t* dx_amax;
  /* Forward pass only, both optional (NULL), round 5: the same side channel for ACTIVATIONS.  x_amax: range word of the
   * layer input x (read: a plane kernel whose input bound lies outside [2^-8, 2^15) carries x times the exact power of
   * two that brings it to [2^14, 2^15) -- f16 planes hold |x| <= 65504 and lose relative precision below 2^-25; inside
   * the window, and without a word, x is carried unscaled as in round 4: then the caller guarantees |x| <= 65504).
   * y_amax: range word of the layer output y (written, same contract as dx_amax: valid on return whatever family ran).
   * Weights are always carried unscaled: |w| <= 65504. */
  const uint32_t* x_amax;
  uint32_t* y_amax;
} odin_conv_desc;

/* ---- runtime ------------------------------------------------------------------------ */
int odin_version(void);
/* diagnostics: kernel family launched last by the calling thread ("...(f16x2)": fp32 operands through
 * the f16 matrix pipe as two planes); used by bench.py to price kernels against the right peak */
const char* odin_debug_last_path(void);
/* CRC-32C (Castagnoli) of host bytes, continuing from `crc` (0 to start): the checksum of the
 * TensorFlow checkpoint / event-file formats the reference saves (base_networks.py:373-390,
 * training/trainer.py:52-71).  Returns the checksum (not an error code). */
uint32_t odin_crc32c(uint32_t crc, const void* data, size_t n);
const char* odin_last_error(void);
int odin_max_slab_rows(void);      /* upper bound of the rows any slab-producing call writes */
/* Which producers keep a range word WITHOUT an extra pass: 1 when the data gradient of layer `d` (dispatched for
 * aux_act, aux present, no oversized column-sum slab) / the fused tail folds max|dx| / max|g_out| into d->dx_amax /
 * d->dy_amax from its own epilogue.  Informational since round 5 (cost model, tests): the word is valid either way. */
int odin_conv2d_dgrad_keeps_range(const odin_conv_desc* d, int aux_act);
int odin_deconv2d_dgrad_keeps_range(const odin_conv_desc* d, int aux_act);
int odin_bernoulli_tail_keeps_range(int is_deconv, const odin_conv_desc* d, int C1);
/* Range words (odin_conv_desc.dy_amax / dx_amax; ODIN_RANGE_WORDS uint32 each): zero `n` words at the top of a
 * step; fold max|t[0..n)| of an fp32 tensor into a word (producers that do not track their outputs themselves, or
 * external callers). */
#define ODIN_RANGE_WORDS 2048
int odin_range_reset(uint32_t* words, int n, void* stream);
int odin_absmax(const float* t, size_t n, uint32_t* word, void* stream);
/* diagnostics: a stream with the traffic of the fused Bernoulli ELBO kernel (out = a + b over n floats; 12 bytes per
 * element) in several launch shapes -- the ceiling bench.py prices that kernel against beside the 8 TB/s peak */
int odin_debug_stream_probe(const float* a, const float* b, float* out, size_t n, int variant, int blocks,
                            void* stream);
/* diagnostics: how many times a consumer had to bound a gradient tensor itself (no range word given) */
int odin_debug_absmax_fallbacks(void);

/* ---- Conv2D (keras.layers.Conv2D, odin/networks/image_networks.py:166-169,463-466) --- */
int odin_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                    const odin_conv_desc* d, void* stream);
/* dx = conv2d_backprop_input(dy) * act'(aux);  replaces tape.gradient
 * (odin/networks/base_networks.py:514-518).  aux = the layer's INPUT activation (post
 * activation of the previous layer) or NULL.  colsum_slab (optional,
 * [*slab_rows_out][Cin]) receives per-workgroup column sums of dx (the bias gradient
 * of the previous layer when that layer is a Conv2DTranspose). */
int odin_conv2d_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                      float* colsum_slab, int* slab_rows_out, const odin_conv_desc* d,
                      void* stream);
/* Slab-producing calls: passing the output pointer (dx / slab) as NULL is a DRY RUN that only
 * reports *slab_rows_out (used to size workspaces).
 * slab[g][kh,kw,Cin,Cout | Cout] partial (dW | db) per workgroup g < *slab_rows_out;
 * finish with odin_slab_reduce.  x = layer input, dy = grad wrt pre-activation output. */
int odin_conv2d_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out,
                      const odin_conv_desc* d, void* stream);

/* ---- Conv2DTranspose (odin/networks/image_networks.py:170-173,497-505) ---------------- */
int odin_deconv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                      const odin_conv_desc* d, void* stream);
int odin_deconv2d_dgrad(const float* dy, const float* w, const float* aux, int aux_act,
                        float* dx, float* colsum_slab, int* slab_rows_out,
                        const odin_conv_desc* d, void* stream);
/* slab[g][kh,kw,Cout,Cin] partial dW (bias grad comes from the producer's colsum slab) */
int odin_deconv2d_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out,
                        const odin_conv_desc* d, void* stream);

/* ---- a layer's whole backward pass in ONE call: weight gradient (wslab as in *_wgrad) + data gradient (dx,
 * colsum_slab as in *_dgrad; same argument meaning, same results bit for bit).  Where both halves run on the
 * small-layer implicit-GEMM kernels they share one launch -- each alone is a latency-bound launch of a few hundred
 * workgroups -- otherwise the call is the two calls above.  Replaces the tape.gradient of one layer
 * (odin/networks/base_networks.py:514-518).
 * Round 5: a Conv2DTranspose(k4, s2) over 32 or 64 output channels whose rows are 8 / 16 / 32 pixels wide back-propagates
 * in ONE launch that fetches, scales and splits dy once for both gradients (bwd_planes.hip; needs aux_act = ELU).  With
 * 32 output channels the results are still bit-identical to the two calls; with 64 the weight-gradient slab is
 * partitioned like the data gradient's tiles (equal to rounding) and has MORE rows than odin_deconv2d_wgrad reports: a
 * dry run of odin_deconv2d_bwd (every pointer NULL, aux_act as it will be passed) reports the rows of the one-call
 * form -- size the slab for the larger of the two. */
int odin_conv2d_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                    float* colsum_slab, int* colsum_rows_out, float* wslab, int* wslab_rows_out,
                    const odin_conv_desc* d, void* stream);
int odin_deconv2d_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                      float* colsum_slab, int* colsum_rows_out, float* wslab, int* wslab_rows_out,
                      const odin_conv_desc* d, void* stream);
/* Dense: either half may be left out (want_wgrad / want_dgrad); dy_amax / dx_amax are the optional range words of dy
 * (read) and dx (written: from the epilogue when odin_dense_dgrad_keeps_range(B, K, N) = 1, by one extra pass
 * otherwise) -- layers with both widths >= 256 run on the
 * f16 matrix pipe as two planes (dense_h.hip) like the 4x4/s2 convolutions. */
int odin_dense_bwd(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                   float* colsum_slab, int* colsum_rows_out, float* wslab, int* wslab_rows_out, int B, int K,
                   int N, int want_wgrad, int want_dgrad, const uint32_t* dy_amax, uint32_t* dx_amax, void* stream);
/* The Dense entry points with the ACTIVATION range words of round 5 (odin_conv_desc.x_amax / y_amax: same contract):
 * x_amax (optional) is read by the two-plane GEMM (forward, and the weight gradient's x operand), y_amax (optional) is
 * valid on return whatever family ran. */
/* 1: a launch of this layer (forward or weight gradient, as dispatched now) reads the range word of the layer input:
 * the caller asks the layer below to keep that word only then (a wrong 0 is harmless: no word = unscaled = round 4). */
int odin_conv2d_reads_x_range(const odin_conv_desc* d);
int odin_deconv2d_reads_x_range(const odin_conv_desc* d);
int odin_dense_reads_x_range(int B, int K, int N);
int odin_dense_fwd_ranged(const float* x, const float* w, const float* bias, float* y, int B, int K, int N, int act,
                          const uint32_t* x_amax, uint32_t* y_amax, void* stream);
int odin_dense_bwd_ranged(const float* x, const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                          float* colsum_slab, int* colsum_rows_out, float* wslab, int* wslab_rows_out, int B, int K,
                          int N, int want_wgrad, int want_dgrad, const uint32_t* dy_amax, uint32_t* dx_amax,
                          const uint32_t* x_amax, void* stream);
int odin_dense_dgrad_keeps_range(int B, int K, int N);

/* ---- fused decoder tail of the TRAINING step: layer (Conv2DTranspose if is_deconv else
 * Conv2D, activation d->act, Cout<=32) -> Conv2D 1x1 linear with C1<=4 maps (w1 [Cout,C1],
 * b1 [C1]) -> Independent(Bernoulli(logits),3).log_prob(target), forward AND backward:
 *   logits (optional out), g_out = dL/d(pre-activation of the layer) for
 *   L = -scale[0]*sum llk, llk_part[b][part] (n_part per sample), and
 *   tail_slab[g][Cout*C1 (dW1) | C1 (db1) | Cout (db of the layer)], g < *slab_rows_out.
 * The [B,OH,OW,Cout] activation never reaches HBM.  Replaces decoder4->decoder6->Bernoulli
 * (odin/networks/image_networks.py:505-511,87-93) + px.log_prob(x)
 * (variational_autoencoder.py:528-530) + their tape.gradient.  g_out==NULL = dry run. */
int odin_bernoulli_tail_fwd_bwd(int is_deconv, const float* x, const float* w, const float* bias,
                                const float* w1, const float* b1, const float* target,
                                float* logits, float* g_out, float* llk_part, int* n_part_out,
                                float* tail_slab, int* slab_rows_out, const float* scale,
                                const odin_conv_desc* d, int C1, void* stream);

/* ---- Dense (keras Dense: base_networks.py:1002-1014; DistributionDense projection:
 * odin/bay/layers/dense_distribution.py:229-238) ------------------------------------- */
int odin_dense_fwd(const float* x, const float* w, const float* bias, float* y, int B, int K,
                   int N, int act, void* stream);
int odin_dense_dgrad(const float* dy, const float* w, const float* aux, int aux_act, float* dx,
                     float* colsum_slab, int* slab_rows_out, int B, int K, int N, void* stream);
/* slab[g][K*N | N] partial (dW | db) */
int odin_dense_wgrad(const float* x, const float* dy, float* slab, int* slab_rows_out, int B,
                     int K, int N, void* stream);

/* dst[i] = sum_{g<rows} src[g*stride + i], i < n; any number of jobs per call. */
typedef struct odin_reduce_job {
  const float* src;
  float* dst;
  int n;      /* elements reduced per row */
  int rows;   /* G */
  int stride; /* floats between consecutive slab rows (0 = n) */
  int pad_;
} odin_reduce_job;
int odin_slab_reduce(const odin_reduce_job* jobs, int n_jobs, void* stream);

/* ---- latent posterior q(z|x) = MVNDiag(loc, softplus(raw))
 * MultivariateNormalLayer.new (odin/bay/layers/continuous.py:459-483), sample =
 * loc + scale*eps, KL: kl_divergence (odin/bay/helpers.py:177-282): analytic=0 -> MC
 * log q(z) - log p(z) at the same z; analytic=1 -> closed form KL(q||p); analytic=2 -> closed
 * form KL(p||q) (`reverse=False`, helpers.py:261-262); free_bits<0 = disabled,
 * else max(kl, free_bits*D).   p [B,2D], eps [B,D] -> z [B,D], kl [B] (after free bits),
 * fbmask [B] (1 where the gradient flows).  capacity (optional DEVICE scalar C(step)): BetaCapacityVAE
 * (odin/bay/vi/autoencoder/beta_vae.py:132-177) -- kl <- |kl - C|, fbmask <- fbmask * sign(kl - C). */
int odin_latent_fwd(const float* p, const float* eps, float* z, float* kl, float* fbmask, int B,
                    int D, int analytic, float free_bits, const float* capacity, void* stream);
/* dp [B,2D] of  L = sum_b klw[0]*kl_b  given dz (+ dz_extra) = dL/dz from the decoder
 * / regularisers (may be NULL) and optional extra grads (dloc_x, dscale_x: total correlation).  klw is a DEVICE scalar
 * (= beta / B) so that graph replays see schedule updates. */
int odin_latent_bwd(const float* p, const float* eps, const float* z, const float* dz,
                    const float* dz_extra, const float* fbmask, const float* klw,
                    const float* dloc_x, const float* dscale_x, float* dp, int B, int D,
                    int analytic, void* stream);

/* ---- the bottleneck as one launch per direction (latent_block.hip): DistributionDense(P -> 2D) ->
 * MVNDiag(loc, softplus(raw)) -> reparameterised sample + KL term -> the decoder's first Dense(D -> N0)
 * (dense_distribution.py DistributionDense; variational_autoencoder.py:515-542; helpers.py:236-286;
 * image_networks.py:494-497), i.e. odin_rng_normal + odin_dense_fwd + odin_latent_fwd + odin_dense_fwd,
 * and in the backward pass odin_dense_dgrad/wgrad + odin_latent_bwd + odin_dense_wgrad/dgrad.
 * odin_latent_block_rows: workgroups (= slab rows the backward launch writes), 0 when the shapes are
 * outside the fused regime (both weight matrices must fit in LDS).
 * forward: h [B,P] encoder output, wl [P,2D], bl [2D]; eps_in [B,D] or NULL = draw the noise from the
 * Philox stream of odin_rng_normal(seed, step_dev) and store it in eps_out; writes p [B,2D], z [B,D],
 * kl [B], fbmask [B] exactly like odin_latent_fwd and y0 = act0(z w0 + b0) [B,N0].
 * backward: g0 [B,N0] = dL/d(pre-activation of that Dense); klw / dz_extra / dloc_x / dscale_x as in
 * odin_latent_bwd; writes dz [B,D], dp [B,2D], dh = (dp wl^T) * act'(h) [B,P] and one partial row per
 * workgroup of slab0 [rows][D*N0 + N0] = (dW0 | db0) and slabl [rows][P*2D + 2D] = (dWl | dbl);
 * dh_amax (optional): the range word of dh (ODIN_RANGE_WORDS, below), max |dh| folded in. */
int odin_latent_block_rows(int B, int P, int D, int N0);
int odin_latent_block_fwd(const float* h, const float* wl, const float* bl, const float* eps_in,
                          float* eps_out, uint64_t seed, const int32_t* step_dev, float* p, float* z,
                          float* kl, float* fbmask, const float* w0, const float* b0, float* y0, int B,
                          int P, int D, int N0, int act0, int analytic, float free_bits,
                          const float* capacity, void* stream);
int odin_latent_block_bwd(const float* g0, const float* w0, const float* z, const float* p,
                          const float* eps, const float* fbmask, const float* klw, const float* dz_extra,
                          const float* dloc_x, const float* dscale_x, const float* wl, const float* h,
                          int h_act, float* dz, float* dp, float* dh, float* slab0, float* slabl, int B,
                          int P, int D, int N0, int analytic, uint32_t* dh_amax, void* stream);

/* ---- the NECK of the 64x64 stacks in one launch per direction (round 6; neck.hip): the encoder's last convolution
 * and projection, the latent block above, the decoder's projection and first Conv2DTranspose
 *   image_networks.py:466-471  Conv2D(64, 4, 2, 'same', act3) on [8,8,64] -> [4,4,64]; Flatten; Dense(P, act4)
 *   dense_distribution.py:339-380 / continuous.py:443-483 / helpers.py:236-286  (as odin_latent_block_*)
 *   image_networks.py:494-502  Dense(D -> 16 C0, act0); Reshape(4,4,C0); Conv2DTranspose(64, 4, 2, 'same', act1)
 * i.e. odin_conv2d_fwd + odin_dense_fwd + odin_latent_block_fwd + odin_deconv2d_fwd of those layers (and in the
 * backward pass their DATA gradients plus the weight gradients of the three small matrices; the two large weight
 * gradients -- conv3: a reduction over all B*16 pixels, projection: over the batch -- stay odin_conv2d_wgrad /
 * odin_dense_wgrad on dy3 / dh4, which the backward launch leaves with their range words).  No layer of the chain
 * couples samples: a workgroup owns two samples.  Same results as the separate calls to fp32 rounding (the conv runs
 * as two f16 planes per operand like the plane kernels: <= 3 * 2^-22 per product).
 * odin_neck_rows: workgroups (= rows of slab1 / slab0 / slabl), 0 when the shapes are outside the fused regime
 * (P in {128, 256}, C0 in {8, 16}, D <= 32). */
typedef struct odin_neck_args {
  int B, P, D, C0;                 /* batch, projection width, latent dims, channels of the decoder's first image */
  int act2, act3, act4, act0, act1;/* activations: of the layer BELOW conv3 (backward: dx *= act2'(x)), conv3, projection, decoder projection, deconv1 */
  int analytic;                    /* KL form as odin_latent_fwd: 0 Monte Carlo, 1 KL(q||p), 2 KL(p||q) */
  float free_bits;                 /* < 0: off */
  uint64_t seed;                   /* Philox key of odin_rng_normal (eps_in == NULL) */
  const int32_t* step_dev;         /* device int: the RNG step */
  /* ---- forward ---- */
  const float* x;                  /* [B,8,8,64] input of conv3 (output of the layer below) */
  const uint32_t* x_amax;          /* its activation range word or NULL */
  const float *w3, *b3;            /* Conv2D (4,4,64,64), [64] */
  float* y3;                       /* [B,4,4,64] */
  const float *w4, *b4;            /* Dense (1024,P), [P] */
  float* y4;                       /* [B,P] */
  const float *wl, *bl;            /* DistributionDense (P,2D), [2D] */
  const float* eps_in;             /* [B,D] or NULL: draw */
  float* eps;                      /* [B,D] written when drawn */
  float *p, *z, *kl, *fbmask;      /* [B,2D], [B,D], [B], [B] */
  const float* capacity;           /* BetaCapacityVAE device scalar or NULL */
  const float *w0, *b0;            /* Dense (D,16*C0), [16*C0] */
  float* y0;                       /* [B,16*C0] = [B,4,4,C0] */
  const float *w1, *b1;            /* Conv2DTranspose (4,4,64,C0), [64] */
  float* y1;                       /* [B,8,8,64]; NULL: the launch stops behind the latent block (p, z, kl, fbmask: the encoder's half) */
  uint32_t* y1_amax;               /* activation range word of y1 (written) or NULL */
  /* ---- backward (forward tensors above are read) ---- */
  const float* dy1;                /* [B,8,8,64] dL/d(pre-activation of deconv1) */
  const float* klw;                /* device scalar: KL weight beta / B */
  const float *dz_extra, *dloc_x, *dscale_x;   /* optional extra terms as odin_latent_bwd */
  float *dz, *dp;                  /* [B,D], [B,2D] */
  float* dh4;                      /* [B,P]  dL/d(pre-activation of the projection) */
  float* dy3;                      /* [B,4,4,64] dL/d(pre-activation of conv3) */
  float* dx;                       /* [B,8,8,64] dL/d(pre-activation of the layer below) = conv3's data gradient * act2'(x) */
  uint32_t *dh4_amax, *dy3_amax, *dx_amax;     /* gradient range words (written) or NULL */
  float* slab1;                    /* [rows][16*64*C0]       partial dW1 (deconv1's bias gradient: column sums of dy1, the caller's) */
  float* slab0;                    /* [rows][D*16*C0 + 16*C0] partial (dW0 | db0) */
  float* slabl;                    /* [rows][P*2D + 2D]       partial (dWl | dbl) */
} odin_neck_args;
/* Weight-gradient calls (odin_conv2d_wgrad / odin_deconv2d_wgrad / odin_dense_wgrad / odin_dense_bwd with want_dx = 0)
 * issued between _begin and _end on ONE stream are declared independent of one another by the caller: where two of
 * them land on the small-layer implicit-GEMM kernels they share a launch (results bit-identical to the separate
 * launches).  _end issues a call that found no partner.  Used behind odin_neck_bwd for conv3's and the projection's
 * weight gradients. */
void odin_wgrad_pair_begin(void);
int odin_wgrad_pair_end(void);
int odin_neck_rows(int B, int P, int D, int C0);
int odin_neck_fwd(const odin_neck_args* a, void* stream);
int odin_neck_bwd(const odin_neck_args* a, void* stream);

/* ---- the discriminator's head with its loss in ONE launch (round 6; thin_dense.hip): Dense(K -> 1) on h [B, K]
 * (factor_discriminator.py:60-95: the last layer of FactorDiscriminator), the mean it feeds, and the layer's backward:
 *   logit[b] = h[b, :] . w + bias
 *   mode 0 (total_correlation, factor_discriminator.py:169-198): out[0] = mean_b logit[b]; dlogit[b] = dlogit_in[b]
 *   mode 1 (dtc_loss, :200-235; rows [0, B/2) = D(z), [B/2, B) = D(permute_dims(z'))):
 *          out[0] = 0.5 (mean softplus(-logit_z) + mean softplus(logit_perm)); dlogit its gradient (-> dlogit_out, optional)
 *   dh[b, k] = dlogit[b] w[k] act'(h[b, k]) with aux_act the activation that produced h (dh == NULL: forward + loss only);
 *   max |dh| is folded into dh_amax (optional); wslab (optional, with dh): odin_disc_head_rows(B, K) rows of [K + 1]:
 *   partial (dW | db), to be summed by odin_slab_reduce.
 * workspace: 16 bytes, 8-byte aligned, zero before the FIRST launch (every launch leaves it zero): the workgroups' loss
 * sums meet there as 64-bit fixed-point numbers (integer addition: independent of the order of arrival, bit reproducible).
 * Replaces odin_dense_fwd + odin_mean / odin_dtc_loss_fwd_bwd + odin_dense_bwd of that layer: same values to fp32 rounding. */
int odin_disc_head_rows(int B, int K);
int odin_disc_head_fwd_bwd(const float* h, const float* w, const float* bias, float* logit, int mode,
                           const float* dlogit_in, float* dlogit_out, float* out, int aux_act, float* dh,
                           uint32_t* dh_amax, float* wslab, int* rows_out, void* workspace, int B, int K, void* stream);

/* ---- observation log-likelihood fused forward+backward
 * Independent(Bernoulli(logits),3).log_prob(x) (image_networks.py:87-93;
 * variational_autoencoder.py:528-530): llk_part[b][part] partial sums (n_part per
 * sample), dlogits = -(x - sigmoid(l)) * scale[0]  (scale = DEVICE scalar 1/B).
 * A NULL logits pointer is a dry run that only reports n_part. */
int odin_elbo_bernoulli_fwd_bwd(const float* logits, const float* x, float* llk_part,
                                float* dlogits, const float* scale, int B, int n_per_sample,
                                int* n_part_out, void* stream);
/* Independent(Normal(loc, scale)) with params = split(h,2,axis=-1)
 * (image_networks.py:95-102): softplus1 = 1 -> scale = softplus(raw + softplus^-1(1))
 * (GaussianLayer, odin/bay/layers/continuous.py:196-260; odin/backend/maths.py:279-281);
 * softplus1 = 2 -> the same (loc | raw) split parameterises
 * QuantizedLogistic(loc, softplus(raw) + e^-7, low=0, high=255, inputs_domain='sigmoid')
 * (image_networks.py:55-71; odin/bay/distributions/quantized.py:50-204): x in [0,1], pixel value
 * y = x*255, log P[Y = y] of the discretised logistic, gradients wrt loc and raw.
 * A NULL h is a dry run that only reports n_part (partials per sample). */
int odin_elbo_gaussian_fwd_bwd(const float* h, const float* x, float* llk_part, float* dh,
                               const float* scale, int B, int n_pix, int C, int softplus1,
                               int* n_part_out, void* stream);
/* The Gaussian head in one pass: Conv2D 1x1 (Cin -> 2C maps, linear; image_networks.py:505-511) ->
 * Independent(Normal(loc, scale)).log_prob(target) (:95-102; softplus1 as above, 0 or 1) forward + backward
 * (examples/vae/vae_audio.py:84-110: the audio VAE's decoder).  h [B*n_pix, Cin] = the activation below the head
 * (activation h_act already applied), w1 [Cin, 2C], b1 [2C], target [B, n_pix, C].  Writes logits [B, n_pix, 2C],
 * dlogits (optional) = -scale * d llk / d logits, dh [B*n_pix, Cin] = (dlogits w1^T) * act'(h), llk_part
 * [B][n_part] (the layout odin_elbo_finalize sums), one row (dW1 | db1) per workgroup of wslab and (optional) one
 * row of Cin column sums of dh per workgroup of colsum_slab (the bias gradient of a Conv2DTranspose below);
 * dh_amax (optional): the range word of dh.  Replaces odin_conv2d_fwd + odin_elbo_gaussian_fwd_bwd +
 * odin_conv2d_wgrad + odin_conv2d_dgrad of that layer (three passes over h) with one.  Cin in {8, 16, 32},
 * C in {1, 3}; a NULL h is a dry run that reports n_part / rows; -2: shapes outside the kernel.
 * softplus1 = 3: the Bernoulli observation instead (image_networks.py:87-93): w1 [Cin, C], logits [B, n_pix, C],
 * llk = Independent(Bernoulli(logits)).log_prob(target) -- a Bernoulli decoder whose last two layers do not fit
 * odin_bernoulli_tail_fwd_bwd's plane kernel (5x5 kernels, rows that are not 8 / 16 / 32 pixels wide). */
/* Between _begin and _end the calling thread's weight gradients of the 4x4 / stride-2 plane layers are collected
 * instead of launched, and _end issues them as ONE multi-layer launch on the stream they were given (they depend on
 * nothing but their own layer's tensors; a backward pass has five of them: -4 launches).  odin_slab_reduce issues
 * whatever is still pending first, so a caller that never calls _end stays correct.  Results are bit-identical to
 * the separate launches.  (Measured on the VAE step: slower than launching each weight gradient right behind the data
 * gradient that produced its dy, which then still sits in the Infinity Cache -- odin_ai_amd/engine.py keeps it off.) */
void odin_wgrad_planes_defer_begin(void);
int odin_wgrad_planes_defer_end(void* stream);
/* tests / diagnostics: the launch size (FLOP) from which the convolutions that fit no plane kernel run on the
 * two-plane implicit GEMM (igemm_h.hip) rather than the fp32 one; 0 = every applicable shape, < 0 = only report.
 * Returns the previous value. */
double odin_debug_igemm_h_min_flop(double flop);
/* block-window plane kernels (blk_planes.hip): the launch size from which they take a 4x4 / stride-2 layer (tests: 0), and
 * a switch (0: off, 1: on, < 0: query) for A/B runs; both return the previous value */
double odin_debug_blk_min_flop(double flop);
int odin_debug_blk_planes(int enable);
/* diagnostics: 1 = the block-window kernels also take the layers the row-window plane kernels serve (A/B runs) */
int odin_debug_blk_first(int on);
/* tests / diagnostics: the number of 64 x 64 tiles from which a Dense weight gradient with both widths >= 256 runs on the
 * LDS-staged kernel (dense_h.hip: dense_hw; default 128; tests: 1; a huge value switches it off for A/B runs); < 0 = only
 * report.  Returns the previous value. */
int odin_debug_dense_hw_min_tiles(int tiles);

/* ---- fused Gaussian tail of the TRAINING step (blk_planes.hip): Conv2DTranspose(k4, s2, 32 -> 32, activation d->act) ->
 * Conv2D 1x1 linear with 2 maps (w1 [32, 2], b1 [2]: loc | raw scale) -> Independent(Normal(loc, raw | softplus1(raw)))
 * .log_prob(target [B, OH, OW, 1]), forward AND backward in one launch, any image size: logits [B, OH, OW, 2] (optional
 * out), g_out = dL/d(pre-activation of the layer) for L = -scale[0] * sum llk, llk_part[b][part] (n_part per sample),
 * tail_slab[g][32 * 2 (dW1) | 2 (db1) | 32 (db of the layer)], g < *slab_rows_out; d->x_amax (optional) / d->dy_amax
 * (optional: receives max |g_out|) as in odin_bernoulli_tail_fwd_bwd.  Replaces decoder4 -> decoder6 ->
 * RVconf(..., 'gaus', projection=False) of the audio VAE (examples/vae/vae_audio.py:84-110; image_networks.py:505-511)
 * + px.log_prob(x) + their tape.gradient.  softplus1: 0 = raw scale, 1 = softplus1.  g_out == NULL = dry run;
 * -2: shapes outside the kernel (odin_gaussian_tail_applicable: C == 1, 32 -> 32 channels, k4 s2). */
int odin_gaussian_tail_applicable(const odin_conv_desc* d, int C);
int odin_gaussian_tail_fwd_bwd(const float* x, const float* w, const float* bias, const float* w1, const float* b1,
                               const float* target, float* logits, float* g_out, float* llk_part, int* n_part_out,
                               float* tail_slab, int* slab_rows_out, const float* scale, const odin_conv_desc* d, int C,
                               int softplus1, void* stream);
int odin_gaussian_head_fwd_bwd(const float* h, const float* w1, const float* b1, const float* target,
                               float* logits, float* dlogits, float* dh, float* llk_part, int* n_part_out,
                               float* wslab, int* rows_out, float* colsum_slab, const float* scale, int B,
                               int n_pix, int Cin, int C, int softplus1, int h_act, uint32_t* dh_amax,
                               void* stream);
/* MixtureQuantizedLogistic(params, n_components=K, n_channels=C, low=0, high=255,
 * inputs_domain='sigmoid') (odin/bay/distributions/quantized.py:206-349; built by
 * _parse_distribution 'mixqlogistic', image_networks.py:72-85): h [B, n_pix, K*n_out] with
 * n_out = 1 + 2C + C(C-1)/2 (mixture logit | loc | raw scale | channel coefficients per component),
 * scale = softplus(raw) + e^-7 (:280-282); llk partials per sample, dh = -scale * d log p / d h.
 * C in {1, 3}, K = 10.  A NULL h is a dry run that only reports n_part. */
int odin_elbo_mixqlogistic_fwd_bwd(const float* h, const float* x, float* llk_part, float* dh,
                                   const float* scale, int B, int n_pix, int C, int K,
                                   int* n_part_out, void* stream);
/* VAEStep.call / VariationalModel.elbo (variational_autoencoder.py:117-126;
 * odin/bay/vi/_base.py:151-194): llk[b] = sum parts; elbo = llk - beta*kl - tc;
 * out[0]=loss=-mean(elbo), out[1]=mean llk, out[2]=mean beta*kl, out[3]=tc term.
 * hyper: DEVICE floats {beta, tc_coef}; tc: optional DEVICE scalar (total correlation or
 * mean discriminator logit), tc term = tc_coef*tc[0] (beta_vae.py:123-129,
 * factor_vae.py:211-228). */
int odin_elbo_finalize(const float* llk_part, int n_part, const float* kl, const float* hyper,
                       const float* tc, float* llk, float* out4, int B, void* stream);
/* out[0] = mean(x[0..n)) (deterministic single-workgroup tree) */
int odin_mean(const float* x, int n, float* out, void* stream);

/* ---- beta-TCVAE total correlation (odin/bay/vi/losses.py:101-157), never materialising
 * the [B,B,D] tensor.  tc_out[0] = TC; grads scaled by coef[0] (DEVICE scalar (beta-1)).
 * tc_out is also the workspace: odin_total_correlation_workspace(B_local, B_global, D) floats
 * (B_local = B_global = B for the single-device form). */
int odin_total_correlation_workspace(int B_local, int B_global, int D);
int odin_total_correlation_fwd_bwd(const float* z, const float* p, float* tc_out, float* dz,
                                   float* dloc, float* dscale, const float* coef, int B, int D,
                                   void* stream);
/* The same estimator with the batch sharded over ranks (losses.py:136-157 couples every
 * pair (j, i) of the GLOBAL batch): this rank evaluates its own rows j (z_local [B_local, D])
 * against ALL posteriors i (p_global [B_global, 2D], all-gathered): tc_out[0] = this rank's share
 * sum_j(...) / B_global (sum the shares of the ranks), dz_local [B_local, D] is complete, and
 * dloc_part / dscale_part [B_global, D] hold this rank's partial sums over its j for EVERY
 * posterior i (reduce-scatter them).  tc_out: odin_total_correlation_workspace(B_local, B_global, D) floats. */
int odin_total_correlation_shard(const float* z_local, const float* p_global, float* tc_out,
                                 float* dz_local, float* dloc_part, float* dscale_part,
                                 const float* coef, int B_local, int B_global, int D, void* stream);
/* permute_dims (odin/bay/vi/utils.py:233-269): out[i,l] = z[perm[i,l], l]; perm int32 [B,D] */
int odin_permute_dims(const float* z, const int32_t* perm, float* out, int B, int D, void* stream);
/* per-column random permutations generated on device (Philox), perm int32 [B,D] */
int odin_random_perm(int32_t* perm, int B, int D, uint64_t seed, const int32_t* step_dev,
                     void* stream);
/* odin_random_perm followed by odin_permute_dims(z, perm, out) as ONE launch (all rows of z local: one GPU). */
int odin_random_permute_dims(int32_t* perm, const float* z, float* out, int B, int D, uint64_t seed,
                             const int32_t* step_dev, void* stream);
/* dtc_loss (odin/bay/vi/autoencoder/factor_discriminator.py:200-235):
 * out[0] = 0.5*(mean softplus(-lz) + mean softplus(lperm)); grads wrt both logit vectors */
int odin_dtc_loss_fwd_bwd(const float* logit_z, const float* logit_perm, float* out,
                          float* dlogit_z, float* dlogit_perm, int n, void* stream);

/* ---- optimiser: tf.optimizers.Adam created at odin/networks/base_networks.py:85-112,
 * applied at :604 -- Keras form, epsilon outside the bias-corrected sqrt.
 * hyper (DEVICE): {alpha_t = lr*sqrt(1-b2^t)/(1-b1^t), beta1, beta2, eps, grad_scale};
 * gnorm2 (optional DEVICE scalar): if non-NULL and clip>0 the gradient is scaled by
 * clip/max(sqrt(gnorm2),clip) (tf.clip_by_global_norm, base_networks.py:588); if
 * non-finite AND flag is non-NULL the update is skipped (nan_gradients_policy, base_networks.py:519-547)
 * and flag[0] is set to 1; with flag == NULL ('ignore') the update is applied whatever the norm holds. */
int odin_adam_step_flat(float* theta, const float* g, float* m, float* v, size_t n,
                        const float* hyper, const float* gnorm2, float clip, int32_t* flag,
                        void* stream);
/* odin_adam_step_flat (no clip, no NaN guard) whose last small gradient pieces are formed INSIDE the launch (round 6:
 * FactorVAE's discriminator step ended in think_wgrad -> slab_reduce -> adam, two 5 us launch floors in front of the update):
 *   x != NULL     the weight gradient of a thin-K Dense layer (K <= 32): (dW [K][N] | db [N]) = (x^T dy | column sums of dy)
 *                 over B rows, written to g + w_off and applied (w_off and (K + 1) N multiples of 4)
 *   slab != NULL  g[slab_off + i] = sum over slab_rows rows of slab[r * slab_stride + i], i < slab_n (rows ascending),
 *                 written and applied (slab_off a multiple of 4)
 *   zero != NULL  zero_n 32-bit words cleared (the step's range words)
 * Every other parameter is updated from g as odin_adam_step_flat does. */
typedef struct odin_adam_fold {
  const float* x; const float* dy; int B, K, N; size_t w_off;
  const float* slab; int slab_rows; size_t slab_stride, slab_n, slab_off;
  void* zero; int zero_n;
} odin_adam_fold;
int odin_adam_step_fold(float* theta, float* g, float* m, float* v, size_t n, const float* hyper,
                        const odin_adam_fold* fold, void* stream);
/* ---- VariationalAutoencoder.marginal_log_prob (variational_autoencoder.py:396-513):
 * n posterior samples per input from one encoder pass: z[k,b,:] = loc_b + softplus(raw_b)*eps[k,b,:],
 * logq[k,b] = log q(z_kb | x_b), logp[k,b] = log N(z_kb; 0, I)   (p: [B,2D], eps/z: [n,B,D]);
 * odin_logmeanexp_rows: out[b] = logsumexp_k in[k,b] - log n  (tf.reduce_logsumexp(axis=0) - C). */
int odin_latent_sample_logprob(const float* p, const float* eps, float* z, float* logq,
                               float* logp, int n, int B, int D, void* stream);
int odin_logmeanexp_rows(const float* in, float* out, int n, int B, void* stream);
/* out[b] = sum_j llk_part[b*n_part + j]: the per-sample log p(x|z) from the partial sums the
 * fused observation kernels write (Distribution.log_prob, variational_autoencoder.py:530) */
int odin_sum_parts(const float* llk_part, int n_part, float* out, int B, void* stream);

/* ---- gradient policies of Networks.optimize (odin/networks/base_networks.py:549-596), applied
 * to the flat gradient buffer between the backward pass and the Adam launch, in this order:
 *  odin_grad_skip_threshold : `skip_update_threshold` (:549-578) -- if ANY gradient element is
 *      >= threshold (and enable[0] != 0, = `step >= when_skip_update`), every gradient of the
 *      step becomes 0 (the reference forms g - g) and skipped_count[0] += 1; the optimiser still
 *      runs (momentum-only update), exactly like apply_gradients on the zeroed list.
 *      hit: DEVICE int scratch (1 = threshold reached), enable / skipped_count may be NULL.
 *  odin_clip_by_norm_segments : `clipnorm` (:579-583), tf.clip_by_norm per variable; seg_offsets
 *      (DEVICE, n_segments + 1 int64) delimit the variables inside the flat buffer.
 *  odin_clip_by_value : `clipvalue` (:592-596), tf.clip_by_value(g, -c, c).  The reference clips
 *      by global norm BEFORE by value (:584-591): when both are requested pass the DEVICE scalar
 *      gnorm2 = sum g^2 (odin_sumsq_flat) and global_clipnorm, and the kernel first scales g by
 *      global_clipnorm / max(sqrt(gnorm2), global_clipnorm); gnorm2 = NULL: value clip only. */
int odin_grad_skip_threshold(float* g, size_t n, float threshold, const int32_t* enable,
                             int32_t* hit, int32_t* skipped_count, void* stream);
int odin_clip_by_norm_segments(float* g, const int64_t* seg_offsets, int n_segments, float clipnorm,
                               void* stream);
int odin_clip_by_value(float* g, size_t n, float clipvalue, const float* gnorm2,
                       float global_clipnorm, void* stream);

/* out[0] = sum g^2 (deterministic two-stage); workspace >= 1024 floats */
int odin_sumsq_flat(const float* g, size_t n, float* workspace, float* out, void* stream);
/* odin_sumsq_flat + odin_adam_step_flat in two launches instead of three: the Adam launch sums
 * the stage-1 partials itself (same fixed order, bit-identical norm) and writes it to gnorm2_out.
 * workspace: >= 1024 floats. */
int odin_sumsq_adam_flat(float* theta, const float* g, float* m, float* v, size_t n,
                         const float* hyper, float* workspace, float* gnorm2_out, float clip,
                         int32_t* flag, void* stream);

/* The same two launches as the LAST launches of a step whose per-step scalars live in a device ring (round 5: no
 * per-step host copy): `ring` [rows][row_floats] holds the rows of the coming steps (row of step s at s mod rows, rows a
 * power of two), `cur` is the row the step's kernels read (`hyper` and `elbo_hyper` point into it), t = the int32 at
 * cur[t_word].  The stage-1 launch copies hyper[0..4] to `staged` (>= 8 floats), the Adam launch reads them from there
 * and loads ring row (t + 1) mod rows into `cur` for the next step.  llk_part != NULL: the ELBO finalisation rides in
 * the first launch as in odin_sumsq_adam_finalize_flat.  Replaces the reference's per-step evaluation of the optimiser's
 * learning-rate schedule / beta annealing inside Networks.optimize (base_networks.py:549-596, beta_vae.py:97-126). */
int odin_sumsq_adam_ring(float* theta, const float* g, float* m, float* v, size_t n, const float* hyper,
                         float* workspace, float* gnorm2_out, float clip, int32_t* flag, const float* llk_part,
                         int n_part, const float* kl, const float* elbo_hyper, const float* tc, float* llk,
                         float* out4, int B, const float* ring, float* cur, float* staged, int rows, int row_floats,
                         int t_word, void* stream);

/* Round 5: the gradient norm's stage-1 launch rides in the slab reduction.  odin_slab_reduce_sumsq = odin_slab_reduce
 * that also leaves sum(g^2) of what it writes as one partial per active workgroup in part[0 .. *n_parts_out): jobs whose
 * dst lies inside [g, g + g_n) count -- the caller guarantees that they tile the flat gradient exactly once -- others
 * (the range-word reset) do not.  stage_src / stage_dst (optional): stage_n <= 256 floats copied by the same launch
 * (the step's hyper-parameter row: the Adam launch advances `hyper` while it still needs this step's scalars).
 * part == NULL: dry run, only *n_parts_out (static for a given job list: capture-safe).
 * odin_adam_ring_parts: norm from those partials, clip scale, NaN guard, Adam, the ring advance of
 * odin_sumsq_adam_ring and (llk_part != NULL) the ELBO finalisation, in ONE launch; `staged` = the staged row,
 * alpha_off / elbo_off = offsets of Adam's five scalars / of the ELBO weights in it (ring == NULL: no advance).  Together they replace
 * odin_slab_reduce + odin_sumsq_adam_ring (three launches -> two; base_networks.py:584-596). */
int odin_slab_reduce_sumsq(const odin_reduce_job* jobs, int n_jobs, const float* g, size_t g_n, float* part,
                           int* n_parts_out, const float* stage_src, float* stage_dst, int stage_n, void* stream);
int odin_adam_ring_parts(float* theta, const float* g, float* m, float* v, size_t n, const float* staged,
                         int alpha_off, int elbo_off, const float* parts, int n_parts, float* gnorm2_out, float clip,
                         int32_t* flag, const float* llk_part, int n_part, const float* kl, const float* tc, float* llk,
                         float* out4, int B, const float* ring, float* cur, int rows, int row_floats, int t_word,
                         void* stream);

/* odin_sumsq_adam_flat whose first launch also finalises the step's ELBO (the arguments of odin_elbo_finalize):
 * one launch less per training step; loss / norm / update bit-identical to the separate calls. */
int odin_sumsq_adam_finalize_flat(float* theta, const float* g, float* m, float* v, size_t n,
                                  const float* hyper, float* workspace, float* gnorm2_out, float clip,
                                  int32_t* flag, const float* llk_part, int n_part, const float* kl,
                                  const float* elbo_hyper, const float* tc, float* llk, float* out4, int B,
                                  void* stream);

/* ---- counter-based RNG (the reference uses TF's Philox via tfd.sample; streams are not
 * reproducible across frameworks, so parity tests pass eps explicitly) ------------------ */
int odin_rng_normal(float* out, size_t n, uint64_t seed, const int32_t* step_dev, void* stream);

/* ---- on-device input pipeline: batch gather from an HBM-resident uint8 dataset
 * [N, n_per] + ImageDataset.normalize (odin/fuel/image_data/_base.py:130-147), replacing the
 * tf.data map/batch of create_dataset (:338-395).  out[b, e] = normalize(premul * data[idx[b], e]);
 * mode 0 'probs': clip(x,0,255)/255 clipped to [1e-6, 1-1e-6]; 1 'tanh': clip(x/255*2-1);
 * 2 'raster': clip(x,0,255); 3: binarised data, premul * x unchanged.  dSprites stores 0/1
 * and uses premul = 255 (fuel/image_data/shapes.py:69-72,80).  n_per % 16 == 0. */
int odin_gather_normalize_u8(const uint8_t* data, const int32_t* idx, float* out, int B, int n_per,
                             float premul, int mode, void* stream);
/* the same for a float32 dataset that is already normalised: out[b, :] = data[idx[b], :]; n_per % 4 == 0.  The
 * batch selection of Networks.fit on an in-memory array (odin/networks/base_networks.py:642-812). */
int odin_gather_rows_f32(const float* data, const int32_t* idx, float* out, int B, int n_per, void* stream);

/* ---- speech front-end: pre-emphasis -> STFT -> |.|^2 -> Slaney mel -> dB
 * (odin/preprocessing/signal.py:955-967,1442-1562,1623-1691,636-680), computed in float64 like
 * the reference, stored as float32.
 * y [B,n_samples] -> out [B,n_frames,n_mels], n_frames = 1 + (n_samples-frame_length)/step;
 * window [frame_length] float64 (already divided by its sum); twiddles [n_fft/2][2] float64 =
 * (cos, -sin)(2 pi k / n_fft); the mel_filters basis [n_mels, n_fft/2+1] as its non-zero band
 * per filter: fb_band [n_mels][3] = {first bin, count, offset into fb_vals}, fb_vals float64;
 * n_fft a power of two in [16, 2048]; top_db < 0 disables the per-utterance floor.
 * log_output: 0 = mel power; 1 = dB (power2db / amplitude_to_DB); 2 = the floored dB mapped to
 * [0, 1] by (dB - max)/top_db + 1 (input scaling for networks without batch normalisation);
 * 3 = ln(mel + 1e-6) (`AudioFeatureLoader(log_mels=True)`, odin/fuel/audio_data.py:222-223).
 * The same launch serves the TF variant (fuel/audio_data.py:17-101,210-270): un-normalised
 * periodic Hann window, preemph = 0, the HTK filterbank of tf.signal.linear_to_mel_weight_matrix. */
int odin_stft_mel_db(const float* y, const double* window, const double* twiddles,
                     const double* fb_vals, const int32_t* fb_band, float* out, int B,
                     int n_samples, int frame_length, int step_length, int n_fft, int n_mels,
                     double preemph, double top_db, int log_output, void* stream);
/* the same, storing only the first n_out_frames frames of every utterance (out [B, n_out_frames, n_mels]; the
 * top_db floor is still taken over all frames): the front-end writes the VAE's [B, T, n_mels, 1] input buffer
 * directly (fuel/audio_data.py:236-260 crops the spectrogram to max_length).  workspace: 8 * B floats or NULL;
 * with a workspace the frame blocks of an utterance are dealt to up to 8 workgroups (one workgroup per
 * utterance leaves most of the chip idle at batch 256) and the top_db floor follows in a second launch. */
int odin_stft_mel_db_frames(const float* y, const double* window, const double* twiddles,
                            const double* fb_vals, const int32_t* fb_band, float* out, int B, int n_samples,
                            int frame_length, int step_length, int n_fft, int n_mels, double preemph,
                            double top_db, int log_output, int n_out_frames, float* workspace, void* stream);

/* ---- data parallel (the reference has no distributed path) --------------------
 * Thin RCCL entry points: one process per GPU, every collective is enqueued on the caller's stream.
 * The 128-byte id of rank 0 (odin_comm_unique_id = ncclGetUniqueId) is handed to the other ranks by the
 * launcher (odin_ai_amd/dist.py broadcasts it through the torch.distributed store); odin_comm_init =
 * ncclCommInitRank.  RCCL is bound at the first call (dlopen of librccl; ODIN_RCCL_LIB overrides the
 * name): a process that never calls these needs no RCCL.
 *   odin_allreduce_flat      in-place SUM of the flat fp32 gradient bucket (Networks.optimize would apply
 *                            ONE gradient per variable, base_networks.py:415-624: the bucket makes the
 *                            replicas' gradients the global-batch gradient);
 *   odin_allgather_flat      recv[world * n] <- every rank's send[n] ([B, 3D] posterior rows for
 *                            total_correlation over the global batch, losses.py:136-157; z' rows for
 *                            permute_dims, vi/utils.py:262-267);
 *   odin_reduce_scatter_flat recv[n] <- sum over ranks of send[rank * n ...] (posterior-side TC gradients). */
int odin_comm_unique_id(void* id128);
int odin_comm_init(void** comm_out, const void* id128, int rank, int world_size);
int odin_comm_destroy(void* comm);
/* the RCCL library bound by this process: ODIN_RCCL_LIB, else the librccl the process has already mapped (the one
 * PyTorch bundles -- never a second copy), else the system one; "" before the first odin_comm_* call */
const char* odin_comm_library(void);
int odin_allreduce_flat(void* comm, float* buf, size_t n, void* stream);
int odin_allgather_flat(void* comm, const float* send, float* recv, size_t n_per_rank, void* stream);
int odin_reduce_scatter_flat(void* comm, const float* send, float* recv, size_t n_per_rank, void* stream);

/* diagnostics only: device buffer (>= 64 int64) that receives in-kernel cycle stamps of the
 * conv kernels' workgroup 0 (NULL disables; never set in production) */
int odin_debug_set_stamps(void* buf);
int odin_debug_set_wgrad_stamps(void* buf);
/* diagnostics: workgroup 0 of odin_neck_fwd / _bwd records 100 MHz wall-clock stamps at its phase boundaries into
 * buf[0..6] / buf[8..17] (int64, device memory); NULL: off */
int odin_debug_set_neck_stamps(void* buf);
/* diagnostics: launch shape of the persistent Bernoulli ELBO kernel (workgroups, chunks in flight per wave, 1 = the
 * software-pipelined body); a non-positive / negative field keeps its value (tools/elbo_sweep6.py) */
int odin_debug_elbo_shape(int blocks, int U, int pipelined);
/* diagnostics: workgroup (0, 0) of odin_stft_mel_db_frames records 100 MHz wall-clock stamps (per pass: start, staged,
 * FFT done, power spectrum done; then the end of its last pass) into buf (int64, device memory, >= 64 entries); NULL: off */
int odin_debug_set_mel_stamps(void* buf);
/* tests / A-B runs: 0 = n_fft 512 on the general front-end kernel instead of the register radix-16 one (mel.hip); < 0 = only
 * report.  Returns the previous value. */
int odin_debug_mel_r16(int enable);
/* tests / A-B runs: 0 = the RGB first layer's forward (4x4, 3 -> 32 channels) on the fp32 matrix instructions instead of two
 * f16 planes (smallc_conv.hip); < 0 = only report.  Returns the previous value. */
int odin_debug_smallc_planes(int enable);
/* diagnostics: the largest weight slice igemm_h keeps in LDS, in 16-value steps of 2 KB (default 32 = 64 KB; 64 measured slower);
 * returns the previous value, a negative argument only reads it */
int odin_debug_igemm_h_ldsw_steps(int steps);

/* ---- HIP-graph helpers (capture a sequence of the calls above, replay per step) ------- */
int odin_graph_begin(void* stream);
int odin_graph_end(void* stream, void** graph_exec_out);
int odin_graph_launch(void* graph_exec, void* stream);
int odin_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* ODIN_HIP_H */
