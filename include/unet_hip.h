/*
 * unet_hip.h -- C ABI of the MI355X (gfx950) U-Net segmentation engine.
 *
 * The reference (deadskull7/One-Stop-for-COVID-19-...) has NO native/FFI boundary of its
 * own: its hot path is Keras calls inside two Python runners.  Each entry point below
 * therefore cites the Keras call site it replaces (paths relative to
 * /root/reference/Scripts/; T1 = task1_preprocessing_plus_unet_with_comments.py,
 * T3 = task3_lung_segmentation_unet.py, which is line-for-line the same graph/recipe).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++/torch types cross the boundary.
 *   - every function returns int32 status: 0 = ok, <0 = error (unet_last_error(ctx)).
 *   - the CALLER owns every tensor buffer (device memory, fp32, NHWC = Keras channels_last).
 *     `ld*` arguments are the pixel stride in floats, so an op can read/write a channel
 *     slice of a wider NHWC buffer (zero-copy skip concatenation, T1:887).
 *   - all launches are asynchronous on the passed hipStream_t (`void* stream`); no hidden
 *     synchronisation, no allocation after unet_ctx_create.  State outside the caller's buffers: the ctx -- it owns two small device
 *     scratch areas (BatchNorm reduction slots; the split weight image of a ConvT launch), so launches through ONE ctx belong on one stream
 *     at a time -- and its options (unet_ctx_set_option).  The library reads NO environment variables: the kernel family is the `algo`
 *     argument of every convolution entry point / of unet_model_create, the graph-level choices are ctx options.
 *   - there is NO CPU fallback: without a gfx950 device unet_ctx_create fails.
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_ABI_VERSION 16

typedef struct unet_ctx unet_ctx;
typedef struct unet_model unet_model;
/* bf16 storage element (raw bit pattern; round-to-nearest-even from fp32) of the mixed-precision path */
typedef uint16_t unet_bf16;
enum { UNET_DTYPE_F32 = 0, UNET_DTYPE_BF16 = 1 };

/* status codes */
enum { UNET_OK = 0, UNET_E_ARG = -1, UNET_E_HIP = -2, UNET_E_SHAPE = -3, UNET_E_STATE = -4, UNET_E_NODEV = -5 };

/* kernel family of the convolutions (all are HIP kernels):
 *   AUTO  : fp32 conv3x3 / ConvT forward, data gradient and weight gradient as three v_mfma_f32_32x32x16_f16 products of a block-scaled two-term fp16 split
 *           (fp32-class accuracy inside the domain DESIGN.md 4g states) wherever the channel counts allow (multiples of 16 / 32), else as MFMA;
 *   MFMA  : STRICT fp32 -- v_mfma_f32_32x32x2_f32 kernels (exact fp32 multiply-add), VALU kernels for the shapes those do not take; the fallback for
 *           tensors outside the split's domain and the on-device reference of its accuracy claims;
 *   NAIVE : fp32 VALU kernels only (cross-check). */
enum { UNET_ALGO_AUTO = 0, UNET_ALGO_NAIVE = 1, UNET_ALGO_MFMA = 2 };

int32_t unet_abi_version(void);
int32_t unet_ctx_create(int32_t device_id, unet_ctx** out);
void unet_ctx_destroy(unet_ctx* ctx);
const char* unet_last_error(const unet_ctx* ctx);
/* 1 = op-level timing with hipEvents (bench.py roofline leg); adds a sync per op */
int32_t unet_ctx_set_profiling(unet_ctx* ctx, int32_t on);
/* Options of a context (defaults = the shipped path).  A model reads them when it is CREATED (unet_model_create), op-level entry points when they launch;
 * so one process can hold models / contexts of different settings side by side.
 *   RELU_BITS (1)          ReLU masks of the data gradients as one bit per element, written by the producing conv (0: the fp32 activation is re-read)
 *   BN_FOLD (2)            decoder BatchNormalization folded into the conv behind it: 0 = explicit statistics / apply passes, 1 = forward + weight gradient +
 *                          backward sums folded, 2 = also the BatchNorm backward applied in the data-gradient epilogue, 3 = as 2 and the classifier's
 *                          16-channel first block too (T2:748-751: +15 % on that step; its folded pre-activations carry more round-off -- the raw
 *                          activations have a large mean -- so at 224 x 224 x 256 about twice as many ReLU decisions differ from a float64 evaluation)
 *   ENC_BN_FUSED (1)       encoder tail backward without a statistics pass (sums from the pooled tensors + closed-form skip term, one fused apply pass)
 *   BN_CONCAT_ANALYTIC (1) decoder BatchNorm statistics: skip half from the encoder layer's sums, only the upsampled half measured
 *   BN_FUSE_STATS (1)      BatchNorm statistics accumulated by the producing conv's epilogue (unet_request_bn_stats honoured)
 *   DETERMINISTIC (0)      1 = no floating-point atomics anywhere, so reruns are bit-identical: the reductions of the stand-alone passes go through per-workgroup slots
 *                          folded in index order; the sums a kernel EPILOGUE takes (BatchNorm statistics, the fused head's loss / gradient sums, the pooled sums -- launches
 *                          with far more workgroups than slots) leave as exact integer window sums, four 64-bit words per value, whose addition is associative
 *                          (ABI v15: the same fused graph as the default mode; 1.6 % of the step, 9 % before).  Domain of an epilogue partial sum: |t| < 2^39,
 *                          bits below 2^-80 dropped; outside it (and for Inf / NaN) the folded value is NaN
 *   HEAD_FUSED (1)         fp32 U-Net, h2 kernels: the 1x1 sigmoid head (T1:913), the loss sums and the per-channel sums of the head's weight gradient come out of
 *                          the epilogue of the last conv3x3 (no pass over its 32-channel output in forward; backward writes dL/d(conv output) from p, the labels
 *                          and one bit per element); 0 = the separate head_fwd / head_bwd passes
 *   SKIP_RAW (1)           fp32 U-Net with BN_FOLD >= 2, ENC_BN_FUSED and BN_CONCAT_ANALYTIC: the second conv of an encoder block (T1:860) writes straight into the skip half
 *                          of its concat (T1:908) and the encoder BatchNorm's output is never stored: max-pool reads the raw tensor, the folded decoder BatchNorm is
 *                          composed with the encoder one (two affine maps in a row are one).  -1 GB of writes per step at 512 x 512 x 16.  0 = the normalised copy is stored
 *   POOL_SUMS_FUSED (1)    fp32 U-Net with ENC_BN_FUSED: the pooled-path sums of an encoder tail's BatchNorm backward come out of the epilogue of the data gradient that
 *                          produces the pooled tensor's gradient (no pass over the pooled tensors); dropout-removed elements are recognised by the -0.0f the forward stored
 *   HEAD_BWD_FUSED (1)     with HEAD_FUSED and RELU_BITS: dL/d(output of the last conv3x3) = dz_p w_c [y_pc > 0] is never written as a tensor -- unet_head_dzm leaves
 *                          {dz_p, 32 mask bits} per pixel (8 bytes instead of 128) and the last conv's data gradient and weight gradient expand that stream while they
 *                          stage it (-1.5 GB of traffic per step at 512 x 512 x 16); 0 = unet_head_dy writes the fp32 tensor
 *   CONV_PP (13; default 1) the conv3x3 forward / data-gradient launches of the shallow levels (K = 32 -> 32 channels) on the persistent two-half schedule of
 *                          kernels_conv_pp.hip: one 512-thread workgroup per CU, the layer's split weight image resident in LDS, one half's MFMAs over the other half's
 *                          loads, split and stores.  Same arithmetic as the h2 kernels (one block exponent per 8 x 32 pixel tile); taken by launches of at least four tiles per
 *                          half-workgroup (2048 tiles: 512 x 512 from batch 2 up).  0 = conv_h2_kernel everywhere; 2 = also smaller launches (tests)
 *   (options 11 / 12 of ABI v13-v14 -- WGRAD_ATOMIC, C1A_RECOMPUTE -- were same-box A/B losers and left the library in v15; DESIGN.md keeps the measurements)
 */
enum { UNET_OPT_RELU_BITS = 1, UNET_OPT_BN_FOLD = 2, UNET_OPT_ENC_BN_FUSED = 3, UNET_OPT_BN_CONCAT_ANALYTIC = 4, UNET_OPT_BN_FUSE_STATS = 5, UNET_OPT_DETERMINISTIC = 6,
       UNET_OPT_HEAD_FUSED = 7, UNET_OPT_SKIP_RAW = 8, UNET_OPT_POOL_SUMS_FUSED = 9, UNET_OPT_HEAD_BWD_FUSED = 10, UNET_OPT_CONV_PP = 13 };
int32_t unet_ctx_set_option(unet_ctx* ctx, int32_t option, int32_t value);
int32_t unet_ctx_get_option(unet_ctx* ctx, int32_t option);   /* >= 0: the value; < 0: error */

/* activations of the conv epilogue and "mask modes" of the backward epilogues (derivative of the activation -- and of the
 * dropout fused behind it -- that produced a stored tensor m):
 *   UNET_MASK_RELU 1[m>0]   UNET_MASK_ELU  m>0 ? 1 : m+1   UNET_MASK_ELU_DROP  m = dropout(elu(z)), keep mask recomputed
 *   from the counter-based RNG stream (rate, seed) of that dropout */
enum { UNET_ACT_NONE = 0, UNET_ACT_RELU = 1, UNET_ACT_ELU = 2 };
enum { UNET_MASK_NONE = 0, UNET_MASK_RELU = 1, UNET_MASK_ELU = 2, UNET_MASK_ELU_DROP = 3, UNET_MASK_RELU_BITS = 9 /* see unet_request_relu_bits */ };

/* ------------------------------------------------------------------------------------
 * Op level.  Replaces: Conv2D(C,(3,3),activation='relu',padding='same')   T1:859-911
 *   y[n,i,j,o] = act(b[o] + sum_{a,b,c} x[n,i+a-1,j+b-1,c] * w[a,b,c,o]);  w is HWIO.
 * act: 'relu' (U-Net, T1:859) or 'elu' (U-Net++, task1_unet_plus_plus.py:876); drop_rate > 0 fuses the Keras Dropout
 * layer that follows the conv (task1_unet_plus_plus.py:862, 877) into the epilogue (inverted dropout, training only).
 * w_ws: device scratch of unet_conv3x3_w_ws_floats(cin, cout) floats for the prepared (split fp16) weight image of the h2 kernels; may be
 * NULL, then only the strict fp32 kernels are used.
 * ---------------------------------------------------------------------------------- */
size_t unet_conv3x3_w_ws_floats(int32_t cin, int32_t cout);
/* which family a forward / data-gradient launch of this shape resolves to: UNET_ALGO_AUTO (the fp16-split h2 kernels), _MFMA (strict fp32 MFMA) or _NAIVE (VALU) */
int32_t unet_conv3x3_pick_algo(int32_t algo, int32_t wd, int32_t cin, int32_t cout);
/* fp32-MFMA-time equivalent of that launch's matrix work: 1 (strict / VALU), 3 * 157.3 / 2500 (h2: three fp16 MFMA products per multiply) */
double unet_conv3x3_exec_ratio(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout);
/* ... and of the weight-gradient launch (unet_conv3x3_bwd_weights with the workspace it asks for) */
double unet_conv3x3_wgrad_exec_ratio(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout);
/* Conv2D / Conv2DTranspose followed by a training-mode BatchNormalization (T1:860-861 `Conv2D(...)(c1)` -> `BatchNormalization()(c1)`, T1:886-888
 * `Conv2DTranspose` -> `concatenate` -> `BatchNormalization`): arms the NEXT unet_conv3x3_fwd / unet_convT2x2_fwd on this context to add the
 * per-channel (sum y, sum y^2) of the values it STORES (c channels; after activation and dropout) to the context's accumulators from its epilogue,
 * where its kernel can (the fp32 h2 kernels); the unet_bn_stats / unet_bn_stats_concat call that MUST follow on that tensor then folds them instead of reading
 * the tensor again (any other kernel ignores the request and that call does its own pass -- same results either way).  c = 0 disarms. */
int32_t unet_request_bn_stats(unet_ctx*, int32_t c);
/* The ReLU mask of a data gradient as ONE BIT per element instead of the stored fp32 activation (the backward of the T1:859-860 conv pairs reads
 * `c1 > 0` only).  unet_request_relu_bits arms the NEXT unet_conv3x3_fwd (act = UNET_ACT_RELU, no dropout) on this context to also write the sign bits
 * of what it stores into `bits` (unet_relu_bits_bytes(n, h, w, cout) bytes of device memory; layout: 64-bit words [n][y][x / 8][c / 32][4], word k of an
 * 8-pixel x 32-channel cell holds bit (x % 8) * 8 + (c % 32) / 4 for the channels with c % 4 == k); that call fails with UNET_E_SHAPE if its kernel
 * cannot (ask unet_relu_bits_supported(algo, h, w, cin, cout) first: cout % 32 == 0, w % 8 == 0, the fp32 h2 kernels).  unet_conv3x3_bwd_data then takes
 * mask_src = bits with mask_mode = UNET_MASK_RELU_BITS where unet_relu_bits_supported(algo, h, w, cout, cin) holds for ITS launch (K = cout, M = cin). */
int32_t unet_relu_bits_supported(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout);
size_t unet_relu_bits_bytes(int32_t n, int32_t h, int32_t wd, int32_t c);
int32_t unet_request_relu_bits(unet_ctx*, void* bits);
/* Inference on one slice (T1:1136-1137: `model.predict` on a single 512 x 512 image) leaves the deep levels with fewer workgroups than the chip has CUs, each walking a long
 * chain of dependent loads over its contraction.  unet_allow_k_slices arms the NEXT unet_conv3x3_fwd / unet_conv3x3_bwd_data on this context (no dropout, no mask, no armed
 * statistics / sign bits, fp32 h2 kernels) to contract 2-4 slices of K side by side into context-owned slabs and add them -- with the bias and the ReLU -- in a second pass,
 * where its grid is that small (fewer than 1.5 workgroups per CU, K >= 256); any other launch ignores it.  Same result up to the order of the fp32 additions, which then
 * follows the grid size: training programs (whose data-parallel ranks must add in the order of the whole batch) never arm it, and a deterministic-mode context ignores it. */
int32_t unet_allow_k_slices(unet_ctx*);
/* Largest private segment (register-spill scratch, bytes per lane) among the h2 conv3x3 / ConvT kernels this context has launched so far: a build whose register-heavy
 * instances fell over their spill cliff shows here (and nowhere in the numerics) -- the GPU test suite holds it under a bound. */
int32_t unet_ctx_max_kernel_scratch_bytes(unet_ctx*);
int32_t unet_conv3x3_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y,
                         int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                         int32_t act, float drop_rate, uint64_t drop_seed, int32_t algo, float* w_ws, void* stream);
/* dx = conv3x3(dy, flip/transposed w) * mask_factor(mask_src): the derivative of the activation (+dropout) of the layer
 * that PRODUCED x is fused here (backward of the T1:859-860 conv pairs).  mask_rate/mask_seed: UNET_MASK_ELU_DROP only.
 * wt_ws: unet_conv3x3_w_ws_floats(cin, cout) floats of scratch for the flipped / transposed weights or their split image. */
int32_t unet_conv3x3_bwd_data(unet_ctx*, const float* dy, const float* w, const float* mask_src,
                              int32_t mask_mode, float mask_rate, uint64_t mask_seed,
                              float* dx, float* wt_ws, int32_t n, int32_t h, int32_t wd,
                              int32_t cin, int32_t cout, int32_t algo, void* stream);
/* BatchNormalization -> Conv2D(3x3) of the decoder blocks (T1:888-889, 895-896, 902-903, 909-910) WITHOUT the normalised tensor: the
 * affine z = scale[c] x + shift[c] (bnp = the float[>=2*cin] scale, shift that unet_bn_finalize_* writes) is folded into the conv --
 * weights scaled per input channel, and because Keras pads z (not x) with zeros, a bias per border class: a pixel on the first / last
 * row or column sees fewer taps of the shift (16 classes, exact).  The forward then reads the raw x.  The weight gradient runs on the raw
 * x as well and is corrected: dw = scale[c] dw_raw + shift[c] S[tap][o], S = db minus the border row / column sums of dy the tap
 * excludes (plus the corner).  Only where unet_conv3x3_bnfold_supported() says so (the h2 kernels: UNET_ALGO_AUTO, cin and cout multiples of 16; cout a
 * divisor of 256); ws: unet_conv3x3_bnfold_ws_floats floats, shared by the two calls of a step; gws: as unet_conv3x3_bwd_weights.
 * bn_bwd_sums (optional, with the kernel w and bnp = scale, shift, mean, invstd): the BatchNorm's backward sums double[2*cin] =
 * (sum dz, sum dz*xhat) as unet_bn_bwd_stats accumulates them, but WITHOUT reading dz or x -- dz is this conv's data gradient, so
 * sum_p dz_c = sum_{tap,o} w[tap][c][o] S[tap][o] and sum_p dz_c x_c = sum_{tap,o} w[tap][c][o] dw_raw[tap][c][o]. */
int32_t unet_conv3x3_bnfold_supported(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout);
size_t unet_conv3x3_bnfold_ws_floats(int32_t n, int32_t cin, int32_t cout);
int32_t unet_conv3x3_bnfold_fwd(unet_ctx*, const float* x, const float* bnp, const float* w, const float* bias, float* y, int32_t n, int32_t h,
                                int32_t wd, int32_t cin, int32_t cout, int32_t act, int32_t algo, float* ws, void* stream);
int32_t unet_conv3x3_bnfold_bwd_weights(unet_ctx*, const float* x, const float* bnp, const float* dy, const float* w, float* dw, float* db,
                                        double* bn_bwd_sums, void* gws, size_t gws_bytes, float* ws, int32_t n, int32_t h, int32_t wd,
                                        int32_t cin, int32_t cout, int32_t algo, void* stream);
/* dw[a,b,c,o] = sum x[n,i+a-1,j+b-1,c]*dy[n,i,j,o];  db[o] = sum dy.  dy already ReLU-masked.
 * ws: split-K scratch (unet_conv3x3_bwd_weights_ws_bytes). dw/db are OVERWRITTEN. */
size_t unet_conv3x3_bwd_weights_ws_bytes(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout);
int32_t unet_conv3x3_bwd_weights(unet_ctx*, const float* x, const float* dy, float* dw, float* db,
                                 void* ws, size_t ws_bytes, int32_t n, int32_t h, int32_t wd,
                                 int32_t cin, int32_t cout, int32_t algo, void* stream);

/* Replaces: Conv2DTranspose(C,(2,2),strides=(2,2),padding='same') + concatenate([u,c])
 * T1:886-887 (and 893-894, 900-901, 907-908).  Kernel layout [2,2,Cout,Cin] (Keras).
 *   u[n,2i+a,2j+b,o] = bias[o] + sum_c x[n,i,j,c] * w[a,b,o,c]
 * y is written with pixel stride ldy (the first Cout channels of the concat buffer). */
int32_t unet_convT2x2_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y,
                          int32_t ldy, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                          int32_t algo, void* stream);
int32_t unet_convT2x2_bwd_data(unet_ctx*, const float* dy, int32_t lddy, const float* w,
                               const float* relu_src, float* dx, int32_t n, int32_t h, int32_t wd,
                               int32_t cin, int32_t cout, int32_t algo, void* stream);
/* dw [2,2,Cout,Cin] and db are OVERWRITTEN; ws: split-K scratch (unet_convT2x2_bwd_weights_ws_bytes) */
size_t unet_convT2x2_bwd_weights_ws_bytes(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout);
int32_t unet_convT2x2_bwd_weights(unet_ctx*, const float* x, const float* dy, int32_t lddy,
                                  float* dw, float* db, void* ws, size_t ws_bytes, int32_t n,
                                  int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo,
                                  void* stream);

/* Replaces: BatchNormalization()  T1:861,867,873,879,888,895,902,909 (eps 1e-3, momentum .99)
 * Training forward = stats -> [optional cross-rank all-reduce of `sums`] -> finalize -> apply.
 *   sums: double[2*C] = (sum x, sum x^2), ACCUMULATED (zero it first: unet_zero).
 *   bnp : float[4*C]  = scale, shift, mean, invstd.
 *   count = elements per channel over the GLOBAL batch (n*h*w*world).
 * The statistics kernels (unet_bn_stats, unet_bn_bwd_stats, unet_maxpool2x2_dropout_bwd_bnstats) stage their atomics in a scratch the
 * CONTEXT owns: issue them on one stream at a time per context; use one context per stream otherwise. */
int32_t unet_bn_stats(unet_ctx*, const float* x, int32_t ldx, double* sums, int64_t pixels,
                      int32_t c, void* stream);
/* Statistics of the decoder's BatchNormalization over concatenate([u, c]) (T1:887-888, 894-895, 901-902, 908-909) when the skip half `c`
 * is the output of an encoder BatchNormalization of the same step: over the batch that output has mean beta and variance
 * gamma^2 var/(var + eps) exactly, so only the c_up channels of `u` (x_up, row stride ldx) are read; the c_skip pairs come from the
 * source layer's sums (src_sums = its double[2*c_skip] AFTER any cross-rank reduction, src_count = its global element count) and
 * parameters.  sums: double[2*(c_up+c_skip)] = (sums, sums of squares) of the concat, ACCUMULATED like unet_bn_stats; the analytic half is
 * scaled by `pixels` (this rank's count) so a cross-rank SUM of `sums` stays correct. */
int32_t unet_bn_stats_concat(unet_ctx*, const float* x_up, int32_t ldx, const double* src_sums, double src_count, const float* src_gamma,
                             const float* src_beta, double* sums, int64_t pixels, int32_t c_up, int32_t c_skip, void* stream);
int32_t unet_bn_finalize_train(unet_ctx*, const double* sums, double count, const float* gamma,
                               const float* beta, float* moving_mean, float* moving_var,
                               float* bnp, int32_t c, void* stream);
int32_t unet_bn_finalize_infer(unet_ctx*, const float* gamma, const float* beta,
                               const float* moving_mean, const float* moving_var, float* bnp,
                               int32_t c, void* stream);
int32_t unet_bn_apply(unet_ctx*, const float* x, int32_t ldx, const float* bnp, float* y,
                      int32_t ldy, int64_t pixels, int32_t c, void* stream);
/* backward: sums = double[2*C] (sum dy, sum dy*xhat), accumulated.  param_grads writes
 * dgamma/dbeta from the LOCAL sums (call before any cross-rank reduction of sums).
 * apply: dx = scale*(dy - sum_dy/count - xhat*sum_dyxhat/count) * mask_factor(x) (UNET_MASK_*: the derivative of what
 * produced the BN input x; UNET_MASK_ELU_DROP needs a dense x). */
int32_t unet_bn_bwd_stats(unet_ctx*, const float* dy, int32_t lddy, const float* x, int32_t ldx,
                          const float* bnp, double* sums, int64_t pixels, int32_t c, void* stream);
int32_t unet_bn_bwd_param_grads(unet_ctx*, const double* sums, float* dgamma, float* dbeta,
                                int32_t c, void* stream);
int32_t unet_bn_bwd_apply(unet_ctx*, const float* dy, int32_t lddy, const float* x, int32_t ldx,
                          const float* bnp, const double* sums, double count, int32_t mask_mode,
                          float mask_rate, uint64_t mask_seed, float* dx, int32_t lddx, int64_t pixels,
                          int32_t c, void* stream);

/* Replaces: MaxPooling2D((2,2)) + Dropout(0.25)  T1:862-863 (868-869, 874-875, 880-881).
 * rate 0 => plain pool.  Dropout keep-mask = counter-based RNG keyed by (seed, element index);
 * the backward recomputes it.  Ties in the max go to the first element in (di,dj) order.
 * bwd: dx[slice] (+)= routed gradient; accumulate=1 adds to what is already in dx (skip grad). */
int32_t unet_maxpool2x2_dropout_fwd(unet_ctx*, const float* x, int32_t ldx, float* y, int32_t n,
                                    int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed,
                                    void* stream);
int32_t unet_maxpool2x2_dropout_bwd(unet_ctx*, const float* x, int32_t ldx, const float* dy,
                                    float* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd,
                                    int32_t c, float rate, uint64_t seed, int32_t accumulate,
                                    void* stream);

/* Fused encoder tail BatchNormalization -> [skip] -> MaxPooling2D -> Dropout (T1:861-863): y = bn(x) into the concat slice
 * (ldy) and pooled = dropout(maxpool(y)) in one pass. */
int32_t unet_bn_apply_maxpool_dropout_fwd(unet_ctx*, const float* x, int32_t ldx, const float* bnp, float* y,
                                          int32_t ldy, float* pooled, int32_t n, int32_t h, int32_t wd,
                                          int32_t c, float rate, uint64_t seed, void* stream);
/* Fused backward of the same: dx[slice] += routed pool/dropout gradient (dx already holds the skip gradient) and
 * sums (double[2C], accumulated) += (sum d, sum d*xhat) of the finished gradient d, xhat = (y-beta)/gamma from the BN
 * output y (gamma != 0).  Replaces unet_maxpool2x2_dropout_bwd(accumulate=1) + unet_bn_bwd_stats. */
int32_t unet_maxpool2x2_dropout_bwd_bnstats(unet_ctx*, const float* y, int32_t ldy, const float* dy, float* dx,
                                            int32_t lddx, const float* gamma, const float* beta, double* sums,
                                            int32_t n, int32_t h, int32_t wd, int32_t c, float rate,
                                            uint64_t seed, void* stream);
/* The same encoder tail WITHOUT a pass for the statistics (fp32): the BatchNorm backward sums of the gradient g = g_skip + route(dy_pooled)
 * come from the pooled tensors alone -- the routed part touches only the arg-max elements, whose BatchNorm output is the pooled activation
 * itself (unet_maxpool2x2_dropout_bwd_sums: sum dy ks, sum dy ks (p/ks - beta)/gamma over 1/4 of the pixels) -- plus a closed-form term for
 * g_skip, which is the skip half of the DECODER BatchNorm's backward output: orthogonal to 1 exactly and to its own xhat up to
 * eps/(var+eps), and the decoder's xhat of a skip channel is gamma_e * invstd_d times the encoder's (unet_bn_bwd_skip_term adds
 * frac * gamma_d S2_d eps invstd_d^2 / gamma_e to sums[c + j]; the dec_* pointers at the decoder layer's skip channels, S2_d its
 * cross-rank-reduced sum dz*xhat, frac = this rank's share 1/world).  gamma == 0 is not supported (as in the fused form above).
 * unet_bn_maxpool_bwd_apply then does pool backward + skip add + BatchNorm backward + ReLU mask of the BN input x in ONE pass:
 * dx[.., c] = 1[x>0] scale (g - k1 - xhat k2); y = BN(x) is recomputed for the arg-max exactly as the forward stored it.
 * g_skip may be NULL (no skip connection: the classifier's Conv -> BN -> MaxPool tails, T2:752-754 -- then the pooled sums are the whole statistics). */
int32_t unet_maxpool2x2_dropout_bwd_sums(unet_ctx*, const float* pooled, const float* dy_pooled, const float* gamma, const float* beta, double* sums,
                                         int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream);
int32_t unet_bn_bwd_skip_term(unet_ctx*, double* sums, const double* dec_sum_dyxhat, const float* dec_invstd, const float* dec_gamma,
                              const float* gamma, int32_t c, double frac, void* stream);
int32_t unet_bn_maxpool_bwd_apply(unet_ctx*, const float* x, int32_t ldx, const float* bnp, const double* sums, double count, const float* g_skip,
                                  int32_t ldg, const float* dy_pooled, float* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c,
                                  float rate, uint64_t seed, void* stream);

/* Replaces: Conv2D(1,(1,1),activation='sigmoid') T1:913 fused with the reductions of
 * bce_dice_loss / dice_coeff T1:784-799.  p = sigmoid(b + x.w).  If y_true != NULL,
 * loss_sums (double[4], accumulated) += (sum bce_elem, sum t*p, sum t, sum p). */
int32_t unet_head_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* p,
                      const float* y_true, double* loss_sums, int64_t pixels, int32_t cin,
                      void* stream);
/* loss_out float[2] = (bce_dice_loss, dice_coeff) from (globally reduced) sums; count = GLOBAL
 * number of label elements. */
int32_t unet_loss_finalize(unet_ctx*, const double* loss_sums, double count, float* loss_out,
                           void* stream);
/* backward of loss + sigmoid + 1x1 conv: dx = dz*w [*(x>0) if relu_mask: x is a ReLU conv output, T1:911], dw = sum dz*x, db = sum dz,
 * dz = dL/dp * p(1-p) with dL/dp from the GLOBAL sums.  dw/db (cin+1 floats) are ACCUMULATED. */
int32_t unet_head_bwd(unet_ctx*, const float* x, const float* w, const float* p, const float* y_true,
                      const double* loss_sums, double count, float* dx, float* dw, float* db,
                      int64_t pixels, int32_t cin, int32_t relu_mask, void* stream);

/* The data gradient of the conv BEHIND `MaxPooling2D((2, 2))` + `Dropout(0.25)` (T1:862-865: c2 = Conv2D(64)(p1) ...) together with the pooled-path sums the
 * BatchNorm in front of that pool needs for its backward (what unet_maxpool2x2_dropout_bwd_sums computes in a pass of its own):
 *   dx = conv3x3(dy, flipped w) [n,h,w,cin]  (= the gradient of the pooled tensor);  sums[2 cin] += (sum dx ks, sum dx ks (p (1 - rate) - beta) / gamma)
 * with p = `pooled`, the OUTPUT of unet_bn_apply_maxpool_dropout_fwd at the same rate: that kernel stores a dropout-removed element as -0.0f (a kept zero as
 * +0.0f), which is how ks = 1 / (1 - rate) | 0 is read back without replaying the random stream.  fp32 UNET_ALGO_AUTO kernels, cin % 32 == 0, not in
 * deterministic mode (unet_conv3x3_bwd_data_pool_sums_supported).  wt_ws as for unet_conv3x3_bwd_data. */
int32_t unet_conv3x3_bwd_data_pool_sums_supported(unet_ctx*, int32_t algo, int32_t wd, int32_t cin, int32_t cout);
int32_t unet_conv3x3_bwd_data_pool_sums(unet_ctx*, const float* dy, const float* w, const float* pooled, const float* gamma, const float* beta, float rate,
                                        float* dx, double* sums, float* wt_ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream);

/* Replaces: `c9 = Conv2D(32, (3, 3), relu)(c9)` + `outputs = Conv2D(1, (1, 1), activation='sigmoid')(c9)` T1:911-913 and the loss sums of
 * bce_dice_loss T1:784-799 in ONE launch (fp32, UNET_ALGO_AUTO kernels, 32 output channels, W % 8 == 0, not in deterministic mode:
 * unet_conv3x3_head_supported): y = relu(conv3x3(x) + bias) [n,h,w,32], p = sigmoid(y . w_head + b_head) [n,h,w]; with y_true
 *   loss_sums[4] += (sum bce, sum t p, sum t, sum p)   -- as unet_head_fwd
 *   head_sums[99] += per channel c: sum a y_c | sum t q y_c | sum q y_c  (a = dBCE/dz = p_clipped - t inside the clip range, q = p (1 - p)), then sum a, sum t q, sum q
 * The head's weight gradient is a combination of head_sums once the batch-global Dice sums are known (unet_head_dy), so the backward needs no pass
 * over y.  ReLU sign bits of y can be requested as for unet_conv3x3_fwd (unet_request_relu_bits).  w_ws: unet_conv3x3_w_ws_floats(cin, 32) floats.
 * y may be NULL: the 32-channel tensor is then not stored (a backward through unet_head_dzm reads p, the sums and the sign bits only; inference reads p). */
int32_t unet_conv3x3_head_supported(unet_ctx*, int32_t algo, int32_t w, int32_t cin, int32_t cout);
int32_t unet_conv3x3_head_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, const float* w_head, const float* b_head, float* p,
                              const float* y_true, double* loss_sums, double* head_sums, int32_t n, int32_t h, int32_t wd, int32_t cin, float* w_ws, void* stream);
/* backward of that head: dy[n,h,w,32] = dz w_head [y > 0] (mask from relu_bits, or from y when relu_bits is NULL), dz as in unet_head_bwd from the
 * GLOBAL loss sums; dw_head[32] / db_head[1] += the combination of head_sums. */
int32_t unet_head_dy(unet_ctx*, const float* p, const float* y_true, const double* loss_sums, double count, const double* head_sums, const float* w_head,
                     const void* relu_bits, const float* y, float* dy, float* dw_head, float* db_head, int32_t n, int32_t h, int32_t wd, void* stream);
/* The same backward without the fp32 tensor dy: it has ONE fp32 degree of freedom and 32 mask bits per pixel, so unet_head_dzm writes the stream
 * dzm[n,h,w] = {float dz, uint32 mask (bit c = y_c > 0)} (8 bytes per pixel instead of 128; relu_bits required; dw_head / db_head += as unet_head_dy) and the two
 * gradients of the conv in front of the head (`Conv2D(32, (3, 3), relu)`, T1:911) expand it while they stage it:
 *   unet_conv3x3_bwd_data_dzm     dx[n,h,w,cin] = conv3x3_bwd_data(dy, w) with w_head folded into the weight image; relu_bits_in (or NULL) = the ReLU bits of the conv's INPUT
 *                                 (unet_request_relu_bits on the launch that produced it); wt_ws as for unet_conv3x3_bwd_data
 *   unet_conv3x3_bwd_weights_dzm  dw[3][3][cin][32], db[32] = conv3x3_bwd_weights(x, dy) (overwritten), the head's weights applied to the finished columns; ws as for
 *                                 unet_conv3x3_bwd_weights(cin, 32)
 * fp32 UNET_ALGO_AUTO kernels, cin = 32, W % 8 == 0 (unet_head_bwd_stream_supported). */
int32_t unet_head_bwd_stream_supported(unet_ctx*, int32_t algo, int32_t wd, int32_t cin);
int32_t unet_head_dzm(unet_ctx*, const float* p, const float* y_true, const double* loss_sums, double count, const double* head_sums, const void* relu_bits, void* dzm,
                      float* dw_head, float* db_head, int32_t n, int32_t h, int32_t wd, void* stream);
int32_t unet_conv3x3_bwd_data_dzm(unet_ctx*, const void* dzm, const float* w, const float* w_head, const void* relu_bits_in, float* dx, float* wt_ws, int32_t n, int32_t h,
                                  int32_t wd, int32_t cin, void* stream);
int32_t unet_conv3x3_bwd_weights_dzm(unet_ctx*, const float* x, const void* dzm, const float* w_head, float* dw, float* db, void* ws, size_t ws_bytes, int32_t n, int32_t h,
                                     int32_t wd, int32_t cin, void* stream);

/* Replaces: Adam(lr=0.0005) step of model.fit T1:1053,1059 -- Keras-2.3 form:
 *   m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2; p -= lr_t*m/(sqrt(v)+eps), lr_t=lr*sqrt(1-b2^t)/(1-b1^t)
 * (lr_t computed by the caller).  One launch over the flat parameter buffer. */
int32_t unet_adam_keras(unet_ctx*, float* p, const float* g, float* m, float* v, int64_t count,
                        float lr_t, float b1, float b2, float eps, float grad_scale, void* stream);

/* Replaces: sm.metrics.IOUScore/FScore/Precision/Recall(threshold=t) evaluate sweeps
 * T1:1205-1211, 1259-1265, 1313-1319: out double[T*3] += (sum gt*pr, sum pr, sum gt), pr=(p>t). */
int32_t unet_seg_metrics_sweep(unet_ctx*, const float* p, const float* gt, const float* thresholds,
                               int32_t nthr, double* out, int64_t count, void* stream);

/* Replaces: the batch slicing of model.fit (T1:1059-1061; Keras takes `x[batch_ids]` of a shuffled index array per step) for a dataset that was
 * uploaded ONCE: dst[i] = src[idx[i]], whole samples of sample_floats floats (a multiple of 4); idx = int64 sample numbers on the device. */
int32_t unet_gather_samples(unet_ctx*, const float* src, const int64_t* idx, float* dst, int64_t n,
                            int64_t sample_floats, void* stream);

int32_t unet_zero(unet_ctx*, void* ptr, size_t bytes, void* stream);
/* concatenate([...]) of a tensor that feeds SEVERAL concats (U-Net++ nested skips, task1_unet_plus_plus.py:891-923):
 * copy a dense/sliced tensor into a channel slice of a concat buffer; and the backward: dst (+)= sum of <= 4 gradient slices */
int32_t unet_copy_slice(unet_ctx*, const float* src, int32_t lds, float* dst, int32_t ldd, int64_t pixels, int32_t c, void* stream);
int32_t unet_accum_slices(unet_ctx*, const float* const* srcs, const int32_t* lds, int32_t nsrc, float* dst, int32_t ldd,
                          int64_t pixels, int32_t c, int32_t accumulate, void* stream);

/* ---- bf16-storage variants of the ops above (ABI v5) ----------------------------------------------------------------
 * Mixed precision of the same graph: activations and activation gradients are unet_bf16 in HBM (half the traffic), parameters,
 * parameter gradients, BN sums (fp64), the head's probabilities / targets and all arithmetic stay fp32; convolutions run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Same argument meaning as the fp32 functions; `ld*` in ELEMENTS.
 * Channel counts: conv3x3 forward cin % 16 == 0 (or the cin == 1 `first` entry points, whose image stays fp32) and cout % 16 == 0,
 * so a layer that is also differentiated needs both multiples of 16; convT cin, cout % 32 == 0; other shapes return UNET_E_SHAPE.  w_ws: device scratch of unet_conv3x3_w_ws_floats(cin, cout)
 * floats (re-laid-out bf16 weights).  BASELINE.json configs[3], configs[4] name bf16. */
int32_t unet_cast_f32_to_bf16(unet_ctx*, const float* src, unet_bf16* dst, int64_t count, void* stream);   /* count % 4 == 0 */
int32_t unet_cast_bf16_to_f32(unet_ctx*, const unet_bf16* src, float* dst, int64_t count, void* stream);
int32_t unet_conv3x3_fwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int32_t n, int32_t h, int32_t wd,
                              int32_t cin, int32_t cout, int32_t act, float drop_rate, uint64_t drop_seed, void* w_ws, void* stream);
int32_t unet_conv3x3_first_fwd_bf16(unet_ctx*, const float* x, const float* w, const float* bias, unet_bf16* y, int32_t n, int32_t h,
                                    int32_t wd, int32_t cout, int32_t act, float drop_rate, uint64_t drop_seed, void* stream);
int32_t unet_conv3x3_bwd_data_bf16(unet_ctx*, const unet_bf16* dy, const float* w, const unet_bf16* mask_src, int32_t mask_mode, float mask_rate,
                                   uint64_t mask_seed, unet_bf16* dx, void* w_ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                                   void* stream);
size_t unet_conv3x3_bwd_weights_ws_bytes_bf16(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout);
int32_t unet_conv3x3_bwd_weights_bf16(unet_ctx*, const unet_bf16* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                                      int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream);
int32_t unet_conv3x3_first_bwd_weights_bf16(unet_ctx*, const float* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                                            int32_t n, int32_t h, int32_t wd, int32_t cout, void* stream);
int32_t unet_convT2x2_fwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int32_t ldy, int32_t n, int32_t h,
                               int32_t wd, int32_t cin, int32_t cout, void* w_ws, void* stream);
int32_t unet_convT2x2_bwd_data_bf16(unet_ctx*, const unet_bf16* dy, int32_t lddy, const float* w, const unet_bf16* relu_src, unet_bf16* dx,
                                    int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* w_ws, void* stream);
size_t unet_convT2x2_bwd_weights_ws_bytes_bf16(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout);
int32_t unet_convT2x2_bwd_weights_bf16(unet_ctx*, const unet_bf16* x, const unet_bf16* dy, int32_t lddy, float* dw, float* db, void* ws,
                                       size_t ws_bytes, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream);
int32_t unet_bn_stats_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, double* sums, int64_t pixels, int32_t c, void* stream);
int32_t unet_bn_stats_concat_bf16(unet_ctx*, const unet_bf16* x_up, int32_t ldx, const double* src_sums, double src_count, const float* src_gamma,
                                  const float* src_beta, double* sums, int64_t pixels, int32_t c_up, int32_t c_skip, void* stream);
int32_t unet_bn_apply_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, const float* bnp, unet_bf16* y, int32_t ldy, int64_t pixels, int32_t c,
                           void* stream);
int32_t unet_bn_bwd_stats_bf16(unet_ctx*, const unet_bf16* dy, int32_t lddy, const unet_bf16* x, int32_t ldx, const float* bnp, double* sums,
                               int64_t pixels, int32_t c, void* stream);
int32_t unet_bn_bwd_apply_bf16(unet_ctx*, const unet_bf16* dy, int32_t lddy, const unet_bf16* x, int32_t ldx, const float* bnp, const double* sums,
                               double count, int32_t mask_mode, float mask_rate, uint64_t mask_seed, unet_bf16* dx, int32_t lddx,
                               int64_t pixels, int32_t c, void* stream);
int32_t unet_maxpool2x2_dropout_fwd_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, unet_bf16* y, int32_t n, int32_t h, int32_t wd, int32_t c,
                                         float rate, uint64_t seed, void* stream);
int32_t unet_maxpool2x2_dropout_bwd_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, const unet_bf16* dy, unet_bf16* dx, int32_t lddx, int32_t n,
                                         int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, int32_t accumulate, void* stream);
int32_t unet_bn_apply_maxpool_dropout_fwd_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, const float* bnp, unet_bf16* y, int32_t ldy,
                                               unet_bf16* pooled, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed,
                                               void* stream);
int32_t unet_maxpool2x2_dropout_bwd_sums_bf16(unet_ctx*, const unet_bf16* pooled, const unet_bf16* dy_pooled, const float* gamma, const float* beta,
                                              double* sums, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream);
int32_t unet_bn_maxpool_bwd_apply_bf16(unet_ctx*, const unet_bf16* x, int32_t ldx, const float* bnp, const double* sums, double count,
                                       const unet_bf16* g_skip, int32_t ldg, const unet_bf16* dy_pooled, unet_bf16* dx, int32_t lddx, int32_t n,
                                       int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream);
int32_t unet_maxpool2x2_dropout_bwd_bnstats_bf16(unet_ctx*, const unet_bf16* y, int32_t ldy, const unet_bf16* dy, unet_bf16* dx, int32_t lddx,
                                                 const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd,
                                                 int32_t c, float rate, uint64_t seed, void* stream);
int32_t unet_head_fwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, float* p, const float* y_true, double* loss_sums,
                           int64_t pixels, int32_t cin, void* stream);
int32_t unet_head_bwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* p, const float* y_true, const double* loss_sums,
                           double count, unet_bf16* dx, float* dw, float* db, int64_t pixels, int32_t cin, int32_t relu_mask, void* stream);
/* dense tail: the flattened activations x (and their gradient dx) are bf16, the 32 hidden units, dy and the weights stay fp32 */
int32_t unet_dense_fwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, float* y, int32_t batch, int32_t k, int32_t n, int32_t act,
                            float drop_rate, uint64_t drop_seed, void* ws, size_t ws_bytes, void* stream);
int32_t unet_dense_bwd_bf16(unet_ctx*, const unet_bf16* x, const float* w, const float* dy, unet_bf16* dx, float* dw, int32_t batch, int32_t k, int32_t n,
                            void* stream);
int32_t unet_copy_slice_bf16(unet_ctx*, const unet_bf16* src, int32_t lds, unet_bf16* dst, int32_t ldd, int64_t pixels, int32_t c, void* stream);
int32_t unet_accum_slices_bf16(unet_ctx*, const unet_bf16* const* srcs, const int32_t* lds, int32_t nsrc, unet_bf16* dst, int32_t ldd,
                               int64_t pixels, int32_t c, int32_t accumulate, void* stream);

/* ---- dense tail of the slice classifier (task2_covid19_classifcation.py:770-776, `T2`) ------------------------------
 * Replaces: Flatten -> Dense(32, relu) -> Dropout(0.4) T2:772-775.  y[b,:] = dropout(act(x[b,:] W + bias)); x [batch,k] row-major
 * (the NHWC pooled tensor IS Keras' channels_last flatten order), W [k,n] (Keras Dense kernel), n a power of two in 4..32,
 * k %% 4 == 0.  Split-K partials in `ws` (unet_dense_ws_bytes), fixed-order reduction -> deterministic. */
size_t unet_dense_ws_bytes(int32_t batch, int32_t k, int32_t n);
int32_t unet_dense_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t k, int32_t n,
                       int32_t act, float drop_rate, uint64_t drop_seed, void* ws, size_t ws_bytes, void* stream);
/* dx[b,:] = dy[b,:] W^T (dx may be NULL), dw = x^T dy.  dy must already carry the derivative of the activation/dropout
 * (unet_cls_head_bwd writes it that way). */
int32_t unet_dense_bwd(unet_ctx*, const float* x, const float* w, const float* dy, float* dx, float* dw, int32_t batch, int32_t k,
                       int32_t n, void* stream);
/* Replaces: Dense(1, sigmoid) T2:776 fused with loss='binary_crossentropy' (T2:829, optional class weights T2:801-803, 835) and the
 * sums of the f1 metric T2:688-703.  p[b] = sigmoid(h[b,:].w + bias).  If y_true != NULL, sums (double[4], accumulated) +=
 * (sum cw(t)*bce, sum round(t*p), sum round(t), sum round(p)). */
int32_t unet_cls_head_fwd(unet_ctx*, const float* h, const float* w, const float* bias, float* p, const float* y_true, float class_w0,
                          float class_w1, double* sums, int32_t batch, int32_t n, void* stream);
/* out float[2] = (loss = sums[0]/count, f1) from (globally reduced) sums; count = GLOBAL batch size */
int32_t unet_cls_loss_finalize(unet_ctx*, const double* sums, double count, float* out, void* stream);
/* backward of loss + sigmoid + Dense(n->1) + the Dropout/ReLU of the hidden layer h = dropout(relu(a)):
 * dw[n], db[1]; dh[batch,n] = dL/da (ready for unet_dense_bwd); dbias_prev[n] = sum_b dh[b,:] (bias gradient of the hidden Dense). */
int32_t unet_cls_head_bwd(unet_ctx*, const float* h, const float* w, const float* p, const float* y_true, float class_w0, float class_w1,
                          double count, float drop_rate, float* dh, float* dw, float* db, float* dbias_prev, int32_t batch, int32_t n,
                          void* stream);

/* ---- image steps in front of the path: byte / integer work, bit-exact against oracle/preprocess_oracle.py ----
 * Replaces: `img = (img - xmin)/(xmax - xmin)` T1:336-337 + `np.uint8(test_img*255)` T1:165-166 (float64 arithmetic, truncation);
 *           `cv2.createCLAHE(clipLimit=3.0, tileGridSize=(8,8)).apply(img)` T1:168-169 (OpenCV clahe.cpp restated; cv2 is not in
 *           this image: parity unpinned); `cts/255` T1:520.  Images are dense [n][h][w] (single channel). */
size_t unet_pre_minmax_ws_bytes(int32_t n);
int32_t unet_pre_minmax_to_u8(unet_ctx*, const float* img, uint8_t* out, int32_t n, int64_t pixels_per_image, void* ws, size_t ws_bytes, void* stream);
int32_t unet_pre_unit_to_u8(unet_ctx*, const float* img, uint8_t* out, int64_t count, void* stream);
int32_t unet_pre_u8_to_unit(unet_ctx*, const uint8_t* src, float* dst, int64_t count, void* stream);
size_t unet_pre_clahe_ws_bytes(int32_t n, int32_t tiles_x, int32_t tiles_y);
int32_t unet_pre_clahe_u8(unet_ctx*, const uint8_t* src, uint8_t* dst, int32_t n, int32_t h, int32_t w, float clip_limit, int32_t tiles_x,
                          int32_t tiles_y, void* ws, size_t ws_bytes, void* stream);
/* cv2.resize on uint8 crops (OpenCV resize.cpp, 8-bit path, restated: parity unpinned).  Replaces
 *   `cv2.resize(img[y:y+h, x:x+w], dsize=(125,250), interpolation=cv2.INTER_AREA)`  T1:235-238, 354-357, 364-367  (interp = 3)
 *   `cv2.resize(cts[i], dsize=(new_dim,new_dim), interpolation=cv2.INTER_LINEAR)`    T1:486-488                     (interp = 1)
 * src: dense [n][sh][sw]; rects: HOST array [n][4] = (x, y, w, h) per image in cv2.boundingRect order, or NULL for the whole image;
 * image i is written to the dh x dw window at column dst_x0 of dst[i] (rows of dst_ld bytes), so the two lung crops of a slice
 * land side by side without a concatenate (T1:358).  Interpolation values are cv2's (INTER_LINEAR = 1, INTER_AREA = 3). */
enum { UNET_RESIZE_LINEAR = 1, UNET_RESIZE_AREA = 3 };
int32_t unet_pre_resize_u8(unet_ctx*, const uint8_t* src, int32_t n, int32_t sh, int32_t sw, const int32_t* rects, uint8_t* dst, int32_t dh,
                           int32_t dw, int32_t dst_ld, int32_t dst_x0, int32_t interp, void* stream);
/* The contour search of `cropper` (T1:219-233, T3:221-235) on the HOST: `cv2.findContours(img, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)`, then
 * `cv2.contourArea(c)` and `cv2.boundingRect(c)` of every contour, for n uint8 slices (non-zero = foreground) in HOST memory [n][h][w].  Border following
 * (Suzuki-Abe, what OpenCV implements; restated: parity unpinned) is a serial walk, so it runs on CPU threads (`threads` <= 0: all cores, one slice per
 * thread at a time).  Per slice i: counts[i] = number of contours; the first min(counts[i], max_contours) areas / rects (x, y, w, h) are written in cv2's
 * output order (hierarchy pre-order, siblings newest first).  The caller does the reference's `np.argsort(areas)` and picks the two largest.
 * ctx may be NULL (no device is touched). */
int32_t unet_pre_contours_u8(unet_ctx*, const uint8_t* host_imgs, int32_t n, int32_t h, int32_t w, int32_t max_contours, double* areas, int32_t* rects,
                             int32_t* counts, int32_t threads);

/* ------------------------------------------------------------------------------------
 * Model level.  Replaces the Keras Model built at T1:853-916 and driven by
 * compile/fit/evaluate/predict (T1:1053-1061, 1101, 1137).  A model is a fixed-shape plan:
 * three op programs (training forward, backward, inference forward) over caller-owned
 * flat buffers.  Programs can be run in [begin,end) slices so a data-parallel host can put
 * collectives between ops (sync points) and overlap gradient all-reduce with backward.
 * ---------------------------------------------------------------------------------- */
enum { UNET_PROG_FWD_TRAIN = 0, UNET_PROG_BWD = 1, UNET_PROG_FWD_INFER = 2 };

typedef struct unet_sync_point {
  int32_t after_op;   /* run ops [.., after_op] then reduce */
  int32_t kind;       /* 0 = bn fwd sums, 1 = loss sums, 2 = bn bwd sums, 3 = grad bucket ready */
  void* ptr;          /* device pointer of the doubles (kinds 0-2) / floats (kind 3) to SUM-reduce */
  int64_t count;      /* number of elements */
  int32_t use_op;     /* kinds 0-2: the first op that READS the reduced values (> after_op).  use_op > after_op + 1 means the ops in between do not depend on
                         the reduction: the host may run it on a side stream beside them and make the compute stream wait just before use_op (the backward
                         programs place an independent weight gradient there).  Kind 3: the number of ops of the program (the optimizer is the reader). */
  int32_t reserved;
} unet_sync_point;

/* The SUM all-reduce of a sync point of kinds 0-2, device side (data parallelism is new: the reference is single-process, T1:1053-1061 runs one
 * model.fit).  Every rank of ONE node owns a receive area in fine-grained HBM that its peers map through HIP IPC; one kernel on `stream` pushes the rank's
 * doubles into every peer's area (xGMI is point to point: the W - 1 copies travel on W - 1 links at once), polls its own area until all W contributions of
 * this call have landed and adds them in rank order -- every rank ends with the same bits.  No host proxy, no ring: one fabric latency per reduction.
 *   unet_comm_create   allocates the area and returns its 64-byte IPC handle in handle_out; the host exchanges the handles of all ranks by any means it has
 *                      (torch.distributed.all_gather_object in engine.py) and passes them, in rank order, to
 *   unet_comm_connect  (handles = world x UNET_COMM_HANDLE_BYTES bytes; the own entry is ignored).
 *   unet_comm_allreduce_f64  buf[0..count) <- sum over ranks, in place (one launch per UNET_COMM_MAX_DOUBLES).  Every rank must issue the same calls in the same order.
 *                      A contribution that does not arrive within the timeout (default 60 s) ends the kernel and latches the communicator's error word:
 *   unet_comm_status   *err_out = 0, or 1 + the rank that was missing (sticky; the reduced values of that call are not valid).  Synchronises `stream`.
 * Gradient buckets (kind 3) are bandwidth-bound and stay with RCCL.  A rank may destroy its communicator once its own last all-reduce has completed (by then every
 * peer has written all it will ever write into this rank's area). */
#define UNET_COMM_HANDLE_BYTES 64
#define UNET_COMM_MAX_WORLD 8
#define UNET_COMM_MAX_DOUBLES 2048
typedef struct unet_comm unet_comm;
int32_t unet_comm_create(unet_ctx*, int32_t rank, int32_t world, unet_comm** out, unsigned char* handle_out);
int32_t unet_comm_connect(unet_comm*, const unsigned char* handles);
int32_t unet_comm_set_timeout_ms(unet_comm*, int32_t ms);
int32_t unet_comm_allreduce_f64(unet_comm*, double* buf, int32_t count, void* stream);
int32_t unet_comm_status(unet_comm*, int32_t* err_out, void* stream);
void unet_comm_destroy(unet_comm*);

/* arch: UNET_ARCH_UNET (T1:853-916), UNET_ARCH_UNETPP (task1_unet_plus_plus.py:858-950) or UNET_ARCH_CLASSIFIER (the Sequential
 * CNN of task2_covid19_classifcation.py:747-776: y_true / p_out are [n] floats, loss_ptr = (binary cross-entropy, f1)) */
enum { UNET_ARCH_UNET = 0, UNET_ARCH_UNETPP = 1, UNET_ARCH_CLASSIFIER = 2 };
/* dtype: UNET_DTYPE_F32, or UNET_DTYPE_BF16 = activations / activation gradients stored as bf16 inside the workspace (the image x,
 * the targets, the probabilities p_out, parameters, gradients and optimizer state stay fp32; in_ch must be 1) */
int32_t unet_model_create(unet_ctx*, int32_t arch, int32_t in_ch, int32_t n, int32_t h, int32_t w,
                          int32_t world_size, int32_t conv_algo, int32_t dtype, unet_model** out);
int32_t unet_model_dtype(const unet_model*);
void unet_model_destroy(unet_model*);
int64_t unet_model_param_count(const unet_model*);   /* trainable floats (7,762,401 for in_ch=1) */
int64_t unet_model_state_count(const unet_model*);   /* BN moving mean/var floats (2,880) */
size_t unet_model_workspace_bytes(const unet_model*, int32_t training);
/* offsets (in floats) of a named tensor inside the flat param / state buffers; name as in
 * Keras order: "c1a/kernel", "bn1/gamma", "bn1/mean", "u6/kernel", "out/bias", ... */
int32_t unet_model_tensor_info(const unet_model*, const char* name, int32_t* is_state,
                               int64_t* offset, int64_t* count);
int32_t unet_model_bind(unet_model*, float* params, float* grads, float* adam_m, float* adam_v,
                        float* bn_state, void* workspace, size_t workspace_bytes);
int32_t unet_model_set_io(unet_model*, const float* x, const float* y_true, float* p_out);
/* loss_out: device float[2] of the caller's that the next forward programs ALSO write (loss, metric) to -- a training loop that collects one pair per step
 * (model.fit reads them once per epoch, T1:1059) hands in a fresh slot per step instead of copying unet_model_loss_ptr behind every step; NULL = none (ABI v15) */
int32_t unet_model_set_loss_out(unet_model*, float* loss_out);
int32_t unet_model_set_dropout(unet_model*, float rate, uint64_t seed);
/* classifier only: weights of class 0 / class 1 in the loss (Keras class_weight, T2:835); default 1, 1 */
int32_t unet_model_set_class_weights(unet_model*, float w0, float w1);
int32_t unet_model_num_ops(const unet_model*, int32_t prog);
int32_t unet_model_sync_points(const unet_model*, int32_t prog, unet_sync_point* out, int32_t cap);
int32_t unet_model_run(unet_model*, int32_t prog, int32_t begin, int32_t end, void* stream);
/* device pointer to float[2] = (loss, dice_coeff) of the last forward with y_true bound */
const float* unet_model_loss_ptr(const unet_model*);
/* intermediate activation / gradient taps for tests ("c1a","bn1","p1","u6","cat6",...) */
int32_t unet_model_tap(const unet_model*, const char* name, int32_t grad, const void** ptr,   /* element size: unet_model_tap_elem_bytes */
                       int32_t* ld, int32_t* n, int32_t* h, int32_t* w, int32_t* c);
int32_t unet_model_tap_elem_bytes(const unet_model*, const char* name, int32_t grad);   /* 4 (float) or 2 (unet_bf16) */
/* profiling: per-op name and accumulated milliseconds since last reset (profiling on) */
int32_t unet_model_op_info(const unet_model*, int32_t prog, int32_t op, const char** name,
                           double* flops, double* bytes, double* ms, int64_t* calls);
int32_t unet_model_reset_timers(unet_model*);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
