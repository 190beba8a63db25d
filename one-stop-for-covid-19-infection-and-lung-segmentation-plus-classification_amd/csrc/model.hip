// Context, op-level convolution entry points and the model-level op programs of the gfx950
// U-Net engine.  The graph mirrors the Keras model of the reference,
// /root/reference/Scripts/task1_preprocessing_plus_unet_with_comments.py:853-916
// (== task3_lung_segmentation_unet.py:850-913): 4 encoder blocks Conv-Conv-BN-[skip]-Pool-Dropout,
// bottleneck Conv-Conv, 4 decoder blocks ConvT-concat([up,skip])-BN-Conv-Conv, 1x1 sigmoid head.
//
// HBM layout: every activation is a dense NHWC fp32 tensor carved from ONE caller-owned
// workspace; the skip concatenation is zero-copy: encoder BN k writes channels [C,2C) and the
// decoder ConvT writes channels [0,C) of the same [N,S,S,2C] "cat" buffer (ld = 2C).
// Parameters / gradients / Adam moments are flat buffers in Keras get_weights() order, so
// Adam is one launch and gradient buckets for all-reduce are contiguous ranges.
#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "common.h"

// =========================================================================================
// context
// =========================================================================================
extern "C" {

int32_t unet_abi_version(void) { return UNET_ABI_VERSION; }

int32_t unet_ctx_create(int32_t device_id, unet_ctx** out) {
  if (!out) return UNET_E_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return UNET_E_NODEV;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return UNET_E_NODEV;
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return UNET_E_NODEV;  // gfx950-only code objects
  unet_ctx* c = new unet_ctx();
  c->device = device_id;
  c->num_cu = prop.multiProcessorCount;
  const size_t sb = sizeof(double) * UNET_BN_SLOTS_DET * UNET_BN_SLOT_DOUBLES;
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (hipSetDevice(device_id) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&c->bn_slots), sb) != hipSuccess ||
      hipMemset(c->bn_slots, 0, sb) != hipSuccess) { (void)hipSetDevice(prev); delete c; return UNET_E_HIP; }
  c->convt_img_bytes = (size_t)8 << 20;             // ConvT weight images up to cin * cout = 512 K (u6 of the U-Net: 128 K)
  if (hipMalloc(&c->convt_img, c->convt_img_bytes) != hipSuccess) { c->convt_img = nullptr; c->convt_img_bytes = 0; }
  c->splitk_ws_bytes = (size_t)32 << 20;            // K-sliced launches have < 384 workgroups of <= 8 x 32 x 32 outputs: 4 slices x 12 MB at most
  if (hipMalloc(&c->splitk_ws, c->splitk_ws_bytes) != hipSuccess) { c->splitk_ws = nullptr; c->splitk_ws_bytes = 0; }
  (void)hipSetDevice(prev);
  *out = c;
  return UNET_OK;
}

void unet_ctx_destroy(unet_ctx* ctx) {
  if (ctx && ctx->bn_slots) (void)hipFree(ctx->bn_slots);
  if (ctx && ctx->convt_img) (void)hipFree(ctx->convt_img);
  if (ctx && ctx->splitk_ws) (void)hipFree(ctx->splitk_ws);
  delete ctx;
}
const char* unet_last_error(const unet_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int32_t unet_ctx_set_profiling(unet_ctx* ctx, int32_t on) { if (!ctx) return UNET_E_ARG; ctx->profiling = on; return UNET_OK; }

static int* ctx_option(unet_ctx* ctx, int32_t option) {
  switch (option) {
    case UNET_OPT_RELU_BITS: return &ctx->opt_relu_bits;
    case UNET_OPT_BN_FOLD: return &ctx->opt_bn_fold;
    case UNET_OPT_ENC_BN_FUSED: return &ctx->opt_enc_bn_fused;
    case UNET_OPT_BN_CONCAT_ANALYTIC: return &ctx->opt_bn_concat_analytic;
    case UNET_OPT_BN_FUSE_STATS: return &ctx->opt_bn_fuse_stats;
    case UNET_OPT_DETERMINISTIC: return &ctx->opt_deterministic;
    case UNET_OPT_HEAD_FUSED: return &ctx->opt_head_fused;
    case UNET_OPT_SKIP_RAW: return &ctx->opt_skip_raw;
    case UNET_OPT_POOL_SUMS_FUSED: return &ctx->opt_pool_sums_fused;
    case UNET_OPT_HEAD_BWD_FUSED: return &ctx->opt_head_bwd_fused;
    case UNET_OPT_CONV_PP: return &ctx->opt_conv_pp;
    default: return nullptr;
  }
}
int32_t unet_ctx_set_option(unet_ctx* ctx, int32_t option, int32_t value) {
  if (!ctx) return UNET_E_ARG;
  int* p = ctx_option(ctx, option);
  const int hi = option == UNET_OPT_BN_FOLD ? 3 : option == UNET_OPT_CONV_PP ? 2 : 1;          // (CONV_PP 2: also the launches too small to fill the persistent grid -- tests)
  if (!p || value < 0 || value > hi) UNET_FAIL(ctx, UNET_E_ARG, "ctx_set_option: option %d value %d", option, value);
  *p = value;
  return UNET_OK;
}
int32_t unet_ctx_get_option(unet_ctx* ctx, int32_t option) {
  if (!ctx) return UNET_E_ARG;
  const int* p = ctx_option(ctx, option);
  if (!p) UNET_FAIL(ctx, UNET_E_ARG, "ctx_get_option: unknown option %d", option);
  return *p;
}

}  // extern "C"

// =========================================================================================
// convolution dispatch (shared by the op-level ABI and the model programs)
// =========================================================================================
// Kernel families: UNET_ALGO_AUTO = the fp16-split h2 kernels wherever the channel counts allow (K, M multiples of 16), else the strict family;
// UNET_ALGO_MFMA = strict fp32: v_mfma_f32_32x32x2_f32 kernels (exact fp32 multiply-add), VALU kernels for the shapes those do not take (Cin = 1, odd
// channel counts); UNET_ALGO_NAIVE = VALU kernels only (on-device cross-check).
static bool use_mfma(int algo, int cin, int cout) {
  if (algo == UNET_ALGO_NAIVE) return false;
  return mfma_conv3x3_supported(cin, cout);
}
// the launch runs on the h2 kernels (and therefore consumes a prepared split weight image; `uws` = scratch for it, may be null -> strict family)
static bool use_h2(int algo, int cin, int cout, const void* uws) { return uws && h2_conv3x3_selected(algo, cin, cout); }

// `w` are Keras-layout weights [3][3][cin][cout] when flip == 0; for the data gradient (flip == 1) the caller passes the FORWARD
// weights [3][3][cout][cin] of the layer and the roles of cin/cout below are already swapped (cin = channels of dy).
static int32_t conv3x3_fwd_dispatch(unet_ctx* ctx, const float* x, const float* w, const float* bias, const float* mask, int mask_mode,
                                    float* y, int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, int algo,
                                    hipStream_t s, float* uws = nullptr, int flip = 0, const float* prepared = nullptr, int ldy = 0) {
  if (ldy && ldy != cout && !(use_h2(algo, cin, cout, uws) && prepared)) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3: a strided output (ldy %d) exists on the h2 kernels with a prepared image only", ldy);
  if (use_h2(algo, cin, cout, uws)) {
    if (prepared) return k_conv3x3_h2_fwd(ctx, x, prepared, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s, 1 << 30, ldy);     // image built by the program's batch launch
    int32_t r = flip ? k_h2_weights(ctx, w, uws, cout, cin, 1, s) : k_h2_weights(ctx, w, uws, cin, cout, 0, s);
    if (r) return r;
    return k_conv3x3_h2_fwd(ctx, x, uws, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
  }
  if (mask_mode >= MASK_BIAS_TAB) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3: the folded-BatchNorm / bit-mask epilogues exist on the h2 kernels only (cin=%d cout=%d algo=%d)", cin, cout, algo);
  if (flip) {                                   // direct algorithms want the flipped/transposed copy
    if (!uws) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 data gradient: needs the weight scratch");
    int32_t r = k_flip_transpose_w3x3(ctx, w, uws, cout, cin, s);
    if (r) return r;
    w = uws;
  }
  if (use_mfma(algo, cin, cout)) return k_conv3x3_mfma_fwd(ctx, x, w, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
  if (cin == 1 && !mask && algo != UNET_ALGO_NAIVE && (cout % 4) == 0 && 256 % (cout / 4) == 0) {
    if (ctx->signs_req && act == ACT_RELU && rate == 0.0f && c1_relu_bits_supported(wd, cout)) {          // (armed: unet_request_relu_bits)
      unsigned long long* q = ctx->signs_req; ctx->signs_req = nullptr;
      int32_t r = k_conv3x3_c1_fwd_bits(ctx, x, w, bias, y, q, n, h, wd, cout, s);
      if (!r) ctx->signs_done = q;
      return r;
    }
    return k_conv3x3_c1_fwd(ctx, x, w, bias, y, n, h, wd, cout, act, rate, seed, s);
  }
  return k_conv3x3_naive_fwd(ctx, x, w, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
}

static int32_t conv3x3_wgrad_dispatch(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws,
                                      size_t ws_bytes, int n, int h, int wd, int cin, int cout, int algo, hipStream_t s) {
  if (algo != UNET_ALGO_NAIVE && mfma_wgrad_supported(cin, cout) && ws && ws_bytes >= mfma_wgrad_ws_bytes(n, h, wd, cin, cout)) {
    if (h2_wgrad_c16_selected(algo, wd, cin, cout) && ws_bytes >= h2_wgrad_c16_ws_bytes(n, h, wd))
      return k_conv3x3_h2_wgrad_c16(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, s);      // pixel pairs as 32 channels: half of the MFMA tile useful instead of a quarter
    if (h2_wgrad_selected(algo, cin, cout) && ws_bytes >= h2_wgrad_ws_bytes(n, h, wd, cin, cout))
      return k_conv3x3_h2_wgrad(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);      // three fp16 MFMA products of the block-scaled two-term split
    return k_conv3x3_mfma_wgrad(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);
  }
  if (cin == 1 && algo != UNET_ALGO_NAIVE && (cout % 4) == 0 && 256 % (cout / 4) == 0 && cout <= 256 && ws && ws_bytes >= c1_wgrad_ws_bytes(cout))
    return k_conv3x3_c1_wgrad(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cout, s);
  return k_conv3x3_naive_wgrad(ctx, x, dy, dw, db, n, h, wd, cin, cout, s);
}

extern "C" {

// Conv2D / Conv2DTranspose -> BatchNormalization in training mode (T1:860-861, 886-888): see common.h (unet_ctx::stats_req_c)
int32_t unet_request_bn_stats(unet_ctx* ctx, int32_t c) {
  if (!ctx || c < 0) UNET_FAIL(ctx, UNET_E_ARG, "request_bn_stats: bad args");
  ctx->stats_req_c = ctx->opt_bn_fuse_stats ? c : 0;          // (deterministic mode: the h2 kernels leave exact window sums -- common.h xsum_add --, the bf16 ones decline)
  return UNET_OK;
}

// ReLU masks as one bit per element (MASK_RELU_BITS, common.h): which (forward conv, data gradient) pairs can use them, how big the bit tensor is, arming
int32_t unet_relu_bits_supported(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  if (cin < 1 || cout < 1 || (cout & 31) || (wd & 7) || h < 1) return 0;
  if (cin == 1) return algo == UNET_ALGO_AUTO && c1_relu_bits_supported(wd, cout) ? 1 : 0;          // the first layer's own kernel (T1:859)
  return h2_conv3x3_selected(algo, cin, cout) ? 1 : 0;
}
size_t unet_relu_bits_bytes(int32_t n, int32_t h, int32_t wd, int32_t c) { return n > 0 && h > 0 && wd > 0 && c > 0 ? (size_t)n * h * wd * c / 8 : 0; }
int32_t unet_request_relu_bits(unet_ctx* ctx, void* bits) {
  if (!ctx) return UNET_E_ARG;
  ctx->signs_req = static_cast<unsigned long long*>(bits); ctx->signs_done = nullptr;
  return UNET_OK;
}
int32_t unet_ctx_max_kernel_scratch_bytes(unet_ctx* ctx) { return ctx ? ctx->max_scratch_bytes : -1; }
int32_t unet_allow_k_slices(unet_ctx* ctx) {
  if (!ctx) return UNET_E_ARG;
  ctx->k_slices_ok = 1;
  return UNET_OK;
}

int32_t unet_conv3x3_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int32_t n, int32_t h,
                         int32_t wd, int32_t cin, int32_t cout, int32_t act, float drop_rate, uint64_t drop_seed, int32_t algo,
                         float* w_ws, void* stream) {
  if (!ctx || !x || !w || !y || n < 1 || h < 1 || wd < 1 || cin < 1 || cout < 1 || act < 0 || act > 2 || drop_rate < 0 || drop_rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_fwd: bad args");
  const void* armed = ctx->signs_req;
  int32_t r = conv3x3_fwd_dispatch(ctx, x, w, bias, nullptr, MASK_NONE, y, n, h, wd, cin, cout, act, drop_rate, drop_seed, algo, as_stream(stream), w_ws, 0);
  ctx->signs_req = nullptr; ctx->k_slices_ok = 0;
  if (!r && armed && ctx->signs_done != armed) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_fwd: armed with unet_request_relu_bits but this launch cannot write them (unet_relu_bits_supported, act = ReLU, no dropout)");
  return r;
}

// T1:911-913 in one launch (include/unet_hip.h): the last conv3x3 + the 1x1 sigmoid head + loss sums + the sums of the head's weight gradient
int32_t unet_conv3x3_head_supported(unet_ctx* ctx, int32_t algo, int32_t wd, int32_t cin, int32_t cout) { return ctx && h2_conv3x3_head_selected(ctx, algo, wd, cin, cout) ? 1 : 0; }
int32_t unet_conv3x3_head_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, const float* w_head, const float* b_head, float* p, const float* y_true,
                              double* loss_sums, double* head_sums, int32_t n, int32_t h, int32_t wd, int32_t cin, float* w_ws, void* stream) {
  if (!ctx || !x || !w || !bias || !w_head || !b_head || !p || !w_ws || n < 1 || h < 1 || wd < 1 || (y_true && (!loss_sums || !head_sums)))          // (y null: the 32-channel tensor is not stored)
    UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_head_fwd: bad args");
  if (!h2_conv3x3_head_selected(ctx, UNET_ALGO_AUTO, wd, cin, 32)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_head_fwd: not supported here (unet_conv3x3_head_supported)");
  unsigned long long* armed = ctx->signs_req;
  int32_t r = k_h2_weights(ctx, w, w_ws, cin, 32, 0, as_stream(stream));
  if (!r) r = k_conv3x3_h2_head_fwd(ctx, x, w_ws, bias, y, w_head, b_head, p, y_true, n, h, wd, cin, as_stream(stream));
  ctx->signs_req = nullptr;
  if (!r && armed && ctx->signs_done != armed) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_head_fwd: armed with unet_request_relu_bits but the launch did not write them");
  if (!r && y_true) r = k_head_fold(ctx, loss_sums, head_sums, as_stream(stream));
  return r;
}
int32_t unet_head_bwd_stream_supported(unet_ctx* ctx, int32_t algo, int32_t wd, int32_t cin) {
  return ctx && cin == 32 && h2_head_bwd_selected(ctx, algo, wd, cin) && h2_wgrad_selected(algo, cin, 32) ? 1 : 0;
}
int32_t unet_head_dzm(unet_ctx* ctx, const float* p, const float* y_true, const double* loss_sums, double count, const double* head_sums, const void* relu_bits, void* dzm,
                      float* dw_head, float* db_head, int32_t n, int32_t h, int32_t wd, void* stream) {
  if (!ctx) return UNET_E_ARG;
  return k_head_dzm(ctx, p, y_true, loss_sums, count, head_sums, static_cast<const unsigned long long*>(relu_bits), dzm, dw_head, db_head, n, h, wd, as_stream(stream));
}
int32_t unet_conv3x3_bwd_data_dzm(unet_ctx* ctx, const void* dzm, const float* w, const float* w_head, const void* relu_bits_in, float* dx, float* wt_ws, int32_t n, int32_t h,
                                  int32_t wd, int32_t cin, void* stream) {
  if (!ctx || !dzm || !w || !w_head || !dx || !wt_ws || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_data_dzm: bad args");
  if (!unet_head_bwd_stream_supported(ctx, UNET_ALGO_AUTO, wd, cin)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bwd_data_dzm: not supported here (unet_head_bwd_stream_supported)");
  int32_t r = k_h2_weights(ctx, w, wt_ws, cin, 32, 1, as_stream(stream), w_head);          // (the head's weights: a per-contraction-channel factor of the image)
  if (r) return r;
  return k_conv3x3_h2_dgrad_dzm(ctx, dzm, wt_ws, static_cast<const float*>(relu_bits_in), relu_bits_in ? MASK_RELU_BITS : MASK_NONE, dx, n, h, wd, cin, as_stream(stream));
}
int32_t unet_conv3x3_bwd_weights_dzm(unet_ctx* ctx, const float* x, const void* dzm, const float* w_head, float* dw, float* db, void* ws, size_t ws_bytes, int32_t n, int32_t h,
                                     int32_t wd, int32_t cin, void* stream) {
  if (!ctx || !x || !dzm || !w_head || !dw || !db || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_weights_dzm: bad args");
  if (!unet_head_bwd_stream_supported(ctx, UNET_ALGO_AUTO, wd, cin)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bwd_weights_dzm: not supported here (unet_head_bwd_stream_supported)");
  return k_conv3x3_h2_wgrad_dzm(ctx, x, dzm, w_head, dw, db, ws, ws_bytes, n, h, wd, cin, as_stream(stream));
}
int32_t unet_head_dy(unet_ctx* ctx, const float* p, const float* y_true, const double* loss_sums, double count, const double* head_sums, const float* w_head, const void* relu_bits,
                     const float* y, float* dy, float* dw_head, float* db_head, int32_t n, int32_t h, int32_t wd, void* stream) {
  if (!ctx) return UNET_E_ARG;
  return k_head_dy(ctx, p, y_true, loss_sums, count, head_sums, w_head, static_cast<const unsigned long long*>(relu_bits), y, dy, dw_head, db_head, n, h, wd, as_stream(stream));
}

int32_t unet_conv3x3_pick_algo(int32_t algo, int32_t wd, int32_t cin, int32_t cout) {
  (void)wd;
  if (h2_conv3x3_selected(algo, cin, cout)) return UNET_ALGO_AUTO;
  if (algo != UNET_ALGO_NAIVE && mfma_conv3x3_supported(cin, cout)) return UNET_ALGO_MFMA;
  return UNET_ALGO_NAIVE;
}

/* fp32-MFMA-time equivalent of a forward or data-gradient launch of this shape: 1 (strict fp32 / VALU kernels), 3 * 157.3 / 2500 (h2: three fp16 MFMA
 * products per multiply on a pipe 2500 / 157.3 times faster than the fp32 MFMA) */
double unet_conv3x3_exec_ratio(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  (void)h; (void)wd;
  return h2_conv3x3_selected(algo, cin, cout) ? 3.0 * 157.3 / 2500.0 : 1.0;
}

/* the same for the weight-gradient launch (given the workspace unet_conv3x3_bwd_weights_ws_bytes asks for) */
double unet_conv3x3_wgrad_exec_ratio(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  (void)h; (void)wd;
  if (algo == UNET_ALGO_NAIVE || !mfma_wgrad_supported(cin, cout)) return 1.0;
  return h2_wgrad_selected(algo, cin, cout) ? 3.0 * 157.3 / 2500.0 : 1.0;
}

// (channel counts below 32 are padded to one 32-channel block in the h2 weight image, either direction)
size_t unet_conv3x3_w_ws_floats(int32_t cin, int32_t cout) { return cin > 0 && cout > 0 ? (size_t)16 * std::max(cin, 32) * std::max(cout, 32) + 64 : 0; }

int32_t unet_conv3x3_bwd_data(unet_ctx* ctx, const float* dy, const float* w, const float* mask_src, int32_t mask_mode, float mask_rate,
                              uint64_t mask_seed, float* dx, float* wt_ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout,
                              int32_t algo, void* stream) {
  if (!ctx || !dy || !w || !dx || !wt_ws || n < 1 || h < 1 || wd < 1 || cin < 1 || cout < 1 || mask_mode < 0 || (mask_mode > 3 && mask_mode != MASK_RELU_BITS) ||
      (mask_mode != MASK_NONE && !mask_src) || mask_rate < 0 || mask_rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_data: bad args");
  if (mask_mode == MASK_RELU_BITS && !unet_relu_bits_supported(algo, h, wd, cout, cin))
    UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bwd_data: no bit-mask form for h=%d w=%d cin=%d cout=%d algo=%d (unet_relu_bits_supported(algo, h, w, cout, cin))", h, wd, cin, cout, algo);
  // data gradient = 3x3 convolution of dy (cout channels) with the flipped/transposed kernel -> cin channels
  const int32_t r = conv3x3_fwd_dispatch(ctx, dy, w, nullptr, mask_src, mask_mode, dx, n, h, wd, cout, cin, ACT_NONE, mask_rate, mask_seed, algo,
                                         as_stream(stream), wt_ws, 1);
  ctx->k_slices_ok = 0;          // (one-shot: a launch that did not run on the h2 kernels must not leave the arm for a later one)
  return r;
}

// the data gradient behind an encoder tail (MaxPooling2D + Dropout, T1:862-863) with that tail's pooled-path BatchNorm-backward sums in the epilogue (include/unet_hip.h)
int32_t unet_conv3x3_bwd_data_pool_sums_supported(unet_ctx* ctx, int32_t algo, int32_t wd, int32_t cin, int32_t cout) { return ctx && h2_pool_sums_selected(ctx, algo, wd, cout, cin) ? 1 : 0; }
int32_t unet_conv3x3_bwd_data_pool_sums(unet_ctx* ctx, const float* dy, const float* w, const float* pooled, const float* gamma, const float* beta, float rate, float* dx, double* sums,
                                        float* wt_ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream) {
  if (!ctx || !dy || !w || !pooled || !gamma || !beta || !dx || !sums || !wt_ws || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_data_pool_sums: bad args");
  if (!h2_pool_sums_selected(ctx, UNET_ALGO_AUTO, wd, cout, cin)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bwd_data_pool_sums: not supported here (unet_conv3x3_bwd_data_pool_sums_supported)");
  int32_t r = k_h2_weights(ctx, w, wt_ws, cin, cout, 1, as_stream(stream));
  if (r) return r;
  return k_conv3x3_h2_dgrad_pool_sums(ctx, dy, wt_ws, pooled, gamma, beta, rate, dx, sums, n, h, wd, cout, cin, as_stream(stream));
}

/* ---- conv3x3 over a BatchNorm-affine input without the normalised tensor (the decoder blocks BN -> Conv, T1:888-889 ...) ---- */
int32_t unet_conv3x3_bnfold_supported(int32_t algo, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  return (h >= 1 && wd >= 1 && cin >= 1 && cout >= 1 && h2_conv3x3_selected(algo, cin, cout) && (cout % 16) == 0 && wgrad_bn_fold_supported(cout)) ? 1 : 0;
}
size_t unet_conv3x3_bnfold_ws_floats(int32_t n, int32_t cin, int32_t cout) {
  if (n < 1 || cin < 1 || cout < 1) return 0;
  return bn_fold_scratch_floats(cin, cout) + unet_conv3x3_w_ws_floats(cin, cout) + wgrad_bn_fold_scratch_floats(n, cout);
}
int32_t unet_conv3x3_bnfold_fwd(unet_ctx* ctx, const float* x, const float* bnp, const float* w, const float* bias, float* y, int32_t n, int32_t h, int32_t wd,
                                int32_t cin, int32_t cout, int32_t act, int32_t algo, float* ws, void* stream) {
  if (!ctx || !x || !bnp || !w || !y || !ws || n < 1 || act < 0 || act > ACT_RELU) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bnfold_fwd: bad args");
  if (!unet_conv3x3_bnfold_supported(algo, h, wd, cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bnfold_fwd: no folded form for h=%d w=%d cin=%d cout=%d (unet_conv3x3_bnfold_supported)", h, wd, cin, cout);
  hipStream_t s = as_stream(stream);
  float* u = ws + bn_fold_scratch_floats(cin, cout);
  int32_t r = k_bn_fold_prepare(ctx, w, bias, bnp, bnp + cin, cin, cout, ws, s, false);
  if (r) return r;
  r = k_h2_weights(ctx, w, u, cin, cout, 0, s, bnp);          // (the image kernel applies the BatchNorm scale per input channel)
  if (r) return r;
  const float* tab = ws + (size_t)9 * cin * cout;
  return k_conv3x3_h2_fwd(ctx, x, u, tab, tab, MASK_BIAS_TAB, y, n, h, wd, cin, cout, act, 0.0f, 0, s);
}
int32_t unet_conv3x3_bnfold_bwd_weights(unet_ctx* ctx, const float* x, const float* bnp, const float* dy, const float* w, float* dw, float* db, double* bn_bwd_sums, void* gws,
                                        size_t gws_bytes, float* ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo, void* stream) {
  if (!ctx || !x || !bnp || !dy || !dw || !db || !ws || n < 1 || h < 1 || wd < 1 || cin < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bnfold_bwd_weights: bad args");
  if (!wgrad_bn_fold_supported(cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_bnfold_bwd_weights: cout=%d (needs a divisor of 256)", cout);
  hipStream_t s = as_stream(stream);
  int32_t r = conv3x3_wgrad_dispatch(ctx, x, dy, dw, db, gws, gws_bytes, n, h, wd, cin, cout, algo, s);
  if (r) return r;
  if (bn_bwd_sums && !w) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bnfold_bwd_weights: bn_bwd_sums needs the kernel w");
  return k_wgrad_bn_fold_fix(ctx, dy, n, h, wd, cin, cout, bnp, bnp + cin, dw, db, ws + bn_fold_scratch_floats(cin, cout) + unet_conv3x3_w_ws_floats(cin, cout), s,
                             bn_bwd_sums ? w : nullptr, bn_bwd_sums ? bnp + 2 * cin : nullptr, bn_bwd_sums ? bnp + 3 * cin : nullptr, bn_bwd_sums);
}

size_t unet_conv3x3_bwd_weights_ws_bytes(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  if (cin == 1 && (cout % 4) == 0 && 256 % (cout / 4) == 0 && cout <= 256) return c1_wgrad_ws_bytes(cout);
  return mfma_wgrad_supported(cin, cout) ? mfma_wgrad_ws_bytes(n, h, wd, cin, cout) : 0;
}

int32_t unet_conv3x3_bwd_weights(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes,
                                 int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo, void* stream) {
  if (!ctx || !x || !dy || !dw || !db || n < 1 || h < 1 || wd < 1 || cin < 1 || cout < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_weights: bad args");
  return conv3x3_wgrad_dispatch(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, algo, as_stream(stream));
}

int32_t unet_convT2x2_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int32_t ldy, int32_t n,
                          int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo, void* stream) {
  if (!ctx || !x || !w || !y || ldy < cout || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "convT_fwd: bad args");
  if (h2_convT_selected(ctx, algo, cin, cout)) return k_convT_h2_fwd(ctx, x, w, bias, y, ldy, n, h, wd, cin, cout, as_stream(stream));
  if (algo != UNET_ALGO_NAIVE && mfma_convT_supported(cin, cout)) return k_convT_mfma_fwd(ctx, x, w, bias, y, ldy, n, h, wd, cin, cout, as_stream(stream));
  return k_convT_naive_fwd(ctx, x, w, bias, y, ldy, n, h, wd, cin, cout, as_stream(stream));
}

int32_t unet_convT2x2_bwd_data(unet_ctx* ctx, const float* dy, int32_t lddy, const float* w, const float* relu_src, float* dx,
                               int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo, void* stream) {
  if (!ctx || !dy || !w || !dx || lddy < cout) UNET_FAIL(ctx, UNET_E_ARG, "convT_bwd_data: bad args");
  if (h2_convT_selected(ctx, algo, cin, cout)) return k_convT_h2_dgrad(ctx, dy, lddy, w, relu_src, dx, n, h, wd, cin, cout, as_stream(stream));
  if (algo != UNET_ALGO_NAIVE && mfma_convT_supported(cin, cout)) return k_convT_mfma_dgrad(ctx, dy, lddy, w, relu_src, dx, n, h, wd, cin, cout, as_stream(stream));
  return k_convT_naive_dgrad(ctx, dy, lddy, w, relu_src, dx, n, h, wd, cin, cout, as_stream(stream));
}

size_t unet_convT2x2_bwd_weights_ws_bytes(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  return std::max(mfma_convT_supported(cin, cout) ? mfma_convT_wgrad_ws_bytes(n, h, wd, cin, cout) : 0, h2_convT_wgrad_ws_bytes(n, h, wd, cin, cout));
}

int32_t unet_convT2x2_bwd_weights(unet_ctx* ctx, const float* x, const float* dy, int32_t lddy, float* dw, float* db, void* ws,
                                  size_t ws_bytes, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, int32_t algo,
                                  void* stream) {
  if (!ctx || !x || !dy || !dw || !db || lddy < cout) UNET_FAIL(ctx, UNET_E_ARG, "convT_bwd_weights: bad args");
  const bool can = mfma_convT_supported(cin, cout) && ws && ws_bytes >= mfma_convT_wgrad_ws_bytes(n, h, wd, cin, cout);
  if (h2_convT_wgrad_selected(algo, cin, cout) && ws && ws_bytes >= h2_convT_wgrad_ws_bytes(n, h, wd, cin, cout))
    return k_convT_h2_wgrad(ctx, x, dy, lddy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, as_stream(stream));      // three fp16 MFMA products of the block-scaled two-term split
  if (algo != UNET_ALGO_NAIVE && can) return k_convT_mfma_wgrad(ctx, x, dy, lddy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, as_stream(stream));
  return k_convT_naive_wgrad(ctx, x, dy, lddy, dw, db, n, h, wd, cin, cout, as_stream(stream));
}


/* ---- bf16-storage variants (activations / activation gradients bf16, everything else fp32).  No algorithm selector: one MFMA
 * kernel family (v_mfma_f32_32x32x16_bf16); unsupported channel counts fail with UNET_E_SHAPE. */
int32_t unet_conv3x3_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int32_t n, int32_t h, int32_t wd,
                              int32_t cin, int32_t cout, int32_t act, float drop_rate, uint64_t drop_seed, void* w_ws, void* stream) {
  if (!ctx || !x || !w || !y || !w_ws || n < 1 || h < 1 || wd < 1 || act < 0 || act > 2 || drop_rate < 0 || drop_rate >= 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_fwd_bf16: bad args");
  return k_conv3x3_bf16_fwd(ctx, x, w, bias, nullptr, MASK_NONE, y, n, h, wd, cin, cout, act, drop_rate, drop_seed, static_cast<unet_bf16*>(w_ws), 0, as_stream(stream));
}
int32_t unet_conv3x3_first_fwd_bf16(unet_ctx* ctx, const float* x, const float* w, const float* bias, unet_bf16* y, int32_t n, int32_t h, int32_t wd,
                                    int32_t cout, int32_t act, float drop_rate, uint64_t drop_seed, void* stream) {
  if (!ctx || !x || !w || !y || n < 1 || h < 1 || wd < 1 || act < 0 || act > 2 || drop_rate < 0 || drop_rate >= 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_first_fwd_bf16: bad args");
  return k_conv3x3_c1_fwd_bf16(ctx, x, w, bias, y, n, h, wd, cout, act, drop_rate, drop_seed, as_stream(stream));
}
int32_t unet_conv3x3_bwd_data_bf16(unet_ctx* ctx, const unet_bf16* dy, const float* w, const unet_bf16* mask_src, int32_t mask_mode, float mask_rate,
                                   uint64_t mask_seed, unet_bf16* dx, void* w_ws, int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream) {
  if (!ctx || !dy || !w || !dx || !w_ws || n < 1 || h < 1 || wd < 1 || mask_mode < 0 || mask_mode > 3 || (mask_mode != MASK_NONE && !mask_src) || mask_rate < 0 || mask_rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_data_bf16: bad args");
  return k_conv3x3_bf16_fwd(ctx, dy, w, nullptr, mask_src, mask_mode, dx, n, h, wd, cout, cin, ACT_NONE, mask_rate, mask_seed, static_cast<unet_bf16*>(w_ws), 1, as_stream(stream));
}
size_t unet_conv3x3_bwd_weights_ws_bytes_bf16(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout) {
  if (cin == 1) return c1_wgrad_ws_bytes(cout);
  return bf16_wgrad_ws_bytes(n, h, wd, cin, cout);
}
int32_t unet_conv3x3_bwd_weights_bf16(unet_ctx* ctx, const unet_bf16* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int32_t n,
                                      int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream) {
  if (!ctx || !x || !dy || !dw || !db || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_bwd_weights_bf16: bad args");
  return k_conv3x3_bf16_wgrad(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, as_stream(stream));
}
int32_t unet_conv3x3_first_bwd_weights_bf16(unet_ctx* ctx, const float* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int32_t n,
                                            int32_t h, int32_t wd, int32_t cout, void* stream) {
  if (!ctx || !x || !dy || !dw || !db || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_first_bwd_weights_bf16: bad args");
  return k_conv3x3_c1_wgrad_bf16(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cout, as_stream(stream));
}
int32_t unet_convT2x2_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int32_t ldy, int32_t n, int32_t h,
                               int32_t wd, int32_t cin, int32_t cout, void* w_ws, void* stream) {
  if (!ctx || !x || !w || !y || !w_ws || ldy < cout || (ldy & 7) || n < 1 || h < 1 || wd < 1) UNET_FAIL(ctx, UNET_E_ARG, "convT_fwd_bf16: bad args");
  return k_convT_bf16_fwd(ctx, x, w, bias, y, ldy, n, h, wd, cin, cout, static_cast<unet_bf16*>(w_ws), as_stream(stream));
}
int32_t unet_convT2x2_bwd_data_bf16(unet_ctx* ctx, const unet_bf16* dy, int32_t lddy, const float* w, const unet_bf16* relu_src, unet_bf16* dx, int32_t n,
                                    int32_t h, int32_t wd, int32_t cin, int32_t cout, void* w_ws, void* stream) {
  if (!ctx || !dy || !w || !dx || !w_ws || lddy < cout || (lddy & 7)) UNET_FAIL(ctx, UNET_E_ARG, "convT_bwd_data_bf16: bad args");
  return k_convT_bf16_dgrad(ctx, dy, lddy, w, relu_src, dx, n, h, wd, cin, cout, static_cast<unet_bf16*>(w_ws), as_stream(stream));
}
size_t unet_convT2x2_bwd_weights_ws_bytes_bf16(int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout) { return bf16_convT_wgrad_ws_bytes(n, h, wd, cin, cout); }
int32_t unet_convT2x2_bwd_weights_bf16(unet_ctx* ctx, const unet_bf16* x, const unet_bf16* dy, int32_t lddy, float* dw, float* db, void* ws, size_t ws_bytes,
                                       int32_t n, int32_t h, int32_t wd, int32_t cin, int32_t cout, void* stream) {
  if (!ctx || !x || !dy || !dw || !db || lddy < cout || (lddy & 7)) UNET_FAIL(ctx, UNET_E_ARG, "convT_bwd_weights_bf16: bad args");
  return k_convT_bf16_wgrad(ctx, x, dy, lddy, dw, db, ws, ws_bytes, n, h, wd, cin, cout, as_stream(stream));
}

}  // extern "C"

// =========================================================================================
// model
// =========================================================================================
namespace {

struct Layer { std::string name; int kind; int cin, cout; };   // kind 0 conv3, 1 convT, 2 bn, 3 conv1, 4 dense
struct TInfo { int is_state; int64_t off, count; };
struct Buf { size_t off = 0; int ld = 0, n = 0, h = 0, w = 0, c = 0; size_t chan_off = 0; int f32 = 0; };   // offsets in 4-byte units into the workspace; f32: stays fp32 in bf16-storage mode

struct Op {
  std::string name;
  std::function<int32_t(hipStream_t)> run;
  double flops = 0, bytes = 0, ms = 0;
  int64_t calls = 0;
};

}  // namespace

struct unet_model {
  unet_ctx* ctx = nullptr;
  int arch = 0, in_ch = 1, N = 0, H = 0, W = 0, world = 1, algo = 0;
  int dt = UNET_DTYPE_F32;                             // storage type of activations / activation gradients
  std::vector<Layer> layers;
  std::map<std::string, TInfo> tinfo;
  int64_t n_params = 0, n_state = 0;
  // bound buffers
  float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr, *state = nullptr;
  char* ws = nullptr; size_t ws_bytes = 0;
  const float *x = nullptr, *yt = nullptr; float* pout = nullptr; float* loss_out2 = nullptr;
  float drop_rate = 0.0f; uint64_t drop_seed = 0;
  float cw0 = 1.0f, cw1 = 1.0f;                       // classifier: class weights of the loss
  size_t off_dense_ws = 0, dense_ws_bytes = 0;
  // workspace plan (offsets in floats)
  std::map<std::string, Buf> act, grad;
  size_t ws_floats_infer = 0, ws_floats_train = 0;
  size_t off_bn_sums = 0, bn_sums_doubles = 0;       // all fwd BN sums (double), then loss sums (4 doubles)
  size_t off_bn_bsums = 0;                            // all bwd BN sums (double)
  std::map<std::string, size_t> bn_sum_off, bn_bsum_off, bnp_off;   // per-BN offsets (doubles / floats)
  size_t off_loss_sums = 0, off_loss_out = 0, off_wt = 0, off_wgrad_ws = 0; size_t wgrad_ws_bytes = 0;
  size_t off_first_tmp = 0;                // bf16 storage with a multi-channel image (configs[4] as written: 224 x 224 x 3): fp32 staging of the first conv's output / output gradient
  // U-Net fp32, every decoder BatchNorm folded both ways: c<k>b (k = 1..4) IS the skip half of cat<10-k> -- the encoder conv writes there (ldy = 2C), the encoder BatchNorm's
  // output is never stored (pool reads the raw tensor; the decoder fold composes the two BatchNorms: bn_comp_off = [scale'][shift'][pre_s][pre_t] x 2C per decoder level)
  bool skip_raw = false; std::map<std::string, size_t> bn_comp_off; size_t off_tap_tmp = 0;
  std::set<std::string> pool_sums_fused;          // pooled tensors whose backward sums come out of the data-gradient epilogue (MASK_POOL_SUMS)
  bool c9b_virtual = false;               // HEAD_BWD_FUSED with sign bits: the fused head launch does not store c9b's output at all (a tap recomputes it)
  bool head_bwd_fused = false;            // ... and its backward as the {dz, mask} stream the two gradients of c9b expand (HEAD_BWD_FUSED; the stream sits at the start of c9b's gradient buffer)
  size_t off_head_sums = 0; bool head_fused = false;          // U-Net, fp32 h2 kernels: c9b + 1x1 head + loss sums in one launch (kernels_conv_h2.hip, HEAD)
  std::map<std::string, size_t> sign_off;            // U-Net fp32 training: activation name -> its one-bit-per-element ReLU mask (MASK_RELU_BITS), offset in floats
  std::map<std::string, size_t> wprep_f, wprep_b;    // U-Net fp32: per-layer scratch of the prepared weights: the split fp16 image of the h2 kernels (forward / data-gradient form),
                                                      // filled by ONE batched launch at the start of a program
  // U-Net fp32: convs whose input BatchNorm is folded into them: conv name -> scratch of (scaled weights, bias table) / of the
  // weight-gradient correction; folded_bn maps the BatchNorm's activation name to (its input buffer, its channel count): never written by the programs
  std::map<std::string, size_t> fold_off, fold_g_off, fold_c_off;     // fold_c_off: + the BatchNorm backward runs in the conv's data-gradient epilogue (coefficients [3][cin])
  std::map<std::string, std::pair<std::string, int>> folded_bn;
  std::map<std::string, size_t> skip_k1_off;          // U-Net fp32: encoder BatchNorm name -> offset (floats) of the decoder's K1 coefficients of its skip channels: the decoder's
                                                      // data gradient did not read the skip tensor, bn_pool_bwd_apply adds K1 * y (common.h: mask_climit)
  std::vector<Op> prog[3];
  std::vector<unet_sync_point> sync[3];
  struct SyncRef { int after_op, kind; bool in_ws; size_t off_bytes; int64_t count; int use_op = -1; };          // use_op -1: the op right behind after_op
  std::vector<SyncRef> syncref[3];

  float* wsf(size_t off) const { return reinterpret_cast<float*>(ws) + off; }
  double* wsd(size_t off_floats) const { return reinterpret_cast<double*>(reinterpret_cast<float*>(ws) + off_floats); }
  float* P(const std::string& n) const { auto& t = tinfo.at(n); return (t.is_state ? state : params) + t.off; }
  float* G(const std::string& n) const { auto& t = tinfo.at(n); return grads + t.off; }
  const float* A(const std::string& n) const { auto& b = act.at(n); return wsf(b.off + b.chan_off); }
  float* Aw(const std::string& n) const { auto& b = act.at(n); return wsf(b.off + b.chan_off); }
  float* D(const std::string& n) const { auto& b = grad.at(n); return wsf(b.off + b.chan_off); }
  // storage-type-agnostic addresses: buffer offsets are in 4-byte units, channel offsets in elements
  int elem_bytes(const Buf& b) const { return (dt && !b.f32) ? 2 : 4; }
  void* bufptr(const Buf& b) const { return ws + b.off * 4 + b.chan_off * elem_bytes(b); }
  void* Av(const std::string& n) const { return bufptr(act.at(n)); }
  void* Dv(const std::string& n) const { return bufptr(grad.at(n)); }
};

namespace {

const int ENC[4] = {32, 64, 128, 256};

void assign_param_offsets(unet_model* m) {
  int64_t po = 0, so = 0;
  for (auto& l : m->layers) {
    if (l.kind == 2) {
      m->tinfo[l.name + "/gamma"] = {0, po, l.cout}; po += l.cout;
      m->tinfo[l.name + "/beta"] = {0, po, l.cout}; po += l.cout;
      m->tinfo[l.name + "/mean"] = {1, so, l.cout}; so += l.cout;
      m->tinfo[l.name + "/var"] = {1, so, l.cout}; so += l.cout;
    } else {
      int64_t kn = (l.kind == 0 ? 9 : l.kind == 1 ? 4 : 1) * (int64_t)l.cin * l.cout;      // conv1 / dense: cin*cout
      m->tinfo[l.name + "/kernel"] = {0, po, kn}; po += kn;
      m->tinfo[l.name + "/bias"] = {0, po, l.cout}; po += l.cout;
    }
  }
  m->n_params = po; m->n_state = so;
}

void build_layers(unet_model* m) {
  auto& L = m->layers;
  int cprev = m->in_ch;
  for (int k = 1; k <= 4; ++k) {
    int c = ENC[k - 1];
    L.push_back({"c" + std::to_string(k) + "a", 0, cprev, c});
    L.push_back({"c" + std::to_string(k) + "b", 0, c, c});
    L.push_back({"bn" + std::to_string(k), 2, c, c});
    cprev = c;
  }
  L.push_back({"c5a", 0, 256, 512});
  L.push_back({"c5b", 0, 512, 512});
  cprev = 512;
  const int dec[4] = {256, 128, 64, 32};
  for (int k = 6; k <= 9; ++k) {
    int c = dec[k - 6];
    L.push_back({"u" + std::to_string(k), 1, cprev, c});
    L.push_back({"bn" + std::to_string(k), 2, 2 * c, 2 * c});
    L.push_back({"c" + std::to_string(k) + "a", 0, 2 * c, c});
    L.push_back({"c" + std::to_string(k) + "b", 0, c, c});
    cprev = c;
  }
  L.push_back({"out", 3, 32, 1});
  assign_param_offsets(m);
}

struct Carver {
  size_t cur = 0; int dt = 0;          // dt: storage type of the activation buffers carved by mk()
  size_t take(size_t floats) { size_t o = cur; cur += (floats + 63) & ~size_t(63); return o; }
};

Buf mk(Carver& cv, int n, int h, int w, int c) { Buf b; const size_t el = (size_t)n * h * w * c; b.off = cv.take(cv.dt ? (el + 1) / 2 : el); b.ld = c; b.n = n; b.h = h; b.w = w; b.c = c; return b; }
Buf slice(const Buf& b, int c0, int c) { Buf s = b; s.chan_off = c0; s.c = c; return s; }

void plan_scratch(unet_model* m, Carver& cv) {          // BN sums / params, loss scalars (the cross-rank sync buffers)
  size_t nd = 0;
  for (auto& l : m->layers) if (l.kind == 2) { m->bn_sum_off[l.name] = nd; nd += 2 * (size_t)l.cout; }
  m->bn_sums_doubles = nd;
  m->off_bn_sums = cv.take((nd + 4 + UNET_HEAD_SUMS) * 2);          // doubles -> 2 floats each; +4 loss sums, + the sums of a fused head (zeroed with them)
  m->off_loss_sums = m->off_bn_sums + nd * 2;
  m->off_head_sums = m->off_loss_sums + 4 * 2;
  size_t nb = 0;
  for (auto& l : m->layers) if (l.kind == 2) { m->bn_bsum_off[l.name] = nb; nb += 2 * (size_t)l.cout; }
  m->off_bn_bsums = cv.take(nb * 2);
  for (auto& l : m->layers) if (l.kind == 2) m->bnp_off[l.name] = cv.take(4 * (size_t)l.cout);
  m->off_loss_out = cv.take(64);
  if (m->dt && m->in_ch != 1) m->off_first_tmp = cv.take((size_t)m->N * m->H * m->W * 32);          // (first conv: 32 channels in the U-Nets, 16 in the classifier)
}

void plan_workspace(unet_model* m) {
  Carver cv; cv.dt = m->dt;
  const int N = m->N;
  plan_scratch(m, cv);
  m->head_fused = !m->dt && h2_conv3x3_head_selected(m->ctx, m->algo, m->W, 32, 32);
  m->skip_raw = !m->dt && m->ctx->opt_skip_raw && m->ctx->opt_bn_fold >= 2 && m->ctx->opt_enc_bn_fused && m->ctx->opt_bn_concat_analytic;
  for (int k = 1; k <= 4 && m->skip_raw; ++k) {
    const int c = ENC[k - 1];          // (exactly the conditions under which plan_workspace folds the decoder BatchNorm forward and backward, and the encoder conv runs on the h2 kernels)
    m->skip_raw = wgrad_bn_fold_supported(c) && h2_conv3x3_selected(m->algo, 2 * c, c) && h2_conv3x3_selected(m->algo, c, 2 * c) && h2_conv3x3_selected(m->algo, c, c);
  }
  // --- activations ---
  int S = m->H, T = m->W;
  for (int k = 1; k <= 4; ++k) {
    int c = ENC[k - 1];
    std::string ks = std::to_string(k), dk = std::to_string(10 - k);
    m->act["c" + ks + "a"] = mk(cv, N, S, T, c);
    if (!m->skip_raw) m->act["c" + ks + "b"] = mk(cv, N, S, T, c);
    Buf cat = mk(cv, N, S, T, 2 * c);
    m->act["cat" + dk] = cat;
    m->act["u" + dk] = slice(cat, 0, c);
    if (m->skip_raw) {
      m->act["c" + ks + "b"] = slice(cat, c, c);          // the raw conv output lives in the concat
      if (k == 1) m->off_tap_tmp = cv.take((size_t)N * S * T * c);          // where a tap of bn<k> (tests, intermediate_output) is materialised on demand
      Buf tb; tb.off = m->off_tap_tmp; tb.ld = c; tb.n = N; tb.h = S; tb.w = T; tb.c = c;
      m->act["bn" + ks] = tb;
      m->bn_comp_off["bn" + dk] = cv.take((size_t)4 * 2 * c);
    } else
    m->act["bn" + ks] = slice(cat, c, c);
    m->act["p" + ks] = mk(cv, N, S / 2, T / 2, c);
    S /= 2; T /= 2;
  }
  m->act["c5a"] = mk(cv, N, S, T, 512);
  m->act["c5b"] = mk(cv, N, S, T, 512);
  const int dec[4] = {256, 128, 64, 32};
  for (int k = 6; k <= 9; ++k) {
    int c = dec[k - 6]; S *= 2; T *= 2;
    std::string ks = std::to_string(k);
    m->act["bn" + ks] = mk(cv, N, S, T, 2 * c);
    m->act["c" + ks + "a"] = mk(cv, N, S, T, c);
    m->act["c" + ks + "b"] = mk(cv, N, S, T, c);
  }
  { size_t wt = 0; for (auto& l : m->layers) if (l.kind == 0) wt = std::max(wt, unet_conv3x3_w_ws_floats(l.cin, l.cout)); m->off_wt = cv.take(wt); }   // transformed-weight scratch
  // per-layer scratch of the prepared weights: the split fp16 image (fp32 storage) or the 9-tap bf16 image
  for (auto& l : m->layers) if (l.kind == 0 && l.cin > 1) m->wprep_f[l.name] = cv.take(m->dt ? ((size_t)9 * l.cin * l.cout + 1) / 2 : unet_conv3x3_w_ws_floats(l.cin, l.cout));
  // decoder BatchNorm folded into the conv that consumes it: whenever that conv runs on the F(2x2,3x3) kernels (the only ones with the border-class bias)
  if (m->ctx->opt_bn_fold) {
    for (int k = 6; k <= 9; ++k) {
      const std::string ks = std::to_string(k), cn = "c" + ks + "a";
      const Buf ob = m->act.at(cn); const int cin = 2 * ob.c, cout = ob.c;
      if (!wgrad_bn_fold_supported(cout)) continue;
      if (m->dt) { if (!bf16_conv3x3_supported(cin, cout) || !bf16_conv3x3_supported(cout, cin)) continue; }      // bf16 storage: the direct MFMA kernel has both epilogues
      else if (!h2_conv3x3_selected(m->algo, cin, cout)) continue;
      m->fold_off[cn] = cv.take(bn_fold_scratch_floats(cin, cout));
      m->folded_bn["bn" + ks] = {"cat" + ks, cin};
    }
  }
  // ConvT layers on the h2 kernels: their split images (forward / data-gradient form) are built with the conv3x3 images in the program's batch launch
  if (!m->dt) for (auto& l : m->layers) if (l.kind == 1 && h2_convT_selected(m->ctx, m->algo, l.cin, l.cout)) m->wprep_f[l.name] = cv.take((h2_convT_img_bytes(l.cin, l.cout) + 3) / 4);
  m->ws_floats_infer = cv.cur;
  if (!m->dt) for (auto& l : m->layers) if (l.kind == 1 && h2_convT_selected(m->ctx, m->algo, l.cin, l.cout)) m->wprep_b[l.name] = cv.take((h2_convT_img_bytes(l.cin, l.cout) + 3) / 4);
  for (auto& kv : m->fold_off) {
    const Buf ob = m->act.at(kv.first); const int cin = 2 * ob.c, cout = ob.c;
    m->fold_g_off[kv.first] = cv.take(wgrad_bn_fold_scratch_floats(N, cout));
    // the data gradient (cout -> cin channels) on the F(2x2,3x3) kernels too: the BatchNorm backward moves into its epilogue
    if (m->ctx->opt_bn_fold >= 2 && (m->dt || h2_conv3x3_selected(m->algo, cout, cin))) m->fold_c_off[kv.first] = cv.take((size_t)3 * cin);
  }
  for (auto& l : m->layers) if (l.kind == 0 && l.cin > 1) m->wprep_b[l.name] = cv.take(m->dt ? ((size_t)9 * l.cin * l.cout + 1) / 2 : unet_conv3x3_w_ws_floats(l.cin, l.cout));
  // fp32: where a decoder block's BatchNorm backward runs in its data-gradient epilogue AND the matching encoder tail takes its fused backward, the K1 * x term of the
  // skip half is added by the latter: coefficients [3][2C] of c<k>a, K1 of skip channel j at [2C + C + j]
  if (!m->dt && m->ctx->opt_enc_bn_fused)
    for (auto& kv : m->fold_c_off) {
      const int k = kv.first[1] - '0'; const int c2 = 2 * m->act.at(kv.first).c;
      if ((c2 / 2) % 32 == 0) m->skip_k1_off["bn" + std::to_string(10 - k)] = kv.second + (size_t)c2 + c2 / 2;
    }
  // ReLU masks as sign bits (MASK_RELU_BITS): the output of a conv that is the mask of the next conv's / ConvT's data gradient -- c<k>a for the conv pairs,
  // c5b ... c8b for the ConvTs -- where producer and consumer both run on the h2 kernels (decided exactly as the launches decide)
  if (!m->dt && m->ctx->opt_relu_bits) {
    auto h2_conv = [&](int w, int K, int M) { (void)w; return h2_conv3x3_selected(m->algo, K, M); };
    for (auto& l : m->layers) {
      if (l.kind != 0) continue;
      const Buf ob = m->act.at(l.name);
      if (l.cin == 1) {                                                                             // c1a: the Cin = 1 kernel writes the bits of c1b's data-gradient mask
        if (m->algo == UNET_ALGO_AUTO && c1_relu_bits_supported(ob.w, l.cout) && h2_conv(ob.w, l.cout, l.cout)) m->sign_off[l.name] = cv.take((size_t)ob.n * ob.h * ob.w * l.cout / 32);
        continue;
      }
      if ((l.cout & 31) || (ob.w & 7) || !h2_conv(ob.w, l.cin, l.cout)) continue;
      const char last = l.name.back();
      bool used = false;
      if (last == 'a') used = h2_conv(ob.w, l.cout, l.cout);                                      // mask of c<k>b's data gradient (K = M = cout)
      else if (l.name == "c5b" || l.name == "c6b" || l.name == "c7b" || l.name == "c8b") used = h2_convT_selected(m->ctx, m->algo, l.cout, l.cout / 2);
      else if (l.name == "c9b") used = m->head_fused;                                             // mask of the fused head's backward (k_head_dy)
      if (used) m->sign_off[l.name] = cv.take((size_t)ob.n * ob.h * ob.w * l.cout / 32);
    }
  }
  // --- training extras: gradient twins ---
  for (auto& kv : m->act) {
    const std::string& nm = kv.first; const Buf& b = kv.second;
    if (nm.rfind("cat", 0) == 0) {
      Buf g = mk(cv, b.n, b.h, b.w, b.c);
      m->grad[nm] = g;
    }
  }
  for (auto& kv : m->act) {
    const std::string& nm = kv.first; const Buf& b = kv.second;
    if (nm.rfind("cat", 0) == 0) continue;
    if (m->skip_raw && nm.size() == 3 && nm[0] == 'b' && nm[2] >= '1' && nm[2] <= '4') {          // bn<1..4>: the activation is only a tap scratch, the gradient IS the skip half of the concat's
      const Buf g = m->grad.at("cat" + std::to_string(10 - (nm[2] - '0')));
      m->grad[nm] = slice(g, g.c / 2, g.c / 2);
    } else if (m->skip_raw && nm.size() == 3 && nm[0] == 'c' && nm[2] == 'b' && nm[1] >= '1' && nm[1] <= '4') {          // c<1..4>b lives in the concat, its gradient is a tensor of its own
      m->grad[nm] = mk(cv, b.n, b.h, b.w, b.c);
    } else
    if (b.chan_off != 0 || b.ld != b.c) {                       // slices of a cat buffer: u<k>, bn<1..4>
      std::string catname;
      if (nm[0] == 'u') catname = "cat" + nm.substr(1);
      else catname = "cat" + std::to_string(10 - std::stoi(nm.substr(2)));
      m->grad[nm] = slice(m->grad.at(catname), (int)b.chan_off, b.c);
    } else {
      m->grad[nm] = mk(cv, b.n, b.h, b.w, b.c);
    }
  }
  size_t wgb = 0;
  int S2 = m->H, T2 = m->W;
  (void)S2; (void)T2;
  // wgrad split-K workspace: max over conv layers at their spatial sizes
  {
    int s = m->H, t = m->W, idx = 0;
    for (auto& l : m->layers) {
      (void)idx;
      if (l.kind != 0) continue;
      // spatial size of this conv layer
      int lvl;
      if (l.name == "c5a" || l.name == "c5b") lvl = 4;
      else { int k = l.name[1] - '0'; lvl = (k <= 4) ? k - 1 : 9 - k; }
      int hs = s >> lvl, ts = t >> lvl;
      wgb = std::max(wgb, m->dt ? unet_conv3x3_bwd_weights_ws_bytes_bf16(m->N, hs, ts, l.cin, l.cout) : unet_conv3x3_bwd_weights_ws_bytes(m->N, hs, ts, l.cin, l.cout));
    }
    for (auto& l : m->layers) {
      if (l.kind != 1) continue;
      int k = l.name[1] - '0';                 // u6..u9: input at level (10-k), i.e. spatial >> (10-k)
      int lvl = 10 - k;
      wgb = std::max(wgb, m->dt ? bf16_convT_wgrad_ws_bytes(m->N, s >> lvl, t >> lvl, l.cin, l.cout) : unet_convT2x2_bwd_weights_ws_bytes(m->N, s >> lvl, t >> lvl, l.cin, l.cout));
    }
  }
  m->wgrad_ws_bytes = wgb;
  m->off_wgrad_ws = cv.take((wgb + 3) / 4);
  m->ws_floats_train = cv.cur;
}

// ---- algorithmic flop / byte model -------------------------------
double nel(const Buf& b) { return (double)b.n * b.h * b.w * b.c; }

#define ADD_OP(vec, nm, fl, by, ...)                          \
  do {                                                        \
    Op _o; _o.name = (nm); _o.flops = (fl); _o.bytes = (by);  \
    _o.run = [=](hipStream_t s) -> int32_t __VA_ARGS__;       \
    (vec).push_back(std::move(_o));                           \
  } while (0)

// Conv2D -> BatchNormalization (T1:860-861; every block of UPP:858-950 and T2:747-776): where a training-mode statistics pass directly follows the
// conv3x3 (fp32 or bf16 storage) that wrote its input, the conv is armed (unet_request_bn_stats) and its kernel, if it can, leaves the sums for that pass to fold
void arm_bn_statistics(unet_model* m) {
  auto& F = m->prog[UNET_PROG_FWD_TRAIN];
  unet_ctx* ctx = m->ctx;
  for (size_t i = 1; i < F.size(); ++i) {
    if (F[i].name.rfind("bn_stats:", 0) != 0 || F[i - 1].name.rfind("conv3x3_fwd:", 0) != 0) continue;
    const std::string bn = F[i].name.substr(9);
    const auto g = m->tinfo.find(bn + "/gamma");
    if (g == m->tinfo.end()) continue;
    const int c = (int)g->second.count;
    auto inner = F[i - 1].run;
    F[i - 1].run = [=](hipStream_t s) -> int32_t { unet_request_bn_stats(ctx, c); return inner(s); };
  }
}

// First conv of a graph in bf16 storage (the image stays fp32): the Cin = 1 kernels (the reference feeds 1-channel slices, T1:853, T2:748) or, for a multi-channel
// image (BASELINE configs[4] names 224 x 224 x 3), the fp32 VALU kernels through an fp32 staging tensor that is rounded to bf16 / widened from it
static int32_t first_conv_fwd_bf16(unet_ctx* ctx, unet_model* m, const std::string& name, const Buf& ob, int cout, int act, float rate, uint64_t seed, hipStream_t s) {
  if (m->in_ch == 1) return k_conv3x3_c1_fwd_bf16(ctx, m->x, m->P(name + "/kernel"), m->P(name + "/bias"), static_cast<unet_bf16*>(m->Av(name)), ob.n, ob.h, ob.w, cout, act, rate, seed, s);
  float* tmp = m->wsf(m->off_first_tmp);
  int32_t r = k_conv3x3_naive_fwd(ctx, m->x, m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, tmp, ob.n, ob.h, ob.w, m->in_ch, cout, act, rate, seed, s);
  if (r) return r;
  return unet_cast_f32_to_bf16(ctx, tmp, static_cast<unet_bf16*>(m->Av(name)), (int64_t)ob.n * ob.h * ob.w * cout, s);
}
static int32_t first_conv_wgrad_bf16(unet_ctx* ctx, unet_model* m, const std::string& name, const Buf& ob, int cout, hipStream_t s) {
  if (m->in_ch == 1) return k_conv3x3_c1_wgrad_bf16(ctx, m->x, static_cast<const unet_bf16*>(m->Dv(name)), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                                    ob.n, ob.h, ob.w, cout, s);
  float* tmp = m->wsf(m->off_first_tmp);
  int32_t r = unet_cast_bf16_to_f32(ctx, static_cast<const unet_bf16*>(m->Dv(name)), tmp, (int64_t)ob.n * ob.h * ob.w * cout, s);
  if (r) return r;
  return k_conv3x3_naive_wgrad(ctx, m->x, tmp, m->G(name + "/kernel"), m->G(name + "/bias"), ob.n, ob.h, ob.w, m->in_ch, cout, s);
}

void build_programs(unet_model* m) {
  unet_ctx* ctx = m->ctx;
  const int algo = m->algo;
  const double gcount = (double)m->world;     // multiplies per-rank element counts into global counts
  const int dt = m->dt;
  const double eb = dt ? 2.0 : 4.0;           // bytes per stored activation element (roofline accounting)
#define CBF(p) static_cast<const unet_bf16*>(p)
#define WBF(p) static_cast<unet_bf16*>(p)
  auto& FT = m->prog[UNET_PROG_FWD_TRAIN];
  auto& FI = m->prog[UNET_PROG_FWD_INFER];
  auto& BW = m->prog[UNET_PROG_BWD];
  const size_t sums_bytes = (m->bn_sums_doubles + 4 + UNET_HEAD_SUMS) * sizeof(double);
  // layers whose 3x3 weights are consumed as a prepared image (decided per layer exactly as the conv dispatch does)
  struct PrepItem { std::string name; int cin, cout, h, w; };
  std::vector<PrepItem> prep_items;
  for (auto& l : m->layers) if (l.kind == 0 && l.cin > 1) { const Buf& ob = m->act.at(l.name); prep_items.push_back({l.name, l.cin, l.cout, ob.h, ob.w}); }
  auto prep_weights_bf16 = [=](int flip, hipStream_t s) -> int32_t {          // bf16 storage: the MFMA weight images of all conv3x3 layers
    unet_wimg_prep_list L; L.n = 0; int ci[UNET_PREP_MAX], co[UNET_PREP_MAX];
    for (auto& it : prep_items) {
      if (!flip && m->fold_off.count(it.name)) continue;          // image built from the scaled weights after the BatchNorm's finalize
      if (L.n >= UNET_PREP_MAX) UNET_FAIL(ctx, UNET_E_STATE, "weight images (bf16): too many layers for one batch");
      L.item[L.n] = unet_wimg_prep{m->P(it.name + "/kernel"), static_cast<unet_bf16*>(static_cast<void*>(m->wsf((flip ? m->wprep_b : m->wprep_f).at(it.name)))), 0, 0, 0, 0, 0, 0, flip, 0};
      ci[L.n] = flip ? it.cout : it.cin; co[L.n] = flip ? it.cin : it.cout; ++L.n;
    }
    return k_wimg_multi(ctx, &L, ci, co, s);
  };
  std::vector<PrepItem> convt_items;
  for (auto& l : m->layers) if (l.kind == 1 && m->wprep_f.count(l.name)) convt_items.push_back({l.name, l.cin, l.cout, 0, 0});
  auto prep_weights = [=](int flip, hipStream_t s) -> int32_t {          // fp32: the split fp16 weight images of every conv3x3 / ConvT launch that runs on the h2 kernels: two launches
    const float* ws_[UNET_PREP_MAX]; const float* cs_[UNET_PREP_MAX]; void* is_[UNET_PREP_MAX]; int ci_[UNET_PREP_MAX], co_[UNET_PREP_MAX], kd_[UNET_PREP_MAX], nx = 0;
    for (int k = 0; k < UNET_PREP_MAX; ++k) cs_[k] = nullptr;
    for (auto& it : prep_items) {
      const int K = flip ? it.cout : it.cin, M = flip ? it.cin : it.cout;          // channels the launch consumes / produces
      if (flip && it.name == "c1a") continue;
      const bool folded = !flip && m->fold_off.count(it.name) != 0;          // image prepared after its BatchNorm's finalize (bn_fold_prepare): only the raw-weight maxima here (kind 4)
      if (!h2_conv3x3_selected(algo, K, M)) continue;
      if (nx >= UNET_PREP_MAX) UNET_FAIL(ctx, UNET_E_STATE, "weight images: too many layers for one batch (a skipped layer would run on an image that was never written)");
      ws_[nx] = m->P(it.name + "/kernel"); is_[nx] = m->wsf((flip ? m->wprep_b : m->wprep_f).at(it.name)); ci_[nx] = it.cin; co_[nx] = it.cout; kd_[nx] = folded ? 4 : flip;
      if (flip && it.name == "c9b" && m->head_bwd_fused) cs_[nx] = m->P("out/kernel");          // the data gradient contracts dz [y > 0] with w_head[c] W[c][.]: the head's weights ride in the image
      ++nx;
    }
    for (auto& it : convt_items) {
      if (nx >= UNET_PREP_MAX) UNET_FAIL(ctx, UNET_E_STATE, "weight images: too many layers for one batch");
      ws_[nx] = m->P(it.name + "/kernel"); is_[nx] = m->wsf((flip ? m->wprep_b : m->wprep_f).at(it.name)); ci_[nx] = it.cin; co_[nx] = it.cout; kd_[nx] = flip ? 3 : 2; ++nx;
    }
    return k_h2_prep_multi(ctx, ws_, cs_, is_, ci_, co_, kd_, nx, s);
  };

  // ------------------------------------------------------------------ forward (train / infer)
  for (int training = 1; training >= 0; --training) {
    auto& F = training ? FT : FI;
    auto& SY = m->syncref[training ? UNET_PROG_FWD_TRAIN : UNET_PROG_FWD_INFER];
    const int tr_ = training;
    ADD_OP(F, "zero_sums", 0, 0, {
      if (!tr_ && !m->yt) return UNET_OK;          // (inference without labels: no statistics, no loss sums -- nothing adds into them)
      return unet_zero(ctx, m->wsf(m->off_bn_sums), sums_bytes, s);
    });
    if (!dt) ADD_OP(F, "weight_images:fwd", 0, 0, { return prep_weights(0, s); });       // all split weight images of the program in one batch of launches
    else ADD_OP(F, "weight_images:fwd", 0, 0, { return prep_weights_bf16(0, s); });
    auto conv = [&](const std::string& name, const std::string& in, int cin, int cout) {
      const Buf ob = m->act.at(name);
      double fl = 2.0 * 9 * cin * cout * (double)ob.n * ob.h * ob.w;
      double by = eb * (double)ob.n * ob.h * ob.w * (cin + cout) + 4.0 * 9.0 * cin * cout;
      ADD_OP(F, "conv3x3_fwd:" + name, fl, by, {
        if (dt) {
          if (in.empty()) return first_conv_fwd_bf16(ctx, m, name, ob, cout, ACT_RELU, 0.0f, 0, s);
          return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(in)), m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, WBF(m->Av(name)), ob.n, ob.h, ob.w, cin, cout, ACT_RELU,
                                    0.0f, 0, WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s, CBF(static_cast<void*>(m->wsf(m->wprep_f.at(name)))));
        }
        const float* xin = in.empty() ? m->x : m->A(in);
        const auto pf = m->wprep_f.find(name);
        const auto so = training ? m->sign_off.find(name) : m->sign_off.end();
        unsigned long long* sg = so == m->sign_off.end() ? nullptr : reinterpret_cast<unsigned long long*>(m->wsf(so->second));
        ctx->signs_req = sg; ctx->signs_done = nullptr;
        ctx->k_slices_ok = training ? 0 : 1;          // (inference: a launch that leaves most CUs idle may slice its contraction -- kernels_conv_h2.hip, SPLITK)
        int32_t r = conv3x3_fwd_dispatch(ctx, xin, m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, m->Aw(name), ob.n, ob.h, ob.w, cin, cout,
                                         ACT_RELU, 0.0f, 0, algo, s, m->wsf(m->off_wt), 0, pf == m->wprep_f.end() ? nullptr : m->wsf(pf->second), ob.ld);          // (ob.ld > cout: c<k>b inside its concat, skip_raw)
        ctx->signs_req = nullptr;
        if (!r && sg && ctx->signs_done != sg) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3_fwd %s: the launch did not write the ReLU sign bits its data gradient was planned with", name.c_str());
        return r;
      });
    };
    // skip_src: the encoder BatchNorm whose output is the second half of the concat `in` -- its statistics are analytic (unet_bn_stats_concat)
    auto bn = [&](const std::string& name, const std::string& in, const std::string& out, int c, bool fuse_pool, const std::string& skip_src = "") {
      const Buf ib = m->act.at(in), ob = m->act.at(out);
      const int64_t pixels = (int64_t)ib.n * ib.h * ib.w;
      const size_t so = m->bn_sum_off.at(name), bo = m->bnp_off.at(name);
      const bool composed = m->skip_raw && !skip_src.empty() && m->bn_comp_off.count(name) != 0;          // decoder BatchNorm over [up | RAW encoder output]: finalize + k_bn_compose in one launch
      if (training) {
        if (!skip_src.empty() && ctx->opt_bn_concat_analytic) {
          const size_t sso = m->bn_sum_off.at(skip_src);
          const int cu = c / 2, cs = c - c / 2;
          ADD_OP(F, "bn_stats:" + name, 0, eb * pixels * cu, {
            if (dt) return unet_bn_stats_concat_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsd(m->off_bn_sums) + sso, (double)pixels * gcount, m->P(skip_src + "/gamma"),
                                                     m->P(skip_src + "/beta"), m->wsd(m->off_bn_sums) + so, pixels, cu, cs, s);
            return unet_bn_stats_concat(ctx, m->A(in), ib.ld, m->wsd(m->off_bn_sums) + sso, (double)pixels * gcount, m->P(skip_src + "/gamma"), m->P(skip_src + "/beta"),
                                        m->wsd(m->off_bn_sums) + so, pixels, cu, cs, s);
          });
        } else {
          ADD_OP(F, "bn_stats:" + name, 0, eb * pixels * c, {
            if (dt) return unet_bn_stats_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
            return unet_bn_stats(ctx, m->A(in), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
          });
        }
        SY.push_back({(int)F.size() - 1, 0, true, (m->off_bn_sums * 4) + so * 8, 2 * (int64_t)c});
        ADD_OP(F, "bn_finalize:" + name, 0, 0, {
          if (composed) return k_bn_finalize_compose(ctx, 1, m->wsd(m->off_bn_sums) + so, (double)pixels * gcount, m->P(name + "/gamma"), m->P(name + "/beta"), m->P(name + "/mean"),
                                                     m->P(name + "/var"), m->wsf(bo), c, m->wsf(m->bnp_off.at(skip_src)), m->wsf(m->bn_comp_off.at(name)), s);
          return unet_bn_finalize_train(ctx, m->wsd(m->off_bn_sums) + so, (double)pixels * gcount, m->P(name + "/gamma"), m->P(name + "/beta"),
                                        m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      } else {
        ADD_OP(F, "bn_finalize_infer:" + name, 0, 0, {
          if (composed) return k_bn_finalize_compose(ctx, 0, nullptr, 0.0, m->P(name + "/gamma"), m->P(name + "/beta"), m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c,
                                                     m->wsf(m->bnp_off.at(skip_src)), m->wsf(m->bn_comp_off.at(name)), s);
          return unet_bn_finalize_infer(ctx, m->P(name + "/gamma"), m->P(name + "/beta"), m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      }
      if (!fuse_pool && !m->folded_bn.count(name)) ADD_OP(F, "bn_apply:" + name, 0, 2 * eb * pixels * c, {
        if (dt) return unet_bn_apply_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsf(bo), WBF(m->Av(out)), ob.ld, pixels, c, s);
        return unet_bn_apply(ctx, m->A(in), ib.ld, m->wsf(bo), m->Aw(out), ob.ld, pixels, c, s);
      });
    };
    std::string prev = "";
    int cprev = m->in_ch;
    for (int k = 1; k <= 4; ++k) {
      int c = ENC[k - 1]; std::string ks = std::to_string(k);
      conv("c" + ks + "a", prev, cprev, c);
      conv("c" + ks + "b", "c" + ks + "a", c, c);
      bn("bn" + ks, "c" + ks + "b", "bn" + ks, c, true);
      const Buf ib = m->act.at("bn" + ks), xb = m->act.at("c" + ks + "b");
      const std::string pin = "bn" + ks, pout = "p" + ks, xin = "c" + ks + "b";
      const int tr = training;
      const size_t bo = m->bnp_off.at("bn" + ks);
      ADD_OP(F, "bn_apply_pool:" + pout, 0, eb * (m->skip_raw ? 1.25 : 2.25) * nel(ib), {
        if (dt) return unet_bn_apply_maxpool_dropout_fwd_bf16(ctx, CBF(m->Av(xin)), xb.ld, m->wsf(bo), WBF(m->Av(pin)), ib.ld, WBF(m->Av(pout)), ib.n, ib.h, ib.w, ib.c,
                                                              tr ? m->drop_rate : 0.0f, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
        return unet_bn_apply_maxpool_dropout_fwd(ctx, m->A(xin), xb.ld, m->wsf(bo), m->skip_raw ? nullptr : m->Aw(pin), ib.ld, m->Aw(pout), ib.n, ib.h, ib.w, ib.c,
                                                 tr ? m->drop_rate : 0.0f, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);          // (skip_raw: the normalised tensor is never stored)
      });
      prev = pout; cprev = c;
    }
    conv("c5a", "p4", 256, 512);
    conv("c5b", "c5a", 512, 512);
    prev = "c5b"; cprev = 512;
    const int dec[4] = {256, 128, 64, 32};
    for (int k = 6; k <= 9; ++k) {
      int c = dec[k - 6]; std::string ks = std::to_string(k);
      const Buf ib = m->act.at(prev), ub = m->act.at("u" + ks);
      const std::string uin = prev, un = "u" + ks; const int ci = cprev;
      const bool arm_up = training && ctx->opt_bn_concat_analytic;          // (the statistics pass of the concat reads only this half then)
      ADD_OP(F, "convT_fwd:" + un, 2.0 * 4 * ci * c * nel(ib) / ib.c, eb * (nel(ib) + nel(ub)), {
        if (arm_up) unet_request_bn_stats(ctx, c);
        if (dt) return k_convT_bf16_fwd(ctx, CBF(m->Av(uin)), m->P(un + "/kernel"), m->P(un + "/bias"), WBF(m->Av(un)), ub.ld, ib.n, ib.h, ib.w, ci, c, WBF(static_cast<void*>(m->wsf(m->off_wt))), s);
        const auto pf = m->wprep_f.find(un);
        if (pf != m->wprep_f.end()) return k_convT_h2_fwd(ctx, m->A(uin), m->P(un + "/kernel"), m->P(un + "/bias"), m->Aw(un), ub.ld, ib.n, ib.h, ib.w, ci, c, s, m->wsf(pf->second));
        return unet_convT2x2_fwd(ctx, m->A(uin), m->P(un + "/kernel"), m->P(un + "/bias"), m->Aw(un), ub.ld, ib.n, ib.h, ib.w, ci, c, algo, s);
      });
      bn("bn" + ks, "cat" + ks, "bn" + ks, 2 * c, false, "bn" + std::to_string(10 - k));      // cat_k = [u_k, bn_{10-k} output]
      if (m->fold_off.count("c" + ks + "a")) {                // BatchNorm folded into the conv: scaled weights + border-class bias, raw concat in
        const std::string cn = "c" + ks + "a", xn = "cat" + ks, bnn = "bn" + ks;
        const Buf ob = m->act.at(cn); const int cin = 2 * c, cout = c;
        const size_t fo = m->fold_off.at(cn), bo = m->bnp_off.at(bnn), uo = m->wprep_f.at(cn);
        const size_t co_ = m->skip_raw ? m->bn_comp_off.at(bnn) : 0, boe = m->skip_raw ? m->bnp_off.at("bn" + std::to_string(10 - k)) : 0;
        ADD_OP(F, "bn_fold_prepare:" + cn, 2.0 * 9 * cin * cout * 2, 4.0 * 9 * cin * cout * 4, {
          const float* sc_ = m->wsf(bo);                        // [scale 2C][shift 2C] of the folded map
          if (m->skip_raw) sc_ = m->wsf(co_);                  // the skip half of the concat is the RAW encoder output: the composite map bn_finalize left (k_bn_finalize_compose)
          (void)boe;
          int32_t r = k_bn_fold_prepare(ctx, m->P(cn + "/kernel"), m->P(cn + "/bias"), sc_, sc_ + cin, cin, cout, m->wsf(fo), s, dt != 0);
          if (r || dt) return r;                                // bf16 storage: the conv below builds its weight image from the scaled fp32 weights
          return k_h2_weights_bound(ctx, m->P(cn + "/kernel"), sc_, m->wsf(uo), cin, cout, s);          // fp32: the image kernel applies the scale per input channel (exponent from max |w| max |scale|)
        });
        ADD_OP(F, "conv3x3_fwd:" + cn, 2.0 * 9 * cin * cout * (double)ob.n * ob.h * ob.w, eb * (double)ob.n * ob.h * ob.w * (cin + cout) + 4.0 * 9.0 * cin * cout, {
          const float* tab = m->wsf(fo) + (size_t)9 * cin * cout;
          if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(xn)), m->wsf(fo), tab, reinterpret_cast<const unet_bf16*>(tab), MASK_BIAS_TAB, WBF(m->Av(cn)), ob.n, ob.h, ob.w, cin, cout, ACT_RELU,
                                            0.0f, 0, WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s, nullptr);
          const auto so = training ? m->sign_off.find(cn) : m->sign_off.end();
          unsigned long long* sg = so == m->sign_off.end() ? nullptr : reinterpret_cast<unsigned long long*>(m->wsf(so->second));
          ctx->signs_req = sg; ctx->signs_done = nullptr;
          ctx->k_slices_ok = training ? 0 : 1;
          int32_t r = k_conv3x3_h2_fwd(ctx, m->A(xn), m->wsf(uo), tab, tab, MASK_BIAS_TAB, m->Aw(cn), ob.n, ob.h, ob.w, cin, cout, ACT_RELU, 0.0f, 0, s);
          ctx->signs_req = nullptr;
          if (!r && sg && ctx->signs_done != sg) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3_fwd %s: the launch did not write the ReLU sign bits its data gradient was planned with", cn.c_str());
          return r;
        });
      } else
      conv("c" + ks + "a", "bn" + ks, 2 * c, c);
      if (k == 9 && m->head_fused) {
        m->c9b_virtual = ctx->opt_head_bwd_fused && m->sign_off.count("c9b") != 0;
        // T1:911-913 in one launch: c9b, the 1x1 sigmoid head, the loss sums and the sums of the head's weight gradient (kernels_conv_h2.hip, HEAD)
        const Buf ob = m->act.at("c9b"); const double px = (double)ob.n * ob.h * ob.w;
        ADD_OP(F, "conv3x3_fwd_head:c9b", 2.0 * 9 * c * c * px + 2.0 * c * px, 4.0 * px * (c + (m->c9b_virtual ? 0 : c)) + 4.0 * 9.0 * c * c + 8.0 * px + (training ? px * c / 8.0 : 0.0), {
          if (!m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_fwd: p_out not set (unet_model_set_io)");
          const auto so = training ? m->sign_off.find("c9b") : m->sign_off.end();
          unsigned long long* sg = so == m->sign_off.end() ? nullptr : reinterpret_cast<unsigned long long*>(m->wsf(so->second));
          ctx->signs_req = sg; ctx->signs_done = nullptr;
          // (c9b_virtual: nothing reads the tensor -- the backward takes p, the sums and the sign bits, inference takes p -- so it is not written: -0.5 GB per step at 512 x 512 x 16)
          int32_t r = k_conv3x3_h2_head_fwd(ctx, m->A("c9a"), m->wsf(m->wprep_f.at("c9b")), m->P("c9b/bias"), m->c9b_virtual ? nullptr : m->Aw("c9b"), m->P("out/kernel"), m->P("out/bias"),
                                            m->pout, m->yt, ob.n, ob.h, ob.w, c, s);
          ctx->signs_req = nullptr;
          if (!r && sg && ctx->signs_done != sg) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3_fwd_head: the launch did not write the ReLU sign bits its backward was planned with");
          return r;
        });
      } else
      conv("c" + ks + "b", "c" + ks + "a", c, c);
      prev = "c" + ks + "b"; cprev = c;
    }
    const Buf hb = m->act.at("c9b");
    const int64_t hp = (int64_t)hb.n * hb.h * hb.w;
    if (m->head_fused) ADD_OP(F, "head_fold", 0, 0, {
      if (!m->yt) return UNET_OK;
      return k_head_fold(ctx, m->wsd(m->off_loss_sums), m->wsd(m->off_head_sums), s);
    });
    else
    ADD_OP(F, "head_fwd", 2.0 * 32 * hp, hp * (eb * 32 + 8.0), {
      if (!m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_fwd: p_out not set (unet_model_set_io)");
      if (dt) return unet_head_fwd_bf16(ctx, CBF(m->Av("c9b")), m->P("out/kernel"), m->P("out/bias"), m->pout, m->yt, m->yt ? m->wsd(m->off_loss_sums) : nullptr, hp, hb.c, s);
      return unet_head_fwd(ctx, m->A("c9b"), m->P("out/kernel"), m->P("out/bias"), m->pout, m->yt, m->yt ? m->wsd(m->off_loss_sums) : nullptr, hp, hb.c, s);
    });
    SY.push_back({(int)F.size() - 1, 1, true, m->off_loss_sums * 4, 4});
    ADD_OP(F, "loss_finalize", 0, 0, {
      if (!m->yt) return UNET_OK;
      return k_loss_finalize(ctx, m->wsd(m->off_loss_sums), (double)hp * gcount, m->wsf(m->off_loss_out), m->loss_out2, s);
    });
  }

  // ------------------------------------------------------------------ backward
  {
    auto& SY = m->syncref[UNET_PROG_BWD];
    size_t nb = 0;
    for (auto& kv : m->bn_bsum_off) nb = std::max(nb, kv.second);
    size_t bs_bytes = 0;
    for (auto& l : m->layers) if (l.kind == 2) bs_bytes += 2 * (size_t)l.cout * sizeof(double);
    ADD_OP(BW, "zero_bwd_sums", 0, 0, {
      int32_t r = unet_zero(ctx, m->wsf(m->off_bn_bsums), bs_bytes, s);
      if (r) return r;
      return unet_zero(ctx, m->G("out/kernel"), (size_t)(m->tinfo.at("out/kernel").count + 1) * sizeof(float), s);
    });
    const Buf hb = m->act.at("c9b");
    const int64_t hp = (int64_t)hb.n * hb.h * hb.w;
    if (!dt) ADD_OP(BW, "weight_images:bwd", 0, 0, { return prep_weights(1, s); });
    else ADD_OP(BW, "weight_images:bwd", 0, 0, { return prep_weights_bf16(1, s); });
    m->head_bwd_fused = m->head_fused && ctx->opt_head_bwd_fused && m->arch == UNET_ARCH_UNET && m->sign_off.count("c9b") && m->wprep_b.count("c9b") && h2_head_bwd_selected(ctx, algo, hb.w, 32) &&
                        h2_wgrad_selected(algo, 32, 32);
    if (m->head_bwd_fused) ADD_OP(BW, "head_dzm", 8.0 * hp, hp * (8.0 + 4.0 + 8.0), {
      if (!m->yt || !m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_dzm: io not set");
      return k_head_dzm(ctx, m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, m->wsd(m->off_head_sums), reinterpret_cast<const unsigned long long*>(m->wsf(m->sign_off.at("c9b"))),
                        m->D("c9b"), m->G("out/kernel"), m->G("out/bias"), hb.n, hb.h, hb.w, s);
    });
    else
    if (m->head_fused) ADD_OP(BW, "head_dy", 2.0 * 32 * hp, hp * (4.0 * 32 + 8.0 + 4.0), {
      if (!m->yt || !m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_dy: io not set");
      const auto so = m->sign_off.find("c9b");
      return k_head_dy(ctx, m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, m->wsd(m->off_head_sums), m->P("out/kernel"),
                       so == m->sign_off.end() ? nullptr : reinterpret_cast<const unsigned long long*>(m->wsf(so->second)), m->A("c9b"), m->D("c9b"),
                       m->G("out/kernel"), m->G("out/bias"), hb.n, hb.h, hb.w, s);
    });
    else
    ADD_OP(BW, "head_bwd", 4.0 * 32 * hp, hp * (eb * 64 + 8.0), {
      if (!m->yt || !m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_bwd: io not set");
      if (dt) return unet_head_bwd_bf16(ctx, CBF(m->Av("c9b")), m->P("out/kernel"), m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, WBF(m->Dv("c9b")),
                                        m->G("out/kernel"), m->G("out/bias"), hp, hb.c, 1, s);
      return unet_head_bwd(ctx, m->A("c9b"), m->P("out/kernel"), m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, m->D("c9b"),
                           m->G("out/kernel"), m->G("out/bias"), hp, hb.c, 1, s);
    });
    // conv backward: wgrad (x, dy) then dgrad (dy -> dx, optional relu mask = activation that produced x)
    // xraw: the conv's input BatchNorm is folded (fold_off): the weight gradient runs on the raw tensor `xraw` and is corrected afterwards
    // Data parallelism: a batch-global reduction (BatchNorm backward sums) sits between the op that produces the local sums and the op that consumes the
    // global ones.  A weight gradient depends on neither, so the program places one BEHIND every such producer: conv_bwd(..., defer_wgrad) parks the layer's
    // weight-gradient op in DEF, flush_def() emits it where a reduction is in flight, and the sync point's use_op tells the host how long it may run beside
    // the compute stream (include/unet_hip.h).  One rank: the same ops in a slightly different order.
    std::vector<Op> DEF;
    auto flush_def = [&]() { for (auto& o : DEF) BW.push_back(std::move(o)); DEF.clear(); };
    auto conv_bwd = [&](const std::string& name, const std::string& in, int cin, int cout, bool want_dx, bool mask_in, const std::string& xraw = "", bool defer_wgrad = false) {
      const Buf ob = m->act.at(name);
      const double px = (double)ob.n * ob.h * ob.w;
      const std::string xsrc = xraw.empty() ? in : xraw;
      auto& WV = (defer_wgrad && xraw.empty()) ? DEF : BW;
      const bool hstream = name == "c9b" && m->head_bwd_fused;          // dy = the head's 8-byte-per-pixel stream
      ADD_OP(WV, "conv3x3_wgrad:" + name, 2.0 * 9 * cin * cout * px, (hstream ? eb * px * cin + 8.0 * px : eb * px * (cin + cout)) + 4.0 * 9.0 * cin * cout, {
        if (dt) {
          if (in.empty()) return first_conv_wgrad_bf16(ctx, m, name, ob, cout, s);
          return k_conv3x3_bf16_wgrad(ctx, CBF(m->Av(xsrc)), CBF(m->Dv(name)), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w,
                                      cin, cout, s);
        }
        const float* xin = xsrc.empty() ? m->x : m->A(xsrc);
        if (name == "c9b" && m->head_bwd_fused)               // dy = the head's {dz, mask} stream (head_dzm above)
          return k_conv3x3_h2_wgrad_dzm(ctx, xin, m->D(name), m->P("out/kernel"), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, cin, s);
        return conv3x3_wgrad_dispatch(ctx, xin, m->D(name), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                      ob.n, ob.h, ob.w, cin, cout, algo, s);
      });
      if (!xraw.empty()) {
        const size_t go = m->fold_g_off.at(name), bo = m->bnp_off.at(in);
        const bool pg = !dt && m->fold_c_off.count(name) != 0;          // the BatchNorm's parameter gradients ride in the same launch (the op below stays as the carrier of the sync point)
        ADD_OP(BW, "wgrad_bn_fold_fix:" + name, 2.0 * 9 * cin * cout, 8.0 * 9 * cin * cout, {
          // ... and the BatchNorm's backward sums (sum dz, sum dz * xhat) come out of W, the raw dw and S: no pass over dz / x (bn_bwd below: stats_done)
          if (dt) return k_wgrad_bn_fold_fix_bf16(ctx, CBF(m->Dv(name)), ob.n, ob.h, ob.w, cin, cout, m->wsf(bo), m->wsf(bo) + cin, m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(go), s,
                                                  m->P(name + "/kernel"), m->wsf(bo) + 2 * cin, m->wsf(bo) + 3 * cin, m->wsd(m->off_bn_bsums) + m->bn_bsum_off.at(in));
          if (m->skip_raw) {                                    // the weight gradient ran on [up | RAW encoder output]: composite scale / shift, and the decoder BatchNorm saw pre_s x + pre_t
            const float* comp = m->wsf(m->bn_comp_off.at(in));
            return k_wgrad_bn_fold_fix(ctx, m->D(name), ob.n, ob.h, ob.w, cin, cout, comp, comp + cin, m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(go), s,
                                       m->P(name + "/kernel"), m->wsf(bo) + 2 * cin, m->wsf(bo) + 3 * cin, m->wsd(m->off_bn_bsums) + m->bn_bsum_off.at(in), comp + 2 * cin, comp + 3 * cin,
                                       pg ? m->G(in + "/gamma") : nullptr, pg ? m->G(in + "/beta") : nullptr);
          }
          return k_wgrad_bn_fold_fix(ctx, m->D(name), ob.n, ob.h, ob.w, cin, cout, m->wsf(bo), m->wsf(bo) + cin, m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(go), s,
                                     m->P(name + "/kernel"), m->wsf(bo) + 2 * cin, m->wsf(bo) + 3 * cin, m->wsd(m->off_bn_bsums) + m->bn_bsum_off.at(in), nullptr, nullptr,
                                     pg ? m->G(in + "/gamma") : nullptr, pg ? m->G(in + "/beta") : nullptr);
        });
      }
      if (!xraw.empty() && m->fold_c_off.count(name)) {
        // BatchNorm backward inside the data gradient: the sums are already there (from the weight gradient), so param grads -> [cross-rank sum] -> coefficients ->
        // dgrad whose epilogue writes dx = K0 dz + K1 x + K2 straight into the gradient of the raw concat; dz itself is never stored
        const std::string bnn = in;
        const size_t so = m->bn_bsum_off.at(bnn), bo = m->bnp_off.at(bnn), co = m->fold_c_off.at(name);
        ADD_OP(BW, "bn_bwd_param_grads:" + bnn, 0, 0, {
          if (!dt) return UNET_OK;                             // fp32: written by wgrad_bn_fold_fix above (one launch less; this op carries the cross-rank sync point)
          return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(bnn + "/gamma"), m->G(bnn + "/beta"), cin, s);
        });
        SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)cin});
        const size_t syi = SY.size() - 1;
        flush_def();                                          // (the block's second conv's weight gradient runs while the sums are reduced)
        SY[syi].use_op = (int)BW.size();
        ADD_OP(BW, "conv3x3_dgrad_bn_bwd:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cout + 2 * cin) + 4.0 * 9.0 * cin * cout, {
          int32_t r = k_bn_bwd_coef(ctx, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, px * gcount, m->wsf(co), cin, s);
          if (r) return r;
          if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(name)), m->P(name + "/kernel"), m->wsf(co), CBF(m->Av(xraw)), MASK_BN_BWD, WBF(m->Dv(xraw)), ob.n, ob.h, ob.w, cout, cin, ACT_NONE, 0.0f, 0,
                                            WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s, CBF(static_cast<void*>(m->wsf(m->wprep_b.at(name)))));
          // the skip half of the concat's gradient leaves without its K1 * x term when the encoder tail's fused backward adds it (it recomputes y = BN(x) anyway):
          // this launch then reads only the upsampled half of the concat (-25 % of its bytes at every level)
          const int climit = m->skip_k1_off.count("bn" + std::to_string(10 - (name[1] - '0'))) ? cin / 2 : (1 << 30);
          return k_conv3x3_h2_fwd(ctx, m->D(name), m->wsf(m->wprep_b.at(name)), m->wsf(co), m->A(xraw), MASK_BN_BWD, m->D(xraw), ob.n, ob.h, ob.w, cout, cin, ACT_NONE, 0.0f, 0, s, climit);
        });
        return;
      }
      if (want_dx && !dt && !mask_in && in.size() == 2 && in[0] == 'p' && ctx->opt_enc_bn_fused && m->wprep_b.count(name) && h2_pool_sums_selected(ctx, algo, ob.w, cout, cin)) {
        // the gradient of a pooled tensor p<k>: the pooled-path sums of the encoder tail's BatchNorm backward ride in this launch's epilogue (MASK_POOL_SUMS) --
        // the pool_bwd_sums op below then only adds the closed-form skip term
        const std::string bnn = "bn" + in.substr(1);
        const size_t so = m->bn_bsum_off.at(bnn);
        m->pool_sums_fused.insert(in);
        ADD_OP(BW, "conv3x3_dgrad_pool_sums:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cout + 2 * cin) + 4.0 * 9.0 * cin * cout, {
          // (the sums stay in the slot copies: the pool_bwd_skip_term op right below folds them out together with the skip term -- nothing between the two uses the slots)
          (void)so;
          return k_conv3x3_h2_dgrad_pool_sums(ctx, m->D(name), m->wsf(m->wprep_b.at(name)), m->A(in), m->P(bnn + "/gamma"), m->P(bnn + "/beta"), m->drop_rate, m->D(in),
                                              nullptr, ob.n, ob.h, ob.w, cout, cin, s);
        });
      } else
      if (want_dx) {
        const bool bits = mask_in && m->sign_off.count(in) != 0;
        ADD_OP(BW, "conv3x3_dgrad:" + name, 2.0 * 9 * cin * cout * px, (hstream ? 8.0 * px + eb * px * (cin + (mask_in ? (bits ? cin / 32.0 : cin) : 0)) : eb * px * (cout + cin + (mask_in ? (bits ? cin / 32.0 : cin) : 0))) + 4.0 * 9.0 * cin * cout, {
          if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(name)), m->P(name + "/kernel"), nullptr, mask_in ? CBF(m->Av(in)) : nullptr, mask_in ? MASK_RELU : MASK_NONE, WBF(m->Dv(in)), ob.n, ob.h,
                                            ob.w, cout, cin, ACT_NONE, 0.0f, 0, WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s, CBF(static_cast<void*>(m->wsf(m->wprep_b.at(name)))));
          const auto pb = m->wprep_b.find(name);
          const auto so = mask_in ? m->sign_off.find(in) : m->sign_off.end();
          if (name == "c9b" && m->head_bwd_fused)
            return k_conv3x3_h2_dgrad_dzm(ctx, m->D(name), m->wsf(pb->second), so != m->sign_off.end() ? m->wsf(so->second) : mask_in ? m->A(in) : nullptr,
                                          so != m->sign_off.end() ? MASK_RELU_BITS : mask_in ? MASK_RELU : MASK_NONE, m->D(in), ob.n, ob.h, ob.w, cin, s);
          if (so != m->sign_off.end())                        // one bit per element instead of the fp32 activation (written by the forward conv that produced `in`)
            return conv3x3_fwd_dispatch(ctx, m->D(name), m->P(name + "/kernel"), nullptr, m->wsf(so->second), MASK_RELU_BITS, m->D(in), ob.n, ob.h,
                                        ob.w, cout, cin, ACT_NONE, 0.0f, 0, algo, s, m->wsf(m->off_wt), 1, pb == m->wprep_b.end() ? nullptr : m->wsf(pb->second));
          return conv3x3_fwd_dispatch(ctx, m->D(name), m->P(name + "/kernel"), nullptr, mask_in ? m->A(in) : nullptr, mask_in ? MASK_RELU : MASK_NONE, m->D(in), ob.n, ob.h,
                                      ob.w, cout, cin, ACT_NONE, 0.0f, 0, algo, s, m->wsf(m->off_wt), 1, pb == m->wprep_b.end() ? nullptr : m->wsf(pb->second));
        });
      }
    };
    // bn backward: dy tensor `dyname` (grad buffer), x tensor `xname` (act), output grad `dxname`
    // stats_done: the (sum dy, sum dy*xhat) sums were already accumulated by the fused pool backward
    auto bn_bwd = [&](const std::string& name, const std::string& dyname, const std::string& xname, const std::string& dxname, int c, int mask,
                      bool stats_done) {
      const Buf gb = m->grad.at(dyname), xb = m->act.at(xname), db = m->grad.at(dxname);
      const int64_t pixels = (int64_t)xb.n * xb.h * xb.w;
      const size_t so = m->bn_bsum_off.at(name), bo = m->bnp_off.at(name);
      if (stats_done) {
        ADD_OP(BW, "bn_bwd_param_grads:" + name, 0, 0, {
          return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(name + "/gamma"), m->G(name + "/beta"), c, s);
        });
      } else {
        ADD_OP(BW, "bn_bwd_stats:" + name, 0, 2 * eb * pixels * c, {
          int32_t r = dt ? unet_bn_bwd_stats_bf16(ctx, CBF(m->Dv(dyname)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s)
                         : unet_bn_bwd_stats(ctx, m->D(dyname), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s);
          if (r) return r;
          return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(name + "/gamma"), m->G(name + "/beta"), c, s);
        });
      }
      SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
      ADD_OP(BW, "bn_bwd_apply:" + name, 0, 3 * eb * pixels * c, {
        if (dt) return unet_bn_bwd_apply_bf16(ctx, CBF(m->Dv(dyname)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount,
                                              mask ? MASK_RELU : MASK_NONE, 0.0f, 0, WBF(m->Dv(dxname)), db.ld, pixels, c, s);
        return unet_bn_bwd_apply(ctx, m->D(dyname), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, mask ? MASK_RELU : MASK_NONE, 0.0f, 0,
                                 m->D(dxname), db.ld, pixels, c, s);
      });
    };
    auto bucket = [&](const std::string& first, const std::string& last_tensor) {
      const TInfo a = m->tinfo.at(first), b = m->tinfo.at(last_tensor);
      SY.push_back({(int)BW.size() - 1, 3, false, (size_t)a.off * 4, b.off + b.count - a.off});
    };
    bool enc_split = false;
    const int dec[4] = {256, 128, 64, 32};
    for (int k = 9; k >= 6; --k) {
      int c = dec[k - 6]; std::string ks = std::to_string(k);
      std::string prev = (k == 6) ? "c5b" : "c" + std::to_string(k - 1) + "b";
      int cprev = (k == 6) ? 512 : dec[k - 7];
      conv_bwd("c" + ks + "b", "c" + ks + "a", c, c, true, true, "", m->fold_c_off.count("c" + ks + "a") != 0);          // (its weight gradient: behind c<k>a's sums)
      conv_bwd("c" + ks + "a", "bn" + ks, 2 * c, c, true, false, m->fold_off.count("c" + ks + "a") ? "cat" + ks : "");
      flush_def();
      if (!m->fold_c_off.count("c" + ks + "a")) bn_bwd("bn" + ks, "bn" + ks, "cat" + ks, "cat" + ks, 2 * c, 0, m->fold_off.count("c" + ks + "a") != 0);
      const Buf ib = m->act.at(prev), ug = m->grad.at("u" + ks);
      const std::string un = "u" + ks;
      ADD_OP(BW, "convT_wgrad:" + un, 2.0 * 4 * cprev * c * nel(ib) / ib.c, eb * (nel(ib) + nel(ug)), {
        if (dt) return k_convT_bf16_wgrad(ctx, CBF(m->Av(prev)), CBF(m->Dv(un)), ug.ld, m->G(un + "/kernel"), m->G(un + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ib.n, ib.h,
                                          ib.w, cprev, c, s);
        return unet_convT2x2_bwd_weights(ctx, m->A(prev), m->D(un), ug.ld, m->G(un + "/kernel"), m->G(un + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                         ib.n, ib.h, ib.w, cprev, c, algo, s);
      });
      ADD_OP(BW, "convT_dgrad:" + un, 2.0 * 4 * cprev * c * nel(ib) / ib.c, eb * (2 * nel(ib) + nel(ug)), {
        if (dt) return k_convT_bf16_dgrad(ctx, CBF(m->Dv(un)), ug.ld, m->P(un + "/kernel"), CBF(m->Av(prev)), WBF(m->Dv(prev)), ib.n, ib.h, ib.w, cprev, c, WBF(static_cast<void*>(m->wsf(m->off_wt))), s);
        const auto so = m->sign_off.find(prev);
        const auto pb = m->wprep_b.find(un);
        const void* img = pb == m->wprep_b.end() ? nullptr : m->wsf(pb->second);
        if (so != m->sign_off.end()) return k_convT_h2_dgrad(ctx, m->D(un), ug.ld, m->P(un + "/kernel"), m->wsf(so->second), m->D(prev), ib.n, ib.h, ib.w, cprev, c, s, 1, img);
        if (img) return k_convT_h2_dgrad(ctx, m->D(un), ug.ld, m->P(un + "/kernel"), m->A(prev), m->D(prev), ib.n, ib.h, ib.w, cprev, c, s, 0, img);
        return unet_convT2x2_bwd_data(ctx, m->D(un), ug.ld, m->P(un + "/kernel"), m->A(prev), m->D(prev), ib.n, ib.h, ib.w, cprev, c, algo, s);
      });
      if (k == 7) bucket("u7/kernel", "out/bias");
      if (k == 6) bucket("u6/kernel", "c6b/bias");
    }
    conv_bwd("c5b", "c5a", 512, 512, true, true);
    conv_bwd("c5a", "p4", 256, 512, true, false, "", ctx->opt_enc_bn_fused != 0);          // (its weight gradient: behind the pooled sums of level 4)
    if (!ctx->opt_enc_bn_fused) bucket("c5a/kernel", "c5b/bias");
    for (int k = 4; k >= 1; --k) {
      int c = ENC[k - 1]; std::string ks = std::to_string(k);
      const Buf xb = m->act.at("bn" + ks), gb = m->grad.at("bn" + ks);
      const std::string bnn = "bn" + ks, pn = "p" + ks;
      const size_t so = m->bn_bsum_off.at(bnn);
      if (ctx->opt_enc_bn_fused) {
        // no pass for the statistics: sums from the pooled tensors + the closed-form skip term, then pool backward + skip add + BatchNorm backward + ReLU mask in ONE pass
        const std::string dn = "bn" + std::to_string(10 - k), cb = "c" + ks + "b";
        const size_t sod = m->bn_bsum_off.at(dn), bod = m->bnp_off.at(dn), bo = m->bnp_off.at(bnn);
        const Buf cbuf = m->act.at(cb), cg = m->grad.at(cb);
        const int64_t pixels = (int64_t)xb.n * xb.h * xb.w;
        const bool sums_done = m->pool_sums_fused.count(pn) != 0;
        ADD_OP(BW, sums_done ? "pool_bwd_skip_term:" + pn : "pool_bwd_sums:" + pn, 0, sums_done ? 0.0 : eb * 0.5 * nel(xb), {
          int32_t r = sums_done ? UNET_OK : dt ? unet_maxpool2x2_dropout_bwd_sums_bf16(ctx, CBF(m->Av(pn)), CBF(m->Dv(pn)), m->P(bnn + "/gamma"), m->P(bnn + "/beta"), m->wsd(m->off_bn_bsums) + so, xb.n, xb.h,
                                                                 xb.w, xb.c, m->drop_rate, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s)
                         : unet_maxpool2x2_dropout_bwd_sums(ctx, m->A(pn), m->D(pn), m->P(bnn + "/gamma"), m->P(bnn + "/beta"), m->wsd(m->off_bn_bsums) + so, xb.n, xb.h, xb.w, xb.c,
                                                            m->drop_rate, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
          if (r) return r;
          if (sums_done)                                        // fold of the epilogue's sums + skip term + parameter gradients: one launch
            return k_enc_tail_finish(ctx, m->wsd(m->off_bn_bsums) + so, m->wsd(m->off_bn_bsums) + sod + 3 * (size_t)c, m->wsf(bod) + 7 * (size_t)c, m->P(dn + "/gamma") + c,
                                     m->P(bnn + "/gamma"), m->G(bnn + "/gamma"), m->G(bnn + "/beta"), c, 1.0 / gcount, s);
          r = unet_bn_bwd_skip_term(ctx, m->wsd(m->off_bn_bsums) + so, m->wsd(m->off_bn_bsums) + sod + 3 * (size_t)c, m->wsf(bod) + 7 * (size_t)c, m->P(dn + "/gamma") + c,
                                    m->P(bnn + "/gamma"), c, 1.0 / gcount, s);
          if (r) return r;
          return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(bnn + "/gamma"), m->G(bnn + "/beta"), c, s);
        });
        SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
        const size_t syi = SY.size() - 1;
        flush_def();                                          // (the weight gradient of the conv that consumed this level's pooled tensor runs while the sums are reduced)
        if (k == 4) bucket("c5a/kernel", "c5b/bias");
        // levels 3 and 4 are complete once c3a's (deferred) weight gradient is out: their 4.4 MB go now, so that what is left for the end of the program -- the one
        // all-reduce nothing can hide -- is the 0.3 MB of levels 1 and 2
        if (k == 2) { bucket("c3a/kernel", "bn4/beta"); enc_split = true; }
        SY[syi].use_op = (int)BW.size();
        ADD_OP(BW, "bn_pool_bwd_apply:" + bnn, 0, eb * 3.25 * nel(xb), {
          if (dt) return unet_bn_maxpool_bwd_apply_bf16(ctx, CBF(m->Av(cb)), cbuf.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, CBF(m->Dv(bnn)), gb.ld, CBF(m->Dv(pn)),
                                                        WBF(m->Dv(cb)), cg.ld, xb.n, xb.h, xb.w, xb.c, m->drop_rate, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
          const auto k1o = m->skip_k1_off.find(bnn);
          return k_bn_maxpool_bwd_apply_k1(ctx, m->A(cb), cbuf.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, m->D(bnn), gb.ld,
                                           k1o == m->skip_k1_off.end() ? nullptr : m->wsf(k1o->second), m->D(pn), m->D(cb), cg.ld,
                                           xb.n, xb.h, xb.w, xb.c, m->drop_rate, m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
        });
      } else {
      ADD_OP(BW, "pool_bwd_bnstats:" + pn, 0, eb * 3.25 * nel(xb), {
        if (dt) return unet_maxpool2x2_dropout_bwd_bnstats_bf16(ctx, CBF(m->Av(bnn)), xb.ld, CBF(m->Dv(pn)), WBF(m->Dv(bnn)), gb.ld, m->P(bnn + "/gamma"), m->P(bnn + "/beta"),
                                                                m->wsd(m->off_bn_bsums) + so, xb.n, xb.h, xb.w, xb.c, m->drop_rate,
                                                                m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
        return unet_maxpool2x2_dropout_bwd_bnstats(ctx, m->A(bnn), xb.ld, m->D(pn), m->D(bnn), gb.ld, m->P(bnn + "/gamma"), m->P(bnn + "/beta"),
                                                   m->wsd(m->off_bn_bsums) + so, xb.n, xb.h, xb.w, xb.c, m->drop_rate,
                                                   m->drop_seed + (uint64_t)k * 0x9E3779B97F4A7C15ull, s);
      });
      bn_bwd(bnn, bnn, "c" + ks + "b", "c" + ks + "b", c, 1, true);
      }
      conv_bwd("c" + ks + "b", "c" + ks + "a", c, c, true, true);
      int cprev = (k == 1) ? m->in_ch : ENC[k - 2];
      conv_bwd("c" + ks + "a", k == 1 ? "" : "p" + std::to_string(k - 1), cprev, c, k > 1, false, "", k > 1 && ctx->opt_enc_bn_fused != 0);
    }
    flush_def();
    bucket("c1a/kernel", enc_split ? "bn2/beta" : "bn4/beta");
  }
#undef CBF
#undef WBF
}

// =========================================================================================
// U-Net++ (nested skips) -- /root/reference/Scripts/task1_unet_plus_plus.py:858-950.
//   encoder k : Conv(C,elu) -> Dropout(.2) -> Conv(C,elu) -> BN -> c_k -> MaxPool            (UPP:876-914)
//   node x    : ConvT(src) ++ skips -> conv_block = [Conv(elu) -> Dropout(.4) -> BN] x 2     (UPP:860-868, 888-924)
//   head      : Conv 1x1 sigmoid on x1_4                                                      (UPP:946-947)
// Dropout is fused into the conv epilogue (counter-based RNG, mask recomputed in backward); a tensor that feeds several
// concats is copied into each concat buffer (copy_slice), its gradient is the sum of the consumers' slices (accum_slices).
// =========================================================================================
struct PPNode { const char* name; int c; const char* src; std::vector<const char*> skips; };
const std::vector<PPNode>& pp_nodes() {
  static const std::vector<PPNode> v = {{"x1_2", 32, "c2", {"c1"}}, {"x2_2", 64, "c3", {"c2"}}, {"x1_3", 32, "x2_2", {"c1", "x1_2"}},
                                        {"x3_2", 128, "c4", {"c3"}}, {"x2_3", 64, "x3_2", {"c2", "x2_2"}},
                                        {"x1_4", 32, "x2_3", {"c1", "x1_2", "x1_3"}}};
  return v;
}
int pp_width(const std::string& t) {
  if (t[0] == 'c') return ENC[t[1] - '1'];
  for (auto& n : pp_nodes()) if (t == n.name) return n.c;
  return 0;
}
int pp_level(const std::string& t) { return t[1] - '1'; }        // c1,x1_* -> 0; c2,x2_* -> 1; ...
constexpr float PP_ENC_DROP = 0.2f, PP_BLOCK_DROP = 0.4f;
// creation order of the reference: enc1, enc2, x1_2, enc3, x2_2, x1_3, enc4, x3_2, x2_3, x1_4 (UPP:876-924)
const char* PP_ORDER[10] = {"e1", "e2", "x1_2", "e3", "x2_2", "x1_3", "e4", "x3_2", "x2_3", "x1_4"};

void build_layers_pp(unet_model* m) {
  auto& L = m->layers;
  for (const char* item : PP_ORDER) {
    std::string it = item;
    if (it[0] == 'e') {
      int k = it[1] - '0', c = ENC[k - 1], cin = k == 1 ? m->in_ch : ENC[k - 2];
      std::string ks = std::to_string(k);
      L.push_back({"c" + ks + "a", 0, cin, c}); L.push_back({"c" + ks + "b", 0, c, c}); L.push_back({"bn" + ks, 2, c, c});
    } else {
      const PPNode* nd = nullptr;
      for (auto& n : pp_nodes()) if (it == n.name) nd = &n;
      int ctot = nd->c; for (auto sk : nd->skips) ctot += pp_width(sk);
      L.push_back({"u" + it.substr(1), 1, pp_width(nd->src), nd->c});
      L.push_back({it + "a", 0, ctot, nd->c}); L.push_back({it + "abn", 2, nd->c, nd->c});
      L.push_back({it + "b", 0, nd->c, nd->c}); L.push_back({it + "bbn", 2, nd->c, nd->c});
    }
  }
  L.push_back({"out", 3, 32, 1});
  assign_param_offsets(m);
}

void plan_workspace_pp(unet_model* m) {
  Carver cv; cv.dt = m->dt;
  const int N = m->N;
  plan_scratch(m, cv);
  auto dims = [&](int lvl, int& hh, int& ww) { hh = m->H >> lvl; ww = m->W >> lvl; };
  int hh, ww;
  size_t wt0 = 0, wgb = 0, tmp_up = 0;
  for (int k = 1; k <= 4; ++k) {
    dims(k - 1, hh, ww); std::string ks = std::to_string(k); int c = ENC[k - 1];
    m->act["c" + ks + "a"] = mk(cv, N, hh, ww, c); m->act["c" + ks + "b"] = mk(cv, N, hh, ww, c);
    m->act["c" + ks] = mk(cv, N, hh, ww, c); m->act["bn" + ks] = m->act["c" + ks];
    if (k <= 3) m->act["p" + ks] = mk(cv, N, hh / 2, ww / 2, c);
  }
  for (auto& nd : pp_nodes()) {
    std::string nm = nd.name; dims(pp_level(nm), hh, ww);
    int ctot = nd.c; for (auto sk : nd.skips) ctot += pp_width(sk);
    Buf cat = mk(cv, N, hh, ww, ctot);
    m->act["cat_" + nm] = cat; m->act["u" + nm.substr(1)] = slice(cat, 0, nd.c);
    m->act[nm + "a"] = mk(cv, N, hh, ww, nd.c); m->act[nm + "abn"] = mk(cv, N, hh, ww, nd.c);
    m->act[nm + "b"] = mk(cv, N, hh, ww, nd.c); m->act[nm] = mk(cv, N, hh, ww, nd.c); m->act[nm + "bbn"] = m->act[nm];
  }
  for (auto& l : m->layers) if (l.kind == 0) wt0 = std::max(wt0, unet_conv3x3_w_ws_floats(l.cin, l.cout));
  m->off_wt = cv.take(wt0);
  // conv_block = [Conv -> Dropout -> BN] x 2 (UPP:860-868): the first BatchNorm feeds only the second conv -> folded into it, fp32 on the
  // F(2x2,3x3) kernels
  if (m->ctx->opt_bn_fold >= 2) {
    for (auto& nd : pp_nodes()) {
      const std::string nm = nd.name; const int c = nd.c;
      if (!wgrad_bn_fold_supported(c)) continue;
      if (m->dt) { if (!bf16_conv3x3_supported(c, c)) continue; }       // bf16 storage: the direct MFMA kernel has the same epilogues
      else if (!h2_conv3x3_selected(m->algo, c, c) || (c % 32)) continue;          // (every U-Net++ width is a multiple of 32)
      m->fold_off[nm + "b"] = cv.take(bn_fold_scratch_floats(c, c));
      m->folded_bn[nm + "abn"] = {nm + "a", c};
    }
  }
  m->ws_floats_infer = cv.cur;
  for (auto& kv : m->fold_off) { const int c = m->act.at(kv.first).c; m->fold_g_off[kv.first] = cv.take(wgrad_bn_fold_scratch_floats(N, c)); m->fold_c_off[kv.first] = cv.take((size_t)3 * c); }
  // ---- training: one dense gradient twin per activation buffer (aliases share it)
  std::map<size_t, std::string> seen;
  for (auto& kv : m->act) {
    const Buf& b = kv.second;
    if (b.chan_off != 0 || b.ld != b.c) continue;                 // slices handled below
    auto it = seen.find(b.off);
    if (it != seen.end()) { m->grad[kv.first] = m->grad.at(it->second); continue; }
    m->grad[kv.first] = mk(cv, b.n, b.h, b.w, b.c); seen[b.off] = kv.first;
  }
  for (auto& nd : pp_nodes()) { std::string nm = nd.name; m->grad["u" + nm.substr(1)] = slice(m->grad.at("cat_" + nm), 0, nd.c); }
  // A level-1 tensor that goes into several concats (c1, x1_2, x1_3: UPP:884-921) LIVES in the slice of its first consumer's concat -- the BatchNorm that produces it
  // writes there -- and is copied (the gradient twins above stay dense) only into the later ones: 3 of the 10 copies per step, the three biggest, go.  (The deeper ones are also ConvT inputs, read densely.)
  for (const char* t : {"c1", "x1_2", "x1_3"}) {
    bool done = false;
    for (auto& nd : pp_nodes()) {
      int off = nd.c;
      for (auto sk : nd.skips) {
        if (!done && std::string(sk) == t) { m->act[t] = slice(m->act.at("cat_" + std::string(nd.name)), off, pp_width(sk)); done = true; }
        off += pp_width(sk);
      }
    }
    if (std::string(t) == "c1") m->act["bn1"] = m->act[t]; else m->act[std::string(t) + "bbn"] = m->act[t];
  }
  for (auto& nd : pp_nodes()) {                                   // ConvT data-gradient staging buffer (largest source tensor)
    const Buf& sb = m->act.at(nd.src); tmp_up = std::max(tmp_up, (size_t)sb.n * sb.h * sb.w * sb.c);
  }
  { Buf t; t.off = cv.take(tmp_up); m->act["tmp_up"] = t; }
  for (auto& l : m->layers) {
    if (l.kind == 0) { const Buf& ob = m->act.at(l.name); wgb = std::max(wgb, m->dt ? unet_conv3x3_bwd_weights_ws_bytes_bf16(N, ob.h, ob.w, l.cin, l.cout) : unet_conv3x3_bwd_weights_ws_bytes(N, ob.h, ob.w, l.cin, l.cout)); }
    if (l.kind == 1) { const Buf& ob = m->act.at(l.name); wgb = std::max(wgb, m->dt ? bf16_convT_wgrad_ws_bytes(N, ob.h / 2, ob.w / 2, l.cin, l.cout) : unet_convT2x2_bwd_weights_ws_bytes(N, ob.h / 2, ob.w / 2, l.cin, l.cout)); }
  }
  m->wgrad_ws_bytes = wgb;
  m->off_wgrad_ws = cv.take((wgb + 3) / 4);
  m->ws_floats_train = cv.cur;
}

void build_programs_pp(unet_model* m) {
  unet_ctx* ctx = m->ctx;
  const int algo = m->algo;
  const double gcount = (double)m->world;
  const int dt = m->dt;
  const double eb = dt ? 2.0 : 4.0;           // bytes per stored activation element (roofline accounting)
  const size_t esz = dt ? 2 : 4;
#define CBF(p) static_cast<const unet_bf16*>(p)
#define WBF(p) static_cast<unet_bf16*>(p)
  auto& BW = m->prog[UNET_PROG_BWD];
  const size_t sums_bytes = (m->bn_sums_doubles + 4 + UNET_HEAD_SUMS) * sizeof(double);
  std::map<std::string, int> lidx;
  for (size_t i = 0; i < m->layers.size(); ++i) lidx[m->layers[i].name] = (int)i;
  auto seed_of = [=](const std::string& conv) { return (uint64_t)(lidx.at(conv) + 1) * 0x9E3779B97F4A7C15ull; };

  // ------------------------------------------------------------------ forward (train / infer)
  for (int training = 1; training >= 0; --training) {
    auto& F = m->prog[training ? UNET_PROG_FWD_TRAIN : UNET_PROG_FWD_INFER];
    auto& SY = m->syncref[training ? UNET_PROG_FWD_TRAIN : UNET_PROG_FWD_INFER];
    const int tr = training;
    ADD_OP(F, "zero_sums", 0, 0, { return unet_zero(ctx, m->wsf(m->off_bn_sums), sums_bytes, s); });
    auto conv = [&](const std::string& name, const std::string& in, int cin, int cout, float rate) {
      const Buf ob = m->act.at(name);
      const uint64_t sd = seed_of(name);
      double px = (double)ob.n * ob.h * ob.w;
      ADD_OP(F, "conv3x3_fwd:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cin + cout) + 4.0 * 9.0 * cin * cout, {
        const float r = (tr && m->drop_rate > 0.0f) ? rate : 0.0f;
        if (dt) {
          if (in.empty()) return first_conv_fwd_bf16(ctx, m, name, ob, cout, ACT_ELU, r, m->drop_seed + sd, s);
          return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(in)), m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, WBF(m->Av(name)), ob.n, ob.h, ob.w, cin, cout, ACT_ELU, r,
                                    m->drop_seed + sd, WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s);
        }
        const float* xin = in.empty() ? m->x : m->A(in);
        return conv3x3_fwd_dispatch(ctx, xin, m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, m->Aw(name), ob.n, ob.h, ob.w, cin, cout,
                                    ACT_ELU, r, m->drop_seed + sd, algo, s, m->wsf(m->off_wt), 0);
      });
    };
    // BN: stats -> [sync] -> finalize -> apply (or apply fused with the 2x2 pool when `pool` names the pooled output)
    auto bn = [&](const std::string& name, const std::string& in, const std::string& out, int c, const std::string& pool) {
      const Buf ib = m->act.at(in), ob = m->act.at(out);
      const int64_t pixels = (int64_t)ib.n * ib.h * ib.w;
      const size_t so = m->bn_sum_off.at(name), bo = m->bnp_off.at(name);
      if (training) {
        ADD_OP(F, "bn_stats:" + name, 0, eb * pixels * c, {
          if (dt) return unet_bn_stats_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
          return unet_bn_stats(ctx, m->A(in), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
        });
        SY.push_back({(int)F.size() - 1, 0, true, (m->off_bn_sums * 4) + so * 8, 2 * (int64_t)c});
        ADD_OP(F, "bn_finalize:" + name, 0, 0, {
          return unet_bn_finalize_train(ctx, m->wsd(m->off_bn_sums) + so, (double)pixels * gcount, m->P(name + "/gamma"), m->P(name + "/beta"),
                                        m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      } else {
        ADD_OP(F, "bn_finalize_infer:" + name, 0, 0, {
          return unet_bn_finalize_infer(ctx, m->P(name + "/gamma"), m->P(name + "/beta"), m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      }
      if (pool.empty() && m->folded_bn.count(name)) {
        // folded into the conv behind it: no normalised tensor
      } else if (pool.empty()) {
        ADD_OP(F, "bn_apply:" + name, 0, 2 * eb * pixels * c, {
          if (dt) return unet_bn_apply_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsf(bo), WBF(m->Av(out)), ob.ld, pixels, c, s);
          return unet_bn_apply(ctx, m->A(in), ib.ld, m->wsf(bo), m->Aw(out), ob.ld, pixels, c, s);
        });
      } else {
        ADD_OP(F, "bn_apply_pool:" + pool, 0, eb * 2.25 * pixels * c, {
          if (dt) return unet_bn_apply_maxpool_dropout_fwd_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsf(bo), WBF(m->Av(out)), ob.ld, WBF(m->Av(pool)), ib.n, ib.h, ib.w, c, 0.0f, 0, s);
          return unet_bn_apply_maxpool_dropout_fwd(ctx, m->A(in), ib.ld, m->wsf(bo), m->Aw(out), ob.ld, m->Aw(pool), ib.n, ib.h, ib.w, c, 0.0f, 0, s);
        });
      }
    };
    for (const char* item : PP_ORDER) {
      std::string it = item;
      if (it[0] == 'e') {
        int k = it[1] - '0', c = ENC[k - 1], cin = k == 1 ? m->in_ch : ENC[k - 2];
        std::string ks = std::to_string(k);
        conv("c" + ks + "a", k == 1 ? "" : "p" + std::to_string(k - 1), cin, c, PP_ENC_DROP);
        conv("c" + ks + "b", "c" + ks + "a", c, c, 0.0f);
        bn("bn" + ks, "c" + ks + "b", "c" + ks, c, k <= 3 ? "p" + ks : "");
      } else {
        const PPNode* nd = nullptr;
        for (auto& n : pp_nodes()) if (it == n.name) nd = &n;
        const std::string un = "u" + it.substr(1), src = nd->src, cat = "cat_" + it;
        const Buf sb = m->act.at(src), cb = m->act.at(cat);
        const int c = nd->c, csrc = pp_width(src);
        ADD_OP(F, "convT_fwd:" + un, 2.0 * 4 * csrc * c * nel(sb) / sb.c, eb * (nel(sb) + 4.0 * nel(sb) / sb.c * c), {
          if (dt) return k_convT_bf16_fwd(ctx, CBF(m->Av(src)), m->P(un + "/kernel"), m->P(un + "/bias"), WBF(m->Av(un)), cb.ld, sb.n, sb.h, sb.w, csrc, c, WBF(static_cast<void*>(m->wsf(m->off_wt))), s);
          return unet_convT2x2_fwd(ctx, m->A(src), m->P(un + "/kernel"), m->P(un + "/bias"), m->Aw(un), cb.ld, sb.n, sb.h, sb.w, csrc, c, algo, s);
        });
        int off = c;
        for (auto sk : nd->skips) {
          const std::string skn = sk; const Buf kb = m->act.at(skn); const int o = off, cw = kb.c;
          if (kb.off == cb.off && kb.chan_off == cb.chan_off + (size_t)o) { off += cw; continue; }          // (the tensor lives in this slice: plan_workspace_pp)
          ADD_OP(F, "copy_slice:" + skn + ">" + it, 0, 2 * eb * nel(kb), {
            if (dt) return unet_copy_slice_bf16(ctx, CBF(m->Av(skn)), kb.ld, WBF(m->Av(cat)) + o, cb.ld, (int64_t)kb.n * kb.h * kb.w, cw, s);
            return unet_copy_slice(ctx, m->A(skn), kb.ld, m->Aw(cat) + o, cb.ld, (int64_t)kb.n * kb.h * kb.w, cw, s);
          });
          off += cw;
        }
        conv(it + "a", cat, cb.c, c, PP_BLOCK_DROP);
        bn(it + "abn", it + "a", it + "abn", c, "");
        if (m->fold_off.count(it + "b")) {
          const std::string cn = it + "b", xn = it + "a", bnn = it + "abn";
          const Buf ob = m->act.at(cn);
          const size_t fo = m->fold_off.at(cn), bo = m->bnp_off.at(bnn);
          const uint64_t sd = seed_of(cn);
          ADD_OP(F, "bn_fold_prepare:" + cn, 2.0 * 9 * c * c * 2, 4.0 * 9 * c * c * 4, {
            return k_bn_fold_prepare(ctx, m->P(cn + "/kernel"), m->P(cn + "/bias"), m->wsf(bo), m->wsf(bo) + c, c, c, m->wsf(fo), s, dt != 0);
          });
          ADD_OP(F, "conv3x3_fwd:" + cn, 2.0 * 9 * c * c * (double)ob.n * ob.h * ob.w, eb * (double)ob.n * ob.h * ob.w * 2 * c + 4.0 * 9.0 * c * c, {
            const float r = (tr && m->drop_rate > 0.0f) ? PP_BLOCK_DROP : 0.0f;
            const float* tab = m->wsf(fo) + (size_t)9 * c * c;
            if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(xn)), m->wsf(fo), tab, reinterpret_cast<const unet_bf16*>(tab), MASK_BIAS_TAB, WBF(m->Av(cn)), ob.n, ob.h, ob.w, c, c, ACT_ELU, r,
                                              m->drop_seed + sd, WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s);
            int32_t e = k_h2_weights(ctx, m->P(cn + "/kernel"), m->wsf(m->off_wt), c, c, 0, s, m->wsf(bo));
            if (e) return e;
            return k_conv3x3_h2_fwd(ctx, m->A(xn), m->wsf(m->off_wt), tab, tab, MASK_BIAS_TAB, m->Aw(cn), ob.n, ob.h, ob.w, c, c, ACT_ELU, r, m->drop_seed + sd, s);
          });
        } else
        conv(it + "b", it + "abn", c, c, PP_BLOCK_DROP);
        bn(it + "bbn", it + "b", it, c, "");
      }
    }
    const Buf hb = m->act.at("x1_4");
    const int64_t hp = (int64_t)hb.n * hb.h * hb.w;
    ADD_OP(F, "head_fwd", 2.0 * 32 * hp, hp * (eb * 32 + 8.0), {
      if (!m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_fwd: p_out not set (unet_model_set_io)");
      if (dt) return unet_head_fwd_bf16(ctx, CBF(m->Av("x1_4")), m->P("out/kernel"), m->P("out/bias"), m->pout, m->yt, m->yt ? m->wsd(m->off_loss_sums) : nullptr, hp, hb.c, s);
      return unet_head_fwd(ctx, m->A("x1_4"), m->P("out/kernel"), m->P("out/bias"), m->pout, m->yt, m->yt ? m->wsd(m->off_loss_sums) : nullptr, hp, hb.c, s);
    });
    SY.push_back({(int)F.size() - 1, 1, true, m->off_loss_sums * 4, 4});
    ADD_OP(F, "loss_finalize", 0, 0, {
      if (!m->yt) return UNET_OK;
      return k_loss_finalize(ctx, m->wsd(m->off_loss_sums), (double)hp * gcount, m->wsf(m->off_loss_out), m->loss_out2, s);
    });
  }

  // ------------------------------------------------------------------ backward (reverse creation order)
  auto& SY = m->syncref[UNET_PROG_BWD];
  size_t bs_bytes = 0;
  for (auto& l : m->layers) if (l.kind == 2) bs_bytes += 2 * (size_t)l.cout * sizeof(double);
  // Gradients summed from several consumers (c1..c4, x1_2, x2_2, x1_3, x3_2, x2_3): every contribution is an accumulate-capable op (accum_slices, the ConvT data
  // gradient's accum, pool backward), so the FIRST one in program order overwrites and the rest accumulate -- no memset of 9 activation-sized buffers per step
  // (0.3 ms) and no read of zeros by the first writer.  first_touch(name) returns the accumulate flag to use and remembers the tensor.
  std::set<std::string> touched;
  auto first_touch = [&](const std::string& t) -> int { return touched.insert(t).second ? 0 : 1; };
  (void)esz;
  ADD_OP(BW, "zero_bwd_sums", 0, 0, {
    int32_t r = unet_zero(ctx, m->wsf(m->off_bn_bsums), bs_bytes, s); if (r) return r;
    return unet_zero(ctx, m->G("out/kernel"), (size_t)(m->tinfo.at("out/kernel").count + 1) * sizeof(float), s);
  });
  const Buf hb = m->act.at("x1_4");
  const int64_t hp = (int64_t)hb.n * hb.h * hb.w;
  ADD_OP(BW, "head_bwd", 4.0 * 32 * hp, hp * (eb * 64 + 8.0), {
    if (!m->yt || !m->pout) UNET_FAIL(ctx, UNET_E_STATE, "head_bwd: io not set");
    if (dt) return unet_head_bwd_bf16(ctx, CBF(m->Av("x1_4")), m->P("out/kernel"), m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, WBF(m->Dv("x1_4")),
                                      m->G("out/kernel"), m->G("out/bias"), hp, hb.c, 0, s);
    return unet_head_bwd(ctx, m->A("x1_4"), m->P("out/kernel"), m->pout, m->yt, m->wsd(m->off_loss_sums), (double)hp * gcount, m->D("x1_4"),
                         m->G("out/kernel"), m->G("out/bias"), hp, hb.c, 0, s);
  });
  // BN backward: dy = grad[dyname] (dense), x = act[xname] = dropout(elu(conv)) or elu(conv); dx = grad[xname] (pre-activation gradient)
  auto bn_bwd = [&](const std::string& name, const std::string& dyname, const std::string& xname, int c, int mask_mode, float rate, uint64_t sd) {
    const Buf gb = m->grad.at(dyname), xb = m->act.at(xname), db = m->grad.at(xname);
    const int64_t pixels = (int64_t)xb.n * xb.h * xb.w;
    const size_t so = m->bn_bsum_off.at(name), bo = m->bnp_off.at(name);
    ADD_OP(BW, "bn_bwd_stats:" + name, 0, 2 * eb * pixels * c, {
      int32_t r = dt ? unet_bn_bwd_stats_bf16(ctx, CBF(m->Dv(dyname)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s)
                     : unet_bn_bwd_stats(ctx, m->D(dyname), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s);
      if (r) return r;
      return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(name + "/gamma"), m->G(name + "/beta"), c, s);
    });
    SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
    ADD_OP(BW, "bn_bwd_apply:" + name, 0, 3 * eb * pixels * c, {
      const bool drop = m->drop_rate > 0.0f && rate > 0.0f;
      if (dt) return unet_bn_bwd_apply_bf16(ctx, CBF(m->Dv(dyname)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount,
                                            (mask_mode == MASK_ELU_DROP && !drop) ? MASK_ELU : mask_mode, drop ? rate : 0.0f, m->drop_seed + sd, WBF(m->Dv(xname)), db.ld, pixels, c, s);
      return unet_bn_bwd_apply(ctx, m->D(dyname), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount,
                               (mask_mode == MASK_ELU_DROP && !drop) ? MASK_ELU : mask_mode, drop ? rate : 0.0f, m->drop_seed + sd, m->D(xname), db.ld,
                               pixels, c, s);
    });
  };
  // conv backward: wgrad (x = act[in] or the network input, dy = grad[name]) + optional dgrad into grad[in] with the mask of act[in]
  auto conv_bwd = [&](const std::string& name, const std::string& in, int cin, int cout, bool want_dx, int mask_mode, float rate, uint64_t sd) {
    const Buf ob = m->act.at(name);
    const double px = (double)ob.n * ob.h * ob.w;
    ADD_OP(BW, "conv3x3_wgrad:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cin + cout) + 4.0 * 9.0 * cin * cout, {
      if (dt) {
        if (in.empty()) return first_conv_wgrad_bf16(ctx, m, name, ob, cout, s);
        return k_conv3x3_bf16_wgrad(ctx, CBF(m->Av(in)), CBF(m->Dv(name)), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, cin, cout, s);
      }
      const float* xin = in.empty() ? m->x : m->A(in);
      return conv3x3_wgrad_dispatch(ctx, xin, m->D(name), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                    ob.n, ob.h, ob.w, cin, cout, algo, s);
    });
    if (want_dx) {
      ADD_OP(BW, "conv3x3_dgrad:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cout + cin + (mask_mode ? cin : 0)) + 4.0 * 9.0 * cin * cout, {
        const bool drop = m->drop_rate > 0.0f && rate > 0.0f;
        const int mm = (mask_mode == MASK_ELU_DROP && !drop) ? MASK_ELU : mask_mode;
        if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(name)), m->P(name + "/kernel"), nullptr, mm ? CBF(m->Av(in)) : nullptr, mm, WBF(m->Dv(in)), ob.n, ob.h, ob.w, cout, cin, ACT_NONE,
                                          drop ? rate : 0.0f, m->drop_seed + sd, WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s);
        return unet_conv3x3_bwd_data(ctx, m->D(name), m->P(name + "/kernel"), mm ? m->A(in) : nullptr, mm, drop ? rate : 0.0f, m->drop_seed + sd, m->D(in),
                                     m->wsf(m->off_wt), ob.n, ob.h, ob.w, cin, cout, algo, s);
      });
    }
  };
  auto bucket = [&](const std::string& first, const std::string& last_tensor) {
    const TInfo a = m->tinfo.at(first), b = m->tinfo.at(last_tensor);
    SY.push_back({(int)BW.size() - 1, 3, false, (size_t)a.off * 4, b.off + b.count - a.off});
  };
  // The gradient of a tensor that went into several concats is the sum of those concats' gradient slices (each concat keeps its own gradient buffer): the slices
  // are only NOTED as the consumers finish and summed by ONE launch (up to 4 sources) right before the tensor's own backward starts -- x1_4, x1_3, x1_2 -> c1
  // as one read-modify-write of 268 MB instead of three
  std::map<std::string, std::vector<std::pair<std::string, int>>> pend;          // tensor -> (concat, channel offset)
  auto flush_pending = [&](const std::string& dstn) {
    auto pi = pend.find(dstn);
    if (pi == pend.end() || pi->second.empty()) return;
    const std::vector<std::pair<std::string, int>> srcv = pi->second; pend.erase(pi);
    const Buf kb = m->act.at(dstn);
    const int cw = kb.c, accs = first_touch(dstn), ns = (int)srcv.size();
    std::string tag;
    for (auto& sv : srcv) tag += (tag.empty() ? "" : "+") + sv.first.substr(4);
    ADD_OP(BW, "accum_slice:" + tag + ">" + dstn, 0, (ns + 1 + accs) * eb * nel(kb), {
      int32_t lds[4];
      if (dt) {
        const unet_bf16* srcs[4];
        for (int k = 0; k < ns; ++k) { srcs[k] = CBF(m->Dv(srcv[k].first)) + srcv[k].second; lds[k] = m->grad.at(srcv[k].first).ld; }
        return unet_accum_slices_bf16(ctx, srcs, lds, ns, WBF(m->Dv(dstn)), m->grad.at(dstn).ld, (int64_t)kb.n * kb.h * kb.w, cw, accs, s);
      }
      const float* srcs[4];
      for (int k = 0; k < ns; ++k) { srcs[k] = m->D(srcv[k].first) + srcv[k].second; lds[k] = m->grad.at(srcv[k].first).ld; }
      return unet_accum_slices(ctx, srcs, lds, ns, m->D(dstn), m->grad.at(dstn).ld, (int64_t)kb.n * kb.h * kb.w, cw, accs, s);
    });
  };
  for (int oi = 9; oi >= 0; --oi) {
    std::string it = PP_ORDER[oi];
    if (it[0] == 'e') {
      int k = it[1] - '0', c = ENC[k - 1], cin = k == 1 ? m->in_ch : ENC[k - 2];
      std::string ks = std::to_string(k), ca = "c" + ks + "a", cb = "c" + ks + "b";
      flush_pending("c" + ks);
      bn_bwd("bn" + ks, "c" + ks, cb, c, MASK_ELU, 0.0f, 0);
      conv_bwd(cb, ca, c, c, true, MASK_ELU_DROP, PP_ENC_DROP, seed_of(ca));
      conv_bwd(ca, k == 1 ? "" : "p" + std::to_string(k - 1), cin, c, k > 1, MASK_NONE, 0.0f, 0);
      if (k > 1) {
        const std::string pk = "p" + std::to_string(k - 1), ck = "c" + std::to_string(k - 1);
        const Buf xb = m->act.at(ck);
        const int accf = first_touch(ck);
        ADD_OP(BW, "pool_bwd:" + pk, 0, eb * 3.25 * nel(xb), {
          const int gld = m->grad.at(ck).ld;                      // (c1 lives in a concat slice, its gradient twin is dense)
          if (dt) return unet_maxpool2x2_dropout_bwd_bf16(ctx, CBF(m->Av(ck)), xb.ld, CBF(m->Dv(pk)), WBF(m->Dv(ck)), gld, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, accf, s);
          return unet_maxpool2x2_dropout_bwd(ctx, m->A(ck), xb.ld, m->D(pk), m->D(ck), gld, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, accf, s);
        });
      }
      if (k == 4) bucket("c4a/kernel", "x2_3bbn/beta");
      if (k == 3) bucket("c3a/kernel", "x1_3bbn/beta");
      if (k == 1) bucket("c1a/kernel", "x1_2bbn/beta");
    } else {
      const PPNode* nd = nullptr;
      for (auto& n : pp_nodes()) if (it == n.name) nd = &n;
      const std::string un = "u" + it.substr(1), src = nd->src, cat = "cat_" + it;
      const Buf sb = m->act.at(src), cb = m->act.at(cat);
      const int c = nd->c, csrc = pp_width(src);
      flush_pending(it);
      bn_bwd(it + "bbn", it, it + "b", c, MASK_ELU_DROP, PP_BLOCK_DROP, seed_of(it + "b"));
      if (m->fold_off.count(it + "b")) {
        // folded BatchNorm (abn -> conv b): weight gradient on the raw x, corrected; the BatchNorm's backward sums from W . dW_raw and S; its backward apply and the
        // ELU / dropout derivative of conv a in the data-gradient epilogue, which writes conv a's pre-activation gradient directly
        const std::string cn = it + "b", xn = it + "a", bnn = it + "abn";
        const Buf ob = m->act.at(cn);
        const double px = (double)ob.n * ob.h * ob.w;
        const size_t go = m->fold_g_off.at(cn), co = m->fold_c_off.at(cn), bo = m->bnp_off.at(bnn), so = m->bn_bsum_off.at(bnn);
        const uint64_t sda = seed_of(xn);
        ADD_OP(BW, "conv3x3_wgrad:" + cn, 2.0 * 9 * c * c * px, eb * px * 2 * c + 4.0 * 9.0 * c * c, {
          if (dt) return k_conv3x3_bf16_wgrad(ctx, CBF(m->Av(xn)), CBF(m->Dv(cn)), m->G(cn + "/kernel"), m->G(cn + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, c, c, s);
          return conv3x3_wgrad_dispatch(ctx, m->A(xn), m->D(cn), m->G(cn + "/kernel"), m->G(cn + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, c, c, algo, s);
        });
        ADD_OP(BW, "wgrad_bn_fold_fix:" + cn, 2.0 * 9 * c * c, 8.0 * 9 * c * c, {
          int32_t r = dt ? k_wgrad_bn_fold_fix_bf16(ctx, CBF(m->Dv(cn)), ob.n, ob.h, ob.w, c, c, m->wsf(bo), m->wsf(bo) + c, m->G(cn + "/kernel"), m->G(cn + "/bias"), m->wsf(go), s,
                                                    m->P(cn + "/kernel"), m->wsf(bo) + 2 * c, m->wsf(bo) + 3 * c, m->wsd(m->off_bn_bsums) + so)
                         : k_wgrad_bn_fold_fix(ctx, m->D(cn), ob.n, ob.h, ob.w, c, c, m->wsf(bo), m->wsf(bo) + c, m->G(cn + "/kernel"), m->G(cn + "/bias"), m->wsf(go), s, m->P(cn + "/kernel"),
                                               m->wsf(bo) + 2 * c, m->wsf(bo) + 3 * c, m->wsd(m->off_bn_bsums) + so);
          if (r) return r;
          return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(bnn + "/gamma"), m->G(bnn + "/beta"), c, s);
        });
        SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
        ADD_OP(BW, "conv3x3_dgrad_bn_bwd:" + cn, 2.0 * 9 * c * c * px, eb * px * 3 * c + 4.0 * 9.0 * c * c, {
          const bool drop = m->drop_rate > 0.0f;
          int32_t r = k_bn_bwd_coef(ctx, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, px * gcount, m->wsf(co), c, s);
          if (r) return r;
          if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(cn)), m->P(cn + "/kernel"), m->wsf(co), CBF(m->Av(xn)), drop ? MASK_BN_BWD_ELU_DROP : MASK_BN_BWD_ELU, WBF(m->Dv(xn)), ob.n, ob.h, ob.w, c, c,
                                            ACT_NONE, drop ? PP_BLOCK_DROP : 0.0f, m->drop_seed + sda, WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s);
          r = k_h2_weights(ctx, m->P(cn + "/kernel"), m->wsf(m->off_wt), c, c, 1, s);
          if (r) return r;
          return k_conv3x3_h2_fwd(ctx, m->D(cn), m->wsf(m->off_wt), m->wsf(co), m->A(xn), drop ? MASK_BN_BWD_ELU_DROP : MASK_BN_BWD_ELU, m->D(xn), ob.n, ob.h, ob.w, c, c, ACT_NONE,
                                    drop ? PP_BLOCK_DROP : 0.0f, m->drop_seed + sda, s);
        });
      } else {
      conv_bwd(it + "b", it + "abn", c, c, true, MASK_NONE, 0.0f, 0);
      bn_bwd(it + "abn", it + "abn", it + "a", c, MASK_ELU_DROP, PP_BLOCK_DROP, seed_of(it + "a"));
      }
      conv_bwd(it + "a", cat, cb.c, c, true, MASK_NONE, 0.0f, 0);
      const Buf ug = m->grad.at(un);
      ADD_OP(BW, "convT_wgrad:" + un, 2.0 * 4 * csrc * c * nel(sb) / sb.c, eb * (nel(sb) + 4.0 * nel(sb) / sb.c * c), {
        if (dt) return k_convT_bf16_wgrad(ctx, CBF(m->Av(src)), CBF(m->Dv(un)), ug.ld, m->G(un + "/kernel"), m->G(un + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, sb.n, sb.h, sb.w, csrc, c, s);
        return unet_convT2x2_bwd_weights(ctx, m->A(src), m->D(un), ug.ld, m->G(un + "/kernel"), m->G(un + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                         sb.n, sb.h, sb.w, csrc, c, algo, s);
      });
      const int accu = first_touch(src);
      ADD_OP(BW, "convT_dgrad:" + un, 2.0 * 4 * csrc * c * nel(sb) / sb.c, eb * (2 * nel(sb) + 4.0 * nel(sb) / sb.c * c), {
        float* tmp = m->wsf(m->act.at("tmp_up").off);
        if (dt) {
          unet_bf16* tb = WBF(static_cast<void*>(tmp));
          int32_t r = k_convT_bf16_dgrad(ctx, CBF(m->Dv(un)), ug.ld, m->P(un + "/kernel"), nullptr, tb, sb.n, sb.h, sb.w, csrc, c, WBF(static_cast<void*>(m->wsf(m->off_wt))), s);
          if (r) return r;
          const unet_bf16* srcs[1] = {tb}; const int32_t lds[1] = {csrc};
          return unet_accum_slices_bf16(ctx, srcs, lds, 1, WBF(m->Dv(src)), m->grad.at(src).ld, (int64_t)sb.n * sb.h * sb.w, csrc, accu, s);
        }
        int32_t r = unet_convT2x2_bwd_data(ctx, m->D(un), ug.ld, m->P(un + "/kernel"), nullptr, tmp, sb.n, sb.h, sb.w, csrc, c, algo, s);
        if (r) return r;
        const float* srcs[1] = {tmp}; const int32_t lds[1] = {csrc};
        return unet_accum_slices(ctx, srcs, lds, 1, m->D(src), m->grad.at(src).ld, (int64_t)sb.n * sb.h * sb.w, csrc, accu, s);
      });
      int off = c;
      for (auto sk : nd->skips) { pend[sk].push_back({cat, off}); off += m->act.at(sk).c; }
      if (it == "x1_4") bucket("u1_4/kernel", "out/bias");
    }
  }
#undef CBF
#undef WBF
}

// =========================================================================================
// Slice classifier -- /root/reference/Scripts/task2_covid19_classifcation.py:747-776 (`T2`), Sequential:
//   block k (C = 16, 32, 64): Conv(C,relu) -> BN -> Conv(C,relu) -> BN -> MaxPool            (T2:748-764)
//   Flatten -> Dense(32, relu) -> Dropout(.4) -> Dense(1, sigmoid)                            (T2:772-776)
// loss binary_crossentropy (+ class weights), metric f1 (T2:688-703, 829).  BN follows the ReLU conv directly, so every conv
// data-gradient lands on a BN output (no activation mask) and every BN backward carries the ReLU mask of its conv.
// =========================================================================================
const int CLS_C[3] = {16, 32, 64};
constexpr int CLS_HIDDEN = 32;
constexpr float CLS_DROP = 0.4f;

void build_layers_cls(unet_model* m) {
  auto& L = m->layers;
  int cprev = m->in_ch;
  for (int k = 1; k <= 3; ++k) {
    const int c = CLS_C[k - 1]; const std::string ks = std::to_string(k);
    L.push_back({"c" + ks + "a", 0, cprev, c}); L.push_back({"bn" + ks + "a", 2, c, c});
    L.push_back({"c" + ks + "b", 0, c, c}); L.push_back({"bn" + ks + "b", 2, c, c});
    cprev = c;
  }
  L.push_back({"fc1", 4, (m->H / 8) * (m->W / 8) * CLS_C[2], CLS_HIDDEN});
  L.push_back({"fc2", 4, CLS_HIDDEN, 1});
  assign_param_offsets(m);
}

void plan_workspace_cls(unet_model* m) {
  Carver cv; cv.dt = m->dt;
  const int N = m->N;
  plan_scratch(m, cv);
  int hh = m->H, ww = m->W;
  size_t wt0 = 0, wgb = 0;
  for (int k = 1; k <= 3; ++k) {
    const int c = CLS_C[k - 1]; const std::string ks = std::to_string(k);
    for (const char* sfx : {"a", "b"}) { m->act["c" + ks + sfx] = mk(cv, N, hh, ww, c); m->act["bn" + ks + sfx] = mk(cv, N, hh, ww, c); }
    m->act["p" + ks] = mk(cv, N, hh / 2, ww / 2, c);
    hh /= 2; ww /= 2;
  }
  { Buf b; b.off = cv.take((size_t)N * CLS_HIDDEN); b.ld = b.c = CLS_HIDDEN; b.n = N; b.h = b.w = 1; b.f32 = 1; m->act["h1"] = b; }      // the hidden units stay fp32 in every mode
  const int K = hh * ww * CLS_C[2];
  m->dense_ws_bytes = unet_dense_ws_bytes(N, K, CLS_HIDDEN);
  m->off_dense_ws = cv.take((m->dense_ws_bytes + 3) / 4);
  for (auto& l : m->layers) if (l.kind == 0) wt0 = std::max(wt0, unet_conv3x3_w_ws_floats(l.cin, l.cout));
  m->off_wt = cv.take(wt0);
  // Conv(relu) -> BN -> Conv (T2:748-751): the first BatchNorm of a block feeds only the block's second conv -> folded into it.  fp32: where that conv
  // runs on the h2 kernels in both directions
  if (m->ctx->opt_bn_fold >= 2) {
    for (int k = 1; k <= 3; ++k) {
      const int c = CLS_C[k - 1]; const std::string cn = "c" + std::to_string(k) + "b";
      if (!wgrad_bn_fold_supported(c)) continue;
      // (bf16 storage: all three blocks, the 16-channel one included -- its storage rounding is far above the fold's extra round-off; re-measured at the end of round 3
      //  with the fused statistics and the pixel-pair weight gradient in place: 58.0 k -> 64.8 k images/s at 224 x 224 x 256, where round 2 had measured -1 %)
      constexpr int bf16_too = 1;
      if (m->dt) { if (!bf16_too || !bf16_conv3x3_supported(c, c)) continue; }
      // 16-channel blocks (T2:748-751 at 224 x 224) only on request (UNET_OPT_BN_FOLD = 3): +15 % on the classifier step, but the raw first-block activations carry a
      // large mean, the folded pre-activations more round-off, and at 224 x 224 x 256 twice the ReLU flips: c1a/kernel 5.2e-3 from the fp64 answer instead of 2.8e-3
      else if (!h2_conv3x3_selected(m->algo, c, c) || ((c % 32) && m->ctx->opt_bn_fold < 3)) continue;
      m->fold_off[cn] = cv.take(bn_fold_scratch_floats(c, c));
      m->folded_bn["bn" + std::to_string(k) + "a"] = {"c" + std::to_string(k) + "a", c};
    }
  }
  m->ws_floats_infer = cv.cur;
  for (auto& kv : m->fold_off) { const int c = m->act.at(kv.first).c; m->fold_g_off[kv.first] = cv.take(wgrad_bn_fold_scratch_floats(N, c)); m->fold_c_off[kv.first] = cv.take((size_t)3 * c); }
  for (auto& kv : m->act) {
    const Buf& b = kv.second;
    if (kv.first == "h1") { Buf g = b; g.off = cv.take((size_t)N * CLS_HIDDEN); m->grad[kv.first] = g; }
    else m->grad[kv.first] = mk(cv, b.n, b.h, b.w, b.c);
  }
  for (auto& l : m->layers)
    if (l.kind == 0) { const Buf& ob = m->act.at(l.name); wgb = std::max(wgb, m->dt ? unet_conv3x3_bwd_weights_ws_bytes_bf16(N, ob.h, ob.w, l.cin, l.cout) : unet_conv3x3_bwd_weights_ws_bytes(N, ob.h, ob.w, l.cin, l.cout)); }
  m->wgrad_ws_bytes = wgb;
  m->off_wgrad_ws = cv.take((wgb + 3) / 4);
  m->ws_floats_train = cv.cur;
}

void build_programs_cls(unet_model* m) {
  unet_ctx* ctx = m->ctx;
  const int algo = m->algo;
  const double gcount = (double)m->world;
  const int dt = m->dt;
  const double eb = dt ? 2.0 : 4.0;
#define CBF(p) static_cast<const unet_bf16*>(p)
#define WBF(p) static_cast<unet_bf16*>(p)
  const size_t sums_bytes = (m->bn_sums_doubles + 4 + UNET_HEAD_SUMS) * sizeof(double);
  const Buf fb = m->act.at("p3");
  const int N = m->N, K = fb.h * fb.w * fb.c;
  const uint64_t fc_seed = 0xC2B2AE3D27D4EB4Full;

  for (int training = 1; training >= 0; --training) {
    auto& F = m->prog[training ? UNET_PROG_FWD_TRAIN : UNET_PROG_FWD_INFER];
    auto& SY = m->syncref[training ? UNET_PROG_FWD_TRAIN : UNET_PROG_FWD_INFER];
    const int tr = training;
    ADD_OP(F, "zero_sums", 0, 0, { return unet_zero(ctx, m->wsf(m->off_bn_sums), sums_bytes, s); });
    auto conv = [&](const std::string& name, const std::string& in, int cin, int cout) {
      const Buf ob = m->act.at(name);
      double px = (double)ob.n * ob.h * ob.w;
      ADD_OP(F, "conv3x3_fwd:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cin + cout) + 4.0 * 9.0 * cin * cout, {
        if (dt) {
          if (in.empty()) return first_conv_fwd_bf16(ctx, m, name, ob, cout, ACT_RELU, 0.0f, 0, s);
          return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(in)), m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, WBF(m->Av(name)), ob.n, ob.h, ob.w, cin, cout, ACT_RELU, 0.0f, 0,
                                    WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s);
        }
        const float* xin = in.empty() ? m->x : m->A(in);
        return conv3x3_fwd_dispatch(ctx, xin, m->P(name + "/kernel"), m->P(name + "/bias"), nullptr, MASK_NONE, m->Aw(name), ob.n, ob.h, ob.w, cin, cout,
                                    ACT_RELU, 0.0f, 0, algo, s, m->wsf(m->off_wt), 0);
      });
    };
    auto bn = [&](const std::string& name, const std::string& in, int c, const std::string& pool) {
      const Buf ib = m->act.at(in), ob = m->act.at(name);
      const int64_t pixels = (int64_t)ib.n * ib.h * ib.w;
      const size_t so = m->bn_sum_off.at(name), bo = m->bnp_off.at(name);
      if (training) {
        ADD_OP(F, "bn_stats:" + name, 0, eb * pixels * c, {
          if (dt) return unet_bn_stats_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
          return unet_bn_stats(ctx, m->A(in), ib.ld, m->wsd(m->off_bn_sums) + so, pixels, c, s);
        });
        SY.push_back({(int)F.size() - 1, 0, true, (m->off_bn_sums * 4) + so * 8, 2 * (int64_t)c});
        ADD_OP(F, "bn_finalize:" + name, 0, 0, {
          return unet_bn_finalize_train(ctx, m->wsd(m->off_bn_sums) + so, (double)pixels * gcount, m->P(name + "/gamma"), m->P(name + "/beta"),
                                        m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      } else {
        ADD_OP(F, "bn_finalize_infer:" + name, 0, 0, {
          return unet_bn_finalize_infer(ctx, m->P(name + "/gamma"), m->P(name + "/beta"), m->P(name + "/mean"), m->P(name + "/var"), m->wsf(bo), c, s);
        });
      }
      if (pool.empty() && m->folded_bn.count(name)) {
        // folded into the conv behind it: no normalised tensor
      } else if (pool.empty()) {
        ADD_OP(F, "bn_apply:" + name, 0, 2 * eb * pixels * c, {
          if (dt) return unet_bn_apply_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsf(bo), WBF(m->Av(name)), ob.ld, pixels, c, s);
          return unet_bn_apply(ctx, m->A(in), ib.ld, m->wsf(bo), m->Aw(name), ob.ld, pixels, c, s);
        });
      } else {
        ADD_OP(F, "bn_apply_pool:" + pool, 0, eb * 2.25 * pixels * c, {
          if (dt) return unet_bn_apply_maxpool_dropout_fwd_bf16(ctx, CBF(m->Av(in)), ib.ld, m->wsf(bo), WBF(m->Av(name)), ob.ld, WBF(m->Av(pool)), ib.n, ib.h, ib.w, c, 0.0f, 0, s);
          return unet_bn_apply_maxpool_dropout_fwd(ctx, m->A(in), ib.ld, m->wsf(bo), m->Aw(name), ob.ld, m->Aw(pool), ib.n, ib.h, ib.w, c, 0.0f, 0, s);
        });
      }
    };
    int cprev = m->in_ch;
    for (int k = 1; k <= 3; ++k) {
      const int c = CLS_C[k - 1]; const std::string ks = std::to_string(k);
      conv("c" + ks + "a", k == 1 ? "" : "p" + std::to_string(k - 1), cprev, c);
      bn("bn" + ks + "a", "c" + ks + "a", c, "");
      if (m->fold_off.count("c" + ks + "b")) {
        const std::string cn = "c" + ks + "b", xn = "c" + ks + "a", bnn = "bn" + ks + "a";
        const Buf ob = m->act.at(cn);
        const size_t fo = m->fold_off.at(cn), bo = m->bnp_off.at(bnn);
        ADD_OP(F, "bn_fold_prepare:" + cn, 2.0 * 9 * c * c * 2, 4.0 * 9 * c * c * 4, {
          return k_bn_fold_prepare(ctx, m->P(cn + "/kernel"), m->P(cn + "/bias"), m->wsf(bo), m->wsf(bo) + c, c, c, m->wsf(fo), s, dt != 0);
        });
        ADD_OP(F, "conv3x3_fwd:" + cn, 2.0 * 9 * c * c * (double)ob.n * ob.h * ob.w, eb * (double)ob.n * ob.h * ob.w * 2 * c + 4.0 * 9.0 * c * c, {
          const float* tab = m->wsf(fo) + (size_t)9 * c * c;
          if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Av(xn)), m->wsf(fo), tab, reinterpret_cast<const unet_bf16*>(tab), MASK_BIAS_TAB, WBF(m->Av(cn)), ob.n, ob.h, ob.w, c, c, ACT_RELU, 0.0f, 0,
                                            WBF(static_cast<void*>(m->wsf(m->off_wt))), 0, s);
          int32_t e = k_h2_weights(ctx, m->P(cn + "/kernel"), m->wsf(m->off_wt), c, c, 0, s, m->wsf(bo));
          if (e) return e;
          return k_conv3x3_h2_fwd(ctx, m->A(xn), m->wsf(m->off_wt), tab, tab, MASK_BIAS_TAB, m->Aw(cn), ob.n, ob.h, ob.w, c, c, ACT_RELU, 0.0f, 0, s);
        });
      } else
      conv("c" + ks + "b", "bn" + ks + "a", c, c);
      bn("bn" + ks + "b", "c" + ks + "b", c, "p" + ks);
      cprev = c;
    }
    ADD_OP(F, "dense_fwd:fc1", 2.0 * N * K * CLS_HIDDEN, eb * (double)N * K + 4.0 * (double)K * CLS_HIDDEN, {
      const float r = (tr && m->drop_rate > 0.0f) ? CLS_DROP : 0.0f;
      if (dt) return unet_dense_fwd_bf16(ctx, CBF(m->Av("p3")), m->P("fc1/kernel"), m->P("fc1/bias"), m->Aw("h1"), N, K, CLS_HIDDEN, ACT_RELU, r, m->drop_seed + fc_seed,
                                         m->wsf(m->off_dense_ws), m->dense_ws_bytes, s);
      return unet_dense_fwd(ctx, m->A("p3"), m->P("fc1/kernel"), m->P("fc1/bias"), m->Aw("h1"), N, K, CLS_HIDDEN, ACT_RELU, r, m->drop_seed + fc_seed,
                            m->wsf(m->off_dense_ws), m->dense_ws_bytes, s);
    });
    ADD_OP(F, "cls_head_fwd", 2.0 * N * CLS_HIDDEN, 4.0 * N * (CLS_HIDDEN + 2), {
      if (!m->pout) UNET_FAIL(ctx, UNET_E_STATE, "cls_head_fwd: p_out not set (unet_model_set_io)");
      return unet_cls_head_fwd(ctx, m->A("h1"), m->P("fc2/kernel"), m->P("fc2/bias"), m->pout, m->yt, m->cw0, m->cw1,
                               m->yt ? m->wsd(m->off_loss_sums) : nullptr, N, CLS_HIDDEN, s);
    });
    SY.push_back({(int)F.size() - 1, 1, true, m->off_loss_sums * 4, 4});
    ADD_OP(F, "loss_finalize", 0, 0, {
      if (!m->yt) return UNET_OK;
      return k_cls_loss_finalize(ctx, m->wsd(m->off_loss_sums), (double)N * gcount, m->wsf(m->off_loss_out), m->loss_out2, s);
    });
  }

  // ------------------------------------------------------------------ backward
  auto& BW = m->prog[UNET_PROG_BWD];
  auto& SY = m->syncref[UNET_PROG_BWD];
  size_t bs_bytes = 0;
  for (auto& l : m->layers) if (l.kind == 2) bs_bytes += 2 * (size_t)l.cout * sizeof(double);
  ADD_OP(BW, "zero_bwd_sums", 0, 0, { return unet_zero(ctx, m->wsf(m->off_bn_bsums), bs_bytes, s); });
  ADD_OP(BW, "cls_head_bwd", 4.0 * N * CLS_HIDDEN, 4.0 * N * (2 * CLS_HIDDEN + 2), {
    if (!m->yt || !m->pout) UNET_FAIL(ctx, UNET_E_STATE, "cls_head_bwd: io not set");
    return unet_cls_head_bwd(ctx, m->A("h1"), m->P("fc2/kernel"), m->pout, m->yt, m->cw0, m->cw1, (double)N * gcount,
                             m->drop_rate > 0.0f ? CLS_DROP : 0.0f, m->D("h1"), m->G("fc2/kernel"), m->G("fc2/bias"), m->G("fc1/bias"), N, CLS_HIDDEN, s);
  });
  ADD_OP(BW, "dense_bwd:fc1", 4.0 * N * K * CLS_HIDDEN, eb * 2.0 * N * K + 4.0 * 2.0 * K * CLS_HIDDEN, {
    if (dt) return unet_dense_bwd_bf16(ctx, CBF(m->Av("p3")), m->P("fc1/kernel"), m->D("h1"), WBF(m->Dv("p3")), m->G("fc1/kernel"), N, K, CLS_HIDDEN, s);
    return unet_dense_bwd(ctx, m->A("p3"), m->P("fc1/kernel"), m->D("h1"), m->D("p3"), m->G("fc1/kernel"), N, K, CLS_HIDDEN, s);
  });
  { const TInfo a = m->tinfo.at("fc1/kernel"), b = m->tinfo.at("fc2/bias"); SY.push_back({(int)BW.size() - 1, 3, false, (size_t)a.off * 4, b.off + b.count - a.off}); }
  auto bn_bwd = [&](const std::string& name, const std::string& xname, int c) {      // dy = grad[name], x = act[xname] (ReLU conv), dx = grad[xname]
    const Buf gb = m->grad.at(name), xb = m->act.at(xname), db = m->grad.at(xname);
    const int64_t pixels = (int64_t)xb.n * xb.h * xb.w;
    const size_t so = m->bn_bsum_off.at(name), bo = m->bnp_off.at(name);
    ADD_OP(BW, "bn_bwd_stats:" + name, 0, 2 * eb * pixels * c, {
      int32_t r = dt ? unet_bn_bwd_stats_bf16(ctx, CBF(m->Dv(name)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s)
                     : unet_bn_bwd_stats(ctx, m->D(name), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, pixels, c, s);
      if (r) return r;
      return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(name + "/gamma"), m->G(name + "/beta"), c, s);
    });
    SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
    ADD_OP(BW, "bn_bwd_apply:" + name, 0, 3 * eb * pixels * c, {
      if (dt) return unet_bn_bwd_apply_bf16(ctx, CBF(m->Dv(name)), gb.ld, CBF(m->Av(xname)), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, MASK_RELU, 0.0f, 0,
                                            WBF(m->Dv(xname)), db.ld, pixels, c, s);
      return unet_bn_bwd_apply(ctx, m->D(name), gb.ld, m->A(xname), xb.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, MASK_RELU, 0.0f, 0,
                               m->D(xname), db.ld, pixels, c, s);
    });
  };
  auto conv_bwd = [&](const std::string& name, const std::string& in, int cin, int cout, bool want_dx) {
    const Buf ob = m->act.at(name);
    const double px = (double)ob.n * ob.h * ob.w;
    ADD_OP(BW, "conv3x3_wgrad:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cin + cout) + 4.0 * 9.0 * cin * cout, {
      if (dt) {
        if (in.empty()) return first_conv_wgrad_bf16(ctx, m, name, ob, cout, s);
        return k_conv3x3_bf16_wgrad(ctx, CBF(m->Av(in)), CBF(m->Dv(name)), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, cin, cout, s);
      }
      const float* xin = in.empty() ? m->x : m->A(in);
      return conv3x3_wgrad_dispatch(ctx, xin, m->D(name), m->G(name + "/kernel"), m->G(name + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes,
                                    ob.n, ob.h, ob.w, cin, cout, algo, s);
    });
    if (want_dx) {
      ADD_OP(BW, "conv3x3_dgrad:" + name, 2.0 * 9 * cin * cout * px, eb * px * (cout + cin) + 4.0 * 9.0 * cin * cout, {
        if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(name)), m->P(name + "/kernel"), nullptr, nullptr, MASK_NONE, WBF(m->Dv(in)), ob.n, ob.h, ob.w, cout, cin, ACT_NONE, 0.0f, 0,
                                          WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s);
        return unet_conv3x3_bwd_data(ctx, m->D(name), m->P(name + "/kernel"), nullptr, MASK_NONE, 0.0f, 0, m->D(in), m->wsf(m->off_wt), ob.n, ob.h, ob.w, cin,
                                     cout, algo, s);
      });
    }
  };
  for (int k = 3; k >= 1; --k) {
    const int c = CLS_C[k - 1], cin = k == 1 ? m->in_ch : CLS_C[k - 2];
    const std::string ks = std::to_string(k), ca = "c" + ks + "a", cb = "c" + ks + "b", ba = "bn" + ks + "a", bb = "bn" + ks + "b", pk = "p" + ks;
    const Buf xb = m->act.at(bb);
    if (ctx->opt_enc_bn_fused) {
      // Conv -> BN -> MaxPool tail (T2:752-754): the BatchNorm's backward sums come from the pooled tensors alone (only arg-max elements carry gradient and their
      // BatchNorm output is the pooled activation), then pool backward + BatchNorm backward + ReLU mask in one pass
      const Buf cbuf = m->act.at(cb), cg = m->grad.at(cb);
      const int64_t pixels = (int64_t)xb.n * xb.h * xb.w;
      const size_t so = m->bn_bsum_off.at(bb), bo = m->bnp_off.at(bb);
      ADD_OP(BW, "pool_bwd_sums:" + pk, 0, eb * 0.5 * nel(xb), {
        int32_t r = dt ? unet_maxpool2x2_dropout_bwd_sums_bf16(ctx, CBF(m->Av(pk)), CBF(m->Dv(pk)), m->P(bb + "/gamma"), m->P(bb + "/beta"), m->wsd(m->off_bn_bsums) + so, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, s)
                       : unet_maxpool2x2_dropout_bwd_sums(ctx, m->A(pk), m->D(pk), m->P(bb + "/gamma"), m->P(bb + "/beta"), m->wsd(m->off_bn_bsums) + so, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, s);
        if (r) return r;
        return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(bb + "/gamma"), m->G(bb + "/beta"), c, s);
      });
      SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
      ADD_OP(BW, "bn_pool_bwd_apply:" + bb, 0, eb * 2.25 * nel(xb), {
        if (dt) return unet_bn_maxpool_bwd_apply_bf16(ctx, CBF(m->Av(cb)), cbuf.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, nullptr, 0, CBF(m->Dv(pk)), WBF(m->Dv(cb)), cg.ld,
                                                      xb.n, xb.h, xb.w, xb.c, 0.0f, 0, s);
        return unet_bn_maxpool_bwd_apply(ctx, m->A(cb), cbuf.ld, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, (double)pixels * gcount, nullptr, 0, m->D(pk), m->D(cb), cg.ld, xb.n, xb.h, xb.w, xb.c,
                                         0.0f, 0, s);
      });
    } else {
    ADD_OP(BW, "pool_bwd:" + pk, 0, eb * 2.25 * nel(xb), {
      if (dt) return unet_maxpool2x2_dropout_bwd_bf16(ctx, CBF(m->Av(bb)), xb.ld, CBF(m->Dv(pk)), WBF(m->Dv(bb)), xb.ld, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, 0, s);
      return unet_maxpool2x2_dropout_bwd(ctx, m->A(bb), xb.ld, m->D(pk), m->D(bb), xb.ld, xb.n, xb.h, xb.w, xb.c, 0.0f, 0, 0, s);
    });
    bn_bwd(bb, cb, c);
    }
    if (m->fold_off.count(cb)) {
      // folded BatchNorm (bn_ka -> conv kb): weight gradient on the raw x, corrected; backward sums from W . dW_raw and S; backward apply + ReLU mask of conv ka in the
      // data-gradient epilogue, which writes conv ka's output gradient directly
      const Buf ob = m->act.at(cb);
      const double px = (double)ob.n * ob.h * ob.w;
      const size_t go = m->fold_g_off.at(cb), co = m->fold_c_off.at(cb), bo = m->bnp_off.at(ba), so = m->bn_bsum_off.at(ba);
      ADD_OP(BW, "conv3x3_wgrad:" + cb, 2.0 * 9 * c * c * px, eb * px * 2 * c + 4.0 * 9.0 * c * c, {
        if (dt) return k_conv3x3_bf16_wgrad(ctx, CBF(m->Av(ca)), CBF(m->Dv(cb)), m->G(cb + "/kernel"), m->G(cb + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, c, c, s);
        return conv3x3_wgrad_dispatch(ctx, m->A(ca), m->D(cb), m->G(cb + "/kernel"), m->G(cb + "/bias"), m->wsf(m->off_wgrad_ws), m->wgrad_ws_bytes, ob.n, ob.h, ob.w, c, c, algo, s);
      });
      ADD_OP(BW, "wgrad_bn_fold_fix:" + cb, 2.0 * 9 * c * c, 8.0 * 9 * c * c, {
        int32_t r = dt ? k_wgrad_bn_fold_fix_bf16(ctx, CBF(m->Dv(cb)), ob.n, ob.h, ob.w, c, c, m->wsf(bo), m->wsf(bo) + c, m->G(cb + "/kernel"), m->G(cb + "/bias"), m->wsf(go), s,
                                                  m->P(cb + "/kernel"), m->wsf(bo) + 2 * c, m->wsf(bo) + 3 * c, m->wsd(m->off_bn_bsums) + so)
                       : k_wgrad_bn_fold_fix(ctx, m->D(cb), ob.n, ob.h, ob.w, c, c, m->wsf(bo), m->wsf(bo) + c, m->G(cb + "/kernel"), m->G(cb + "/bias"), m->wsf(go), s, m->P(cb + "/kernel"),
                                             m->wsf(bo) + 2 * c, m->wsf(bo) + 3 * c, m->wsd(m->off_bn_bsums) + so);
        if (r) return r;
        return unet_bn_bwd_param_grads(ctx, m->wsd(m->off_bn_bsums) + so, m->G(ba + "/gamma"), m->G(ba + "/beta"), c, s);
      });
      SY.push_back({(int)BW.size() - 1, 2, true, m->off_bn_bsums * 4 + so * 8, 2 * (int64_t)c});
      ADD_OP(BW, "conv3x3_dgrad_bn_bwd:" + cb, 2.0 * 9 * c * c * px, eb * px * 3 * c + 4.0 * 9.0 * c * c, {
        int32_t r = k_bn_bwd_coef(ctx, m->wsf(bo), m->wsd(m->off_bn_bsums) + so, px * gcount, m->wsf(co), c, s);
        if (r) return r;
        if (dt) return k_conv3x3_bf16_fwd(ctx, CBF(m->Dv(cb)), m->P(cb + "/kernel"), m->wsf(co), CBF(m->Av(ca)), MASK_BN_BWD_RELU, WBF(m->Dv(ca)), ob.n, ob.h, ob.w, c, c, ACT_NONE, 0.0f, 0,
                                          WBF(static_cast<void*>(m->wsf(m->off_wt))), 1, s);
        r = k_h2_weights(ctx, m->P(cb + "/kernel"), m->wsf(m->off_wt), c, c, 1, s);
        if (r) return r;
        return k_conv3x3_h2_fwd(ctx, m->D(cb), m->wsf(m->off_wt), m->wsf(co), m->A(ca), MASK_BN_BWD_RELU, m->D(ca), ob.n, ob.h, ob.w, c, c, ACT_NONE, 0.0f, 0, s);
      });
    } else {
    conv_bwd(cb, ba, c, c, true);
    bn_bwd(ba, ca, c);
    }
    conv_bwd(ca, k == 1 ? "" : "p" + std::to_string(k - 1), cin, c, k > 1);
  }
  { const TInfo a = m->tinfo.at("c1a/kernel"), b = m->tinfo.at("bn3b/beta"); SY.push_back({(int)BW.size() - 1, 3, false, (size_t)a.off * 4, b.off + b.count - a.off}); }
#undef CBF
#undef WBF
}

void resolve_sync(unet_model* m) {
  for (int p = 0; p < 3; ++p) {
    m->sync[p].clear();
    for (auto& r : m->syncref[p]) {
      unet_sync_point sp;
      sp.after_op = r.after_op; sp.kind = r.kind; sp.count = r.count; sp.reserved = 0;
      sp.use_op = r.kind == 3 ? (int)m->prog[p].size() : (r.use_op > r.after_op ? r.use_op : r.after_op + 1);
      sp.ptr = r.in_ws ? (void*)(m->ws ? m->ws + r.off_bytes : nullptr) : (void*)(m->grads ? (char*)m->grads + r.off_bytes : nullptr);
      m->sync[p].push_back(sp);
    }
  }
}

}  // namespace

extern "C" {

int32_t unet_model_create(unet_ctx* ctx, int32_t arch, int32_t in_ch, int32_t n, int32_t h, int32_t w, int32_t world_size,
                          int32_t conv_algo, int32_t dtype, unet_model** out) {
  if (!ctx || !out) return UNET_E_ARG;
  *out = nullptr;
  if (dtype != UNET_DTYPE_F32 && dtype != UNET_DTYPE_BF16) UNET_FAIL(ctx, UNET_E_ARG, "model_create: unknown dtype %d", dtype);
  if (arch != UNET_ARCH_UNET && arch != UNET_ARCH_UNETPP && arch != UNET_ARCH_CLASSIFIER) UNET_FAIL(ctx, UNET_E_ARG, "model_create: unknown arch %d", arch);
  const int mult = arch == UNET_ARCH_UNET ? 16 : 8;      // 4 pool levels (T1:862-880) / 3 used pool levels (UPP: p4 is dead)
  if (in_ch < 1 || n < 1 || h < mult || w < mult || (h % mult) || (w % mult) || world_size < 1)
    UNET_FAIL(ctx, UNET_E_SHAPE, "model_create: need n>=1 and h,w multiples of %d; got n=%d h=%d w=%d", mult, n, h, w);
  unet_model* m = new unet_model();
  m->ctx = ctx; m->arch = arch; m->in_ch = in_ch; m->N = n; m->H = h; m->W = w; m->world = world_size; m->algo = conv_algo; m->dt = dtype;
  if (arch == UNET_ARCH_UNET) { build_layers(m); plan_workspace(m); build_programs(m); }
  else if (arch == UNET_ARCH_UNETPP) { build_layers_pp(m); plan_workspace_pp(m); build_programs_pp(m); }
  else { build_layers_cls(m); plan_workspace_cls(m); build_programs_cls(m); }
  arm_bn_statistics(m);
  *out = m;
  return UNET_OK;
}

void unet_model_destroy(unet_model* m) {
  if (!m) return;
  delete m;
}
int64_t unet_model_param_count(const unet_model* m) { return m ? m->n_params : 0; }
int64_t unet_model_state_count(const unet_model* m) { return m ? m->n_state : 0; }
int32_t unet_model_dtype(const unet_model* m) { return m ? m->dt : UNET_E_ARG; }
size_t unet_model_workspace_bytes(const unet_model* m, int32_t training) {
  if (!m) return 0;
  return (training ? m->ws_floats_train : m->ws_floats_infer) * sizeof(float);
}

int32_t unet_model_tensor_info(const unet_model* m, const char* name, int32_t* is_state, int64_t* offset, int64_t* count) {
  if (!m || !name) return UNET_E_ARG;
  auto it = m->tinfo.find(name);
  if (it == m->tinfo.end()) return UNET_E_ARG;
  if (is_state) *is_state = it->second.is_state;
  if (offset) *offset = it->second.off;
  if (count) *count = it->second.count;
  return UNET_OK;
}

int32_t unet_model_bind(unet_model* m, float* params, float* grads, float* adam_m, float* adam_v, float* bn_state, void* workspace,
                        size_t workspace_bytes) {
  if (!m || !params || !bn_state || !workspace) UNET_FAIL(m ? m->ctx : nullptr, UNET_E_ARG, "model_bind: null buffer");
  size_t need = (grads ? m->ws_floats_train : m->ws_floats_infer) * sizeof(float);
  if (workspace_bytes < need) UNET_FAIL(m->ctx, UNET_E_ARG, "model_bind: workspace %zu < %zu bytes", workspace_bytes, need);
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) || (reinterpret_cast<uintptr_t>(params) & 15)) UNET_FAIL(m->ctx, UNET_E_ARG, "model_bind: buffers must be 256-B (workspace) / 16-B aligned");
  m->params = params; m->grads = grads; m->adam_m = adam_m; m->adam_v = adam_v; m->state = bn_state;
  m->ws = static_cast<char*>(workspace); m->ws_bytes = workspace_bytes;
  resolve_sync(m);
  return UNET_OK;
}

int32_t unet_model_set_io(unet_model* m, const float* x, const float* y_true, float* p_out) {
  if (!m) return UNET_E_ARG;
  m->x = x; m->yt = y_true; m->pout = p_out;
  return UNET_OK;
}

int32_t unet_model_set_loss_out(unet_model* m, float* loss_out) {
  if (!m) return UNET_E_ARG;
  m->loss_out2 = loss_out;
  return UNET_OK;
}

int32_t unet_model_set_dropout(unet_model* m, float rate, uint64_t seed) {
  if (!m || rate < 0 || rate >= 1) return UNET_E_ARG;
  m->drop_rate = rate; m->drop_seed = seed;
  return UNET_OK;
}

int32_t unet_model_set_class_weights(unet_model* m, float w0, float w1) {
  if (!m || m->arch != UNET_ARCH_CLASSIFIER || !(w0 >= 0) || !(w1 >= 0)) return UNET_E_ARG;
  m->cw0 = w0; m->cw1 = w1;
  return UNET_OK;
}

int32_t unet_model_num_ops(const unet_model* m, int32_t prog) { return (!m || prog < 0 || prog > 2) ? UNET_E_ARG : (int32_t)m->prog[prog].size(); }

int32_t unet_model_sync_points(const unet_model* m, int32_t prog, unet_sync_point* out, int32_t cap) {
  if (!m || prog < 0 || prog > 2) return UNET_E_ARG;
  int n = (int)m->sync[prog].size();
  if (out) for (int i = 0; i < n && i < cap; ++i) out[i] = m->sync[prog][i];
  return n;
}

int32_t unet_model_run(unet_model* m, int32_t prog, int32_t begin, int32_t end, void* stream) {
  if (!m || prog < 0 || prog > 2) return UNET_E_ARG;
  unet_ctx* ctx = m->ctx;
  auto& P = m->prog[prog];
  if (begin < 0 || end > (int)P.size() || begin > end) UNET_FAIL(ctx, UNET_E_ARG, "model_run: bad op range [%d,%d) of %zu", begin, end, P.size());
  if (!m->ws || !m->params) UNET_FAIL(ctx, UNET_E_STATE, "model_run: buffers not bound");
  if (prog == UNET_PROG_BWD && !m->grads) UNET_FAIL(ctx, UNET_E_STATE, "model_run: backward needs a grads buffer");
  if (!m->x) UNET_FAIL(ctx, UNET_E_STATE, "model_run: input not set");
  hipStream_t s = as_stream(stream);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->profiling) { UNET_HIP(ctx, hipEventCreate(&e0)); UNET_HIP(ctx, hipEventCreate(&e1)); }
  // (weight gradients on a second stream beside the data-gradient chain were measured 1-3 % SLOWER on this chip -- two matrix kernels sharing the CUs
  //  cost each other more than their gaps are worth; round 4: even the ~25 tiny launches of the split weight images, forked onto a side stream beside the
  //  HBM-bound first-layer kernel, cost 0.05 ms per step (same-box A/B 16.56 -> 16.61 ms) -- so every launch of a model stays on the caller's stream)
  if (begin == 0 && prog != UNET_PROG_BWD) { ctx->stats_req_c = 0; ctx->stats_in_slots = nullptr; ctx->stats_in_slots_c = 0; ctx->signs_req = nullptr; ctx->k_slices_ok = 0; }   // a program starts clean whatever an aborted run left armed
  for (int i = begin; i < end; ++i) {
    if (ctx->profiling) UNET_HIP(ctx, hipEventRecord(e0, s));
    const int32_t r = P[i].run(s);
    if (r) { ctx->err = P[i].name + ": " + ctx->err; if (ctx->profiling) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); } return r; }
    if (ctx->profiling) {
      UNET_HIP(ctx, hipEventRecord(e1, s));
      UNET_HIP(ctx, hipEventSynchronize(e1));
      float ms = 0; UNET_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
      P[i].ms += ms; P[i].calls += 1;
    }
  }
  if (ctx->profiling) { (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); }
  return UNET_OK;
}

const float* unet_model_loss_ptr(const unet_model* m) { return (m && m->ws) ? m->wsf(m->off_loss_out) : nullptr; }

int32_t unet_model_tap(const unet_model* m, const char* name, int32_t grad, const void** ptr, int32_t* ld, int32_t* n, int32_t* h,
                       int32_t* w, int32_t* c) {
  if (!m || !name || !m->ws) return UNET_E_ARG;
  auto& mp = grad ? m->grad : m->act;
  auto it = mp.find(name);
  if (it == mp.end()) return UNET_E_ARG;
  const Buf& b = it->second;
  const auto fb = m->folded_bn.find(name);
  if (grad && fb != m->folded_bn.end()) {                  // dz is consumed in the data-gradient epilogue, never stored
    const std::string nm = name;
    std::string conv = nm.rfind("bn", 0) == 0 ? "c" + nm.substr(2) + "a" : nm.substr(0, nm.size() - 3) + "b";      // U-Net: bnK -> cKa; U-Net++: <node>abn -> <node>b
    if (m->arch == UNET_ARCH_CLASSIFIER) conv = "c" + nm.substr(2, 1) + "b";                                             // classifier: bnKa -> cKb
    if (m->fold_c_off.count(conv)) return UNET_E_STATE;
  }
  const std::string nm_ = name;
  if (!grad && m->c9b_virtual && nm_ == "c9b") {
    // the fused head launch did not store this tensor: a tap recomputes it from c9a with the weight image of the last forward (bias: the current parameter)
    const Buf& xb = m->act.at("c9a");
    int32_t r = k_conv3x3_h2_fwd(m->ctx, m->A("c9a"), m->wsf(m->wprep_f.at("c9b")), m->P("c9b/bias"), nullptr, MASK_NONE, const_cast<float*>(m->A("c9b")), xb.n, xb.h, xb.w, xb.c, b.c, ACT_RELU,
                                 0.0f, 0, nullptr);
    if (r) return r;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return UNET_E_HIP;
  }
  if (grad && m->head_bwd_fused && nm_ == "c9b") return UNET_E_STATE;          // the head's backward leaves the {dz, mask} stream there, not the tensor (HEAD_BWD_FUSED)
  if (!grad && m->skip_raw && nm_.size() == 3 && nm_[0] == 'b' && nm_[2] >= '1' && nm_[2] <= '4') {
    // skip_raw: an encoder BatchNorm's output is never stored -- a tap materialises it from the raw conv output (inside the concat) into the tap scratch
    const std::string cb = std::string("c") + nm_[2] + "b";
    const Buf& xb = m->act.at(cb);
    int32_t r = unet_bn_apply(m->ctx, m->A(cb), xb.ld, m->wsf(m->bnp_off.at(name)), const_cast<float*>(m->A(name)), b.ld, (int64_t)b.n * b.h * b.w, b.c, nullptr);
    if (r) return r;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return UNET_E_HIP;
  }
  if (!grad && fb != m->folded_bn.end()) {
    // the programs never write this tensor (its BatchNorm is folded into the next conv): a tap materialises it from the layer's input and the
    // scale / shift of the last forward, on the null stream, and waits for it (skip_raw: the composite map -- the concat holds the raw encoder output)
    const Buf& xb = m->act.at(fb->second.first);
    const float* bnp_ = m->skip_raw && m->bn_comp_off.count(name) ? m->wsf(m->bn_comp_off.at(name)) : m->wsf(m->bnp_off.at(name));
    int32_t r = m->dt ? unet_bn_apply_bf16(m->ctx, static_cast<const unet_bf16*>(m->Av(fb->second.first)), xb.ld, m->wsf(m->bnp_off.at(name)), static_cast<unet_bf16*>(m->Av(name)), b.ld,
                                           (int64_t)b.n * b.h * b.w, fb->second.second, nullptr)
                      : unet_bn_apply(m->ctx, m->A(fb->second.first), xb.ld, bnp_, const_cast<float*>(m->A(name)), b.ld, (int64_t)b.n * b.h * b.w, fb->second.second, nullptr);
    if (r) return r;
    if (hipStreamSynchronize(nullptr) != hipSuccess) return UNET_E_HIP;
  }
  if (ptr) *ptr = m->bufptr(b);
  if (ld) *ld = b.ld; if (n) *n = b.n; if (h) *h = b.h; if (w) *w = b.w; if (c) *c = b.c;
  return UNET_OK;
}

/* bytes per element of a tap (4, or 2 for bf16-stored tensors) */
int32_t unet_model_tap_elem_bytes(const unet_model* m, const char* name, int32_t grad) {
  if (!m || !name) return UNET_E_ARG;
  auto& mp = grad ? m->grad : m->act;
  auto it = mp.find(name);
  return it == mp.end() ? UNET_E_ARG : m->elem_bytes(it->second);
}

int32_t unet_model_op_info(const unet_model* m, int32_t prog, int32_t op, const char** name, double* flops, double* bytes, double* ms,
                           int64_t* calls) {
  if (!m || prog < 0 || prog > 2 || op < 0 || op >= (int)m->prog[prog].size()) return UNET_E_ARG;
  const Op& o = m->prog[prog][op];
  if (name) *name = o.name.c_str();
  if (flops) *flops = o.flops; if (bytes) *bytes = o.bytes; if (ms) *ms = o.ms; if (calls) *calls = o.calls;
  return UNET_OK;
}

int32_t unet_model_reset_timers(unet_model* m) {
  if (!m) return UNET_E_ARG;
  for (auto& P : m->prog) for (auto& o : P) { o.ms = 0; o.calls = 0; }
  return UNET_OK;
}

}  // extern "C"
