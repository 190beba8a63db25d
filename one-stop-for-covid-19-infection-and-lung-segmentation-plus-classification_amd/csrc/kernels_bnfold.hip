// BatchNormalization -> Conv2D(3x3) without the normalised tensor (decoder blocks T1:888-889, 895-896, 902-903, 909-910; the
// conv_block of U-Net++ UPP:860-868; the classifier's Conv -> BN -> Conv T2:748-751): the small kernels around the convolution --
//   forward : scaled weights w * scale[c], the bias per border class (k_bn_fold_prepare)
//   backward: border sums of dy, the tap sums S, the weight-gradient correction, the BatchNorm's backward sums from W . dW_raw and W . S
//             (k_wgrad_bn_fold_fix), the three coefficients of dx = K0 dz + K1 x + K2 (k_bn_bwd_coef)
// The convolutions themselves are the h2 kernels (kernels_conv_h2.hip: MASK_BIAS_TAB / MASK_BN_BWD* epilogues) or the bf16 ones.
#include <algorithm>

#include "common.h"

// ---- BatchNorm on the INPUT of a conv folded into the conv (k_bn_fold_prepare) ------------------------------------------------------------
namespace {
__global__ void bn_fold_scale_kernel(const float* __restrict__ w, const float* __restrict__ scale, float* __restrict__ ws, int cin, int cout4, long long total4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / cout4) % cin);
    const float sc = scale[c];
    float4 v = reinterpret_cast<const float4*>(w)[i];
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
    reinterpret_cast<float4*>(ws)[i] = v;
  }
}
// T[tap][o] = sum_c w[tap][c][o] * shift[c] in two levels: grid (cout / 64, 9 taps, slices of 32 input channels) writes part[slice][tap][o]
// (256 threads = 64 couts x 4 channel sub-slices), then one workgroup per 64 couts sums the slices and combines the taps per border class:
// table[cls][o] = bias[o] + sum of T[a][b][o] over the taps (a, b) that stay inside the image for class cls.
__global__ __launch_bounds__(256) void bn_fold_taps_kernel(const float* __restrict__ w, const float* __restrict__ shift, float* __restrict__ part, int cin, int cout) {
  __shared__ float s_part[4][64];
  const int o = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6, tap = blockIdx.y, c0 = blockIdx.z * 32;
  float acc = 0.f;
  if (o < cout) {
    const float* p = w + ((long long)tap * cin) * cout + o;
    for (int c = c0 + sl; c < min(c0 + 32, cin); c += 4) acc = fmaf(p[(long long)c * cout], shift[c], acc);
  }
  s_part[sl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sl == 0 && o < cout)
    part[((long long)blockIdx.z * 9 + tap) * cout + o] = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}
// grid cout / 64, 576 threads = 9 taps x 64 couts: every thread sums ITS tap over the slices, the nine meet in LDS, the first 64 threads write the 16 classes
__global__ __launch_bounds__(576) void bn_fold_table_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ table, int slices, int cout) {
  __shared__ float s_t[9][64];
  const int l = threadIdx.x & 63, tap = threadIdx.x >> 6, o = blockIdx.x * 64 + l;
  float a = 0.f;
  if (o < cout) for (int k = 0; k < slices; ++k) a += part[((long long)k * 9 + tap) * cout + o];
  s_t[tap][l] = a;
  __syncthreads();
  if (tap != 0 || o >= cout) return;
  const float b0 = bias ? bias[o] : 0.f;
#pragma unroll
  for (int cls = 0; cls < 16; ++cls) {
    const int rs = cls >> 2, cs = cls & 3;
    float v = 0.f;
#pragma unroll
    for (int ta = 0; ta < 3; ++ta)
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const bool out = (ta == 0 && (rs & 1)) || (ta == 2 && (rs & 2)) || (tb == 0 && (cs & 1)) || (tb == 2 && (cs & 2));
        if (!out) v += s_t[ta * 3 + tb][l];
      }
    table[(long long)cls * cout + o] = b0 + v;
  }
}
// coefficients of the BatchNorm backward as one affine map of (dz, x): dx = sc (dz - k1 - xhat k2) = K0 dz + K1 x + K2, k1 = sum dz / count, k2 = sum dz xhat / count
__global__ void bn_bwd_coef_kernel(const float* __restrict__ bnp, const double* __restrict__ sums, double inv_count, float* __restrict__ coef, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sc = bnp[c], mean = bnp[2 * C + c], istd = bnp[3 * C + c];
  const double k1 = sums[c] * inv_count, k2 = sums[C + c] * inv_count;
  coef[c] = (float)sc; coef[C + c] = (float)(-sc * istd * k2); coef[2 * C + c] = (float)(sc * (mean * istd * k2 - k1));
}
}  // namespace

namespace {
__global__ void bn_compose_kernel(const float* __restrict__ dec, const float* __restrict__ enc, float* __restrict__ comp, int C) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, C2 = 2 * C;
  if (j >= C2) return;
  const float sd = dec[j], td = dec[C2 + j];
  if (j < C) { comp[j] = sd; comp[C2 + j] = td; comp[2 * C2 + j] = 1.0f; comp[3 * C2 + j] = 0.0f; }
  else {
    const float se = enc[j - C], te = enc[C + (j - C)];
    comp[j] = sd * se; comp[C2 + j] = fmaf(sd, te, td); comp[2 * C2 + j] = se; comp[3 * C2 + j] = te;
  }
}
}  // namespace
int32_t k_bn_compose(unet_ctx* ctx, const float* bnp_dec, const float* bnp_enc, float* comp, int c, hipStream_t s) {
  if (!bnp_dec || !bnp_enc || !comp || c < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_compose: bad args");
  hipLaunchKernelGGL(bn_compose_kernel, dim3((unsigned)((2 * c + 127) / 128)), dim3(128), 0, s, bnp_dec, bnp_enc, comp, c);
  UNET_CHECK_LAUNCH(ctx, "bn_compose");
  return UNET_OK;
}

int32_t k_bn_bwd_coef(unet_ctx* ctx, const float* bnp, const double* sums, double count, float* coef, int c, hipStream_t s) {
  if (!bnp || !sums || !coef || c < 1 || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_coef: bad args");
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((unsigned)((c + 127) / 128)), dim3(128), 0, s, bnp, sums, 1.0 / count, coef, c);
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_coef");
  return UNET_OK;
}

size_t bn_fold_scratch_floats(int cin, int cout) { return (size_t)9 * cin * cout + 16 * (size_t)cout + (size_t)((cin + 31) / 32) * 9 * cout; }
// scratch = [w_scaled 9*cin*cout][table 16*cout][tap partials]: bn_fold_scratch_floats(cin, cout)
// want_scaled = false: only the border-class bias table (the h2 weight image multiplies by scale[c] itself: k_h2_weights(..., cs = scale))
int32_t k_bn_fold_prepare(unet_ctx* ctx, const float* w, const float* bias, const float* scale, const float* shift, int cin, int cout, float* scratch, hipStream_t s, bool want_scaled) {
  if (!w || !scale || !shift || !scratch || cin < 1 || cout < 4 || (cout & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_fold_prepare: bad args");
  float* w_scaled = scratch; float* table = scratch + (size_t)9 * cin * cout; float* part = table + 16 * (size_t)cout;
  const long long total4 = 9LL * cin * cout / 4;
  const int slices = (cin + 31) / 32;
  if (want_scaled) hipLaunchKernelGGL(bn_fold_scale_kernel, dim3((unsigned)std::min<long long>((total4 + 255) / 256, 2048)), dim3(256), 0, s, w, scale, w_scaled, cin, cout / 4, total4);
  hipLaunchKernelGGL(bn_fold_taps_kernel, dim3((unsigned)((cout + 63) / 64), 9, (unsigned)slices), dim3(256), 0, s, w, shift, part, cin, cout);
  hipLaunchKernelGGL(bn_fold_table_kernel, dim3((unsigned)((cout + 63) / 64)), dim3(576), 0, s, part, bias, table, slices, cout);
  UNET_CHECK_LAUNCH(ctx, "bn_fold_prepare");
  return UNET_OK;
}

// ---- weight gradient of a conv whose input BatchNorm was folded into it (common.h: k_bn_fold_prepare) --------------------------------------
// The gradient kernels ran on the raw x; with z = scale[c] * x + shift[c] inside the image and 0 outside,
//   dW[a][b][c][o] = scale[c] * dW_raw[a][b][c][o] + shift[c] * S[a][b][o],   S[a][b][o] = sum of dy[.., o] over the pixels whose tap (a, b) stays inside
// = db[o] minus the border row / column the tap excludes plus the corner both exclude.
namespace {
// out[n * SEG + seg][8][C]: sums of dy over (segment seg of) row 0, row H-1, column 0, column W-1 and the corners (0,0) (0,W-1) (H-1,0) (H-1,W-1)
// of image n.  grid (8 * SEG, N)
constexpr int BORDER_SEG = 4;
template <typename T>
__global__ __launch_bounds__(256) void border_sums_kernel(const T* __restrict__ dy, float* __restrict__ out, int H, int W, int C) {
  __shared__ float s_p[256];
  const int kind = blockIdx.x & 7, seg = blockIdx.x >> 3, n = blockIdx.y;
  const int co = threadIdx.x % C, sl = threadIdx.x / C, nsl = 256 / C;
  const T* img = dy + (long long)n * H * W * C;
  float acc = 0.f;
  if (kind < 2) {
    const T* r = img + (long long)(kind ? H - 1 : 0) * W * C;
    const int per = (W + BORDER_SEG - 1) / BORDER_SEG, j1 = min(W, (seg + 1) * per);
    for (int j = seg * per + sl; j < j1; j += nsl) acc += ld1(r + (long long)j * C + co);
  } else if (kind < 4) {
    const T* q = img + (long long)(kind == 3 ? W - 1 : 0) * C;
    const int per = (H + BORDER_SEG - 1) / BORDER_SEG, i1 = min(H, (seg + 1) * per);
    for (int i = seg * per + sl; i < i1; i += nsl) acc += ld1(q + (long long)i * W * C + co);
  } else if (sl == 0 && seg == 0) { const int i = (kind & 2) ? H - 1 : 0, j = (kind & 1) ? W - 1 : 0; acc = ld1(img + ((long long)i * W + j) * C + co); }
  s_p[threadIdx.x] = acc;
  __syncthreads();
  if (sl == 0) { for (int k = 1; k < nsl; ++k) acc += s_p[k * C + co]; out[(((long long)n * BORDER_SEG + seg) * 8 + kind) * C + co] = acc; }
}
// S[tap][o] from db and the border sums (NS = images x segments of them).  grid (9, C / 64), 1024 threads = 64 channels x 16 slices of NS (a batch of 256
// images is 1024 entries: with 4 slices the three dependent loads per entry made this 0.2 ms of pure latency)
__global__ __launch_bounds__(1024) void fold_tap_sums_kernel(const float* __restrict__ border, const float* __restrict__ db, float* __restrict__ S, int NS, int C) {
  __shared__ float s_r[3][16][64];
  const int tap = blockIdx.x, a = tap / 3, b = tap - a * 3, l = threadIdx.x & 63, sl = threadIdx.x >> 6, o = blockIdx.y * 64 + l;
  const int er = a == 0 ? 0 : (a == 2 ? 1 : -1), ec = b == 0 ? 0 : (b == 2 ? 1 : -1);      // excluded row (0: first, 1: last), column
  float kr = 0.f, kc = 0.f, kk = 0.f;
  if (o < C)
    for (int n = sl; n < NS; n += 16) {
      const float* p = border + (long long)n * 8 * C + o;
      if (er >= 0) kr += p[er * C];
      if (ec >= 0) kc += p[(2 + ec) * C];
      if (er >= 0 && ec >= 0) kk += p[(4 + er * 2 + ec) * C];
    }
  s_r[0][sl][l] = kr; s_r[1][sl][l] = kc; s_r[2][sl][l] = kk;
  __syncthreads();
  if (sl == 0 && o < C) {
    kr = kc = kk = 0.f;
    for (int k = 0; k < 16; ++k) { kr += s_r[0][k][l]; kc += s_r[1][k][l]; kk += s_r[2][k][l]; }          // fixed order
    S[tap * C + o] = ((db[o] - kr) - kc) + kk;
  }
}
// (sum dz, sum dz * xhat) of the folded BatchNorm's backward WITHOUT reading dz or x: dz is the data gradient of this conv, so per input channel c
//   sum_p dz_c(p)        = sum_{tap,o} W[tap][c][o] * S[tap][o]
//   sum_p dz_c(p) x_c(p) = sum_{tap,o} W[tap][c][o] * dW_raw[tap][c][o]        (dW_raw = the weight gradient on the raw x, before the correction)
// and sum dz * xhat = istd * (sum dz x - mean * sum dz).  grid cin, 256 threads over the 9 * cout (tap, o) pairs; added into sums[2 * cin] (doubles).
// pre_s / pre_t (or null): the BatchNorm's input was x = pre_s x_raw + pre_t of the tensor the weight gradient ran on, so sum dz x = pre_s sum dz x_raw + pre_t sum dz
__global__ __launch_bounds__(256) void fold_bn_bwd_sums_kernel(const float* __restrict__ w, const float* dw_raw /* may be dw_fix */, const float* __restrict__ S,
                                                               const float* __restrict__ mean, const float* __restrict__ istd, double* __restrict__ sums, int cin, int cout,
                                                               const float* __restrict__ pre_s, const float* __restrict__ pre_t,
                                                               // round 4: the same sweep over channel c's (tap, o) pairs also applies the weight-gradient correction (fold_fix_kernel's
                                                               // expression) and leaves the BatchNorm's parameter gradients -- two launches less per decoder level
                                                               float* dw_fix, const float* __restrict__ scale, const float* __restrict__ shift, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta) {
  __shared__ double s_a[256], s_b[256];
  const int c = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < 9 * cout; i += 256) {
    const int tap = i / cout, o = i - tap * cout;
    const long long j = ((long long)tap * cin + c) * cout + o;
    const float wv = w[j], dr = dw_raw[j], sv = S[i];
    a = fmaf(wv, dr, a); b = fmaf(wv, sv, b);
    if (dw_fix) dw_fix[j] = fmaf(scale[c], dr, shift[c] * sv);
  }
  s_a[threadIdx.x] = (double)a; s_b[threadIdx.x] = (double)b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s_a[threadIdx.x] += s_a[threadIdx.x + st]; s_b[threadIdx.x] += s_b[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double dzx = pre_s ? (double)pre_s[c] * s_a[0] + (double)pre_t[c] * s_b[0] : s_a[0];
    const double t1 = sums[c] + s_b[0], t2 = sums[cin + c] + (double)istd[c] * (dzx - (double)mean[c] * s_b[0]);
    sums[c] = t1; sums[cin + c] = t2;
    if (dgamma) { dbeta[c] = (float)t1; dgamma[c] = (float)t2; }          // (bn_bwd_param_grads_kernel: the LOCAL sums -- gradients are summed over ranks with their bucket)
  }
}
__global__ void fold_fix_kernel(float* __restrict__ dw, const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ S, int cin, int cout4,
                                long long total4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % cout4); const long long r = i / cout4; const int c = (int)(r % cin), tap = (int)(r / cin);
    const float sc = scale[c], sh = shift[c];
    const float4 sv = reinterpret_cast<const float4*>(S)[tap * cout4 + q];
    float4 v = reinterpret_cast<float4*>(dw)[i];
    v.x = fmaf(sc, v.x, sh * sv.x); v.y = fmaf(sc, v.y, sh * sv.y); v.z = fmaf(sc, v.z, sh * sv.z); v.w = fmaf(sc, v.w, sh * sv.w);
    reinterpret_cast<float4*>(dw)[i] = v;
  }
}
}  // namespace

bool wgrad_bn_fold_supported(int cout) { return cout >= 4 && cout <= 256 && 256 % cout == 0; }
size_t wgrad_bn_fold_scratch_floats(int n, int cout) { return (size_t)(n > 0 ? n : 0) * BORDER_SEG * 8 * cout + 9 * (size_t)cout; }
// w / mean / istd / bn_bwd_sums (all or none): also accumulate the folded BatchNorm's backward sums (sum dz, sum dz * xhat) -- from W, the raw dw and S
template <typename T>
static int32_t wgrad_bn_fold_fix_impl(unet_ctx* ctx, const T* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                                      float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums, const float* pre_s = nullptr,
                                      const float* pre_t = nullptr, float* dgamma = nullptr, float* dbeta = nullptr) {
  if (!dy || !scale || !shift || !dw || !db || !scratch || !wgrad_bn_fold_supported(cout)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad_bn_fold_fix: bad args (cout=%d)", cout);
  float* border = scratch; float* S = scratch + (size_t)n * BORDER_SEG * 8 * cout;
  hipLaunchKernelGGL(border_sums_kernel<T>, dim3(8 * BORDER_SEG, (unsigned)n), dim3(256), 0, s, dy, border, h, wd, cout);
  hipLaunchKernelGGL(fold_tap_sums_kernel, dim3(9, (unsigned)((cout + 63) / 64)), dim3(1024), 0, s, border, db, S, n * BORDER_SEG, cout);
  if (bn_bwd_sums) {
    if (!w || !mean || !istd) UNET_FAIL(ctx, UNET_E_ARG, "wgrad_bn_fold_fix: the BatchNorm backward sums need w, mean, istd");
    hipLaunchKernelGGL(fold_bn_bwd_sums_kernel, dim3((unsigned)cin), dim3(256), 0, s, w, dw, S, mean, istd, bn_bwd_sums, cin, cout, pre_s, pre_t, dw, scale, shift, dgamma, dbeta);
    UNET_CHECK_LAUNCH(ctx, "wgrad_bn_fold_fix");
    return UNET_OK;                                          // (the correction rode along)
  }
  const long long total4 = 9LL * cin * cout / 4;
  hipLaunchKernelGGL(fold_fix_kernel, dim3((unsigned)std::min<long long>((total4 + 255) / 256, 2048)), dim3(256), 0, s, dw, scale, shift, S, cin, cout / 4, total4);
  UNET_CHECK_LAUNCH(ctx, "wgrad_bn_fold_fix");
  return UNET_OK;
}
int32_t k_wgrad_bn_fold_fix(unet_ctx* ctx, const float* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                            float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums, const float* pre_s, const float* pre_t,
                            float* dgamma, float* dbeta) {
  if ((pre_s == nullptr) != (pre_t == nullptr) || (dgamma == nullptr) != (dbeta == nullptr) || (dgamma && !bn_bwd_sums)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad_bn_fold_fix: pre_s / pre_t and dgamma / dbeta go in pairs");
  return wgrad_bn_fold_fix_impl(ctx, dy, n, h, wd, cin, cout, scale, shift, dw, db, scratch, s, w, mean, istd, bn_bwd_sums, pre_s, pre_t, dgamma, dbeta);
}
int32_t k_wgrad_bn_fold_fix_bf16(unet_ctx* ctx, const unet_bf16* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                                 float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums) {
  return wgrad_bn_fold_fix_impl(ctx, dy, n, h, wd, cin, cout, scale, shift, dw, db, scratch, s, w, mean, istd, bn_bwd_sums);
}
