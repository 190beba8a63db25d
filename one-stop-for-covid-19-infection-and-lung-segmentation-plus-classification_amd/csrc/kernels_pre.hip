// The two cleanly defined image steps in front of the U-Net path, on the GPU:
//   min-max normalisation of a slice + np.uint8(x * 255)      T1:336-337, T1:165-166
//   clahe_enhancer: cv2.createCLAHE(clipLimit=3.0, tileGridSize=(8,8)).apply      T1:163-171
//   uint8 / 255 back to a [0,1] float image                                        T1:520
// Byte / integer work, bit-exact against oracle/preprocess_oracle.py (which restates OpenCV's clahe.cpp: parity unpinned, cv2 is not
// in this image).  HBM-bound and tiny (3 bytes per pixel): one pass for the tile LUTs, one for the blend.  This file is compiled with
// -ffp-contract=off (csrc/Makefile): OpenCV's C++ evaluates the blend without fused multiply-adds, and HIP's __fmul_rn / __fadd_rn are
// plain operators that the default -ffp-contract=fast would still fuse (it changed 1 pixel in 4000 by one level on 17-pixel tiles).
#include "common.h"

namespace {
constexpr int TPB = 256;

__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

__global__ void minmax_init_kernel(unsigned* mm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { mm[2 * i] = 0xFFFFFFFFu; mm[2 * i + 1] = 0u; }
}
// per-image min / max: blockIdx.y = image, ordered-integer atomics (exact: min / max do not round)
__global__ __launch_bounds__(TPB) void minmax_kernel(const float* __restrict__ img, unsigned* __restrict__ mm, long long pixels) {
  const float* p = img + (long long)blockIdx.y * pixels;
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < pixels; i += (long long)gridDim.x * TPB) { const float v = p[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(mm + 2 * blockIdx.y, f2ord(mn)); atomicMax(mm + 2 * blockIdx.y + 1, f2ord(mx)); }
}
// np.uint8((img - min)/(max - min) * 255): float64 arithmetic as numpy does on the reference's float64 slices, truncation
__global__ __launch_bounds__(TPB) void norm_to_u8_kernel(const float* __restrict__ img, const unsigned* __restrict__ mm, uint8_t* __restrict__ out, long long pixels) {
  const double mn = (double)ord2f(mm[2 * blockIdx.y]), mx = (double)ord2f(mm[2 * blockIdx.y + 1]);
  const double d = mx - mn;
  const float* p = img + (long long)blockIdx.y * pixels; uint8_t* q = out + (long long)blockIdx.y * pixels;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < pixels; i += (long long)gridDim.x * TPB)
    q[i] = (uint8_t)(int)(__dmul_rn(__ddiv_rn((double)p[i] - mn, d), 255.0));
}
__global__ __launch_bounds__(TPB) void unit_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, long long count) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < count; i += (long long)gridDim.x * TPB) out[i] = (uint8_t)(int)(__dmul_rn((double)img[i], 255.0));
}
__global__ __launch_bounds__(TPB) void u8_to_unit_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long long count) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < count; i += (long long)gridDim.x * TPB) dst[i] = (float)__ddiv_rn((double)src[i], 255.0);
}

// one workgroup per (image, tile): histogram -> clip -> redistribute -> cdf -> LUT (clahe.cpp CLAHE_CalcLut_Body)
__global__ __launch_bounds__(TPB) void clahe_lut_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ lut, int H, int W, int tiles_x, int tiles_y,
                                                       int th, int tw, int clip, float lut_scale) {
  __shared__ int s_hist[256];
  __shared__ int s_scan[256];
  __shared__ int s_red[TPB / 64];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x % (tiles_x * tiles_y), n = blockIdx.x / (tiles_x * tiles_y);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const uint8_t* img = src + (long long)n * H * W;
  s_hist[tid] = 0;
  __syncthreads();
  const int area = th * tw;
  for (int i = tid; i < area; i += TPB) {
    const int r = i / tw, c = i - r * tw;
    int y = ty * th + r, x = tx * tw + c;
    if (y >= H) y = 2 * (H - 1) - y;                       // BORDER_REFLECT_101 padding up to a multiple of the grid
    if (x >= W) x = 2 * (W - 1) - x;
    atomicAdd(&s_hist[img[(long long)y * W + x]], 1);
  }
  __syncthreads();
  int hcount = s_hist[tid];
  if (clip > 0) {
    int excess = hcount > clip ? hcount - clip : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) excess += __shfl_xor(excess, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = excess;
    __syncthreads();
    const int clipped = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const int batch = clipped / 256, residual = clipped - batch * 256;
    hcount = (hcount > clip ? clip : hcount) + batch;
    if (residual > 0) {
      const int step = 256 / residual > 1 ? 256 / residual : 1;
      if (tid % step == 0 && tid / step < residual) ++hcount;        // bins 0, step, 2 step, ... get one of the `residual` counts each
    }
  }
  s_scan[tid] = hcount;
  __syncthreads();
#pragma unroll
  for (int o = 1; o < 256; o <<= 1) {                       // inclusive prefix sum (exact: integers)
    const int v = tid >= o ? s_scan[tid - o] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  float v = rintf(__fmul_rn((float)s_scan[tid], lut_scale));          // saturate_cast<uchar>(sum * lutScale): round half to even, clamp
  v = fminf(fmaxf(v, 0.0f), 255.0f);
  lut[((long long)n * tiles_x * tiles_y + tile) * 256 + tid] = (uint8_t)(int)v;
}

// bilinear blend of the four neighbouring tile LUTs (clahe.cpp CLAHE_Interpolation_Body), one thread per pixel
__global__ __launch_bounds__(TPB) void clahe_blend_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ lut, uint8_t* __restrict__ dst, int N, int H,
                                                         int W, int tiles_x, int tiles_y, float inv_tw, float inv_th) {
  const long long total = (long long)N * H * W;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int x = (int)(i % W); const long long t = i / W; const int y = (int)(t % H); const int n = (int)(t / H);
    const float txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f), tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f);
    int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
    const float xa = __fsub_rn(txf, (float)tx1), ya = __fsub_rn(tyf, (float)ty1);
    const float xa1 = __fsub_rn(1.0f, xa), ya1 = __fsub_rn(1.0f, ya);
    int tx2 = tx1 + 1, ty2 = ty1 + 1;
    tx1 = tx1 < 0 ? 0 : tx1; ty1 = ty1 < 0 ? 0 : ty1;
    tx2 = tx2 > tiles_x - 1 ? tiles_x - 1 : tx2; ty2 = ty2 > tiles_y - 1 ? tiles_y - 1 : ty2;
    const int v = src[i];
    const uint8_t* L = lut + (long long)n * tiles_x * tiles_y * 256 + v;
    const float l11 = (float)L[(ty1 * tiles_x + tx1) * 256], l12 = (float)L[(ty1 * tiles_x + tx2) * 256];
    const float l21 = (float)L[(ty2 * tiles_x + tx1) * 256], l22 = (float)L[(ty2 * tiles_x + tx2) * 256];
    const float top = __fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), bot = __fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa));
    float r = rintf(__fadd_rn(__fmul_rn(top, ya1), __fmul_rn(bot, ya)));
    r = fminf(fmaxf(r, 0.0f), 255.0f);
    dst[i] = (uint8_t)(int)r;
  }
}

inline unsigned blocks_for(long long items, int cap) { long long b = (items + TPB - 1) / TPB; return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b)); }
// ---- cv2.resize on uint8 crops (resize.cpp hal::resize, 8-bit single channel), one thread per destination pixel ------------------------
// Every thread rebuilds the few table entries it needs (OpenCV precomputes them per row / column): same doubles, same float32
// operations in the same order, so the result is the bit pattern oracle/preprocess_oracle.py:resize_u8 produces.
constexpr int RESIZE_RECTS = 96;                                   // rectangles per launch, passed by value
struct resize_rects { int x[RESIZE_RECTS], y[RESIZE_RECTS], w[RESIZE_RECTS], h[RESIZE_RECTS]; };

// computeResizeAreaTab for one destination index: optional partial cell on the left (s1 - 1), full cells [s1, s2), optional partial on the right (s2)
struct area_span { int s1, s2; float a_left, a_mid, a_right; bool left, right; };
__device__ __forceinline__ area_span area_cells(int d, double scale, int ssize) {
  area_span r;
  const double f1 = d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  r.s1 = s1; r.s2 = s2;
  r.left = (s1 - f1) > 1e-3;
  r.a_left = (float)((s1 - f1) / cell);
  r.a_mid = (float)(1.0 / cell);
  r.right = (f2 - s2) > 1e-3;
  r.a_right = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
  return r;
}
__device__ __forceinline__ int sat_short(float v) { return max(-32768, min(32767, __float2int_rn(v))); }
// the bilinear coefficient pair of one destination index (INTER_RESIZE_COEF_BITS = 11); `clamp` = the x-axis treatment of the borders
__device__ __forceinline__ void linear_coef(int d, int ssize, double scale, double inv_scale, bool area_mode, bool clamp, int* so, int* c0, int* c1) {
  int s; float f;
  if (!area_mode) { f = (float)((d + 0.5) * scale - 0.5); s = (int)floorf(f); f = f - (float)s; }
  else { s = (int)floor(d * scale); f = (float)((d + 1) - (s + 1) * inv_scale); f = f <= 0.f ? 0.f : f - floorf(f); }
  if (clamp) {
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  }
  *so = s; *c0 = sat_short((1.f - f) * 2048.f); *c1 = sat_short(f * 2048.f);
}

__global__ __launch_bounds__(TPB) void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, resize_rects rc, int img0, uint8_t* __restrict__ dst, int dh,
                                                       int dw, int dst_ld, int dst_x0, int interp) {
  const int li = blockIdx.y;
  const int rx = rc.x[li], ry = rc.y[li], rw = rc.w[li], rh = rc.h[li];
  const int idx = blockIdx.x * TPB + threadIdx.x;
  if (idx >= dh * dw) return;
  const int dy = idx / dw, dx = idx - dy * dw;
  const uint8_t* S = src + (long long)(img0 + li) * sh * sw + (long long)ry * sw + rx;
  const double inv_x = (double)dw / rw, inv_y = (double)dh / rh;
  const double scale_x = 1.0 / inv_x, scale_y = 1.0 / inv_y;
  const int isx = (int)rint(scale_x), isy = (int)rint(scale_y);
  const bool fast = fabs(scale_x - isx) < 2.220446049250313e-16 && fabs(scale_y - isy) < 2.220446049250313e-16;
  if (interp == UNET_RESIZE_LINEAR && fast && isx == 2 && isy == 2) interp = UNET_RESIZE_AREA;
  int out;
  if (interp == UNET_RESIZE_AREA && scale_x >= 1.0 && scale_y >= 1.0) {
    if (fast) {
      int sum = 0;
      for (int ky = 0; ky < isy; ++ky)
        for (int kx = 0; kx < isx; ++kx) sum += S[(long long)(dy * isy + ky) * sw + dx * isx + kx];
      out = __float2int_rn((float)sum * (1.f / (float)(isx * isy)));
    } else {
      const area_span xs = area_cells(dx, scale_x, rw), ys = area_cells(dy, scale_y, rh);
      float acc = 0.f; bool first = true;
      const int y_lo = ys.left ? ys.s1 - 1 : ys.s1, y_hi = ys.right ? ys.s2 : ys.s2 - 1;
      for (int sy = y_lo; sy <= y_hi; ++sy) {
        const float beta = (sy < ys.s1) ? ys.a_left : (sy < ys.s2 ? ys.a_mid : ys.a_right);
        const uint8_t* row = S + (long long)sy * sw;
        float buf = 0.f;
        if (xs.left) buf = buf + (float)row[xs.s1 - 1] * xs.a_left;
        for (int sx = xs.s1; sx < xs.s2; ++sx) buf = buf + (float)row[sx] * xs.a_mid;
        if (xs.right) buf = buf + (float)row[xs.s2] * xs.a_right;
        acc = first ? beta * buf : acc + beta * buf;
        first = false;
      }
      out = __float2int_rn(acc);
    }
  } else {
    const bool area_mode = interp == UNET_RESIZE_AREA;
    int sx, a0, a1, sy, b0, b1;
    linear_coef(dx, rw, scale_x, inv_x, area_mode, true, &sx, &a0, &a1);
    linear_coef(dy, rh, scale_y, inv_y, area_mode, false, &sy, &b0, &b1);
    const int x1 = min(sx + 1, rw - 1);
    const int r0 = max(0, min(sy, rh - 1)), r1 = max(0, min(sy + 1, rh - 1));
    const uint8_t* p0 = S + (long long)r0 * sw; const uint8_t* p1 = S + (long long)r1 * sw;
    const int h0 = p0[sx] * a0 + p0[x1] * a1, h1 = p1[sx] * a0 + p1[x1] * a1;
    out = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  }
  dst[(long long)(img0 + li) * dh * dst_ld + (long long)dy * dst_ld + dst_x0 + dx] = (uint8_t)max(0, min(255, out));
}
}  // namespace

extern "C" {

size_t unet_pre_minmax_ws_bytes(int32_t n) { return (size_t)(n > 0 ? n : 0) * 2 * sizeof(unsigned); }

int32_t unet_pre_minmax_to_u8(unet_ctx* ctx, const float* img, uint8_t* out, int32_t n, int64_t pixels, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx || !img || !out || n < 1 || pixels < 1 || !ws || ws_bytes < unet_pre_minmax_ws_bytes(n)) UNET_FAIL(ctx, UNET_E_ARG, "pre_minmax_to_u8: bad args");
  hipStream_t s = as_stream(stream);
  unsigned* mm = static_cast<unsigned*>(ws);
  hipLaunchKernelGGL(minmax_init_kernel, dim3((n + 63) / 64), dim3(64), 0, s, mm, n);
  const unsigned bx = blocks_for(pixels, 256);
  hipLaunchKernelGGL(minmax_kernel, dim3(bx, n), dim3(TPB), 0, s, img, mm, (long long)pixels);
  hipLaunchKernelGGL(norm_to_u8_kernel, dim3(bx, n), dim3(TPB), 0, s, img, mm, out, (long long)pixels);
  UNET_CHECK_LAUNCH(ctx, "pre_minmax_to_u8"); return UNET_OK;
}

int32_t unet_pre_unit_to_u8(unet_ctx* ctx, const float* img, uint8_t* out, int64_t count, void* stream) {
  if (!ctx || !img || !out || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "pre_unit_to_u8: bad args");
  hipLaunchKernelGGL(unit_to_u8_kernel, dim3(blocks_for(count, 2048)), dim3(TPB), 0, as_stream(stream), img, out, (long long)count);
  UNET_CHECK_LAUNCH(ctx, "pre_unit_to_u8"); return UNET_OK;
}

int32_t unet_pre_u8_to_unit(unet_ctx* ctx, const uint8_t* src, float* dst, int64_t count, void* stream) {
  if (!ctx || !src || !dst || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "pre_u8_to_unit: bad args");
  hipLaunchKernelGGL(u8_to_unit_kernel, dim3(blocks_for(count, 2048)), dim3(TPB), 0, as_stream(stream), src, dst, (long long)count);
  UNET_CHECK_LAUNCH(ctx, "pre_u8_to_unit"); return UNET_OK;
}

size_t unet_pre_clahe_ws_bytes(int32_t n, int32_t tiles_x, int32_t tiles_y) { return (n > 0 && tiles_x > 0 && tiles_y > 0) ? (size_t)n * tiles_x * tiles_y * 256 : 0; }

int32_t unet_pre_clahe_u8(unet_ctx* ctx, const uint8_t* src, uint8_t* dst, int32_t n, int32_t h, int32_t w, float clip_limit, int32_t tiles_x,
                          int32_t tiles_y, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx || !src || !dst || n < 1 || h < 1 || w < 1 || tiles_x < 1 || tiles_y < 1 || clip_limit < 0 || !ws || ws_bytes < unet_pre_clahe_ws_bytes(n, tiles_x, tiles_y))
    UNET_FAIL(ctx, UNET_E_ARG, "pre_clahe_u8: bad args");
  int eh = h, ew = w;
  if ((w % tiles_x) != 0 || (h % tiles_y) != 0) { eh = h + (tiles_y - h % tiles_y); ew = w + (tiles_x - w % tiles_x); }      // OpenCV pads `tiles - size % tiles`
  if (eh - h >= h || ew - w >= w) UNET_FAIL(ctx, UNET_E_SHAPE, "pre_clahe_u8: image %d x %d too small for a %d x %d grid (reflect-101 padding)", h, w, tiles_y, tiles_x);
  const int th = eh / tiles_y, tw = ew / tiles_x, area = th * tw;
  const float lut_scale = 255.0f / (float)area;
  int clip = 0;
  if (clip_limit > 0.0f) { clip = (int)((double)clip_limit * area / 256); if (clip < 1) clip = 1; }
  hipStream_t s = as_stream(stream);
  uint8_t* lut = static_cast<uint8_t*>(ws);
  hipLaunchKernelGGL(clahe_lut_kernel, dim3((unsigned)(n * tiles_x * tiles_y)), dim3(TPB), 0, s, src, lut, h, w, tiles_x, tiles_y, th, tw, clip, lut_scale);
  hipLaunchKernelGGL(clahe_blend_kernel, dim3(blocks_for((long long)n * h * w, 4096)), dim3(TPB), 0, s, src, lut, dst, n, h, w, tiles_x, tiles_y, 1.0f / (float)tw,
                     1.0f / (float)th);
  UNET_CHECK_LAUNCH(ctx, "pre_clahe_u8"); return UNET_OK;
}

int32_t unet_pre_resize_u8(unet_ctx* ctx, const uint8_t* src, int32_t n, int32_t sh, int32_t sw, const int32_t* rects, uint8_t* dst, int32_t dh, int32_t dw,
                           int32_t dst_ld, int32_t dst_x0, int32_t interp, void* stream) {
  if (!ctx || !src || !dst || n < 1 || sh < 1 || sw < 1 || dh < 1 || dw < 1 || dst_x0 < 0 || dst_ld < dst_x0 + dw) UNET_FAIL(ctx, UNET_E_ARG, "pre_resize_u8: bad args");
  if (interp != UNET_RESIZE_LINEAR && interp != UNET_RESIZE_AREA) UNET_FAIL(ctx, UNET_E_ARG, "pre_resize_u8: interpolation %d (only INTER_LINEAR = 1 and INTER_AREA = 3)", interp);
  if ((long long)dh * dw > 0x7FFFFFFFLL) UNET_FAIL(ctx, UNET_E_SHAPE, "pre_resize_u8: destination too large");
  for (int i = 0; rects && i < n; ++i) {                             // rects is a HOST array: x, y, w, h per image (cv2.boundingRect order)
    const int32_t* r = rects + 4 * i;
    if (r[0] < 0 || r[1] < 0 || r[2] < 1 || r[3] < 1 || (long long)r[0] + r[2] > sw || (long long)r[1] + r[3] > sh)
      UNET_FAIL(ctx, UNET_E_SHAPE, "pre_resize_u8: rectangle %d = (%d, %d, %d, %d) is empty or leaves the %d x %d image", i, r[0], r[1], r[2], r[3], sh, sw);
  }
  hipStream_t s = as_stream(stream);
  for (int i0 = 0; i0 < n; i0 += RESIZE_RECTS) {
    const int cnt = (n - i0) < RESIZE_RECTS ? (n - i0) : RESIZE_RECTS;
    resize_rects rc;
    for (int i = 0; i < cnt; ++i) {
      if (rects) { const int32_t* r = rects + 4 * (i0 + i); rc.x[i] = r[0]; rc.y[i] = r[1]; rc.w[i] = r[2]; rc.h[i] = r[3]; }
      else { rc.x[i] = 0; rc.y[i] = 0; rc.w[i] = sw; rc.h[i] = sh; }
    }
    hipLaunchKernelGGL(resize_u8_kernel, dim3((unsigned)(((long long)dh * dw + TPB - 1) / TPB), (unsigned)cnt), dim3(TPB), 0, s, src, sh, sw, rc, i0, dst, dh, dw, dst_ld,
                       dst_x0, interp);
  }
  UNET_CHECK_LAUNCH(ctx, "pre_resize_u8"); return UNET_OK;
}

}  // extern "C"
