// Dense tail of the slice classifier (/root/reference/Scripts/task2_covid19_classifcation.py:772-776):
//   Flatten -> Dense(50176 -> 32, relu) -> Dropout(0.4) -> Dense(32 -> 1, sigmoid), binary cross-entropy (optionally class weighted),
//   the f1 metric sums (T2:688-703).
// The wide layer is a skinny GEMM (K = 50176, N = 32, M = batch): 0.8 GFLOP at batch 256 against 58 MB of operands -> HBM-bound,
// so it runs on the vector ALUs with every global access a full line; split-K partials + a fixed-order reduction keep it
// deterministic.  Backward fuses the data- and weight-gradient: one thread owns a row W[k,:] (N registers) and its gradient row.
#include <algorithm>

#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int DK = 256;      // rows of W per workgroup (split-K chunk)

// part[kc][b][o] = sum_{k in chunk kc} x[b][k] * W[k][o]
template <typename T>
__global__ __launch_bounds__(TPB) void dense_fwd_partial_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ part, int B, int K, int N) {
  extern __shared__ float smem[];
  float* s_w = smem;                 // [DK][N]
  float* s_x = smem + DK * N;        // [rows][DK]
  const int rows = TPB / N;
  const int tid = threadIdx.x, o = tid % N, r = tid / N;
  const int k0 = blockIdx.x * DK, kn = min(DK, K - k0);
  const int b0 = blockIdx.y * rows;
  for (int i = tid; i < DK * N / 4; i += TPB) {
    const int row = (i * 4) / N;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < kn) v = *reinterpret_cast<const float4*>(w + (long long)k0 * N + i * 4);
    *reinterpret_cast<float4*>(s_w + i * 4) = v;
  }
  for (int i = tid; i < rows * DK / 4; i += TPB) {
    const int rr = (i * 4) / DK, kk = (i * 4) % DK;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b0 + rr < B && kk < kn) v = ld4(x + (long long)(b0 + rr) * K + k0 + kk);   // K % 4 == 0
    *reinterpret_cast<float4*>(s_x + i * 4) = v;
  }
  __syncthreads();
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const float* xr = s_x + r * DK;
#pragma unroll 4
  for (int k = 0; k < DK; k += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + k);          // broadcast within the row group
    a0 = fmaf(xv.x, s_w[(k + 0) * N + o], a0); a1 = fmaf(xv.y, s_w[(k + 1) * N + o], a1);
    a2 = fmaf(xv.z, s_w[(k + 2) * N + o], a2); a3 = fmaf(xv.w, s_w[(k + 3) * N + o], a3);
  }
  if (b0 + r < B) part[((long long)blockIdx.x * B + b0 + r) * N + o] = (a0 + a1) + (a2 + a3);
}

// y[b][o..o+3] = dropout(act(sum_kc part + bias))
__global__ __launch_bounds__(TPB) void dense_fwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias,
                                                               float* __restrict__ y, int B, int N, int chunks, int act, float rate,
                                                               unsigned long long seed) {
  const long long quads = (long long)B * N / 4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < quads; i += (long long)gridDim.x * TPB) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    int c = 0;
    for (; c + 1 < chunks; c += 2) {
      const float4 a = *reinterpret_cast<const float4*>(part + ((long long)c * B * N) + i * 4);
      const float4 b = *reinterpret_cast<const float4*>(part + ((long long)(c + 1) * B * N) + i * 4);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
    }
    if (c < chunks) { const float4 a = *reinterpret_cast<const float4*>(part + ((long long)c * B * N) + i * 4); s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    const int o = (int)((i * 4) % N);
    const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v = make_float4(apply_act(s0.x + s1.x + bb.x, act), apply_act(s0.y + s1.y + bb.y, act), apply_act(s0.z + s1.z + bb.z, act),
                           apply_act(s0.w + s1.w + bb.w, act));
    if (rate > 0.0f) { const float4 ks = keep_scale(i, rate, seed); v.x *= ks.x; v.y *= ks.y; v.z *= ks.z; v.w *= ks.w; }
    *reinterpret_cast<float4*>(y + i * 4) = v;
  }
}

// One thread = one row k of W: dx[b][k] = sum_o dy[b][o] W[k][o]; dW[k][o] = sum_b x[b][k] dy[b][o].
// dy rows are staged through LDS in tiles of DB rows and read as broadcasts; x and dx accesses are coalesced along k.
constexpr int DB = 128;
template <int N, typename T>
__global__ __launch_bounds__(TPB) void dense_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                        T* __restrict__ dx, float* __restrict__ dw, int B, int K) {
  __shared__ __attribute__((aligned(16))) float s_dy[DB * N];
  const int k = blockIdx.x * TPB + threadIdx.x;
  const bool ok = k < K;
  float wr[N], acc[N];
#pragma unroll
  for (int o = 0; o < N; o += 4) {
    const float4 v = ok ? *reinterpret_cast<const float4*>(w + (long long)k * N + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    wr[o] = v.x; wr[o + 1] = v.y; wr[o + 2] = v.z; wr[o + 3] = v.w;
    acc[o] = acc[o + 1] = acc[o + 2] = acc[o + 3] = 0.f;
  }
  for (int b0 = 0; b0 < B; b0 += DB) {
    const int nb = min(DB, B - b0);
    __syncthreads();
    for (int i = threadIdx.x; i < nb * N / 4; i += TPB)
      *reinterpret_cast<float4*>(s_dy + i * 4) = *reinterpret_cast<const float4*>(dy + (long long)b0 * N + i * 4);
    __syncthreads();
    if (!ok) continue;
    float xv = ld1(x + (long long)b0 * K + k);
    for (int b = 0; b < nb; ++b) {
      const float xn = (b + 1 < nb) ? ld1(x + (long long)(b0 + b + 1) * K + k) : 0.f;       // next row's load in flight during the FMAs
      float d = 0.f;
#pragma unroll
      for (int o = 0; o < N; o += 4) {
        const float4 g = *reinterpret_cast<const float4*>(s_dy + b * N + o);
        d = fmaf(g.x, wr[o], d); d = fmaf(g.y, wr[o + 1], d); d = fmaf(g.z, wr[o + 2], d); d = fmaf(g.w, wr[o + 3], d);
        acc[o] = fmaf(xv, g.x, acc[o]); acc[o + 1] = fmaf(xv, g.y, acc[o + 1]); acc[o + 2] = fmaf(xv, g.z, acc[o + 2]); acc[o + 3] = fmaf(xv, g.w, acc[o + 3]);
      }
      if (dx) st1(dx + (long long)(b0 + b) * K + k, d);
      xv = xn;
    }
  }
  if (ok) {
#pragma unroll
    for (int o = 0; o < N; o += 4) *reinterpret_cast<float4*>(dw + (long long)k * N + o) = make_float4(acc[o], acc[o + 1], acc[o + 2], acc[o + 3]);
  }
}

__device__ __forceinline__ float bce_clip(float p, float t, float* pc_out, bool* inrange) {       // Keras binary_crossentropy on probabilities
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  const float pc = fminf(fmaxf(p, lo), hi);
  *pc_out = pc; *inrange = (p >= lo) && (p <= hi);
  const float z = logf(pc / (1.0f - pc));
  return fmaxf(z, 0.0f) - z * t + log1pf(expf(-fabsf(z)));
}

// single workgroup: p[b] = sigmoid(h[b,:] . w + bias); sums += (sum cw*bce, sum round(t*p), sum round(t), sum round(p))
__global__ __launch_bounds__(TPB) void cls_head_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ p, const float* __restrict__ yt, float cw0, float cw1,
                                                           double* sums, int B, int N) {
  float sl = 0.f, stp = 0.f, st = 0.f, sp = 0.f;
  for (int b = threadIdx.x; b < B; b += TPB) {
    float z = bias[0];
    for (int o = 0; o < N; o += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(h + (long long)b * N + o);
      const float4 wv = *reinterpret_cast<const float4*>(w + o);
      z = fmaf(hv.x, wv.x, z); z = fmaf(hv.y, wv.y, z); z = fmaf(hv.z, wv.z, z); z = fmaf(hv.w, wv.w, z);
    }
    const float pr = 1.0f / (1.0f + expf(-z));
    p[b] = pr;
    if (yt) {
      const float t = yt[b]; float pc; bool inr;
      sl += (t >= 0.5f ? cw1 : cw0) * bce_clip(pr, t, &pc, &inr);
      stp += rintf(fminf(fmaxf(t * pr, 0.f), 1.f)); st += rintf(fminf(fmaxf(t, 0.f), 1.f)); sp += rintf(fminf(fmaxf(pr, 0.f), 1.f));
    }
  }
  if (!yt) return;
  __shared__ float red[4][TPB / 64];
  sl = wave_sum(sl); stp = wave_sum(stp); st = wave_sum(st); sp = wave_sum(sp);
  const int wv_ = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv_] = sl; red[1][wv_] = stp; red[2][wv_] = st; red[3][wv_] = sp; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int k = 0; k < TPB / 64; ++k) s += red[threadIdx.x][k];
    atomicAdd(sums + threadIdx.x, (double)s);
  }
}

__global__ void cls_loss_finalize_kernel(const double* sums, double count, float* out, float* out2) {
  const double eps = 1e-7;                                  // K.epsilon()
  const double prec = sums[1] / (sums[3] + eps), rec = sums[1] / (sums[2] + eps);
  const float l = (float)(sums[0] / count), f = (float)(2.0 * (prec * rec) / (prec + rec + eps));
  out[0] = l; out[1] = f;
  if (out2) { out2[0] = l; out2[1] = f; }
}

// single workgroup, N <= 32: thread (o, r) walks batch rows r, r+rows, ...:
//   dz_b = cw(t_b) (p_b - t_b) / count (inside the clip range); dw[o] = sum_b dz_b h[b][o]; db = sum_b dz_b
//   dh[b][o] = dz_b w[o] * (h[b][o] > 0 ? scale : 0)   (h = dropout(relu(a)): positive <=> kept and a > 0); db1[o] = sum_b dh[b][o]
__global__ __launch_bounds__(TPB) void cls_head_bwd_kernel(const float* __restrict__ h, const float* __restrict__ w, const float* __restrict__ p,
                                                           const float* __restrict__ yt, float cw0, float cw1, float inv_count, float scale,
                                                           float* __restrict__ dh, float* dw, float* db, float* db1, int B, int N) {
  const int rows = TPB / N;
  const int o = threadIdx.x % N, r = threadIdx.x / N;
  const float wo = w[o];
  float aw = 0.f, ab = 0.f, a1 = 0.f;
  for (int b = r; b < B; b += rows) {
    const float pr = p[b], t = yt[b]; float pc; bool inr;
    (void)bce_clip(pr, t, &pc, &inr);
    const float dz = inr ? (t >= 0.5f ? cw1 : cw0) * (pc - t) * inv_count : 0.0f;
    const float hv = h[(long long)b * N + o];
    const float g = hv > 0.0f ? dz * wo * scale : 0.0f;
    dh[(long long)b * N + o] = g;
    aw = fmaf(dz, hv, aw); ab += dz; a1 += g;
  }
  __shared__ float s_aw[TPB], s_ab[TPB], s_a1[TPB];
  s_aw[threadIdx.x] = aw; s_ab[threadIdx.x] = ab; s_a1[threadIdx.x] = a1;
  __syncthreads();
  if (r == 0) {
    for (int k = 1; k < rows; ++k) { aw += s_aw[o + k * N]; ab += s_ab[o + k * N]; a1 += s_a1[o + k * N]; }
    dw[o] = aw; db1[o] = a1;
    if (o == 0) db[0] = ab;
  }
}

bool dense_n_ok(int n) { return n >= 4 && n <= 32 && (n & (n - 1)) == 0; }

}  // namespace

extern "C" {

size_t unet_dense_ws_bytes(int32_t batch, int32_t k, int32_t n) {
  if (batch < 1 || k < 1 || n < 1) return 0;
  return (size_t)((k + DK - 1) / DK) * batch * n * sizeof(float);
}

extern "C++" template <typename T> static int32_t dense_fwd_impl(unet_ctx* ctx, const T* x, const float* w, const float* bias, float* y, int32_t batch, int32_t k, int32_t n,
                       int32_t act, float drop_rate, uint64_t drop_seed, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx || !x || !w || !y || batch < 1 || k < 4 || (k & 3) || !dense_n_ok(n) || act < 0 || act > 2 || drop_rate < 0 || drop_rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "dense_fwd: bad args (k %% 4 == 0, n a power of two in 4..32)");
  if (!ws || ws_bytes < unet_dense_ws_bytes(batch, k, n)) UNET_FAIL(ctx, UNET_E_ARG, "dense_fwd: workspace too small");
  const int chunks = (k + DK - 1) / DK, rows = TPB / n;
  const size_t lds = (size_t)(DK * n + rows * DK) * sizeof(float);
  hipLaunchKernelGGL(dense_fwd_partial_kernel<T>, dim3(chunks, (batch + rows - 1) / rows), dim3(TPB), lds, as_stream(stream), x, w, static_cast<float*>(ws), batch, k, n);
  UNET_CHECK_LAUNCH(ctx, "dense_fwd_partial");
  const long long quads = (long long)batch * n / 4;
  hipLaunchKernelGGL(dense_fwd_reduce_kernel, dim3((unsigned)std::min<long long>((quads + TPB - 1) / TPB, 2048)), dim3(TPB), 0, as_stream(stream),
                     static_cast<const float*>(ws), bias, y, batch, n, chunks, act, drop_rate, (unsigned long long)drop_seed);
  UNET_CHECK_LAUNCH(ctx, "dense_fwd_reduce");
  return UNET_OK;
}
int32_t unet_dense_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int32_t batch, int32_t k, int32_t n, int32_t act,
                       float drop_rate, uint64_t drop_seed, void* ws, size_t ws_bytes, void* stream) {
  return dense_fwd_impl(ctx, x, w, bias, y, batch, k, n, act, drop_rate, drop_seed, ws, ws_bytes, stream);
}
int32_t unet_dense_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, float* y, int32_t batch, int32_t k, int32_t n, int32_t act,
                            float drop_rate, uint64_t drop_seed, void* ws, size_t ws_bytes, void* stream) {
  return dense_fwd_impl(ctx, x, w, bias, y, batch, k, n, act, drop_rate, drop_seed, ws, ws_bytes, stream);
}

extern "C++" template <typename T> static int32_t dense_bwd_impl(unet_ctx* ctx, const T* x, const float* w, const float* dy, T* dx, float* dw, int32_t batch, int32_t k, int32_t n,
                       void* stream) {
  if (!ctx || !x || !w || !dy || !dw || batch < 1 || k < 1 || !dense_n_ok(n)) UNET_FAIL(ctx, UNET_E_ARG, "dense_bwd: bad args");
  const dim3 grid((k + TPB - 1) / TPB), block(TPB);
  hipStream_t s = as_stream(stream);
  switch (n) {
    case 4: hipLaunchKernelGGL((dense_bwd_kernel<4, T>), grid, block, 0, s, x, w, dy, dx, dw, batch, k); break;
    case 8: hipLaunchKernelGGL((dense_bwd_kernel<8, T>), grid, block, 0, s, x, w, dy, dx, dw, batch, k); break;
    case 16: hipLaunchKernelGGL((dense_bwd_kernel<16, T>), grid, block, 0, s, x, w, dy, dx, dw, batch, k); break;
    default: hipLaunchKernelGGL((dense_bwd_kernel<32, T>), grid, block, 0, s, x, w, dy, dx, dw, batch, k); break;
  }
  UNET_CHECK_LAUNCH(ctx, "dense_bwd");
  return UNET_OK;
}
int32_t unet_dense_bwd(unet_ctx* ctx, const float* x, const float* w, const float* dy, float* dx, float* dw, int32_t batch, int32_t k, int32_t n, void* stream) {
  return dense_bwd_impl(ctx, x, w, dy, dx, dw, batch, k, n, stream);
}
int32_t unet_dense_bwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* dy, unet_bf16* dx, float* dw, int32_t batch, int32_t k, int32_t n,
                            void* stream) {
  return dense_bwd_impl(ctx, x, w, dy, dx, dw, batch, k, n, stream);
}

int32_t unet_cls_head_fwd(unet_ctx* ctx, const float* h, const float* w, const float* bias, float* p, const float* y_true, float class_w0,
                          float class_w1, double* sums, int32_t batch, int32_t n, void* stream) {
  if (!ctx || !h || !w || !bias || !p || batch < 1 || n < 4 || (n & 3) || (y_true && !sums)) UNET_FAIL(ctx, UNET_E_ARG, "cls_head_fwd: bad args");
  hipLaunchKernelGGL(cls_head_fwd_kernel, dim3(1), dim3(TPB), 0, as_stream(stream), h, w, bias, p, y_true, class_w0, class_w1, sums, batch, n);
  UNET_CHECK_LAUNCH(ctx, "cls_head_fwd");
  return UNET_OK;
}

extern "C++" int32_t k_cls_loss_finalize(unet_ctx* ctx, const double* sums, double count, float* out, float* out2, hipStream_t s) {
  if (!ctx || !sums || !out || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "cls_loss_finalize: bad args");
  hipLaunchKernelGGL(cls_loss_finalize_kernel, dim3(1), dim3(1), 0, s, sums, count, out, out2);
  UNET_CHECK_LAUNCH(ctx, "cls_loss_finalize");
  return UNET_OK;
}
int32_t unet_cls_loss_finalize(unet_ctx* ctx, const double* sums, double count, float* out, void* stream) {
  return k_cls_loss_finalize(ctx, sums, count, out, nullptr, as_stream(stream));
}

int32_t unet_cls_head_bwd(unet_ctx* ctx, const float* h, const float* w, const float* p, const float* y_true, float class_w0, float class_w1,
                          double count, float drop_rate, float* dh, float* dw, float* db, float* dbias_prev, int32_t batch, int32_t n,
                          void* stream) {
  if (!ctx || !h || !w || !p || !y_true || !dh || !dw || !db || !dbias_prev || batch < 1 || !dense_n_ok(n) || count < 1 || drop_rate < 0 || drop_rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "cls_head_bwd: bad args");
  hipLaunchKernelGGL(cls_head_bwd_kernel, dim3(1), dim3(TPB), 0, as_stream(stream), h, w, p, y_true, class_w0, class_w1, (float)(1.0 / count),
                     1.0f / (1.0f - drop_rate), dh, dw, db, dbias_prev, batch, n);
  UNET_CHECK_LAUNCH(ctx, "cls_head_bwd");
  return UNET_OK;
}

}  // extern "C"
