// HBM-bound kernels of the U-Net hot path for gfx950: BatchNorm (stats / apply / backward),
// 2x2 max-pool + dropout, 1x1 sigmoid head fused with the BCE+Dice reductions and its
// backward, Keras-form Adam, thresholded segmentation metric sums.
// All tensors NHWC fp32; every global access is a 16-byte (float4) lane access on a
// channel-contiguous run, so a wave touches whole 128-B lines.  `ld` = pixel stride in
// floats (channel slices of the zero-copy concat buffers).
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 resident blocks; grid-stride beyond that
// BN statistics end in 2C fp64 atomics per block on a handful of cache lines: measured 0.2 ms of pure
// atomic serialisation at 2048 blocks, so the reduction kernels run 2 blocks per CU instead
// Workgroups of the reduction kernels.  Every workgroup ends in 2C fp64 atomics (spread over UNET_BN_SLOTS copies of the target), so
// more workgroups buy memory-level parallelism and pay atomic traffic; measured optima per kernel:
constexpr int HEAD_BLOCKS = 8192;   // head_fwd: one load in flight per thread, wants many workgroups (0.20 -> 0.18 ms); head_bwd ends in 33 atomics on two cache lines: 1024 (0.25 -> 0.215)
constexpr int BN_STATS_BLOCKS = 512;          // one tensor in        (0.86 -> 0.68 ms per step with the slots)
constexpr int BN_BWD_STATS_BLOCKS = 1024;     // two tensors in       (1.06 -> 0.79)
constexpr int POOL_BWD_STATS_BLOCKS = 768;    // read-modify-write    (0.73 -> 0.63)
static_assert(BN_STATS_BLOCKS <= UNET_BN_SLOTS_DET && BN_BWD_STATS_BLOCKS <= UNET_BN_SLOTS_DET && POOL_BWD_STATS_BLOCKS <= UNET_BN_SLOTS_DET, "deterministic mode: one slot copy per workgroup");


// ---------------------------------------------------------------------------------------
// BatchNorm statistics.  MODE 0: (sum x, sum x^2).  MODE 1: (sum dy, sum dy*xhat).
// lanes-per-pixel lpp = C/4; a block covers ppb = 256/lpp pixels per iteration.
// fp32 per-thread partials (<= a few hundred terms each), block tree, fp64 global atomics.
// ---------------------------------------------------------------------------------------
template <int MODE, typename T>
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const T* __restrict__ a, int lda,
                                                       const T* __restrict__ x, int ldx,
                                                       const float* __restrict__ bnp, double* sums,
                                                       long long pixels, int C, int nslots) {
  const int lpp = C >> 2, ppb = TPB / lpp;
  const int tid = threadIdx.x, q = tid % lpp, pl = tid / lpp;
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  float4 mean = s1, istd = s1;
  if (MODE == 1) { mean = ld4(bnp + 2 * C + q * 4); istd = ld4(bnp + 3 * C + q * 4); }
  if (pl < ppb) {
    const long long step = (long long)gridDim.x * ppb;
    auto accum = [&](const float4& v, const float4& xv) {
      if (MODE == 0) {
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
      } else {
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x += v.x * (xv.x - mean.x) * istd.x; s2.y += v.y * (xv.y - mean.y) * istd.y;
        s2.z += v.z * (xv.z - mean.z) * istd.z; s2.w += v.w * (xv.w - mean.w) * istd.w;
      }
    };
    long long p = (long long)blockIdx.x * ppb + pl;
    for (; p + 3 * step < pixels; p += 4 * step) {            // 4 independent 16-B loads in flight per lane
      float4 v0 = ld4(a + p * lda + q * 4), v1 = ld4(a + (p + step) * lda + q * 4);
      float4 v2 = ld4(a + (p + 2 * step) * lda + q * 4), v3 = ld4(a + (p + 3 * step) * lda + q * 4);
      float4 x0 = v0, x1 = v1, x2 = v2, x3 = v3;
      if (MODE == 1) {
        x0 = ld4(x + p * ldx + q * 4); x1 = ld4(x + (p + step) * ldx + q * 4);
        x2 = ld4(x + (p + 2 * step) * ldx + q * 4); x3 = ld4(x + (p + 3 * step) * ldx + q * 4);
      }
      accum(v0, x0); accum(v1, x1); accum(v2, x2); accum(v3, x3);
    }
    for (; p < pixels; p += step) {
      float4 v = ld4(a + p * lda + q * 4);
      float4 xv = v;
      if (MODE == 1) xv = ld4(x + p * ldx + q * 4);
      accum(v, xv);
    }
  }
  __shared__ float4 sh1[TPB], sh2[TPB];
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  if (tid < lpp) {
    for (int k = 1; k < ppb; ++k) {
      float4 u = sh1[tid + k * lpp], w = sh2[tid + k * lpp];
      s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w;
      s2.x += w.x; s2.y += w.y; s2.z += w.z; s2.w += w.w;
    }
    double* d1 = sums + (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES + q * 4; double* d2 = d1 + C;    // `sums` = the slot copies (deterministic mode: one writer per copy)
    atomicAdd(d1 + 0, (double)s1.x); atomicAdd(d1 + 1, (double)s1.y);
    atomicAdd(d1 + 2, (double)s1.z); atomicAdd(d1 + 3, (double)s1.w);
    atomicAdd(d2 + 0, (double)s2.x); atomicAdd(d2 + 1, (double)s2.y);
    atomicAdd(d2 + 2, (double)s2.z); atomicAdd(d2 + 3, (double)s2.w);
  }
}

// enc / comp (or null): the layer is a decoder BatchNorm over concat([up, BN_enc(y)]) whose skip half holds the RAW y (k_bn_compose, kernels_bnfold.hip): the composite map
// [scale' C][shift' C][pre_s C][pre_t C] of the fold is written in the same launch (C = 2 C_enc: channels C/2 .. C-1 are the skip half)
__device__ __forceinline__ void bn_compose_one(const float* __restrict__ enc, float* __restrict__ comp, int C, int c, float sd, float td) {
  const int ce = C >> 1;
  if (c < ce) { comp[c] = sd; comp[C + c] = td; comp[2 * C + c] = 1.0f; comp[3 * C + c] = 0.0f; }
  else { const float se = enc[c - ce], te = enc[ce + (c - ce)]; comp[c] = sd * se; comp[C + c] = fmaf(sd, te, td); comp[2 * C + c] = se; comp[3 * C + c] = te; }
}
__global__ void bn_finalize_train_kernel(const double* sums, double count, const float* gamma,
                                         const float* beta, float* mm, float* mv, float* bnp, int C,
                                         float momentum, float eps, const float* __restrict__ enc = nullptr, float* __restrict__ comp = nullptr) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean = sums[c] / count;
  double var = sums[C + c] / count - mean * mean;
  if (var < 0) var = 0;
  float istd = (float)(1.0 / sqrt(var + (double)eps));
  float sc = gamma[c] * istd;
  bnp[c] = sc; bnp[C + c] = beta[c] - (float)mean * sc; bnp[2 * C + c] = (float)mean; bnp[3 * C + c] = istd;
  if (comp) bn_compose_one(enc, comp, C, c, sc, beta[c] - (float)mean * sc);
  double unbiased = var * (count / (count > 1.0 ? count - 1.0 : 1.0));
  mm[c] = mm[c] * momentum + (float)mean * (1.0f - momentum);
  mv[c] = mv[c] * momentum + (float)unbiased * (1.0f - momentum);
}

__global__ void bn_finalize_infer_kernel(const float* gamma, const float* beta, const float* mm,
                                         const float* mv, float* bnp, int C, float eps, const float* __restrict__ enc = nullptr, float* __restrict__ comp = nullptr) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float istd = 1.0f / sqrtf(mv[c] + eps);
  float sc = gamma[c] * istd;
  bnp[c] = sc; bnp[C + c] = beta[c] - mm[c] * sc; bnp[2 * C + c] = mm[c]; bnp[3 * C + c] = istd;
  if (comp) bn_compose_one(enc, comp, C, c, sc, beta[c] - mm[c] * sc);
}

template <typename T>
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const T* __restrict__ x, int ldx,
                                                       const float* __restrict__ bnp,
                                                       T* __restrict__ y, int ldy,
                                                       long long pixels, int C) {
  const int lpp = C >> 2;
  if (TPB % lpp == 0) {          // a thread keeps its channel quad: parameters live in registers, no per-element 64-bit div/mod
    const int q = threadIdx.x % lpp, ppb = TPB / lpp;
    const float4 sc = ld4(bnp + q * 4), sh = ld4(bnp + C + q * 4);
    const long long step = (long long)gridDim.x * ppb;
    long long p = (long long)blockIdx.x * ppb + threadIdx.x / lpp;
    for (; p + step < pixels; p += 2 * step) {                   // two independent 16-B loads in flight per lane
      const float4 v0 = ld4(x + p * ldx + q * 4), v1 = ld4(x + (p + step) * ldx + q * 4);
      st4(y + p * ldy + q * 4, make_float4(fmaf(v0.x, sc.x, sh.x), fmaf(v0.y, sc.y, sh.y), fmaf(v0.z, sc.z, sh.z), fmaf(v0.w, sc.w, sh.w)));
      st4(y + (p + step) * ldy + q * 4, make_float4(fmaf(v1.x, sc.x, sh.x), fmaf(v1.y, sc.y, sh.y), fmaf(v1.z, sc.z, sh.z), fmaf(v1.w, sc.w, sh.w)));
    }
    if (p < pixels) {
      const float4 v = ld4(x + p * ldx + q * 4);
      st4(y + p * ldy + q * 4, make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)));
    }
    return;
  }
  const long long total = pixels * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    int q = (int)(i % lpp); long long p = i / lpp;
    float4 v = ld4(x + p * ldx + q * 4), sc = ld4(bnp + q * 4), sh = ld4(bnp + C + q * 4);
    st4(y + p * ldy + q * 4, make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y),
                                          fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)));
  }
}


__global__ void bn_bwd_param_grads_kernel(const double* sums, float* dgamma, float* dbeta, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)sums[c]; dgamma[c] = (float)sums[C + c];
}

// mask_mode: derivative of what produced the BN input x (MASK_RELU: T1:860; MASK_ELU / MASK_ELU_DROP: U-Net++ conv_block,
// where x = dropout(elu(conv)) and the keep mask is recomputed from the Philox stream of that dropout)
template <int MM, typename T>       // MASK_* of the producer of x, compile-time so the U-Net (ReLU) instance carries no Philox/ELU code
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const T* __restrict__ dy, int lddy,
                                                           const T* __restrict__ x, int ldx,
                                                           const float* __restrict__ bnp,
                                                           const double* __restrict__ sums, double inv_count,
                                                           T* __restrict__ dx, int lddx,
                                                           long long pixels, int C, float rate, unsigned long long seed) {
  constexpr int mask_mode = MM;
  const int lpp = C >> 2;
  // dx = scale * (dy - k1 - xhat * k2) * mask'(x);  k1 = sum(dy)/count, k2 = sum(dy*xhat)/count
  auto one = [&](long long p, int q, const float4& sc, const float4& mean, const float4& istd, const float4& k1, const float4& k2) {
    const float4 g = ld4(dy + p * lddy + q * 4), xv = ld4(x + p * ldx + q * 4);
    float4 k4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (mask_mode == MASK_ELU_DROP) k4 = keep_scale(p * lpp + q, rate, seed);       // x is dense here: quad index == p*lpp + q
    float4 r;
    r.x = sc.x * (g.x - k1.x - (xv.x - mean.x) * istd.x * k2.x) * mask_factor(xv.x, mask_mode, k4.x, rate);
    r.y = sc.y * (g.y - k1.y - (xv.y - mean.y) * istd.y * k2.y) * mask_factor(xv.y, mask_mode, k4.y, rate);
    r.z = sc.z * (g.z - k1.z - (xv.z - mean.z) * istd.z * k2.z) * mask_factor(xv.z, mask_mode, k4.z, rate);
    r.w = sc.w * (g.w - k1.w - (xv.w - mean.w) * istd.w * k2.w) * mask_factor(xv.w, mask_mode, k4.w, rate);
    st4(dx + p * lddx + q * 4, r);
  };
  auto params = [&](int q, float4& sc, float4& mean, float4& istd, float4& k1, float4& k2) {
    sc = ld4(bnp + q * 4); mean = ld4(bnp + 2 * C + q * 4); istd = ld4(bnp + 3 * C + q * 4);
    const double* s1 = sums + q * 4; const double* s2 = sums + C + q * 4;
    k1 = make_float4((float)(s1[0] * inv_count), (float)(s1[1] * inv_count), (float)(s1[2] * inv_count), (float)(s1[3] * inv_count));
    k2 = make_float4((float)(s2[0] * inv_count), (float)(s2[1] * inv_count), (float)(s2[2] * inv_count), (float)(s2[3] * inv_count));
  };
  float4 sc, mean, istd, k1, k2;
  if (TPB % lpp == 0) {          // a thread keeps its channel quad: the 5 parameter quads (8 of them doubles) are loaded once
    const int q = threadIdx.x % lpp, ppb = TPB / lpp;
    params(q, sc, mean, istd, k1, k2);
    const long long step = (long long)gridDim.x * ppb;
    for (long long p = (long long)blockIdx.x * ppb + threadIdx.x / lpp; p < pixels; p += step) one(p, q, sc, mean, istd, k1, k2);
    return;
  }
  const long long total = pixels * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int q = (int)(i % lpp); const long long p = i / lpp;
    params(q, sc, mean, istd, k1, k2);
    one(p, q, sc, mean, istd, k1, k2);
  }
}

// ---------------------------------------------------------------------------------------
// 2x2 max-pool (+ inverted dropout).  One thread = one pooled pixel x 4 channels.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(TPB) void pool_fwd_kernel(const T* __restrict__ x, int ldx,
                                                       T* __restrict__ y, int N, int H, int W, int C,
                                                       float rate, uint64_t seed) {
  const int lpp = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)lpp, tu = pu / (unsigned)Wo;    // 32-bit index math (launchers check total < 2^31)
    const int q = (int)(iu - pu * (unsigned)lpp), jo = (int)(pu - tu * (unsigned)Wo), io = (int)(tu % (unsigned)Ho);
    const long long p = pu, n = tu / (unsigned)Ho;
    const T* b = x + ((n * H + 2 * io) * W + 2 * jo) * (long long)ldx + q * 4;
    float4 a0 = ld4(b), a1 = ld4(b + ldx), a2 = ld4(b + (long long)W * ldx), a3 = ld4(b + (long long)(W + 1) * ldx);
    float4 m = make_float4(fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x)), fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y)),
                           fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z)), fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w)));
    if (rate > 0.0f) { float4 k = keep_scale(i, rate, seed); m.x *= k.x; m.y *= k.y; m.z *= k.z; m.w *= k.w; }
    st4(y + p * C + q * 4, m);
  }
}

__device__ __forceinline__ int argmax4(float a, float b, float c, float d) {
  int k = 0; float m = a;
  if (b > m) { m = b; k = 1; }
  if (c > m) { m = c; k = 2; }
  if (d > m) { m = d; k = 3; }
  return k;
}

template <bool ACC, typename T>
__global__ __launch_bounds__(TPB) void pool_bwd_kernel(const T* __restrict__ x, int ldx,
                                                       const T* __restrict__ dy, T* dx, int lddx,
                                                       int N, int H, int W, int C, float rate, uint64_t seed) {
  const int lpp = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)lpp, tu = pu / (unsigned)Wo;    // 32-bit index math (launchers check total < 2^31)
    const int q = (int)(iu - pu * (unsigned)lpp), jo = (int)(pu - tu * (unsigned)Wo), io = (int)(tu % (unsigned)Ho);
    const long long p = pu, n = tu / (unsigned)Ho;
    long long pix = (n * H + 2 * io) * W + 2 * jo;
    const T* b = x + pix * ldx + q * 4;
    float4 a0 = ld4(b), a1 = ld4(b + ldx), a2 = ld4(b + (long long)W * ldx), a3 = ld4(b + (long long)(W + 1) * ldx);
    float4 g = ld4(dy + p * C + q * 4);
    if (rate > 0.0f) { float4 k = keep_scale(i, rate, seed); g.x *= k.x; g.y *= k.y; g.z *= k.z; g.w *= k.w; }
    int kx = argmax4(a0.x, a1.x, a2.x, a3.x), ky = argmax4(a0.y, a1.y, a2.y, a3.y);
    int kz = argmax4(a0.z, a1.z, a2.z, a3.z), kw = argmax4(a0.w, a1.w, a2.w, a3.w);
    T* o = dx + pix * lddx + q * 4;
    const long long offs[4] = {0, lddx, (long long)W * lddx, (long long)(W + 1) * lddx};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = make_float4(kx == k ? g.x : 0.f, ky == k ? g.y : 0.f, kz == k ? g.z : 0.f, kw == k ? g.w : 0.f);
      if (ACC) { float4 e = ld4(o + offs[k]); v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w; }
      st4(o + offs[k], v);
    }
  }
}

// Fused encoder tail (T1:861-863): y = BN(x) written into the skip slice of the concat buffer AND
// p = dropout(maxpool2x2(y)) in one pass -- saves re-reading y for the pool.
template <typename T>
__global__ __launch_bounds__(TPB) void bn_pool_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ bnp,
                                                          T* __restrict__ y, int ldy, T* __restrict__ pooled, int N, int H,
                                                          int W, int C, float rate, uint64_t seed) {
  const int lpp = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const unsigned iu = (unsigned)i, pu = iu / (unsigned)lpp, tu = pu / (unsigned)Wo;    // 32-bit index math (launchers check total < 2^31)
    const int q = (int)(iu - pu * (unsigned)lpp), jo = (int)(pu - tu * (unsigned)Wo), io = (int)(tu % (unsigned)Ho);
    const long long p = pu, n = tu / (unsigned)Ho;
    const long long pix = (n * H + 2 * io) * W + 2 * jo;
    const T* b = x + pix * ldx + q * 4;
    const float4 sc = ld4(bnp + q * 4), sh = ld4(bnp + C + q * 4);
    const long long offs[4] = {0, 1, (long long)W, (long long)W + 1};
    float4 m;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float4 v = ld4(b + offs[k] * ldx);
      const float4 r = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
      if (y) st4(y + (pix + offs[k]) * ldy + q * 4, r);          // (y null: only the pooled tensor is wanted -- the normalised one is folded into its consumers)
      if (k == 0) m = r;
      else { m.x = fmaxf(m.x, r.x); m.y = fmaxf(m.y, r.y); m.z = fmaxf(m.z, r.z); m.w = fmaxf(m.w, r.w); }
    }
    // (an element dropout removes is stored as -0.0f, a kept one that happens to be zero as +0.0f: the data gradient behind this tensor reads the keep mask off the
    //  stored value, MASK_POOL_SUMS in kernels_conv_h2.hip; every other reader sees a zero either way)
    if (rate > 0.0f) { float4 kk = keep_scale(i, rate, seed); m.x = kk.x == 0.f ? -0.0f : m.x * kk.x + 0.0f; m.y = kk.y == 0.f ? -0.0f : m.y * kk.y + 0.0f;
                       m.z = kk.z == 0.f ? -0.0f : m.z * kk.z + 0.0f; m.w = kk.w == 0.f ? -0.0f : m.w * kk.w + 0.0f; }
    st4(pooled + p * C + q * 4, m);
  }
}

// Fused encoder backward head: pool/dropout backward accumulated into the skip-gradient slice, PLUS the BatchNorm
// backward statistics (sum d, sum d*xhat) of the finished gradient d, with xhat = (y - beta)/gamma recovered from the BN
// OUTPUT y that the pool backward reads anyway -- saves the separate 2-tensor statistics pass (gamma == 0 is not supported
// by this fused form: xhat cannot be recovered from a constant output).
template <typename T>
__global__ __launch_bounds__(TPB) void pool_bwd_bnstats_kernel(const T* __restrict__ y, int ldy, const T* __restrict__ dyp,
                                                               T* dx, int lddx, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, double* sums, int N, int H, int W,
                                                               int C, float rate, uint64_t seed, int nslots) {
  const int lpp = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * lpp;
  const int tid = threadIdx.x, q = tid % lpp;                        // lpp divides TPB: a thread keeps its channel quad
  const float4 g4 = ld4(gamma + q * 4), b4 = ld4(beta + q * 4);
  const float4 ig = make_float4(g4.x != 0.f ? 1.f / g4.x : 0.f, g4.y != 0.f ? 1.f / g4.y : 0.f, g4.z != 0.f ? 1.f / g4.z : 0.f,
                                g4.w != 0.f ? 1.f / g4.w : 0.f);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  for (long long i = (long long)blockIdx.x * TPB + tid; i < total; i += (long long)gridDim.x * TPB) {
    const unsigned pu = (unsigned)i / (unsigned)lpp, tu = pu / (unsigned)Wo;             // 32-bit index math (launcher checks total < 2^31)
    const int jo = (int)(pu - tu * (unsigned)Wo), io = (int)(tu % (unsigned)Ho);
    const long long p = pu, n = tu / (unsigned)Ho;
    long long pix = (n * H + 2 * io) * W + 2 * jo;
    const T* b = y + pix * ldy + q * 4;
    float4 a[4] = {ld4(b), ld4(b + ldy), ld4(b + (long long)W * ldy), ld4(b + (long long)(W + 1) * ldy)};
    float4 g = ld4(dyp + p * C + q * 4);
    if (rate > 0.0f) { float4 k = keep_scale(i, rate, seed); g.x *= k.x; g.y *= k.y; g.z *= k.z; g.w *= k.w; }
    int kx = argmax4(a[0].x, a[1].x, a[2].x, a[3].x), ky = argmax4(a[0].y, a[1].y, a[2].y, a[3].y);
    int kz = argmax4(a[0].z, a[1].z, a[2].z, a[3].z), kw = argmax4(a[0].w, a[1].w, a[2].w, a[3].w);
    T* o = dx + pix * lddx + q * 4;
    const long long offs[4] = {0, lddx, (long long)W * lddx, (long long)(W + 1) * lddx};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 v = ld4(o + offs[k]);
      v.x += kx == k ? g.x : 0.f; v.y += ky == k ? g.y : 0.f; v.z += kz == k ? g.z : 0.f; v.w += kw == k ? g.w : 0.f;
      st4(o + offs[k], v);
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x += v.x * (a[k].x - b4.x) * ig.x; s2.y += v.y * (a[k].y - b4.y) * ig.y;
      s2.z += v.z * (a[k].z - b4.z) * ig.z; s2.w += v.w * (a[k].w - b4.w) * ig.w;
    }
  }
  __shared__ float4 sh1[TPB], sh2[TPB];
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  if (tid < lpp) {
    for (int k = 1; k < TPB / lpp; ++k) {
      float4 u = sh1[tid + k * lpp], w2 = sh2[tid + k * lpp];
      s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w; s2.x += w2.x; s2.y += w2.y; s2.z += w2.z; s2.w += w2.w;
    }
    double* d1 = sums + (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES + q * 4; double* d2 = d1 + C;    // `sums` = the slot copies (deterministic mode: one writer per copy)
    atomicAdd(d1 + 0, (double)s1.x); atomicAdd(d1 + 1, (double)s1.y); atomicAdd(d1 + 2, (double)s1.z); atomicAdd(d1 + 3, (double)s1.w);
    atomicAdd(d2 + 0, (double)s2.x); atomicAdd(d2 + 1, (double)s2.y); atomicAdd(d2 + 2, (double)s2.z); atomicAdd(d2 + 3, (double)s2.w);
  }
}

// Encoder backward tail without a pass for the statistics (fp32 U-Net).  The gradient reaching the encoder BatchNorm output y is
// g = g_skip + route(dyp): g_skip comes out of the DECODER BatchNorm's backward (the skip half of its dx), dyp through max-pool + dropout.
//   * sum g_skip = 0 and sum g_skip * xhat = gamma_d S2_d eps istd_d^2 / gamma_e analytically (bn_bwd_skip_term_kernel: a BatchNorm backward output is
//     orthogonal to 1 and, up to eps / (var + eps), to its own xhat, and the decoder's xhat of a skip channel is gamma_e istd_d times the encoder's);
//   * the pooled path only touches the arg-max elements, whose y is the pooled activation p itself: sum over the POOLED tensors (1/4 of the pixels)
//     of dyp * ks and dyp * ks * (p / ks - beta) / gamma (pool_bwd_sums_kernel).
// With the sums known, ONE pass does pool backward + skip add + BatchNorm backward + ReLU mask: pool_bn_bwd_apply_kernel.
template <typename T>
__global__ __launch_bounds__(TPB) void pool_bwd_sums_kernel(const T* __restrict__ pooled, const T* __restrict__ dyp, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, double* sums, long long total, int C, float rate, uint64_t seed, int nslots) {
  const int lpp = C >> 2, tid = threadIdx.x, q = tid % lpp;                        // lpp divides TPB: a thread keeps its channel quad
  const float4 g4 = ld4(gamma + q * 4), b4 = ld4(beta + q * 4);
  const float4 ig = make_float4(g4.x != 0.f ? 1.f / g4.x : 0.f, g4.y != 0.f ? 1.f / g4.y : 0.f, g4.z != 0.f ? 1.f / g4.z : 0.f, g4.w != 0.f ? 1.f / g4.w : 0.f);
  const float unkeep = 1.0f - rate;                                                // p = max * keep_scale: the kept maxima are p * (1 - rate)
  float4 s1 = make_float4(0, 0, 0, 0), s2 = s1;
  for (long long i = (long long)blockIdx.x * TPB + tid; i < total; i += (long long)gridDim.x * TPB) {
    float4 g = ld4(dyp + i * 4);
    const float4 pv = ld4(pooled + i * 4);
    if (rate > 0.0f) { const float4 k = keep_scale(i, rate, seed); g.x *= k.x; g.y *= k.y; g.z *= k.z; g.w *= k.w; }
    s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
    s2.x += g.x * (pv.x * unkeep - b4.x) * ig.x; s2.y += g.y * (pv.y * unkeep - b4.y) * ig.y;
    s2.z += g.z * (pv.z * unkeep - b4.z) * ig.z; s2.w += g.w * (pv.w * unkeep - b4.w) * ig.w;
  }
  __shared__ float4 sh1[TPB], sh2[TPB];
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  if (tid < lpp) {
    for (int k = 1; k < TPB / lpp; ++k) {
      float4 u = sh1[tid + k * lpp], w2 = sh2[tid + k * lpp];
      s1.x += u.x; s1.y += u.y; s1.z += u.z; s1.w += u.w; s2.x += w2.x; s2.y += w2.y; s2.z += w2.z; s2.w += w2.w;
    }
    double* d1 = sums + (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES + q * 4; double* d2 = d1 + C;    // `sums` = the slot copies (deterministic mode: one writer per copy)
    atomicAdd(d1 + 0, (double)s1.x); atomicAdd(d1 + 1, (double)s1.y); atomicAdd(d1 + 2, (double)s1.z); atomicAdd(d1 + 3, (double)s1.w);
    atomicAdd(d2 + 0, (double)s2.x); atomicAdd(d2 + 1, (double)s2.y); atomicAdd(d2 + 2, (double)s2.z); atomicAdd(d2 + 3, (double)s2.w);
  }
}
// sums[C + j] += frac * gamma_d[j] * S2_d[j] * eps * istd_d[j]^2 / gamma_e[j]   (pointers of the decoder layer already offset to its skip half)
__global__ void bn_bwd_skip_term_kernel(double* __restrict__ sums, const double* __restrict__ dec_s2, const float* __restrict__ dec_istd, const float* __restrict__ dec_gamma,
                                        const float* __restrict__ gamma, int C, double frac, float eps) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= C || gamma[j] == 0.f) return;
  const double is = dec_istd[j];
  sums[C + j] += frac * (double)dec_gamma[j] * dec_s2[j] * (double)eps * is * is / (double)gamma[j];
}
// bn_slot_fold + bn_bwd_skip_term + bn_bwd_param_grads of an encoder tail in ONE launch (the pooled sums sit in the slot copies, left there by a data-gradient epilogue)
__global__ void enc_tail_finish_kernel(double* __restrict__ slots, double* __restrict__ sums, const double* __restrict__ dec_s2, const float* __restrict__ dec_istd,
                                       const float* __restrict__ dec_gamma, const float* __restrict__ gamma, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, double frac,
                                       float eps, int nslots, int xs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= C) return;
  double s1 = 0.0, s2 = 0.0;
  if (xs) { s1 = xsum_take(slots, j, nslots); s2 = xsum_take(slots, C + j, nslots); }
  else
  for (int k0 = 0; k0 < nslots; k0 += 8) {                  // eight independent load pairs in flight, summed in index order
    double v1[8], v2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { double* p = slots + (size_t)(k0 + k) * UNET_BN_SLOT_DOUBLES + j; v1[k] = p[0]; p[0] = 0.0; v2[k] = p[C]; p[C] = 0.0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { s1 += v1[k]; s2 += v2[k]; }
  }
  double t1 = sums[j] + s1, t2 = sums[C + j] + s2;
  if (gamma[j] != 0.f) { const double is = dec_istd[j]; t2 += frac * (double)dec_gamma[j] * dec_s2[j] * (double)eps * is * is / (double)gamma[j]; }
  sums[j] = t1; sums[C + j] = t2;
  dbeta[j] = (float)t1; dgamma[j] = (float)t2;
}
// one thread = one pooled pixel x 4 channels: y = BN(x) recomputed exactly as the forward stored it (arg-max), total gradient, BatchNorm backward, ReLU mask
template <typename T>
__global__ __launch_bounds__(TPB) void pool_bn_bwd_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ bnp, const double* __restrict__ sums, double inv_count,
                                                                const T* __restrict__ gskip, int ldg, const T* __restrict__ dyp, T* __restrict__ dx, int lddx, int N, int H,
                                                                int W, int C, float rate, uint64_t seed, const float* __restrict__ skip_k1) {
  const int lpp = C >> 2, Ho = H >> 1, Wo = W >> 1;
  const long long total = (long long)N * Ho * Wo * lpp;
  const int tid = threadIdx.x, q = tid % lpp;                        // lpp divides TPB: a thread keeps its channel quad
  const float4 sc = ld4(bnp + q * 4), sh = ld4(bnp + C + q * 4), mean = ld4(bnp + 2 * C + q * 4), istd = ld4(bnp + 3 * C + q * 4);
  const double* s1 = sums + q * 4; const double* s2 = sums + C + q * 4;
  const float4 k1 = make_float4((float)(s1[0] * inv_count), (float)(s1[1] * inv_count), (float)(s1[2] * inv_count), (float)(s1[3] * inv_count));
  const float4 k2 = make_float4((float)(s2[0] * inv_count), (float)(s2[1] * inv_count), (float)(s2[2] * inv_count), (float)(s2[3] * inv_count));
  // skip_k1: the decoder's data gradient left out the K1 * y term of its folded-BatchNorm backward for the skip channels (it did not read the skip tensor): g_skip here
  // holds K0 dz + K2 and the term is added from the y = BN(x) this kernel recomputes anyway
  const float4 kk = skip_k1 ? ld4(skip_k1 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long i = (long long)blockIdx.x * TPB + tid; i < total; i += (long long)gridDim.x * TPB) {
    const unsigned pu = (unsigned)i / (unsigned)lpp, tu = pu / (unsigned)Wo;             // 32-bit index math (launcher checks total < 2^31)
    const int jo = (int)(pu - tu * (unsigned)Wo), io = (int)(tu % (unsigned)Ho);
    const long long p = pu, n = tu / (unsigned)Ho;
    const long long pix = (n * H + 2 * io) * W + 2 * jo;
    const long long offs[4] = {0, 1, (long long)W, (long long)W + 1};
    float4 xv[4], yv[4], gs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { xv[k] = ld4(x + (pix + offs[k]) * ldx + q * 4); gs[k] = gskip ? ld4(gskip + (pix + offs[k]) * ldg + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f); }
    float4 g = ld4(dyp + p * C + q * 4);
    if (rate > 0.0f) { const float4 k = keep_scale(i, rate, seed); g.x *= k.x; g.y *= k.y; g.z *= k.z; g.w *= k.w; }
#pragma unroll
    for (int k = 0; k < 4; ++k) yv[k] = make_float4(fmaf(xv[k].x, sc.x, sh.x), fmaf(xv[k].y, sc.y, sh.y), fmaf(xv[k].z, sc.z, sh.z), fmaf(xv[k].w, sc.w, sh.w));
    const int kx = argmax4(yv[0].x, yv[1].x, yv[2].x, yv[3].x), ky = argmax4(yv[0].y, yv[1].y, yv[2].y, yv[3].y);
    const int kz = argmax4(yv[0].z, yv[1].z, yv[2].z, yv[3].z), kw = argmax4(yv[0].w, yv[1].w, yv[2].w, yv[3].w);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float tx = fmaf(kk.x, yv[k].x, gs[k].x) + (kx == k ? g.x : 0.f), ty = fmaf(kk.y, yv[k].y, gs[k].y) + (ky == k ? g.y : 0.f);
      const float tz = fmaf(kk.z, yv[k].z, gs[k].z) + (kz == k ? g.z : 0.f), tw = fmaf(kk.w, yv[k].w, gs[k].w) + (kw == k ? g.w : 0.f);
      float4 r;
      r.x = xv[k].x > 0.f ? sc.x * (tx - k1.x - (xv[k].x - mean.x) * istd.x * k2.x) : 0.f;
      r.y = xv[k].y > 0.f ? sc.y * (ty - k1.y - (xv[k].y - mean.y) * istd.y * k2.y) : 0.f;
      r.z = xv[k].z > 0.f ? sc.z * (tz - k1.z - (xv[k].z - mean.z) * istd.z * k2.z) : 0.f;
      r.w = xv[k].w > 0.f ? sc.w * (tw - k1.w - (xv[k].w - mean.w) * istd.w * k2.w) : 0.f;
      st4(dx + (pix + offs[k]) * lddx + q * 4, r);
    }
  }
}

// ---------------------------------------------------------------------------------------
// 1x1 conv + sigmoid head fused with the loss reductions; and its backward.
// lpp = cin/4 lanes per pixel (power of two <= 64), xor-shuffle dot product.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bce_elem(float p, float t, float* pc_out, bool* inrange) {
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  float pc = fminf(fmaxf(p, lo), hi);
  *pc_out = pc; *inrange = (p >= lo) && (p <= hi);
  float z = logf(pc / (1.0f - pc));
  return fmaxf(z, 0.0f) - z * t + log1pf(expf(-fabsf(z)));
}

template <typename T>
__global__ __launch_bounds__(TPB) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ pout,
                                                       const float* __restrict__ yt, double* sums,
                                                       long long pixels, int cin, int nslots) {
  if (nslots) sums += (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES;          // deterministic mode: `sums` = the slot copies, one writer per copy
  const int lpp = cin >> 2;
  const int sub = threadIdx.x & (lpp - 1);
  const float4 wv = ld4(w + sub * 4);
  const float b = bias[0];
  float sb = 0, stp = 0, st = 0, sp = 0;
  const long long gt0 = ((long long)blockIdx.x * TPB + threadIdx.x) / lpp;
  const long long gstride = ((long long)gridDim.x * TPB) / lpp;
  const long long iters = (pixels + gstride - 1) / gstride;   // uniform trip count: shuffles stay converged
  for (long long it = 0; it < iters; ++it) {
    long long p = gt0 + it * gstride;
    bool ok = p < pixels;
    float4 v = ok ? ld4(x + p * cin + sub * 4) : make_float4(0, 0, 0, 0);
    float d = v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
    for (int o = lpp >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
    if (ok && sub == 0) {
      float pr = 1.0f / (1.0f + expf(-(d + b)));
      pout[p] = pr;
      if (yt) {
        float t = yt[p], pc; bool inr;
        sb += bce_elem(pr, t, &pc, &inr); stp += t * pr; st += t; sp += pr;
      }
    }
  }
  if (yt) {
    __shared__ float red[4][TPB / 64];
    sb = wave_sum(sb); stp = wave_sum(stp); st = wave_sum(st); sp = wave_sum(sp);
    int wv_ = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv_] = sb; red[1][wv_] = stp; red[2][wv_] = st; red[3][wv_] = sp; }
    __syncthreads();
    if (threadIdx.x < 4) {
      float s = 0;
      for (int k = 0; k < TPB / 64; ++k) s += red[threadIdx.x][k];
      atomicAdd(sums + threadIdx.x, (double)s);
    }
  }
}

// Same op, LPP lanes per pixel known at compile time: a lane group walks LPP pixels per iteration (LPP independent loads in flight)
// and lane `sub` then does the sigmoid / BCE / Dice arithmetic of pixel `sub` -- in the kernel above only one lane in LPP does that
// (transcendental-bound: the same 0.18 ms in fp32 and in bf16); here every lane does, and the probability stores are contiguous.
template <typename T, int LPP>
__global__ __launch_bounds__(TPB) void head_fwd_lpp_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ pout,
                                                           const float* __restrict__ yt, double* sums, long long pixels, int nslots) {
  if (nslots) sums += (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES;          // deterministic mode: `sums` = the slot copies, one writer per copy
  constexpr int cin = LPP * 4;
  const int sub = threadIdx.x & (LPP - 1);
  const float4 wv = ld4(w + sub * 4);
  const float b = bias[0];
  float sb = 0, stp = 0, st = 0, sp = 0;
  const long long g0 = ((long long)blockIdx.x * TPB + threadIdx.x) / LPP * LPP;          // first pixel of this lane group
  const long long gstride = (long long)gridDim.x * TPB;                                  // pixels per sweep of the whole grid
  const long long iters = (pixels + gstride - 1) / gstride;                               // uniform trip count: shuffles stay converged
  for (long long it = 0; it < iters; ++it) {
    const long long base = g0 + it * gstride;
    float4 v[LPP];
#pragma unroll
    for (int k = 0; k < LPP; ++k) v[k] = base + k < pixels ? ld4(x + (base + k) * cin + sub * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < LPP; ++k) {
      float d = v[k].x * wv.x + v[k].y * wv.y + v[k].z * wv.z + v[k].w * wv.w;
#pragma unroll
      for (int o = LPP >> 1; o > 0; o >>= 1) d += __shfl_xor(d, o, 64);
      mine = (k == sub) ? d : mine;
    }
    const long long p = base + sub;
    if (p < pixels) {
      const float pr = 1.0f / (1.0f + expf(-(mine + b)));
      pout[p] = pr;
      if (yt) {
        float t = yt[p], pc; bool inr;
        sb += bce_elem(pr, t, &pc, &inr); stp += t * pr; st += t; sp += pr;
      }
    }
  }
  if (yt) {
    __shared__ float red[4][TPB / 64];
    sb = wave_sum(sb); stp = wave_sum(stp); st = wave_sum(st); sp = wave_sum(sp);
    int wv_ = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv_] = sb; red[1][wv_] = stp; red[2][wv_] = st; red[3][wv_] = sp; }
    __syncthreads();
    if (threadIdx.x < 4) {
      float s = 0;
      for (int k = 0; k < TPB / 64; ++k) s += red[threadIdx.x][k];
      atomicAdd(sums + threadIdx.x, (double)s);
    }
  }
}

__global__ void loss_finalize_kernel(const double* sums, double count, float* out, float* out2) {
  double dice = (2.0 * sums[1] + 1.0) / (sums[2] + sums[3] + 1.0);
  const float l = (float)(0.5 * (sums[0] / count) + 0.5 * (1.0 - dice)), d = (float)dice;
  out[0] = l; out[1] = d;
  if (out2) { out2[0] = l; out2[1] = d; }                     // (unet_model_set_loss_out: the caller's own copy of this step's pair -- no copy kernel behind the step)
}

template <typename T>
__global__ __launch_bounds__(TPB) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ pin, const float* __restrict__ yt,
                                                       const double* __restrict__ sums, double inv_count,
                                                       T* __restrict__ dx, float* dw, float* db,
                                                       long long pixels, int cin, int relu_mask, double* slots, int nslots) {
  const int lpp = cin >> 2;
  const int sub = threadIdx.x & (lpp - 1);
  const float4 wv = ld4(w + sub * 4);
  const double S = sums[2] + sums[3] + 1.0;
  const float dice = (float)((2.0 * sums[1] + 1.0) / S), invS = (float)(1.0 / S);
  const float hb = (float)(0.5 * inv_count);
  float4 aw = make_float4(0, 0, 0, 0); float ab = 0;
  const long long gt0 = ((long long)blockIdx.x * TPB + threadIdx.x) / lpp;
  const long long gstride = ((long long)gridDim.x * TPB) / lpp;
  for (long long p = gt0; p < pixels; p += gstride) {
    float pr = pin[p], t = yt[p], pc; bool inr;
    (void)bce_elem(pr, t, &pc, &inr);
    // d(BCE)/dz = (p - t) inside the clip range (the p(1-p) of the sigmoid cancels the 1/(p(1-p)) of the
    // log terms analytically -- no 1-p cancellation noise near saturation); the Dice part keeps p(1-p)
    float dz = (inr ? hb * (pc - t) : 0.0f) - 0.5f * (2.0f * t - dice) * invS * pr * (1.0f - pr);
    float4 v = ld4(x + p * cin + sub * 4);
    st4(dx + p * cin + sub * 4, make_float4((!relu_mask || v.x > 0) ? dz * wv.x : 0.f, (!relu_mask || v.y > 0) ? dz * wv.y : 0.f,
                                            (!relu_mask || v.z > 0) ? dz * wv.z : 0.f, (!relu_mask || v.w > 0) ? dz * wv.w : 0.f));
    aw.x += dz * v.x; aw.y += dz * v.y; aw.z += dz * v.z; aw.w += dz * v.w;
    if (sub == 0) ab += dz;
  }
  // reduce lanes sharing `sub` inside the wave, then the 4 waves through LDS
  for (int o = lpp; o < 64; o <<= 1) {
    aw.x += __shfl_xor(aw.x, o, 64); aw.y += __shfl_xor(aw.y, o, 64);
    aw.z += __shfl_xor(aw.z, o, 64); aw.w += __shfl_xor(aw.w, o, 64);
    ab += __shfl_xor(ab, o, 64);
  }
  __shared__ float4 rw[TPB / 64][64]; __shared__ float rb[TPB / 64];
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
  if (lane < lpp) rw[wv_][lane] = aw;
  if (lane == 0) rb[wv_] = ab;
  __syncthreads();
  if (threadIdx.x < lpp) {
    float4 s = rw[0][threadIdx.x];
    for (int k = 1; k < TPB / 64; ++k) { float4 u = rw[k][threadIdx.x]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
    if (nslots) {                                          // deterministic mode: one writer per slot copy ([cin] dw, then db), folded in index order afterwards
      double* d = slots + (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES + threadIdx.x * 4;
      atomicAdd(d + 0, (double)s.x); atomicAdd(d + 1, (double)s.y); atomicAdd(d + 2, (double)s.z); atomicAdd(d + 3, (double)s.w);
    } else {
      atomicAdd(dw + threadIdx.x * 4 + 0, s.x); atomicAdd(dw + threadIdx.x * 4 + 1, s.y);
      atomicAdd(dw + threadIdx.x * 4 + 2, s.z); atomicAdd(dw + threadIdx.x * 4 + 3, s.w);
    }
  }
  if (threadIdx.x == 0) {
    float s = 0; for (int k = 0; k < TPB / 64; ++k) s += rb[k];
    if (nslots) atomicAdd(slots + (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES + cin, (double)s); else atomicAdd(db, s);
  }
}

// ---- the fused head (kernels_conv_h2.hip: conv_h2_kernel<..., HEAD>) -------------------------------------------------
// fold of the 103 sums it left in the slot copies (cleared for the next launch): [96, 100) -> loss_sums, the rest -> head_sums[99]
__global__ void head_fold_kernel(double* __restrict__ slots, double* __restrict__ loss_sums, double* __restrict__ head_sums, int nslots, int xs) {
  const int i = threadIdx.x;
  if (i >= 103) return;
  double s = 0.0;
  if (xs) s = xsum_take(slots, i, nslots);
  else
  for (int k0 = 0; k0 < nslots; k0 += 16) {
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { double* p = slots + (size_t)(k0 + k) * UNET_BN_SLOT_DOUBLES + i; v[k] = *p; *p = 0.0; }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
  if (i < 96) head_sums[i] += s; else if (i < 100) loss_sums[i - 96] += s; else head_sums[96 + (i - 100)] += s;
}

// dz of pixel p from the stored probability, the label and the batch-global sums (the same expression as head_bwd_kernel)
__device__ __forceinline__ float head_dz(float pr, float t, float hb, float dice, float invS) {
  const float lo = 1e-7f, hi = 1.0f - 1e-7f;
  const float pc = fminf(fmaxf(pr, lo), hi);
  const bool inr = (pr >= lo) && (pr <= hi);
  return (inr ? hb * (pc - t) : 0.0f) - 0.5f * (2.0f * t - dice) * invS * pr * (1.0f - pr);
}

// dy[p][c] = dz_p w_c [y_pc > 0], 32 channels: a thread writes one 16-B quad; the mask is one bit per element (the layout conv_h2_kernel writes:
// per (row, 8 pixels) four 64-bit words, bit (pixel % 8) * 8 + quad of word k = channel quad * 4 + k) or, without bits, y itself.
// Workgroup 0 also finishes the head's own gradient: dz = hb a - invS t q + 0.5 invS dice q, so
//   dw_c = hb S1_c - invS S2_c + 0.5 invS dice S3_c  with the three per-channel sums the forward epilogue took; db the same with the scalar sums
__global__ __launch_bounds__(TPB) void head_dy_kernel(const float* __restrict__ pin, const float* __restrict__ yt, const double* __restrict__ sums, double inv_count,
                                                      const double* __restrict__ hs, const float* __restrict__ w, const unsigned long long* __restrict__ bits,
                                                      const float* __restrict__ y, float* __restrict__ dy, float* dw, float* db, long long pixels, int wd) {
  const double S = sums[2] + sums[3] + 1.0;
  const float dice = (float)((2.0 * sums[1] + 1.0) / S), invS = (float)(1.0 / S);
  const float hb = (float)(0.5 * inv_count);
  if (blockIdx.x == 0 && threadIdx.x < 33) {
    const int c = threadIdx.x;
    const double g = c < 32 ? (0.5 * inv_count) * hs[c] - (1.0 / S) * hs[32 + c] + 0.5 * (1.0 / S) * ((2.0 * sums[1] + 1.0) / S) * hs[64 + c]
                            : (0.5 * inv_count) * hs[96] - (1.0 / S) * hs[97] + 0.5 * (1.0 / S) * ((2.0 * sums[1] + 1.0) / S) * hs[98];
    if (c < 32) dw[c] += (float)g; else db[0] += (float)g;
  }
  const int q = threadIdx.x & 7;
  const float4 wv = ld4(w + q * 4);
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < pixels * 8; i += (long long)gridDim.x * TPB) {
    const long long p = i >> 3;
    const float dz = head_dz(pin[p], yt[p], hb, dice, invS);
    bool m0, m1, m2, m3;
    if (bits) {
      const long long row = p / wd; const int x = (int)(p - row * wd);
      const unsigned long long* bw = bits + (row * (wd >> 3) + (x >> 3)) * 4;
      const int sh = (x & 7) * 8 + q;
      m0 = (bw[0] >> sh) & 1; m1 = (bw[1] >> sh) & 1; m2 = (bw[2] >> sh) & 1; m3 = (bw[3] >> sh) & 1;
    } else {
      const float4 v = ld4(y + i * 4);
      m0 = v.x > 0.f; m1 = v.y > 0.f; m2 = v.z > 0.f; m3 = v.w > 0.f;
    }
    st4(dy + i * 4, make_float4(m0 ? dz * wv.x : 0.f, m1 ? dz * wv.y : 0.f, m2 ? dz * wv.z : 0.f, m3 ? dz * wv.w : 0.f));
  }
}

// The same gradient as a rank-1 stream: dy[p][c] = dz_p w_c [y_pc > 0] has ONE fp32 degree of freedom and 32 mask bits per pixel.  out[p] = {dz_p, mask_p}
// (bit c of mask_p = y_pc > 0; 8 bytes instead of the 128 head_dy_kernel writes): the data gradient and the weight gradient of the last conv3x3 expand it while they
// stage it (kernels_conv_h2.hip EPI 3, kernels_wgrad_h2.hip VDY) -- w_c goes into the data gradient's weight image / the weight gradient's column scale.
// A thread: one pixel.  The sign bits of 8 pixels x 32 channels are four 64-bit words (layout above): byte x % 8 of word k holds channels k, 4 + k, 8 + k, ...
__global__ __launch_bounds__(TPB) void head_dzm_kernel(const float* __restrict__ pin, const float* __restrict__ yt, const double* __restrict__ sums, double inv_count,
                                                       const double* __restrict__ hs, const unsigned long long* __restrict__ bits, uint2* __restrict__ out, float* dw, float* db,
                                                       long long pixels, int wd) {
  const double S = sums[2] + sums[3] + 1.0;
  const float dice = (float)((2.0 * sums[1] + 1.0) / S), invS = (float)(1.0 / S);
  const float hb = (float)(0.5 * inv_count);
  if (blockIdx.x == 0 && threadIdx.x < 33) {
    const int c = threadIdx.x;
    const double g = c < 32 ? (0.5 * inv_count) * hs[c] - (1.0 / S) * hs[32 + c] + 0.5 * (1.0 / S) * ((2.0 * sums[1] + 1.0) / S) * hs[64 + c]
                            : (0.5 * inv_count) * hs[96] - (1.0 / S) * hs[97] + 0.5 * (1.0 / S) * ((2.0 * sums[1] + 1.0) / S) * hs[98];
    if (c < 32) dw[c] += (float)g; else db[0] += (float)g;
  }
  for (long long p = (long long)blockIdx.x * TPB + threadIdx.x; p < pixels; p += (long long)gridDim.x * TPB) {
    const float dz = head_dz(pin[p], yt[p], hb, dice, invS);
    const long long row = p / wd; const int x = (int)(p - row * wd);
    const unsigned long long* bw = bits + (row * (wd >> 3) + (x >> 3)) * 4;
    const int sh = (x & 7) * 8;
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned b = (unsigned)(bw[k] >> sh) & 0xFFu;          // bit Q = channel 4 Q + k  ->  spread to every fourth bit
      b = (b | (b << 12)) & 0x000F000Fu; b = (b | (b << 6)) & 0x03030303u; b = (b | (b << 3)) & 0x11111111u;
      m |= b << k;
    }
    out[p] = make_uint2(__float_as_uint(dz), m);
  }
}

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps, float gs) {
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
    float4 pp = ld4(p + i * 4), gg = ld4(g + i * 4), mm = ld4(m + i * 4), vv = ld4(v + i * 4);
    float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x * gs, gg.y * gs, gg.z * gs, gg.w * gs};
    float M[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      M[k] = b1 * M[k] + (1.0f - b1) * G[k];
      V[k] = b2 * V[k] + (1.0f - b2) * G[k] * G[k];
      P[k] = P[k] - lr_t * M[k] / (sqrtf(V[k]) + eps);
    }
    st4(p + i * 4, make_float4(P[0], P[1], P[2], P[3]));
    st4(m + i * 4, make_float4(M[0], M[1], M[2], M[3]));
    st4(v + i * 4, make_float4(V[0], V[1], V[2], V[3]));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    long long i = (n4 << 2) + threadIdx.x;
    float G = g[i] * gs, M = b1 * m[i] + (1.0f - b1) * G, V = b2 * v[i] + (1.0f - b2) * G * G;
    m[i] = M; v[i] = V; p[i] = p[i] - lr_t * M / (sqrtf(V) + eps);
  }
}

constexpr int THR_CHUNK = 8;
__global__ __launch_bounds__(TPB) void metrics_sweep_kernel(const float* __restrict__ p, const float* __restrict__ gt,
                                                            const float* __restrict__ thr, int nthr, double* out,
                                                            long long n, int nslots) {
  if (nslots) out += (size_t)(blockIdx.x % nslots) * UNET_BN_SLOT_DOUBLES;          // deterministic mode: `out` = the slot copies, one writer (blockIdx.x) per copy
  const int t0 = blockIdx.y * THR_CHUNK;
  float th[THR_CHUNK], tp[THR_CHUNK], pr[THR_CHUNK], sg = 0;
#pragma unroll
  for (int k = 0; k < THR_CHUNK; ++k) { th[k] = (t0 + k < nthr) ? thr[t0 + k] : 2.0f; tp[k] = 0; pr[k] = 0; }
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
    float pv = p[i], g = gt[i];
    sg += g;
#pragma unroll
    for (int k = 0; k < THR_CHUNK; ++k) { bool on = pv > th[k]; tp[k] += on ? g : 0.f; pr[k] += on ? 1.f : 0.f; }
  }
  __shared__ float red[2 * THR_CHUNK + 1][TPB / 64];
  const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < THR_CHUNK; ++k) {
    float a = wave_sum(tp[k]), b = wave_sum(pr[k]);
    if (lane == 0) { red[2 * k][wv_] = a; red[2 * k + 1][wv_] = b; }
  }
  sg = wave_sum(sg);
  if (lane == 0) red[2 * THR_CHUNK][wv_] = sg;
  __syncthreads();
  if (threadIdx.x < 2 * THR_CHUNK + 1) {
    float s = 0;
    for (int k = 0; k < TPB / 64; ++k) s += red[threadIdx.x][k];
    if (threadIdx.x == 2 * THR_CHUNK) {
      for (int k = 0; k < THR_CHUNK; ++k) if (t0 + k < nthr) atomicAdd(out + (t0 + k) * 3 + 2, (double)s);
    } else {
      int k = threadIdx.x >> 1;
      if (t0 + k < nthr) atomicAdd(out + (t0 + k) * 3 + (threadIdx.x & 1), (double)s);
    }
  }
}

// dst[slice] = src[slice] (materialise a skip tensor inside a concat buffer) and dst[slice] (+)= sum of up to 4 gradient slices
template <typename T>
__global__ __launch_bounds__(TPB) void copy_slice_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd,
                                                         long long pixels, int C) {
  const int lpp = C >> 2; const long long total = pixels * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    int q = (int)(i % lpp); long long p = i / lpp;
    st4(dst + p * ldd + q * 4, ld4(src + p * lds + q * 4));
  }
}
template <typename T> struct SliceList { const T* p[4]; int ld[4]; int n; };
template <typename T>
__global__ __launch_bounds__(TPB) void accum_slices_kernel(SliceList<T> sl, T* __restrict__ dst, int ldd, long long pixels, int C,
                                                           int accumulate) {
  const int lpp = C >> 2; const long long total = pixels * lpp;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    int q = (int)(i % lpp); long long p = i / lpp;
    float4 s = accumulate ? ld4(dst + p * ldd + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < sl.n; ++k) { const float4 v = ld4(sl.p[k] + p * sl.ld[k] + q * 4); s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    st4(dst + p * ldd + q * 4, s);
  }
}

inline int grid_for(long long work_items) {
  long long b = cdiv64(work_items, TPB);
  return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}
__global__ __launch_bounds__(TPB) void zero_kernel(float4* __restrict__ p, long long n4, int tail_words) {          // n4 16-byte pieces + up to three 4-byte words behind them
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && (int)threadIdx.x < tail_words) reinterpret_cast<float*>(p + n4)[threadIdx.x] = 0.f;
}

// sums[i] += sum over the slot copies IN INDEX ORDER; the copies are cleared for the next launch.  OUT = double (statistics) or float (the head's
// weight gradient in deterministic mode)
template <typename OUT>
__global__ void bn_slot_fold_kernel(double* __restrict__ slots, OUT* __restrict__ sums, int n2c, int nslots, int xs = 0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2c) return;
  if (xs) { sums[i] += (OUT)xsum_take(slots, i, nslots); return; }          // (exact window sums left by a kernel epilogue in deterministic mode, common.h)
  double s = 0.0;
  for (int k0 = 0; k0 < nslots; k0 += 16) {                 // (nslots is a multiple of 16) sixteen independent loads in flight, summed in index order
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { double* p = slots + (size_t)(k0 + k) * UNET_BN_SLOT_DOUBLES + i; v[k] = *p; *p = 0.0; }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
  }
  sums[i] += (OUT)s;
}

// Statistics of a concat([up, skip]) whose skip half is the OUTPUT of an earlier training-mode BatchNorm: y = gamma * xhat + beta has, over the
// batch, mean(y) = beta and var(y) = gamma^2 * var / (var + eps) exactly, so its (sum, sum of squares) follow from the source layer's sums
// without reading the tensor.  Thread i < c_up folds the measured sums of the up half out of the slot copies (as bn_slot_fold_kernel), thread
// c_up + j writes the analytic pair of skip channel j.  Layout of `sums`: [c_up + c_skip sums][c_up + c_skip sums of squares].
__global__ void bn_fold_concat_kernel(double* __restrict__ slots, double* __restrict__ sums, int c_up, int c_skip, const double* __restrict__ src_sums,
                                      double src_count, const float* __restrict__ src_gamma, const float* __restrict__ src_beta, double pixels, float eps, int nslots, int xs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int c2 = c_up + c_skip;
  if (i < c_up && xs) {
    sums[i] += xsum_take(slots, i, nslots); sums[c2 + i] += xsum_take(slots, c_up + i, nslots);
  } else if (i < c_up) {
    double s1 = 0.0, s2 = 0.0;
    for (int k0 = 0; k0 < nslots; k0 += 8) {                // eight independent load pairs in flight, summed in index order
      double v1[8], v2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { double* p = slots + (size_t)(k0 + k) * UNET_BN_SLOT_DOUBLES + i; v1[k] = p[0]; p[0] = 0.0; v2[k] = p[c_up]; p[c_up] = 0.0; }
#pragma unroll
      for (int k = 0; k < 8; ++k) { s1 += v1[k]; s2 += v2[k]; }
    }
    sums[i] += s1; sums[c2 + i] += s2;
  } else if (i < c2) {
    const int j = i - c_up;
    const double mean = src_sums[j] / src_count;
    double var = src_sums[c_skip + j] / src_count - mean * mean;
    if (var < 0) var = 0;
    const double g = (double)src_gamma[j], b = (double)src_beta[j];
    sums[i] += pixels * b;
    sums[c2 + i] += pixels * (g * g * var / (var + (double)eps) + b * b);
  }
}

inline bool bn_c_ok(int c) { return c >= 4 && (c % 4) == 0 && (c / 4) <= TPB; }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" {

extern "C++" template <typename T> static int32_t bn_stats_impl(unet_ctx* ctx, const T* x, int32_t ldx, double* sums, int64_t pixels, int32_t c, void* stream) {
  if (!ctx || !x || !sums || !bn_c_ok(c) || ldx < c || (ldx & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_stats: bad args c=%d ldx=%d", c, ldx);
  ctx->stats_req_c = 0;                                    // (a request no launch took up)
  if (ctx->stats_in_slots && (ctx->stats_in_slots != (const void*)x || ctx->stats_in_slots_c != c)) {
    // the slot copies hold the fused statistics of ANOTHER tensor (an aborted program, a partial op range, a caller that skipped the call that must follow an armed
    // conv): recover instead of failing for ever -- clear the slots on the stream and take the statistics by the normal pass
    UNET_HIP(ctx, hipMemsetAsync(ctx->bn_slots, 0, sizeof(double) * UNET_BN_SLOTS_DET * UNET_BN_SLOT_DOUBLES, as_stream(stream)));
    ctx->stats_in_slots = nullptr; ctx->stats_in_slots_c = 0;
  }
  if (!ctx->stats_in_slots) {
    int ppb = TPB / (c / 4);
    int grid = (int)std::min<int64_t>(cdiv64(pixels, (int64_t)ppb * 16), BN_STATS_BLOCKS); if (grid < 1) grid = 1;
    hipLaunchKernelGGL((bn_stats_kernel<0, T>), dim3(grid), dim3(TPB), 0, as_stream(stream), x, ldx, (const T*)nullptr, 0, nullptr, ctx->bn_slots, (long long)pixels, c, ctx->bn_nslots());
  }
  const bool xs = ctx->stats_in_slots && ctx->stats_in_slots_xs;          // (left by a conv epilogue in deterministic mode: exact window sums in the first UNET_BN_SLOTS copies)
  ctx->stats_in_slots = nullptr;                           // (else: the conv that wrote x left them there -- fold only)
  hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((2 * c + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, sums, 2 * c, xs ? UNET_BN_SLOTS : ctx->bn_nslots(), xs ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "bn_stats"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_stats_concat_impl(unet_ctx* ctx, const T* x_up, int32_t ldx, const double* src_sums, double src_count, const float* src_gamma,
                                                                       const float* src_beta, double* sums, int64_t pixels, int32_t c_up, int32_t c_skip, void* stream) {
  if (!ctx || !x_up || !sums || !src_sums || !src_gamma || !src_beta || !bn_c_ok(c_up) || c_skip < 1 || ldx < c_up || (ldx & 3) || src_count < 1 || pixels < 1)
    UNET_FAIL(ctx, UNET_E_ARG, "bn_stats_concat: bad args c_up=%d c_skip=%d ldx=%d", c_up, c_skip, ldx);
  ctx->stats_req_c = 0;
  if (ctx->stats_in_slots && (ctx->stats_in_slots != (const void*)x_up || ctx->stats_in_slots_c != c_up)) {          // (see bn_stats_impl: recover, do not fail for ever)
    UNET_HIP(ctx, hipMemsetAsync(ctx->bn_slots, 0, sizeof(double) * UNET_BN_SLOTS_DET * UNET_BN_SLOT_DOUBLES, as_stream(stream)));
    ctx->stats_in_slots = nullptr; ctx->stats_in_slots_c = 0;
  }
  if (!ctx->stats_in_slots) {
    int ppb = TPB / (c_up / 4);
    int grid = (int)std::min<int64_t>(cdiv64(pixels, (int64_t)ppb * 16), BN_STATS_BLOCKS); if (grid < 1) grid = 1;
    hipLaunchKernelGGL((bn_stats_kernel<0, T>), dim3(grid), dim3(TPB), 0, as_stream(stream), x_up, ldx, (const T*)nullptr, 0, nullptr, ctx->bn_slots, (long long)pixels, c_up, ctx->bn_nslots());
  }
  const bool xs = ctx->stats_in_slots && ctx->stats_in_slots_xs;
  ctx->stats_in_slots = nullptr;                           // (else: the ConvT that wrote the up half left its sums there)
  hipLaunchKernelGGL(bn_fold_concat_kernel, dim3((c_up + c_skip + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, sums, c_up, c_skip, src_sums, src_count, src_gamma,
                     src_beta, (double)pixels, 1e-3f, xs ? UNET_BN_SLOTS : ctx->bn_nslots(), xs ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "bn_stats_concat"); return UNET_OK;
}

int32_t unet_bn_finalize_train(unet_ctx* ctx, const double* sums, double count, const float* gamma, const float* beta,
                               float* mm, float* mv, float* bnp, int32_t c, void* stream) {
  if (!sums || !gamma || !beta || !mm || !mv || !bnp || c < 1 || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_finalize_train: bad args");
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((c + 127) / 128), dim3(128), 0, as_stream(stream), sums, count, gamma, beta, mm, mv, bnp, c, 0.99f, 1e-3f);
  UNET_CHECK_LAUNCH(ctx, "bn_finalize_train"); return UNET_OK;
}

int32_t unet_bn_finalize_infer(unet_ctx* ctx, const float* gamma, const float* beta, const float* mm, const float* mv,
                               float* bnp, int32_t c, void* stream) {
  if (!gamma || !beta || !mm || !mv || !bnp || c < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_finalize_infer: bad args");
  hipLaunchKernelGGL(bn_finalize_infer_kernel, dim3((c + 127) / 128), dim3(128), 0, as_stream(stream), gamma, beta, mm, mv, bnp, c, 1e-3f);
  UNET_CHECK_LAUNCH(ctx, "bn_finalize_infer"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_apply_impl(unet_ctx* ctx, const T* x, int32_t ldx, const float* bnp, T* y, int32_t ldy, int64_t pixels,
                      int32_t c, void* stream) {
  if (!x || !bnp || !y || !bn_c_ok(c) || ldx < c || ldy < c || ((ldx | ldy) & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_apply: bad args");
  hipLaunchKernelGGL(bn_apply_kernel<T>, dim3(grid_for(pixels * (c / 4))), dim3(TPB), 0, as_stream(stream), x, ldx, bnp, y, ldy, (long long)pixels, c);
  UNET_CHECK_LAUNCH(ctx, "bn_apply"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_bwd_stats_impl(unet_ctx* ctx, const T* dy, int32_t lddy, const T* x, int32_t ldx, const float* bnp,
                          double* sums, int64_t pixels, int32_t c, void* stream) {
  if (!ctx || !dy || !x || !bnp || !sums || !bn_c_ok(c) || ((ldx | lddy) & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_stats: bad args");
  int ppb = TPB / (c / 4);
  int grid = (int)std::min<int64_t>(cdiv64(pixels, (int64_t)ppb * 16), BN_BWD_STATS_BLOCKS); if (grid < 1) grid = 1;
  hipLaunchKernelGGL((bn_stats_kernel<1, T>), dim3(grid), dim3(TPB), 0, as_stream(stream), dy, lddy, x, ldx, bnp, ctx->bn_slots, (long long)pixels, c, ctx->bn_nslots());
  hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((2 * c + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, sums, 2 * c, ctx->bn_nslots());
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_stats"); return UNET_OK;
}

int32_t unet_bn_bwd_param_grads(unet_ctx* ctx, const double* sums, float* dgamma, float* dbeta, int32_t c, void* stream) {
  if (!sums || !dgamma || !dbeta || c < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_param_grads: bad args");
  hipLaunchKernelGGL(bn_bwd_param_grads_kernel, dim3((c + 127) / 128), dim3(128), 0, as_stream(stream), sums, dgamma, dbeta, c);
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_param_grads"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_bwd_apply_impl(unet_ctx* ctx, const T* dy, int32_t lddy, const T* x, int32_t ldx, const float* bnp,
                          const double* sums, double count, int32_t mask_mode, float mask_rate, uint64_t mask_seed, T* dx,
                          int32_t lddx, int64_t pixels, int32_t c, void* stream) {
  if (!dy || !x || !bnp || !sums || !dx || !bn_c_ok(c) || count < 1 || ((ldx | lddy | lddx) & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_apply: bad args");
  if (mask_mode < 0 || mask_mode > 3 || (mask_mode == MASK_ELU_DROP && (ldx != c || mask_rate < 0 || mask_rate >= 1))) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_apply: bad mask mode (MASK_ELU_DROP needs a dense x)");
#define UNET_LAUNCH_BNBA(MM_)                                                                                                        \
  hipLaunchKernelGGL((bn_bwd_apply_kernel<MM_, T>), dim3(grid_for(pixels * (c / 4))), dim3(TPB), 0, as_stream(stream), dy, lddy, x, ldx, bnp, \
                     sums, 1.0 / count, dx, lddx, (long long)pixels, c, mask_rate, (unsigned long long)mask_seed)
  switch (mask_mode) {
    case MASK_NONE: UNET_LAUNCH_BNBA(MASK_NONE); break;
    case MASK_RELU: UNET_LAUNCH_BNBA(MASK_RELU); break;
    case MASK_ELU: UNET_LAUNCH_BNBA(MASK_ELU); break;
    default: UNET_LAUNCH_BNBA(MASK_ELU_DROP); break;
  }
#undef UNET_LAUNCH_BNBA
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_apply"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t maxpool_fwd_impl(unet_ctx* ctx, const T* x, int32_t ldx, T* y, int32_t n, int32_t h, int32_t wd,
                                    int32_t c, float rate, uint64_t seed, void* stream) {
  if (!x || !y || (c & 3) || (h & 1) || (wd & 1) || ldx < c || (ldx & 3) || rate < 0 || rate >= 1) UNET_FAIL(ctx, UNET_E_ARG, "maxpool fwd: bad args (h,w even; c%%4==0)");
  long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  hipLaunchKernelGGL(pool_fwd_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, as_stream(stream), x, ldx, y, n, h, wd, c, rate, seed);
  UNET_CHECK_LAUNCH(ctx, "maxpool fwd"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t maxpool_bwd_impl(unet_ctx* ctx, const T* x, int32_t ldx, const T* dy, T* dx, int32_t lddx,
                                    int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed,
                                    int32_t accumulate, void* stream) {
  if (!x || !dy || !dx || (c & 3) || (h & 1) || (wd & 1) || ((ldx | lddx) & 3) || rate < 0 || rate >= 1) UNET_FAIL(ctx, UNET_E_ARG, "maxpool bwd: bad args");
  long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  int grid = grid_for(total);
  if (accumulate) hipLaunchKernelGGL((pool_bwd_kernel<true, T>), dim3(grid), dim3(TPB), 0, as_stream(stream), x, ldx, dy, dx, lddx, n, h, wd, c, rate, seed);
  else hipLaunchKernelGGL((pool_bwd_kernel<false, T>), dim3(grid), dim3(TPB), 0, as_stream(stream), x, ldx, dy, dx, lddx, n, h, wd, c, rate, seed);
  UNET_CHECK_LAUNCH(ctx, "maxpool bwd"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_apply_maxpool_impl(unet_ctx* ctx, const T* x, int32_t ldx, const float* bnp, T* y, int32_t ldy,
                                          T* pooled, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed,
                                          void* stream) {
  if (!x || !bnp || !pooled || !bn_c_ok(c) || (h & 1) || (wd & 1) || ldx < c || (y && (ldy < c || (ldy & 3))) || (ldx & 3) || rate < 0 || rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "bn_apply_maxpool fwd: bad args (h,w even; c%%4==0)");
  long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  hipLaunchKernelGGL(bn_pool_fwd_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, as_stream(stream), x, ldx, bnp, y, ldy, pooled, n, h, wd, c, rate, seed);
  UNET_CHECK_LAUNCH(ctx, "bn_apply_maxpool fwd"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t maxpool_bwd_bnstats_impl(unet_ctx* ctx, const T* y, int32_t ldy, const T* dy, T* dx, int32_t lddx,
                                            const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd,
                                            int32_t c, float rate, uint64_t seed, void* stream) {
  if (!ctx || !y || !dy || !dx || !gamma || !beta || !sums || (c & 3) || TPB % (c / 4) || (h & 1) || (wd & 1) || ((ldy | lddx) & 3) || rate < 0 || rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "maxpool bwd + bn stats: bad args (c/4 must divide 256)");
  long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  int grid = (int)std::min<long long>(cdiv64(total, TPB), POOL_BWD_STATS_BLOCKS); if (grid < 1) grid = 1;
  hipLaunchKernelGGL(pool_bwd_bnstats_kernel<T>, dim3(grid), dim3(TPB), 0, as_stream(stream), y, ldy, dy, dx, lddx, gamma, beta, ctx->bn_slots, n, h, wd, c, rate, seed, ctx->bn_nslots());
  hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((2 * c + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, sums, 2 * c, ctx->bn_nslots());
  UNET_CHECK_LAUNCH(ctx, "maxpool bwd + bn stats"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t maxpool_bwd_sums_impl(unet_ctx* ctx, const T* pooled, const T* dy_pooled, const float* gamma, const float* beta, double* sums, int32_t n,
                                                                        int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) {
  if (!ctx || !pooled || !dy_pooled || !gamma || !beta || !sums || (c & 3) || c < 4 || TPB % (c / 4) || (h & 1) || (wd & 1) || rate < 0 || rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "maxpool bwd sums: bad args (c/4 must divide 256)");
  const long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  int grid = (int)std::min<long long>(cdiv64(total, TPB * 4), BN_STATS_BLOCKS); if (grid < 1) grid = 1;
  hipLaunchKernelGGL(pool_bwd_sums_kernel<T>, dim3(grid), dim3(TPB), 0, as_stream(stream), pooled, dy_pooled, gamma, beta, ctx->bn_slots, total, c, rate, seed, ctx->bn_nslots());
  hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((2 * c + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, sums, 2 * c, ctx->bn_nslots());
  UNET_CHECK_LAUNCH(ctx, "maxpool bwd sums"); return UNET_OK;
}

int32_t unet_bn_bwd_skip_term(unet_ctx* ctx, double* sums, const double* dec_sum_dyxhat, const float* dec_invstd, const float* dec_gamma, const float* gamma, int32_t c, double frac,
                              void* stream) {
  if (!ctx || !sums || !dec_sum_dyxhat || !dec_invstd || !dec_gamma || !gamma || c < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_skip_term: bad args");
  hipLaunchKernelGGL(bn_bwd_skip_term_kernel, dim3((c + 127) / 128), dim3(128), 0, as_stream(stream), sums, dec_sum_dyxhat, dec_invstd, dec_gamma, gamma, c, frac, 1e-3f);
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_skip_term"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t bn_maxpool_bwd_apply_impl(unet_ctx* ctx, const T* x, int32_t ldx, const float* bnp, const double* sums, double count, const T* g_skip,
                                                                            int32_t ldg, const T* dy_pooled, T* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c, float rate,
                                                                            uint64_t seed, void* stream, const float* skip_k1 = nullptr) {
  if (!ctx || !x || !bnp || !sums || !dy_pooled || !dx || (c & 3) || c < 4 || TPB % (c / 4) || (h & 1) || (wd & 1) || ((ldx | lddx) & 3) || (g_skip && (ldg & 3)) || count < 1 || rate < 0 || rate >= 1)
    UNET_FAIL(ctx, UNET_E_ARG, "bn + maxpool bwd apply: bad args (c/4 must divide 256)");
  const long long total = (long long)n * (h / 2) * (wd / 2) * (c / 4);
  if (total >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "pooling kernels index with 32 bits: %lld element quads is too many", total);
  hipLaunchKernelGGL(pool_bn_bwd_apply_kernel<T>, dim3(grid_for(total)), dim3(TPB), 0, as_stream(stream), x, ldx, bnp, sums, 1.0 / count, g_skip, ldg, dy_pooled, dx, lddx, n, h,
                     wd, c, rate, seed, skip_k1);
  UNET_CHECK_LAUNCH(ctx, "bn + maxpool bwd apply"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t head_fwd_impl(unet_ctx* ctx, const T* x, const float* w, const float* bias, float* p, const float* y_true,
                      double* loss_sums, int64_t pixels, int32_t cin, void* stream) {
  if (!x || !w || !bias || !p || (cin & 3) || !pow2(cin / 4) || cin / 4 > 64 || (y_true && !loss_sums)) UNET_FAIL(ctx, UNET_E_ARG, "head_fwd: bad args (cin/4 must be a power of two <= 64)");
  // deterministic mode: the four loss sums go through the context's slot copies (one workgroup per copy) and are folded in index order
  const bool det = ctx->opt_deterministic && y_true;
  const int nslots = det ? ctx->bn_nslots() : 0;
  double* target = det ? ctx->bn_slots : loss_sums;
  if (cin == 32) {            // the U-Net / U-Net++ heads
    const int lpp_blocks = det ? nslots : 2048;     // 8 loads in flight per lane: 2048 workgroups measured best (0.119 ms fp32 / 0.085 bf16; 8192: 0.141)
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(cdiv64(pixels, TPB * 2), lpp_blocks));
    hipLaunchKernelGGL((head_fwd_lpp_kernel<T, 8>), dim3(grid), dim3(TPB), 0, as_stream(stream), x, w, bias, p, y_true, target, (long long)pixels, nslots);
  } else {
    hipLaunchKernelGGL(head_fwd_kernel<T>, dim3((unsigned)std::max<long long>(1, std::min<long long>(cdiv64(pixels * (cin / 4) / 4, TPB), det ? nslots : HEAD_BLOCKS))), dim3(TPB), 0, as_stream(stream), x, w, bias, p, y_true, target, (long long)pixels, cin, nslots);
  }
  if (det) hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3(1), dim3(128), 0, as_stream(stream), ctx->bn_slots, loss_sums, 4, nslots);
  UNET_CHECK_LAUNCH(ctx, "head_fwd"); return UNET_OK;
}

extern "C++" int32_t k_loss_finalize(unet_ctx* ctx, const double* loss_sums, double count, float* loss_out, float* loss_out2, hipStream_t s) {
  if (!loss_sums || !loss_out || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "loss_finalize: bad args");
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, loss_sums, count, loss_out, loss_out2);
  UNET_CHECK_LAUNCH(ctx, "loss_finalize"); return UNET_OK;
}
int32_t unet_loss_finalize(unet_ctx* ctx, const double* loss_sums, double count, float* loss_out, void* stream) {
  return k_loss_finalize(ctx, loss_sums, count, loss_out, nullptr, as_stream(stream));
}

extern "C++" template <typename T> static int32_t head_bwd_impl(unet_ctx* ctx, const T* x, const float* w, const float* p, const float* y_true,
                      const double* loss_sums, double count, T* dx, float* dw, float* db, int64_t pixels, int32_t cin,
                      int32_t relu_mask, void* stream) {
  if (!x || !w || !p || !y_true || !loss_sums || !dx || !dw || !db || (cin & 3) || !pow2(cin / 4) || cin / 4 > 64 || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "head_bwd: bad args");
  const int nslots = ctx->opt_deterministic ? ctx->bn_nslots() : 0;
  hipLaunchKernelGGL(head_bwd_kernel<T>, dim3(std::min(grid_for(pixels * (cin / 4) / 4), 1024)), dim3(TPB), 0, as_stream(stream), x, w, p, y_true, loss_sums, 1.0 / count, dx, dw, db, (long long)pixels, cin, relu_mask,
                     ctx->bn_slots, nslots);
  if (nslots) {                                            // (dw and db are adjacent in the flat gradient buffer or not: two folds)
    hipLaunchKernelGGL(bn_slot_fold_kernel<float>, dim3(1), dim3(128), 0, as_stream(stream), ctx->bn_slots, dw, cin, nslots);
    hipLaunchKernelGGL(bn_slot_fold_kernel<float>, dim3(1), dim3(128), 0, as_stream(stream), ctx->bn_slots + cin, db, 1, nslots);
  }
  UNET_CHECK_LAUNCH(ctx, "head_bwd"); return UNET_OK;
}

extern "C++" int32_t k_slot_fold(unet_ctx* ctx, double* sums, int count, hipStream_t s, bool xs) {
  hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((count + 127) / 128), dim3(128), 0, s, ctx->bn_slots, sums, count, xs ? UNET_BN_SLOTS : ctx->bn_nslots(), xs ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "slot_fold"); return UNET_OK;
}
extern "C++" int32_t k_enc_tail_finish(unet_ctx* ctx, double* sums, const double* dec_sum_dyxhat, const float* dec_invstd, const float* dec_gamma, const float* gamma, float* dgamma,
                                       float* dbeta, int c, double frac, hipStream_t s) {
  if (!sums || !dec_sum_dyxhat || !dec_invstd || !dec_gamma || !gamma || !dgamma || !dbeta || c < 1) UNET_FAIL(ctx, UNET_E_ARG, "enc_tail_finish: bad args");
  hipLaunchKernelGGL(enc_tail_finish_kernel, dim3((c + 127) / 128), dim3(128), 0, s, ctx->bn_slots, sums, dec_sum_dyxhat, dec_invstd, dec_gamma, gamma, dgamma, dbeta, c, frac, 1e-3f, UNET_BN_SLOTS,
                     ctx->opt_deterministic ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "enc_tail_finish"); return UNET_OK;
}
extern "C++" int32_t k_bn_finalize_compose(unet_ctx* ctx, int training, const double* sums, double count, const float* gamma, const float* beta, float* mm, float* mv, float* bnp, int c,
                                           const float* enc_bnp, float* comp, hipStream_t s) {
  if (!gamma || !beta || !mm || !mv || !bnp || !enc_bnp || !comp || c < 2 || (c & 1) || (training && (!sums || count < 1))) UNET_FAIL(ctx, UNET_E_ARG, "bn_finalize_compose: bad args");
  if (training) hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((c + 127) / 128), dim3(128), 0, s, sums, count, gamma, beta, mm, mv, bnp, c, 0.99f, 1e-3f, enc_bnp, comp);
  else hipLaunchKernelGGL(bn_finalize_infer_kernel, dim3((c + 127) / 128), dim3(128), 0, s, gamma, beta, mm, mv, bnp, c, 1e-3f, enc_bnp, comp);
  UNET_CHECK_LAUNCH(ctx, "bn_finalize_compose"); return UNET_OK;
}
extern "C++" int32_t k_head_fold(unet_ctx* ctx, double* loss_sums, double* head_sums, hipStream_t s) {
  if (!loss_sums || !head_sums) UNET_FAIL(ctx, UNET_E_ARG, "head_fold: bad args");
  hipLaunchKernelGGL(head_fold_kernel, dim3(1), dim3(128), 0, s, ctx->bn_slots, loss_sums, head_sums, UNET_BN_SLOTS, ctx->opt_deterministic ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "head_fold"); return UNET_OK;
}

extern "C++" int32_t k_head_dzm(unet_ctx* ctx, const float* p, const float* t, const double* loss_sums, double count, const double* head_sums, const unsigned long long* bits,
                               void* dzm, float* dw, float* db, int n, int h, int wd, hipStream_t s) {
  if (!p || !t || !loss_sums || !head_sums || !bits || !dzm || !dw || !db || count < 1 || (wd & 7)) UNET_FAIL(ctx, UNET_E_ARG, "head_dzm: bad args");
  const long long pixels = (long long)n * h * wd;
  hipLaunchKernelGGL(head_dzm_kernel, dim3(std::min(grid_for(pixels), 4096)), dim3(TPB), 0, s, p, t, loss_sums, 1.0 / count, head_sums, bits, static_cast<uint2*>(dzm), dw, db, pixels, wd);
  UNET_CHECK_LAUNCH(ctx, "head_dzm"); return UNET_OK;
}

extern "C++" int32_t k_head_dy(unet_ctx* ctx, const float* p, const float* t, const double* loss_sums, double count, const double* head_sums, const float* w, const unsigned long long* bits,
                  const float* y, float* dy, float* dw, float* db, int n, int h, int wd, hipStream_t s) {
  if (!p || !t || !loss_sums || !head_sums || !w || (!bits && !y) || !dy || !dw || !db || count < 1 || (bits && (wd & 7))) UNET_FAIL(ctx, UNET_E_ARG, "head_dy: bad args");
  const long long pixels = (long long)n * h * wd;
  const int head_dy_blocks = 16384;          // (a thread: 8 iterations at 512 x 512 x 16)
  hipLaunchKernelGGL(head_dy_kernel, dim3(std::min(grid_for(pixels * 8 / 4), head_dy_blocks)), dim3(TPB), 0, s, p, t, loss_sums, 1.0 / count, head_sums, w, bits, y, dy, dw, db, pixels, wd);
  UNET_CHECK_LAUNCH(ctx, "head_dy"); return UNET_OK;
}

int32_t unet_adam_keras(unet_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t count, float lr_t, float b1,
                        float b2, float eps, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "adam: bad args");
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count / 4 + 1)), dim3(TPB), 0, as_stream(stream), p, g, m, v, (long long)count, lr_t, b1, b2, eps, grad_scale);
  UNET_CHECK_LAUNCH(ctx, "adam"); return UNET_OK;
}

int32_t unet_seg_metrics_sweep(unet_ctx* ctx, const float* p, const float* gt, const float* thresholds, int32_t nthr,
                               double* out, int64_t count, void* stream) {
  if (!p || !gt || !thresholds || !out || nthr < 1 || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "metrics_sweep: bad args");
  int gx = (int)std::min<int64_t>(cdiv64(count, TPB * 8), 1024); if (gx < 1) gx = 1;
  // deterministic mode: thresholds in rounds of what one slot copy holds; one workgroup column per copy, folded in index order
  if (ctx && ctx->opt_deterministic) {
    const int nslots = ctx->bn_nslots(), per = UNET_BN_SLOT_DOUBLES / 3 / THR_CHUNK * THR_CHUNK;
    for (int t0 = 0; t0 < nthr; t0 += per) {
      const int nt = std::min(per, nthr - t0);
      hipLaunchKernelGGL(metrics_sweep_kernel, dim3(gx, (nt + THR_CHUNK - 1) / THR_CHUNK), dim3(TPB), 0, as_stream(stream), p, gt, thresholds + t0, nt, ctx->bn_slots, (long long)count, nslots);
      hipLaunchKernelGGL(bn_slot_fold_kernel<double>, dim3((3 * nt + 127) / 128), dim3(128), 0, as_stream(stream), ctx->bn_slots, out + 3 * t0, 3 * nt, nslots);
    }
    UNET_CHECK_LAUNCH(ctx, "metrics_sweep"); return UNET_OK;
  }
  hipLaunchKernelGGL(metrics_sweep_kernel, dim3(gx, (nthr + THR_CHUNK - 1) / THR_CHUNK), dim3(TPB), 0, as_stream(stream), p, gt, thresholds, nthr, out, (long long)count, 0);
  UNET_CHECK_LAUNCH(ctx, "metrics_sweep"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t copy_slice_impl(unet_ctx* ctx, const T* src, int32_t lds, T* dst, int32_t ldd, int64_t pixels, int32_t c, void* stream) {
  if (!src || !dst || (c & 3) || lds < c || ldd < c || ((lds | ldd) & 3)) UNET_FAIL(ctx, UNET_E_ARG, "copy_slice: bad args");
  hipLaunchKernelGGL(copy_slice_kernel<T>, dim3(grid_for(pixels * (c / 4))), dim3(TPB), 0, as_stream(stream), src, lds, dst, ldd, (long long)pixels, c);
  UNET_CHECK_LAUNCH(ctx, "copy_slice"); return UNET_OK;
}

extern "C++" template <typename T> static int32_t accum_slices_impl(unet_ctx* ctx, const T* const* srcs, const int32_t* lds, int32_t nsrc, T* dst, int32_t ldd, int64_t pixels,
                          int32_t c, int32_t accumulate, void* stream) {
  if (!srcs || !lds || !dst || nsrc < 1 || nsrc > 4 || (c & 3) || ldd < c || (ldd & 3)) UNET_FAIL(ctx, UNET_E_ARG, "accum_slices: bad args (1..4 sources)");
  SliceList<T> sl; sl.n = nsrc;
  for (int k = 0; k < 4; ++k) { sl.p[k] = k < nsrc ? srcs[k] : nullptr; sl.ld[k] = k < nsrc ? lds[k] : 0; if (k < nsrc && (!srcs[k] || lds[k] < c || (lds[k] & 3))) UNET_FAIL(ctx, UNET_E_ARG, "accum_slices: bad source %d", k); }
  hipLaunchKernelGGL(accum_slices_kernel<T>, dim3(grid_for(pixels * (c / 4))), dim3(TPB), 0, as_stream(stream), sl, dst, ldd, (long long)pixels, c, accumulate);
  UNET_CHECK_LAUNCH(ctx, "accum_slices"); return UNET_OK;
}

int32_t unet_zero(unet_ctx* ctx, void* ptr, size_t bytes, void* stream) {
  if (!ptr) UNET_FAIL(ctx, UNET_E_ARG, "zero: null");
  // a plain kernel instead of hipMemsetAsync: the runtime's fill path costs a ~22 us bubble in the stream at every call
  // (kernel trace), and the training step zeroes its reduction scratch several times
  if ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0 && (bytes & 3) == 0 && bytes > 0) {
    const long long n4 = (long long)(bytes / 16);
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(std::max<long long>(n4, 1))), dim3(TPB), 0, as_stream(stream), static_cast<float4*>(ptr), n4, (int)((bytes & 15) / 4));
    UNET_CHECK_LAUNCH(ctx, "zero");
    return UNET_OK;
  }
  UNET_HIP(ctx, hipMemsetAsync(ptr, 0, bytes, as_stream(stream)));
  return UNET_OK;
}


// C entry points of the storage-templated ops: fp32 and bf16 (activation pointers only; parameters, sums and the head's
// probabilities / targets stay fp32 / fp64)
int32_t unet_bn_stats(unet_ctx* ctx, const float* x, int32_t ldx, double* sums, int64_t pixels, int32_t c, void* stream) { return bn_stats_impl(ctx, x, ldx, sums, pixels, c, stream); }
int32_t unet_bn_stats_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, double* sums, int64_t pixels, int32_t c, void* stream) { return bn_stats_impl(ctx, x, ldx, sums, pixels, c, stream); }
int32_t unet_bn_stats_concat(unet_ctx* ctx, const float* x_up, int32_t ldx, const double* src_sums, double src_count, const float* src_gamma, const float* src_beta, double* sums,
                             int64_t pixels, int32_t c_up, int32_t c_skip, void* stream) {
  return bn_stats_concat_impl(ctx, x_up, ldx, src_sums, src_count, src_gamma, src_beta, sums, pixels, c_up, c_skip, stream);
}
int32_t unet_bn_stats_concat_bf16(unet_ctx* ctx, const unet_bf16* x_up, int32_t ldx, const double* src_sums, double src_count, const float* src_gamma, const float* src_beta,
                                  double* sums, int64_t pixels, int32_t c_up, int32_t c_skip, void* stream) {
  return bn_stats_concat_impl(ctx, x_up, ldx, src_sums, src_count, src_gamma, src_beta, sums, pixels, c_up, c_skip, stream);
}
int32_t unet_bn_apply(unet_ctx* ctx, const float* x, int32_t ldx, const float* bnp, float* y, int32_t ldy, int64_t pixels, int32_t c, void* stream) { return bn_apply_impl(ctx, x, ldx, bnp, y, ldy, pixels, c, stream); }
int32_t unet_bn_apply_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, const float* bnp, unet_bf16* y, int32_t ldy, int64_t pixels, int32_t c, void* stream) { return bn_apply_impl(ctx, x, ldx, bnp, y, ldy, pixels, c, stream); }
int32_t unet_bn_bwd_stats(unet_ctx* ctx, const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* bnp, double* sums, int64_t pixels, int32_t c, void* stream) { return bn_bwd_stats_impl(ctx, dy, lddy, x, ldx, bnp, sums, pixels, c, stream); }
int32_t unet_bn_bwd_stats_bf16(unet_ctx* ctx, const unet_bf16* dy, int32_t lddy, const unet_bf16* x, int32_t ldx, const float* bnp, double* sums, int64_t pixels, int32_t c, void* stream) { return bn_bwd_stats_impl(ctx, dy, lddy, x, ldx, bnp, sums, pixels, c, stream); }
int32_t unet_bn_bwd_apply(unet_ctx* ctx, const float* dy, int32_t lddy, const float* x, int32_t ldx, const float* bnp, const double* sums, double count, int32_t mask_mode, float mask_rate, uint64_t mask_seed, float* dx, int32_t lddx, int64_t pixels, int32_t c, void* stream) { return bn_bwd_apply_impl(ctx, dy, lddy, x, ldx, bnp, sums, count, mask_mode, mask_rate, mask_seed, dx, lddx, pixels, c, stream); }
int32_t unet_bn_bwd_apply_bf16(unet_ctx* ctx, const unet_bf16* dy, int32_t lddy, const unet_bf16* x, int32_t ldx, const float* bnp, const double* sums, double count, int32_t mask_mode, float mask_rate, uint64_t mask_seed, unet_bf16* dx, int32_t lddx, int64_t pixels, int32_t c, void* stream) { return bn_bwd_apply_impl(ctx, dy, lddy, x, ldx, bnp, sums, count, mask_mode, mask_rate, mask_seed, dx, lddx, pixels, c, stream); }
int32_t unet_maxpool2x2_dropout_fwd(unet_ctx* ctx, const float* x, int32_t ldx, float* y, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_fwd_impl(ctx, x, ldx, y, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, unet_bf16* y, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_fwd_impl(ctx, x, ldx, y, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_bwd(unet_ctx* ctx, const float* x, int32_t ldx, const float* dy, float* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, int32_t accumulate, void* stream) { return maxpool_bwd_impl(ctx, x, ldx, dy, dx, lddx, n, h, wd, c, rate, seed, accumulate, stream); }
int32_t unet_maxpool2x2_dropout_bwd_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, const unet_bf16* dy, unet_bf16* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, int32_t accumulate, void* stream) { return maxpool_bwd_impl(ctx, x, ldx, dy, dx, lddx, n, h, wd, c, rate, seed, accumulate, stream); }
int32_t unet_bn_apply_maxpool_dropout_fwd(unet_ctx* ctx, const float* x, int32_t ldx, const float* bnp, float* y, int32_t ldy, float* pooled, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return bn_apply_maxpool_impl(ctx, x, ldx, bnp, y, ldy, pooled, n, h, wd, c, rate, seed, stream); }
int32_t unet_bn_apply_maxpool_dropout_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, const float* bnp, unet_bf16* y, int32_t ldy, unet_bf16* pooled, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return bn_apply_maxpool_impl(ctx, x, ldx, bnp, y, ldy, pooled, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_bwd_sums(unet_ctx* ctx, const float* pooled, const float* dy_pooled, const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_bwd_sums_impl(ctx, pooled, dy_pooled, gamma, beta, sums, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_bwd_sums_bf16(unet_ctx* ctx, const unet_bf16* pooled, const unet_bf16* dy_pooled, const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_bwd_sums_impl(ctx, pooled, dy_pooled, gamma, beta, sums, n, h, wd, c, rate, seed, stream); }
extern "C++" int32_t k_bn_maxpool_bwd_apply_k1(unet_ctx* ctx, const float* x, int ldx, const float* bnp, const double* sums, double count, const float* g_skip, int ldg, const float* skip_k1,
                                               const float* dy_pooled, float* dx, int lddx, int n, int h, int wd, int c, float rate, uint64_t seed, hipStream_t s) {
  if (skip_k1 && !g_skip) UNET_FAIL(ctx, UNET_E_ARG, "bn + maxpool bwd apply: skip_k1 without a skip gradient");
  return bn_maxpool_bwd_apply_impl(ctx, x, ldx, bnp, sums, count, g_skip, ldg, dy_pooled, dx, lddx, n, h, wd, c, rate, seed, (void*)s, skip_k1);
}
int32_t unet_bn_maxpool_bwd_apply(unet_ctx* ctx, const float* x, int32_t ldx, const float* bnp, const double* sums, double count, const float* g_skip, int32_t ldg, const float* dy_pooled, float* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return bn_maxpool_bwd_apply_impl(ctx, x, ldx, bnp, sums, count, g_skip, ldg, dy_pooled, dx, lddx, n, h, wd, c, rate, seed, stream); }
int32_t unet_bn_maxpool_bwd_apply_bf16(unet_ctx* ctx, const unet_bf16* x, int32_t ldx, const float* bnp, const double* sums, double count, const unet_bf16* g_skip, int32_t ldg, const unet_bf16* dy_pooled, unet_bf16* dx, int32_t lddx, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return bn_maxpool_bwd_apply_impl(ctx, x, ldx, bnp, sums, count, g_skip, ldg, dy_pooled, dx, lddx, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_bwd_bnstats(unet_ctx* ctx, const float* y, int32_t ldy, const float* dy, float* dx, int32_t lddx, const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_bwd_bnstats_impl(ctx, y, ldy, dy, dx, lddx, gamma, beta, sums, n, h, wd, c, rate, seed, stream); }
int32_t unet_maxpool2x2_dropout_bwd_bnstats_bf16(unet_ctx* ctx, const unet_bf16* y, int32_t ldy, const unet_bf16* dy, unet_bf16* dx, int32_t lddx, const float* gamma, const float* beta, double* sums, int32_t n, int32_t h, int32_t wd, int32_t c, float rate, uint64_t seed, void* stream) { return maxpool_bwd_bnstats_impl(ctx, y, ldy, dy, dx, lddx, gamma, beta, sums, n, h, wd, c, rate, seed, stream); }
int32_t unet_head_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* p, const float* y_true, double* loss_sums, int64_t pixels, int32_t cin, void* stream) { return head_fwd_impl(ctx, x, w, bias, p, y_true, loss_sums, pixels, cin, stream); }
int32_t unet_head_fwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, float* p, const float* y_true, double* loss_sums, int64_t pixels, int32_t cin, void* stream) { return head_fwd_impl(ctx, x, w, bias, p, y_true, loss_sums, pixels, cin, stream); }
int32_t unet_head_bwd(unet_ctx* ctx, const float* x, const float* w, const float* p, const float* y_true, const double* loss_sums, double count, float* dx, float* dw, float* db, int64_t pixels, int32_t cin, int32_t relu_mask, void* stream) { return head_bwd_impl(ctx, x, w, p, y_true, loss_sums, count, dx, dw, db, pixels, cin, relu_mask, stream); }
int32_t unet_head_bwd_bf16(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* p, const float* y_true, const double* loss_sums, double count, unet_bf16* dx, float* dw, float* db, int64_t pixels, int32_t cin, int32_t relu_mask, void* stream) { return head_bwd_impl(ctx, x, w, p, y_true, loss_sums, count, dx, dw, db, pixels, cin, relu_mask, stream); }
int32_t unet_copy_slice(unet_ctx* ctx, const float* src, int32_t lds, float* dst, int32_t ldd, int64_t pixels, int32_t c, void* stream) { return copy_slice_impl(ctx, src, lds, dst, ldd, pixels, c, stream); }
int32_t unet_copy_slice_bf16(unet_ctx* ctx, const unet_bf16* src, int32_t lds, unet_bf16* dst, int32_t ldd, int64_t pixels, int32_t c, void* stream) { return copy_slice_impl(ctx, src, lds, dst, ldd, pixels, c, stream); }
int32_t unet_accum_slices(unet_ctx* ctx, const float* const* srcs, const int32_t* lds, int32_t nsrc, float* dst, int32_t ldd, int64_t pixels, int32_t c, int32_t accumulate, void* stream) { return accum_slices_impl(ctx, srcs, lds, nsrc, dst, ldd, pixels, c, accumulate, stream); }
int32_t unet_accum_slices_bf16(unet_ctx* ctx, const unet_bf16* const* srcs, const int32_t* lds, int32_t nsrc, unet_bf16* dst, int32_t ldd, int64_t pixels, int32_t c, int32_t accumulate, void* stream) { return accum_slices_impl(ctx, srcs, lds, nsrc, dst, ldd, pixels, c, accumulate, stream); }

namespace {
// dst[i] = src[idx[i]] for whole samples of `sf4` float4s: the mini-batch of a shuffled epoch, taken from a dataset that lives in HBM
__global__ __launch_bounds__(TPB) void gather_samples_kernel(const float4* __restrict__ src, const long long* __restrict__ idx, float4* __restrict__ dst, long long n, long long sf4) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n * sf4; i += (long long)gridDim.x * TPB) {
    const long long s = i / sf4, e = i - s * sf4;
    dst[i] = src[idx[s] * sf4 + e];
  }
}
}  // namespace
int32_t unet_gather_samples(unet_ctx* ctx, const float* src, const int64_t* idx, float* dst, int64_t n, int64_t sample_floats, void* stream) {
  if (!src || !idx || !dst || n < 1 || sample_floats < 4 || (sample_floats & 3)) UNET_FAIL(ctx, UNET_E_ARG, "gather_samples: bad args (sample_floats must be a multiple of 4)");
  hipLaunchKernelGGL(gather_samples_kernel, dim3(grid_for(n * (sample_floats / 4))), dim3(TPB), 0, as_stream(stream), reinterpret_cast<const float4*>(src),
                     reinterpret_cast<const long long*>(idx), reinterpret_cast<float4*>(dst), (long long)n, (long long)(sample_floats / 4));
  UNET_CHECK_LAUNCH(ctx, "gather_samples"); return UNET_OK;
}

namespace {
__global__ __launch_bounds__(TPB) void cast_f32_bf16_kernel(const float* __restrict__ src, unet_bf16* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) st4(dst + i * 4, ld4(src + i * 4));
}
__global__ __launch_bounds__(TPB) void cast_bf16_f32_kernel(const unet_bf16* __restrict__ src, float* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) st4(dst + i * 4, ld4(src + i * 4));
}
}  // namespace
int32_t unet_cast_f32_to_bf16(unet_ctx* ctx, const float* src, unet_bf16* dst, int64_t count, void* stream) {
  if (!src || !dst || count < 4 || (count & 3)) UNET_FAIL(ctx, UNET_E_ARG, "cast: count must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(count / 4)), dim3(TPB), 0, as_stream(stream), src, dst, (long long)(count / 4));
  UNET_CHECK_LAUNCH(ctx, "cast_f32_to_bf16"); return UNET_OK;
}
int32_t unet_cast_bf16_to_f32(unet_ctx* ctx, const unet_bf16* src, float* dst, int64_t count, void* stream) {
  if (!src || !dst || count < 4 || (count & 3)) UNET_FAIL(ctx, UNET_E_ARG, "cast: count must be a positive multiple of 4");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(count / 4)), dim3(TPB), 0, as_stream(stream), src, dst, (long long)(count / 4));
  UNET_CHECK_LAUNCH(ctx, "cast_bf16_to_f32"); return UNET_OK;
}

}  // extern "C"
