// Direct (non-MFMA) HIP convolution kernels: the Cin=1 first layer (HBM-bound, final form),
// plus straightforward one-output-per-thread kernels for 3x3 conv / 2x2 transposed conv and
// their gradients.  The direct kernels serve (1) shapes the MFMA kernels do not cover and
// (2) as an on-device cross-check of the MFMA path (UNET_ALGO_NAIVE).  They are HIP kernels,
// not a CPU fallback.
#include "common.h"

namespace {
constexpr int TPB = 256;

// y[pix][co] = act(b[co] + sum_{tap,ci} x[pix+tap][ci] * w[tap][ci][co]); lanes run along co.
__global__ __launch_bounds__(TPB) void conv3x3_naive_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ mask, int mask_mode,
                                                            float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int act,
                                                            float rate, unsigned long long seed) {
  const long long total = (long long)N * H * W * Cout;
  for (long long idx = (long long)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (long long)gridDim.x * TPB) {
    int co = (int)(idx % Cout); long long pix = idx / Cout;
    int j = (int)(pix % W); long long t = pix / W; int i = (int)(t % H); long long n = t / H;
    float acc = bias ? bias[co] : 0.0f;
    for (int a = 0; a < 3; ++a) {
      int ii = i + a - 1; if (ii < 0 || ii >= H) continue;
      for (int b = 0; b < 3; ++b) {
        int jj = j + b - 1; if (jj < 0 || jj >= W) continue;
        const float* xp = x + ((n * H + ii) * W + jj) * (long long)Cin;
        const float* wp = w + (long long)((a * 3 + b) * Cin) * Cout + co;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(xp[ci], wp[(long long)ci * Cout], acc);
      }
    }
    acc = apply_act(acc, act);
    float ks = 1.0f;
    if (rate > 0.0f && (mask_mode == MASK_NONE || mask_mode == MASK_ELU_DROP)) {     // same Philox stream as the float4 kernels
      const float4 k4 = keep_scale(idx >> 2, rate, seed); const int e = (int)(idx & 3);
      ks = e == 0 ? k4.x : e == 1 ? k4.y : e == 2 ? k4.z : k4.w;
    }
    if (mask_mode == MASK_NONE) { if (rate > 0.0f) acc *= ks; }
    else acc *= mask_factor(mask[idx], mask_mode, ks, rate);
    y[idx] = acc;
  }
}

// First layer, Cin = 1 (c1a, T1:859): HBM-bound (AI ~2 F/B).  8 lanes per pixel, each lane owns
// 4 output channels with its 9x4 weights in registers; a wave writes 8 pixels = 1 KiB contiguous.
template <typename T>
__global__ __launch_bounds__(TPB) void conv3x3_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, T* __restrict__ y,
                                                         int N, int H, int W, int Cout, int act, float rate, unsigned long long seed) {
  const int lpp = Cout >> 2;
  const int sub = threadIdx.x % lpp;
  float4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wr[t] = *reinterpret_cast<const float4*>(w + t * Cout + sub * 4);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + sub * 4) : make_float4(0, 0, 0, 0);
  const long long pixels = (long long)N * H * W;
  const long long g0 = ((long long)blockIdx.x * TPB + threadIdx.x) / lpp, gs = ((long long)gridDim.x * TPB) / lpp;
  for (long long p = g0; p < pixels; p += gs) {
    const unsigned pu = (unsigned)p, tq = pu / (unsigned)W;             // 32-bit index math (the launcher guarantees pixels < 2^31):
    const int j = (int)(pu - tq * (unsigned)W), i = (int)(tq % (unsigned)H);   // the 64-bit div/mod pair cost more than the 9 FMAs
    float4 acc = b4;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int ii = i + a - 1;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        int jj = j + b - 1;
        float v = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? x[p + (a - 1) * W + (b - 1)] : 0.0f;
        const float4 k = wr[a * 3 + b];
        acc.x = fmaf(v, k.x, acc.x); acc.y = fmaf(v, k.y, acc.y); acc.z = fmaf(v, k.z, acc.z); acc.w = fmaf(v, k.w, acc.w);
      }
    }
    acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
    if (rate > 0.0f) { const float4 ks = keep_scale(p * lpp + sub, rate, seed); acc.x *= ks.x; acc.y *= ks.y; acc.z *= ks.z; acc.w *= ks.w; }
    st4(y + p * Cout + sub * 4, acc);
  }
}

// The same layer with FOUR consecutive pixels of a row per thread: 3 x 6 image values instead of 4 x 9 loads, four independent accumulator chains and
// four stores in flight per thread (the one-pixel form is bound by its load -> FMA -> store latency chain: 3.0 TB/s on a layer whose only real traffic
// is the 537 MB it writes).  W % 4 == 0.
template <typename T>
__global__ __launch_bounds__(TPB) void conv3x3_c1x4_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, T* __restrict__ y,
                                                           int N, int H, int W, int Cout, int act, float rate, unsigned long long seed) {
  const int lpp = Cout >> 2;
  const int sub = threadIdx.x % lpp;
  float4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wr[t] = *reinterpret_cast<const float4*>(w + t * Cout + sub * 4);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + sub * 4) : make_float4(0, 0, 0, 0);
  const unsigned W4 = (unsigned)W >> 2;
  const long long groups = (long long)N * H * W4;
  const long long g0 = ((long long)blockIdx.x * TPB + threadIdx.x) / lpp, gs = ((long long)gridDim.x * TPB) / lpp;
  for (long long gi = g0; gi < groups; gi += gs) {
    const unsigned gu = (unsigned)gi, row = gu / W4;                     // (pixels < 2^31: the launcher checks)
    const int j0 = (int)(gu - row * W4) * 4, i = (int)(row % (unsigned)H);
    const long long p0 = (long long)row * W + j0;
    float v[3][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ii = i + a - 1;
      const bool rok = ii >= 0 && ii < H;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int jj = j0 + b - 1;
        v[a][b] = (rok && jj >= 0 && jj < W) ? x[p0 + (a - 1) * W + (b - 1)] : 0.0f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 acc = b4;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const float xv = v[a][k + b]; const float4 kk = wr[a * 3 + b];
          acc.x = fmaf(xv, kk.x, acc.x); acc.y = fmaf(xv, kk.y, acc.y); acc.z = fmaf(xv, kk.z, acc.z); acc.w = fmaf(xv, kk.w, acc.w);
        }
      acc.x = apply_act(acc.x, act); acc.y = apply_act(acc.y, act); acc.z = apply_act(acc.z, act); acc.w = apply_act(acc.w, act);
      const long long p = p0 + k;
      if (rate > 0.0f) { const float4 ks = keep_scale(p * lpp + sub, rate, seed); acc.x *= ks.x; acc.y *= ks.y; acc.z *= ks.z; acc.w *= ks.w; }
      st4(y + p * Cout + sub * 4, acc);
    }
  }
}

// The same layer also leaving the ReLU sign bits of what it stores (MASK_RELU_BITS, common.h: the mask of the next conv's data gradient in 1/32 of the bytes).
// Cout = 32, W % 8 == 0, ReLU, no dropout.  A wave owns 8 pixels of a row x 4 rows; lane (g, sub) = (lane / 8, lane % 8) computes channel quad `sub` of pixel
// column g, one row per step (the 6 x 3 input window of the four rows stays in registers) -- so the wave's ballot of "component q > 0" in step t IS word q of
// the (8 pixels x 32 channels) cell of row t (bit = (pixel % 8) * 8 + quad), and each step's store is one contiguous KB.
__global__ __launch_bounds__(TPB) void conv3x3_c1_bits_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                              unsigned long long* __restrict__ signs, int N, int H, int W) {
  const int lane = threadIdx.x & 63, sub = lane & 7, gl = lane >> 3;
  float4 wr[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wr[t] = *reinterpret_cast<const float4*>(w + t * 32 + sub * 4);
  const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + sub * 4) : make_float4(0, 0, 0, 0);
  const unsigned W8 = (unsigned)W >> 3, H4 = ((unsigned)H + 3) >> 2;
  const long long items = (long long)N * H4 * W8;
  const long long s0 = ((long long)blockIdx.x * TPB + threadIdx.x) >> 6, ss = ((long long)gridDim.x * TPB) >> 6;
  for (long long it = s0; it < items; it += ss) {
    const unsigned iu = (unsigned)it, rq = iu / W8;                      // (n, row quad)
    const int xc = (int)(iu - rq * W8), n = (int)(rq / H4), i0 = (int)(rq - (unsigned)n * H4) * 4;
    const int j = xc * 8 + gl;
    const float* img = x + (long long)n * H * W;
    float v[6][3];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int ii = i0 + a - 1;
      const bool rok = ii >= 0 && ii < H;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int jj = j + b - 1;
        v[a][b] = (rok && jj >= 0 && jj < W) ? img[(long long)ii * W + jj] : 0.0f;
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (i0 + t >= H) break;                                            // (wave-uniform)
      float4 acc = b4;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const float xv = v[t + a][b]; const float4 kk = wr[a * 3 + b];
          acc.x = fmaf(xv, kk.x, acc.x); acc.y = fmaf(xv, kk.y, acc.y); acc.z = fmaf(xv, kk.z, acc.z); acc.w = fmaf(xv, kk.w, acc.w);
        }
      acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
      const long long row = (long long)n * H + i0 + t;
      st4(y + (row * W + j) * 32 + sub * 4, acc);
      const unsigned long long q0 = __builtin_amdgcn_ballot_w64(acc.x > 0.f), q1 = __builtin_amdgcn_ballot_w64(acc.y > 0.f);
      const unsigned long long q2 = __builtin_amdgcn_ballot_w64(acc.z > 0.f), q3 = __builtin_amdgcn_ballot_w64(acc.w > 0.f);
      if (lane < 4) signs[(row * W8 + xc) * 4 + lane] = lane == 0 ? q0 : lane == 1 ? q1 : lane == 2 ? q2 : q3;
    }
  }
}

// wt[t'][co][ci] = w[8-t'][ci][co]: weights of the data-gradient convolution.
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cin, int Cout) {
  const int total = 9 * Cin * Cout;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    int ci = i % Cin; int r = i / Cin; int co = r % Cout; int t = r / Cout;
    wt[i] = w[((8 - t) * Cin + ci) * Cout + co];
  }
}

// dw[tap][ci][co] = sum_pix x[pix+tap][ci]*dy[pix][co]; grid.y splits the pixel range, float atomics.
// The bias gradient rides along as a virtual row (r == 9*Cin): db[co] = sum_pix dy[pix][co].
__global__ __launch_bounds__(TPB) void conv3x3_naive_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* dw, float* db, int N, int H, int W, int Cin, int Cout) {
  const int rows = 9 * Cin + 1;
  const long long total = (long long)rows * Cout;
  const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int co = (int)(idx % Cout); const int r = (int)(idx / Cout);
  const long long pixels = (long long)N * H * W;
  const long long chunk = (pixels + gridDim.y - 1) / gridDim.y;
  const long long p0 = chunk * blockIdx.y, p1 = (p0 + chunk < pixels) ? p0 + chunk : pixels;
  float acc = 0.0f;
  if (r == 9 * Cin) {
    for (long long p = p0; p < p1; ++p) acc += dy[p * Cout + co];
    atomicAdd(db + co, acc);
    return;
  }
  const int ci = r % Cin, tap = r / Cin, da = tap / 3 - 1, dbb = tap % 3 - 1;
  for (long long p = p0; p < p1; ++p) {
    int j = (int)(p % W); long long t = p / W; int i = (int)(t % H);
    int ii = i + da, jj = j + dbb;
    if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
    acc = fmaf(x[(p + da * W + dbb) * Cin + ci], dy[p * Cout + co], acc);
  }
  atomicAdd(dw + (long long)r * Cout + co, acc);
}

// Weight gradient of the Cin = 1 first layer (c1a): HBM-bound (reads dy once: 128 B/pixel).
// 8 lanes per pixel (one float4 of dy each), 9 x-neighbours broadcast from L1; per-thread
// accumulators [9 taps + bias] x float4, xor-shuffle + LDS tree per block, then block partials
// [grid][10][Cout] that reduce_c1_kernel sums in a fixed order (deterministic).
constexpr int C1_BLOCKS = 1024;
template <typename T, bool X4>
__global__ __launch_bounds__(TPB) void conv3x3_c1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                               float* __restrict__ part, int N, int H, int W, int Cout) {
  const int lpp = Cout >> 2;
  const int sub = threadIdx.x % lpp;
  float4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long pixels = (long long)N * H * W;
  const long long g0 = ((long long)blockIdx.x * TPB + threadIdx.x) / lpp, gs = ((long long)gridDim.x * TPB) / lpp;
  if (X4) {
    // four consecutive pixels of a row per thread: four dy loads in flight, 3 x 6 image values for the 4 x 9 products
    const unsigned W4 = (unsigned)W >> 2;
    const long long groups = (long long)N * H * W4;
    for (long long gi = g0; gi < groups; gi += gs) {
      const unsigned gu = (unsigned)gi, row = gu / W4;
      const int j0 = (int)(gu - row * W4) * 4, i = (int)(row % (unsigned)H);
      const long long p0 = (long long)row * W + j0;
      float4 g[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = ld4(dy + (p0 + k) * Cout + sub * 4);
      float v[3][6];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int ii = i + a - 1;
        const bool rok = ii >= 0 && ii < H;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
          const int jj = j0 + b - 1;
          v[a][b] = (rok && jj >= 0 && jj < W) ? x[p0 + (a - 1) * W + (b - 1)] : 0.0f;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            const float xv = v[a][k + b];
            float4& r = acc[a * 3 + b];
            r.x = fmaf(xv, g[k].x, r.x); r.y = fmaf(xv, g[k].y, r.y); r.z = fmaf(xv, g[k].z, r.z); r.w = fmaf(xv, g[k].w, r.w);
          }
        acc[9].x += g[k].x; acc[9].y += g[k].y; acc[9].z += g[k].z; acc[9].w += g[k].w;
      }
    }
  } else
  for (long long p = g0; p < pixels; p += gs) {
    const unsigned pu = (unsigned)p, tq = pu / (unsigned)W;             // 32-bit index math (pixels < 2^31, checked by the launcher)
    const int j = (int)(pu - tq * (unsigned)W), i = (int)(tq % (unsigned)H);
    const float4 g = ld4(dy + p * Cout + sub * 4);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int ii = i + a - 1;
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int jj = j + b - 1;
        const float v = (ii >= 0 && ii < H && jj >= 0 && jj < W) ? x[p + (a - 1) * W + (b - 1)] : 0.0f;
        float4& r = acc[a * 3 + b];
        r.x = fmaf(v, g.x, r.x); r.y = fmaf(v, g.y, r.y); r.z = fmaf(v, g.z, r.z); r.w = fmaf(v, g.w, r.w);
      }
    }
    acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
  }
  __shared__ float4 red[TPB / 64][10][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    float4 r = acc[t];
    for (int o = lpp; o < 64; o <<= 1) {
      r.x += __shfl_xor(r.x, o, 64); r.y += __shfl_xor(r.y, o, 64); r.z += __shfl_xor(r.z, o, 64); r.w += __shfl_xor(r.w, o, 64);
    }
    if (lane < lpp) red[wv][t][lane] = r;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 10 * lpp; e += TPB) {
    const int t = e / lpp, q = e % lpp;
    float4 r = red[0][t][q];
    for (int k = 1; k < TPB / 64; ++k) { const float4 u = red[k][t][q]; r.x += u.x; r.y += u.y; r.z += u.z; r.w += u.w; }
    *reinterpret_cast<float4*>(part + ((long long)blockIdx.x * 10 + t) * Cout + q * 4) = r;
  }
}

// two-level fixed-order reduction of the block partials: level 1 sums groups of 32 blocks
__global__ void reduce_c1_kernel(const float* __restrict__ part, float* __restrict__ out, float* __restrict__ out_b, int nblocks, int per_group,
                                 int Cout, int final_level) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 10 * Cout) return;
  const int k0 = blockIdx.y * per_group; int k1 = k0 + per_group; if (k1 > nblocks) k1 = nblocks;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = k0;
  for (; k + 3 < k1; k += 4) {
    s0 += part[(long long)k * 10 * Cout + e]; s1 += part[(long long)(k + 1) * 10 * Cout + e];
    s2 += part[(long long)(k + 2) * 10 * Cout + e]; s3 += part[(long long)(k + 3) * 10 * Cout + e];
  }
  for (; k < k1; ++k) s0 += part[(long long)k * 10 * Cout + e];
  const float s = (s0 + s1) + (s2 + s3);
  if (!final_level) out[(long long)blockIdx.y * 10 * Cout + e] = s;
  else if (e < 9 * Cout) out[e] = s; else out_b[e - 9 * Cout] = s;
}

// ---- transposed conv 2x2 stride 2 (kernel [2][2][Cout][Cin]) --------------------------
__global__ __launch_bounds__(TPB) void convT_naive_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y, int ldy,
                                                              int N, int h, int wd, int Cin, int Cout) {
  const int Ho = 2 * h, Wo = 2 * wd;
  const long long total = (long long)N * Ho * Wo * Cout;
  for (long long idx = (long long)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (long long)gridDim.x * TPB) {
    int o = (int)(idx % Cout); long long pix = idx / Cout;
    int xo = (int)(pix % Wo); long long t = pix / Wo; int yo = (int)(t % Ho); long long n = t / Ho;
    int i = yo >> 1, a = yo & 1, j = xo >> 1, b = xo & 1;
    const float* xp = x + ((n * h + i) * wd + j) * (long long)Cin;
    const float* wp = w + ((long long)((a * 2 + b) * Cout + o)) * Cin;
    float acc = bias ? bias[o] : 0.0f;
    for (int c = 0; c < Cin; ++c) acc = fmaf(xp[c], wp[c], acc);
    y[pix * ldy + o] = acc;
  }
}

template <bool MASK>
__global__ __launch_bounds__(TPB) void convT_naive_dgrad_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w,
                                                                const float* __restrict__ mask, float* __restrict__ dx,
                                                                int N, int h, int wd, int Cin, int Cout) {
  const int Wo = 2 * wd, Ho = 2 * h;
  const long long total = (long long)N * h * wd * Cin;
  for (long long idx = (long long)blockIdx.x * TPB + threadIdx.x; idx < total; idx += (long long)gridDim.x * TPB) {
    int c = (int)(idx % Cin); long long pix = idx / Cin;
    int j = (int)(pix % wd); long long t = pix / wd; int i = (int)(t % h); long long n = t / h;
    float acc = 0.0f;
    for (int ab = 0; ab < 4; ++ab) {
      const float* gp = dy + ((n * Ho + 2 * i + (ab >> 1)) * Wo + 2 * j + (ab & 1)) * (long long)lddy;
      const float* wp = w + (long long)ab * Cout * Cin + c;
      for (int o = 0; o < Cout; ++o) acc = fmaf(gp[o], wp[(long long)o * Cin], acc);
    }
    if (MASK) acc = mask[idx] > 0.0f ? acc : 0.0f;
    dx[idx] = acc;
  }
}

// dw[ab][o][c] = sum_{n,i,j} dy[n,2i+a,2j+b,o]*x[n,i,j,c];  db[o] = sum dy (virtual column c == Cin, ab == 0 rows
// of every ab so each output pixel is counted once).
__global__ __launch_bounds__(TPB) void convT_naive_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int lddy,
                                                                float* dw, float* db, int N, int h, int wd, int Cin, int Cout) {
  const long long total = (long long)4 * Cout * (Cin + 1);
  const long long idx = (long long)blockIdx.x * TPB + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % (Cin + 1)); long long r = idx / (Cin + 1); const int o = (int)(r % Cout); const int ab = (int)(r / Cout);
  const int Wo = 2 * wd, Ho = 2 * h;
  const long long pixels = (long long)N * h * wd;
  const long long chunk = (pixels + gridDim.y - 1) / gridDim.y;
  const long long p0 = chunk * blockIdx.y, p1 = (p0 + chunk < pixels) ? p0 + chunk : pixels;
  float acc = 0.0f;
  for (long long p = p0; p < p1; ++p) {
    int j = (int)(p % wd); long long t = p / wd; int i = (int)(t % h); long long n = t / h;
    float g = dy[((n * Ho + 2 * i + (ab >> 1)) * Wo + 2 * j + (ab & 1)) * (long long)lddy + o];
    acc = (c == Cin) ? acc + g : fmaf(g, x[p * Cin + c], acc);
  }
  if (c == Cin) atomicAdd(db + o, acc);
  else atomicAdd(dw + ((long long)ab * Cout + o) * Cin + c, acc);
}

inline int grid_for(long long items, int cap = 4096) {
  long long b = cdiv64(items, TPB);
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}
}  // namespace

int32_t k_conv3x3_naive_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, const float* mask, int mask_mode, float* y,
                            int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, hipStream_t s) {
  long long total = (long long)n * h * wd * cout;
  if (!mask) mask_mode = MASK_NONE;
  hipLaunchKernelGGL(conv3x3_naive_kernel, dim3(grid_for(total, 1 << 20)), dim3(TPB), 0, s, x, w, bias, mask, mask_mode, y, n, h, wd, cin, cout, act,
                     rate, (unsigned long long)seed);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_naive_fwd"); return UNET_OK;
}

template <typename T>
static int32_t c1_fwd_impl(unet_ctx* ctx, const float* x, const float* w, const float* bias, T* y, int n, int h, int wd,
                         int cout, int act, float rate, uint64_t seed, hipStream_t s) {
  if ((cout & 3) || TPB % (cout / 4) || (long long)n * h * wd >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_c1: cout=%d / %d x %d x %d pixels unsupported", cout, n, h, wd);
  long long threads = (long long)n * h * wd * (cout / 4);
  if ((wd & 3) == 0) {                                   // four pixels of a row per thread
    hipLaunchKernelGGL(conv3x3_c1x4_kernel<T>, dim3(grid_for(threads / 4, 4096)), dim3(TPB), 0, s, x, w, bias, y, n, h, wd, cout, act, rate, (unsigned long long)seed);
    UNET_CHECK_LAUNCH(ctx, "conv3x3_c1x4_fwd"); return UNET_OK;
  }
  hipLaunchKernelGGL(conv3x3_c1_kernel<T>, dim3(grid_for(threads / 2 + 1, 2048)), dim3(TPB), 0, s, x, w, bias, y, n, h, wd, cout, act, rate,
                     (unsigned long long)seed);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_c1_fwd"); return UNET_OK;
}
int32_t k_conv3x3_c1_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, int act, float rate,
                         uint64_t seed, hipStream_t s) { return c1_fwd_impl(ctx, x, w, bias, y, n, h, wd, cout, act, rate, seed, s); }
bool c1_relu_bits_supported(int wd, int cout) { return cout == 32 && wd >= 8 && (wd & 7) == 0; }
// ... + the ReLU sign bits of y (layout of MASK_RELU_BITS, M = 32): ReLU only
int32_t k_conv3x3_c1_fwd_bits(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, unsigned long long* signs, int n, int h, int wd, int cout, hipStream_t s) {
  if (!c1_relu_bits_supported(wd, cout) || !signs || (long long)n * h * wd >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_c1 + sign bits: cout=%d W=%d unsupported", cout, wd);
  const long long items = (long long)n * ((h + 3) / 4) * (wd / 8);          // one wave per (8 pixels x 4 rows)
  hipLaunchKernelGGL(conv3x3_c1_bits_kernel, dim3(grid_for(items * 64, 4096)), dim3(TPB), 0, s, x, w, bias, y, signs, n, h, wd);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_c1_bits_fwd"); return UNET_OK;
}
int32_t k_conv3x3_c1_fwd_bf16(unet_ctx* ctx, const float* x, const float* w, const float* bias, unet_bf16* y, int n, int h, int wd, int cout, int act,
                              float rate, uint64_t seed, hipStream_t s) { return c1_fwd_impl(ctx, x, w, bias, y, n, h, wd, cout, act, rate, seed, s); }

int32_t k_flip_transpose_w3x3(unet_ctx* ctx, const float* w, float* wt, int cin, int cout, hipStream_t s) {
  hipLaunchKernelGGL(flip_transpose_kernel, dim3(grid_for(9LL * cin * cout, 2048)), dim3(TPB), 0, s, w, wt, cin, cout);
  UNET_CHECK_LAUNCH(ctx, "flip_transpose"); return UNET_OK;
}

int32_t k_conv3x3_naive_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, int n, int h, int wd,
                              int cin, int cout, hipStream_t s) {
  UNET_HIP(ctx, hipMemsetAsync(dw, 0, sizeof(float) * 9 * cin * cout, s));
  UNET_HIP(ctx, hipMemsetAsync(db, 0, sizeof(float) * cout, s));
  long long outs = (long long)(9 * cin + 1) * cout, pixels = (long long)n * h * wd;
  int gx = (int)cdiv64(outs, TPB);
  long long want = (1 << 18) / (outs < 1 ? 1 : outs) + 1;    // aim for ~256K threads in flight
  int gy = (int)std::min<long long>(std::max<long long>(want, 1), std::max<long long>(pixels / 64, 1));
  hipLaunchKernelGGL(conv3x3_naive_wgrad_kernel, dim3(gx, gy), dim3(TPB), 0, s, x, dy, dw, db, n, h, wd, cin, cout);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_naive_wgrad"); return UNET_OK;
}

int32_t k_convT_naive_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int ldy, int n, int h,
                          int wd, int cin, int cout, hipStream_t s) {
  long long total = (long long)n * 4 * h * wd * cout;
  hipLaunchKernelGGL(convT_naive_fwd_kernel, dim3(grid_for(total, 1 << 20)), dim3(TPB), 0, s, x, w, bias, y, ldy, n, h, wd, cin, cout);
  UNET_CHECK_LAUNCH(ctx, "convT_naive_fwd"); return UNET_OK;
}

int32_t k_convT_naive_dgrad(unet_ctx* ctx, const float* dy, int lddy, const float* w, const float* mask, float* dx, int n,
                            int h, int wd, int cin, int cout, hipStream_t s) {
  long long total = (long long)n * h * wd * cin;
  dim3 g(grid_for(total, 1 << 20)), b(TPB);
  if (mask) hipLaunchKernelGGL(convT_naive_dgrad_kernel<true>, g, b, 0, s, dy, lddy, w, mask, dx, n, h, wd, cin, cout);
  else hipLaunchKernelGGL(convT_naive_dgrad_kernel<false>, g, b, 0, s, dy, lddy, w, mask, dx, n, h, wd, cin, cout);
  UNET_CHECK_LAUNCH(ctx, "convT_naive_dgrad"); return UNET_OK;
}

int32_t k_convT_naive_wgrad(unet_ctx* ctx, const float* x, const float* dy, int lddy, float* dw, float* db, int n, int h,
                            int wd, int cin, int cout, hipStream_t s) {
  UNET_HIP(ctx, hipMemsetAsync(dw, 0, sizeof(float) * 4 * cin * cout, s));
  UNET_HIP(ctx, hipMemsetAsync(db, 0, sizeof(float) * cout, s));
  long long outs = 4LL * cout * (cin + 1), pixels = (long long)n * h * wd;
  int gx = (int)cdiv64(outs, TPB);
  long long want = (1 << 18) / outs + 1;
  int gy = (int)std::min<long long>(std::max<long long>(want, 1), std::max<long long>(pixels / 64, 1));
  hipLaunchKernelGGL(convT_naive_wgrad_kernel, dim3(gx, gy), dim3(TPB), 0, s, x, dy, lddy, dw, db, n, h, wd, cin, cout);
  UNET_CHECK_LAUNCH(ctx, "convT_naive_wgrad"); return UNET_OK;
}

size_t c1_wgrad_ws_bytes(int cout) { return (size_t)(C1_BLOCKS + 32) * 10 * cout * sizeof(float); }

template <typename T>
static int32_t c1_wgrad_impl(unet_ctx* ctx, const float* x, const T* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                           int wd, int cout, hipStream_t s) {
  if ((cout & 3) || TPB % (cout / 4) || (cout / 4) > 64 || (long long)n * h * wd >= (1LL << 31)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3_c1_wgrad: cout=%d unsupported", cout);
  if (!ws || ws_bytes < c1_wgrad_ws_bytes(cout)) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3_c1_wgrad: workspace too small");
  long long groups = ((long long)n * h * wd * (cout / 4) + TPB - 1) / TPB;
  int blocks = (int)std::min<long long>(C1_BLOCKS, std::max<long long>(groups, 1));
  if ((wd & 3) == 0) hipLaunchKernelGGL((conv3x3_c1_wgrad_kernel<T, true>), dim3(blocks), dim3(TPB), 0, s, x, dy, static_cast<float*>(ws), n, h, wd, cout);
  else hipLaunchKernelGGL((conv3x3_c1_wgrad_kernel<T, false>), dim3(blocks), dim3(TPB), 0, s, x, dy, static_cast<float*>(ws), n, h, wd, cout);
  float* part = static_cast<float*>(ws); float* part2 = part + (size_t)C1_BLOCKS * 10 * cout;
  const int ngroups = (blocks + 31) / 32;
  hipLaunchKernelGGL(reduce_c1_kernel, dim3((10 * cout + 127) / 128, ngroups), dim3(128), 0, s, part, part2, nullptr, blocks, 32, cout, 0);
  hipLaunchKernelGGL(reduce_c1_kernel, dim3((10 * cout + 127) / 128, 1), dim3(128), 0, s, part2, dw, db, ngroups, ngroups, cout, 1);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_c1_wgrad"); return UNET_OK;
}
int32_t k_conv3x3_c1_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cout,
                           hipStream_t s) { return c1_wgrad_impl(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cout, s); }
int32_t k_conv3x3_c1_wgrad_bf16(unet_ctx* ctx, const float* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd,
                                int cout, hipStream_t s) { return c1_wgrad_impl(ctx, x, dy, dw, db, ws, ws_bytes, n, h, wd, cout, s); }
