// fp32 3x3 convolution (forward + data gradient) on the BF16 matrix cores with fp32 accuracy: the "x3" kernels.
//
// gfx950 has no TF32 and its fp32 MFMA runs at the vector rate (157 TFLOP/s); v_mfma_f32_32x32x16_bf16 runs 16x faster.  Every fp32 number is
// EXACTLY the sum of three bf16 numbers, x = h + m + l with h = RN_bf16(x), m = RN_bf16(x - h), l = x - h - m (8 + 8 + 8 significand bits cover
// fp32's 24; the subtractions are exact), so
//     x * w = xh wh + (xh wm + xm wh) + (xh wl + xm wm + xl wh) + O(2^-26 |x w|)
// and six bf16 MFMAs with fp32 accumulation reproduce the fp32 product sum: 6/16 of the fp32-MFMA time per multiply.  Measured on the box
// (tools/probe/split3_probe.hip, K = 16 ... 4608, against float64): relative L2 error 3.2e-8 ... 1.09e-6 for the six products vs 7.7e-8 ... 1.22e-6 for
// v_mfma_f32_32x32x2_f32 -- the same accuracy class; dropping the three smallest products would cost 4.5e-6 and is not done.
//
// The kernel is the implicit GEMM of kernels_bf16.hip (A = weights 32 x 16, B = 32 pixels of a row, lane = 16 consecutive output channels of a pixel)
// with fp32 tensors on both sides:
//   * the input patch is fetched as fp32 (16-B pieces, branch-free buffer loads), split into the three bf16 planes in registers
//     (3 v_cvt_pk_bf16_f32 + 4 shift/and + 4 v_sub per pair: 5.5 VALU per element against >= 27 * NB bf16 MFMAs that consume it) and written to
//     three LDS planes [plane][pixel][16 channels] -- conflict-free ds_write_b64 / ds_read_b128;
//   * the weights are split ONCE per launch into the exact LDS image ([group][chunk][tap][n-block][plane][k-half][row][8], x3_wimg_kernel) and
//     staged with linear 16-B copies;
//   * per 16-channel chunk a wave issues 9 taps x RW rows x NB blocks x 6 products from 3 (RW + 2) * 3 + 27 NB ds_read_b128: one LDS operand
//     byte feeds 2-3x more MFMA work than in the one-product bf16 kernel, and 10x more than in the fp32 Winograd kernels -- the loop is MFMA-bound;
//   * small terms are accumulated first (mm, hl, lh, hm, mh, hh) into the same fp32 accumulator.
// LDS is single-buffered (3 planes of a 18 x 34 patch + 3 planes of weights = 86 / 114 KB): the next chunk travels global -> registers under
// the MFMAs and is split + stored between two barriers (~10 % of a chunk's MFMA time; the 512-register budget of a 1-wave-per-SIMD kernel pays for
// the prefetch registers).  Epilogues are the fp32 path's (bias / ReLU / ELU / fused dropout / producer masks / folded-BatchNorm modes).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace {

typedef unet_bf16x8 bf16x8;
typedef unet_f32x16 f32x16;

__host__ __device__ inline int cperm(int m) { return ((m >> 2) & 1) * 16 + (m & 3) + 4 * (m >> 3); }       // as kernels_bf16.hip: lane -> 16 consecutive channels

// x = h + m + l for two values at once; returns the packed bf16 pairs
__device__ __forceinline__ void split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16x2(a, b);
  const float ra = a - bf16_lo(h), rb = b - bf16_hi(h);
  m = pack_bf16x2(ra, rb);
  l = pack_bf16x2(ra - bf16_lo(m), rb - bf16_hi(m));
}

// ---------------------------------------------------------------------------------------------------------------------
// weight image: img[((((((g*nchunks + chunk)*9 + tap)*NB + nb)*3 + plane)*2 + half)*32 + m][j] = plane(W(tap, k, mm))
//   k = chunk*16 + half*8 + j,  mm = (g*NB + nb)*32 + cperm(m),  W(tap,k,mm) = w[(flip ? 8-tap : tap)*tap_stride + k*sk + mm*sm]
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void x3_wimg_body(const float* __restrict__ w, unet_bf16* __restrict__ img, int NB, int nchunks, long long tap_stride, int tap_flip,
                                             long long sk, long long sm, long long total, int M) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {          // e indexes (g, chunk, tap, nb, half, m)
    long long r = e;
    const int m = (int)(r & 31); r >>= 5;
    const int half = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % NB); r /= NB;
    const int tap = (int)(r % 9); r /= 9;
    const int chunk = (int)(r % nchunks); const int g = (int)(r / nchunks);
    const long long k0 = (long long)chunk * 16 + half * 8;
    const long long mm = ((long long)g * NB + nb) * 32 + cperm(m);
    const float* src = w + (long long)(tap_flip ? 8 - tap : tap) * tap_stride + k0 * sk + mm * sm;
    unsigned hh[4], mmid[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = mm < M ? src[(2 * j) * sk] : 0.0f, b = mm < M ? src[(2 * j + 1) * sk] : 0.0f;          // rows past M (a 32-row tile of a 16-channel layer) are zero
      split3(a, b, hh[j], mmid[j], ll[j]);
    }
    const long long base = (((((long long)g * nchunks + chunk) * 9 + tap) * NB + nb) * 3 * 2 + half) * 32 + m;      // plane 0; planes are 2 * 32 rows apart
    *reinterpret_cast<uint4*>(img + (base + 0 * 64) * 8) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *reinterpret_cast<uint4*>(img + (base + 1 * 64) * 8) = make_uint4(mmid[0], mmid[1], mmid[2], mmid[3]);
    *reinterpret_cast<uint4*>(img + (base + 2 * 64) * 8) = make_uint4(ll[0], ll[1], ll[2], ll[3]);
  }
}
__global__ __launch_bounds__(256) void x3_wimg_kernel(const float* __restrict__ w, unet_bf16* __restrict__ img, int NB, int nchunks, long long tap_stride, int tap_flip,
                                                      long long sk, long long sm, long long total, int M) {
  x3_wimg_body(w, img, NB, nchunks, tap_stride, tap_flip, sk, sm, total, M);
}
__global__ __launch_bounds__(256) void x3_wimg_multi_kernel(unet_wimg_prep_list L) {          // all layers of a program in one launch (blockIdx.y = layer)
  const unet_wimg_prep& p = L.item[blockIdx.y];
  x3_wimg_body(p.w, p.img, p.nb, p.nchunks, p.tap_stride, p.flip, p.sk, p.sm, p.total8, p.m);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

template <int NB, int RW, bool GEN, int WPS, bool DB>
__global__ __launch_bounds__(256, WPS) void conv_x3_kernel(const float* __restrict__ x, const unet_bf16* __restrict__ wimg, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, float* __restrict__ y, int N, int H, int W, int K, int M, int act,
                                                           int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y, int groups,
                                                           int total_blocks) {
  constexpr int TH = 4 * RW;                             // tile rows: RW per wave
  constexpr int PR = TH + 2, PWD = 34, NPIX = PR * PWD;
  constexpr int PLANE = NPIX * 32;                       // bytes of one bf16 plane of the 16-channel pixel patch
  constexpr int IN_BYTES = 3 * PLANE, W_BYTES = 9 * NB * 3 * 2 * 32 * 16;
  constexpr int PPIECES = NPIX * 4;                      // 16-B fp32 pieces of the patch: (pixel, channel quad)
  constexpr int PL = (PPIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [DB ? 2 : 1][IN_BYTES]: the three bf16 planes of the patch

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block map (workgroup b runs on XCD b % 8): an XCD gets a contiguous range of work items -- the channel groups of one spatial tile,
  // then the neighbouring tiles -- so a patch (and the weight image of a group) is fetched into ONE L2
  const int per = gridDim.x >> 3;
  const int wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wi >= total_blocks) return;
  const int g = wi % groups; int t = wi / groups;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; const int n = t / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int nchunks = K / 16;

  f32x16 acc[RW][NB];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (long long)n * H * W * K, (long long)H * W * K * 4);
  // the weight image of this channel group: [chunk][tap][nb][plane][half][32 rows][16 B]; the A fragments are read straight from L2 (every workgroup of
  // the group, on every CU, reads the same 27 * NB KB per chunk: they never leave the L2), 16 B per lane, 1 KiB contiguous per wave-instruction
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(wimg) + (long long)g * nchunks * (W_BYTES / 2), 0, nchunks * W_BYTES, 0x00020000);
  const int wlane = hi * 512 + l31 * 16;
  int poff[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = tid + k * 256;
    const int q = idx & 3, pix = idx >> 2;
    const int r = pix / PWD, c = pix - r * PWD;
    const int gy = y0 + r - 1, gx = x0 + c - 1;
    const bool ok = idx < PPIECES && gy >= 0 && gy < H && gx >= 0 && gx < W;
    poff[k] = ok ? ((gy * W + gx) * K + q * 4) * 4 : UNET_OOB;          // halo and overhang pieces read 0 (out-of-range buffer offset)
  }
  unet_u32x4 preg[PL];
  auto issue_loads = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) preg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, poff[k], chunk * 64, 0);
  };
  auto store_lds = [&](char* s_in) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int idx = tid + k * 256;
      if (idx < PPIECES) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3(__uint_as_float(preg[k][0]), __uint_as_float(preg[k][1]), h0, m0, l0);
        split3(__uint_as_float(preg[k][2]), __uint_as_float(preg[k][3]), h1, m1, l1);
        char* p = s_in + idx * 8;                                       // plane-local layout [pixel][16 channels] bf16: piece (pixel, quad) -> 8 bytes
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(p + 2 * PLANE) = make_uint2(l0, l1);
      }
    }
  };
  // MFMAs of the taps (ky, kx) for kx in [KX0, KX1) of one 16-channel chunk
  auto compute = [&](const char* s_in, int chunk, auto kx0, auto kx1) __attribute__((always_inline)) {
    const int wbase = chunk * W_BYTES;
#pragma unroll
    for (int kx = decltype(kx0)::value; kx < decltype(kx1)::value; ++kx) {
      bf16x8 px[3][RW + 2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) px[p][rr] = lds_frag(s_in + p * PLANE + (((wave * RW + rr) * PWD + l31 + kx) * 32 + hi * 16));
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        bf16x8 wf[NB][3];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < 3; ++p)
            wf[nb][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, wbase + (((ky * 3 + kx) * NB + nb) * 3 + p) * 1024, 0));
        // six products per (row, block), small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int pr = 0; pr < 6; ++pr) {
          constexpr int PW[6] = {1, 0, 2, 0, 1, 0}, PX[6] = {1, 2, 0, 1, 0, 0};          // (weight plane, pixel plane): mm, hl, lh, hm, mh, hh
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][PW[pr]], px[PX[pr]][r + ky], acc[r][nb], 0, 0, 0);
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  issue_loads(0);
  store_lds(smem);
  __syncthreads();
  if (DB) {
    // two patch buffers: the next chunk is split and stored between the kx = 1 and kx = 2 taps of the current one (its loads were issued a chunk's MFMAs
    // earlier), one barrier per chunk
    for (int chunk = 0; chunk < nchunks - 1; ++chunk) {
      char* cur = smem + (chunk & 1) * IN_BYTES; char* nxt = smem + ((chunk & 1) ^ 1) * IN_BYTES;
      issue_loads(chunk + 1);
      compute(cur, chunk, I0{}, I2{});
      store_lds(nxt);
      compute(cur, chunk, I2{}, I3{});
      __syncthreads();
    }
    compute(smem + ((nchunks - 1) & 1) * IN_BYTES, nchunks - 1, I0{}, I3{});
  } else {
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk + 1 < nchunks) issue_loads(chunk + 1);
      compute(smem, chunk, I0{}, I3{});
      if (chunk + 1 < nchunks) {
        __syncthreads();                                   // every wave is done reading this chunk's planes
        store_lds(smem);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: lane (l31, hi) holds, for pixel column l31 of each of its RW rows, channels mb + 0..15
  const int px_ = x0 + l31;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int mb = (g * NB + nb) * 32 + hi * 16;
    if (mb >= M) continue;                                 // zero-padded rows of a tile that overhangs M (M % 16 == 0)
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + mb + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[q * 4] = b4.x; bv[q * 4 + 1] = b4.y; bv[q * 4 + 2] = b4.z; bv[q * 4 + 3] = b4.w;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int py = y0 + wave * RW + r;
      if (py >= H || px_ >= W) continue;
      const long long o = (((long long)n * H + py) * W + px_) * M + mb;
      float v[16], mv[16];
      const bool want_m = mask_mode != MASK_NONE && mask_mode != MASK_BIAS_TAB;
      if (want_m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 m4 = *reinterpret_cast<const float4*>(mask + o + q * 4);
          mv[q * 4] = m4.x; mv[q * 4 + 1] = m4.y; mv[q * 4 + 2] = m4.z; mv[q * 4 + 3] = m4.w;
        }
      }
      if (mask_mode >= MASK_BN_BWD) {
        // data gradient of a conv whose input BatchNorm is folded: dx = K0 dz + K1 x + K2, x read where a ReLU layer reads its mask
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 k1 = *reinterpret_cast<const float4*>(bias + M + mb + q * 4), k2 = *reinterpret_cast<const float4*>(bias + 2 * M + mb + q * 4);
          v[q * 4] = fmaf(bv[q * 4], acc[r][nb][q * 4], fmaf(k1.x, mv[q * 4], k2.x));
          v[q * 4 + 1] = fmaf(bv[q * 4 + 1], acc[r][nb][q * 4 + 1], fmaf(k1.y, mv[q * 4 + 1], k2.y));
          v[q * 4 + 2] = fmaf(bv[q * 4 + 2], acc[r][nb][q * 4 + 2], fmaf(k1.z, mv[q * 4 + 2], k2.z));
          v[q * 4 + 3] = fmaf(bv[q * 4 + 3], acc[r][nb][q * 4 + 3], fmaf(k1.w, mv[q * 4 + 3], k2.w));
        }
        if (mask_mode == MASK_BN_BWD_RELU) {               // x = relu(conv): the gradient stops where it was clipped
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = mv[i] > 0.f ? v[i] : 0.f;
        }
      } else if (mask_mode == MASK_BIAS_TAB && (py == 0 || py == H - 1 || px_ == 0 || px_ == W - 1)) {
        // forward of such a conv: border pixels see fewer taps of the BatchNorm shift -- the bias vector of their border class (`mask` = table [16][M])
        const int cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px_ == 0) | ((px_ == W - 1) << 1));
        const float* tb = mask + (long long)cls * M + mb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(tb + q * 4);
          v[q * 4] = acc[r][nb][q * 4] + b4.x; v[q * 4 + 1] = acc[r][nb][q * 4 + 1] + b4.y; v[q * 4 + 2] = acc[r][nb][q * 4 + 2] + b4.z; v[q * 4 + 3] = acc[r][nb][q * 4 + 3] + b4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[r][nb][i] + bv[i];
      }
      if (!GEN) {
        if (act == ACT_RELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (mask_mode == MASK_RELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = mv[i] > 0.f ? v[i] : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = apply_act(v[i], act);
        if (mask_mode >= MASK_BN_BWD) {                    // (v already holds K0 dz + K1 x + K2) then the ELU (+ dropout) derivative of x's producer
          if (mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP) {
            const int mm = mask_mode == MASK_BN_BWD_ELU_DROP ? MASK_ELU_DROP : MASK_ELU;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 ks4 = make_float4(1.f, 1.f, 1.f, 1.f);
              if (mm == MASK_ELU_DROP) ks4 = keep_scale((o >> 2) + q, rate, seed);
              v[q * 4] *= mask_factor(mv[q * 4], mm, ks4.x, rate); v[q * 4 + 1] *= mask_factor(mv[q * 4 + 1], mm, ks4.y, rate);
              v[q * 4 + 2] *= mask_factor(mv[q * 4 + 2], mm, ks4.z, rate); v[q * 4 + 3] *= mask_factor(mv[q * 4 + 3], mm, ks4.w, rate);
            }
          }
        } else if (mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) {
          if (rate > 0.0f) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 ks4 = keep_scale((o >> 2) + q, rate, seed);
              v[q * 4] *= ks4.x; v[q * 4 + 1] *= ks4.y; v[q * 4 + 2] *= ks4.z; v[q * 4 + 3] *= ks4.w;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float4 ks4 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (mask_mode == MASK_ELU_DROP) ks4 = keep_scale((o >> 2) + q, rate, seed);
            v[q * 4] *= mask_factor(mv[q * 4], mask_mode, ks4.x, rate); v[q * 4 + 1] *= mask_factor(mv[q * 4 + 1], mask_mode, ks4.y, rate);
            v[q * 4 + 2] *= mask_factor(mv[q * 4 + 2], mask_mode, ks4.z, rate); v[q * 4 + 3] *= mask_factor(mv[q * 4 + 3], mask_mode, ks4.w, rate);
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(y + o + q * 4) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
    }
  }
}

template <int NB, int RW, int WPS, bool DB>
int32_t launch_x3(unet_ctx* ctx, const float* x, const unet_bf16* wimg, const float* bias, const float* mask, int mask_mode, float* y, int n, int h, int wd, int K,
                  int M, int act, float rate, unsigned long long seed, hipStream_t s) {
  constexpr int TH = 4 * RW;
  constexpr int NPIX = (TH + 2) * 34;
  constexpr size_t smem = (size_t)(DB ? 2 : 1) * 3 * NPIX * 32;
  if (!mask) mask_mode = MASK_NONE;
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH, groups = (M + 32 * NB - 1) / (32 * NB);
  const long long total = (long long)tiles_x * tiles_y * n * groups;
  if (total >= (1LL << 28)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv x3: too many tiles");
  const unsigned grid = (unsigned)(8 * ((total + 7) / 8));
  const bool gen = act == ACT_ELU || rate > 0.0f || mask_mode == MASK_ELU || mask_mode == MASK_ELU_DROP || mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP;
  auto go = [&](auto kern) -> int32_t {
    if (smem > 65536) UNET_BIG_LDS(ctx, kern, smem, "conv_x3");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, x, wimg, bias, mask, y, n, h, wd, K, M, act, mask_mode, rate, seed, tiles_x, tiles_y, groups, (int)total);
    return UNET_OK;
  };
  int32_t r;
  if (gen) r = go(conv_x3_kernel<NB, RW, true, WPS, DB>); else r = go(conv_x3_kernel<NB, RW, false, WPS, DB>);
  if (r) return r;
  UNET_CHECK_LAUNCH(ctx, "conv_x3");
  return UNET_OK;
}

int x3_mode() {
  static const int on = [] { const char* e = getenv("UNET_X3"); return e ? atoi(e) : 1; }();          // A/B switch: 0 = the fp32-MFMA Winograd kernels
  return on;
}

}  // namespace

// n-blocks of 32 output channels per workgroup for a layer with M output channels (both the image and the launch use it)
static int x3_nb(int M) { return (M % 64) == 0 ? 2 : 1; }

// a launch with K contraction channels and M output channels runs on the x3 kernels (both the weight preparation and the launch ask this)
bool x3_conv3x3_selected(int K, int M) { return x3_mode() != 0 && K >= 16 && (K % 16) == 0 && M >= 32 && (M % 32) == 0; }

// the split weight image of a layer with (forward) dimensions cin x cout; flip = the data-gradient form (contraction over cout, flipped taps).
// `img` needs 54 * K * M bytes (x3_wimg_bytes) -- it fits the 16 * cin * cout floats every caller reserves for transformed weights
size_t x3_wimg_bytes(int K, int M) {
  const int nb = x3_nb(M), groups = (M + 32 * nb - 1) / (32 * nb);
  return (size_t)groups * (K / 16) * 9 * nb * 3 * 2 * 32 * 16;
}

static void x3_prep_item(const float* w, unet_bf16* img, int cin, int cout, int flip, unet_wimg_prep* p) {
  const int K = flip ? cout : cin, M = flip ? cin : cout;
  const int nb = x3_nb(M), groups = (M + 32 * nb - 1) / (32 * nb), nchunks = K / 16;
  p->w = w; p->img = img; p->tap_stride = (long long)cin * cout; p->flip = flip;
  p->sk = flip ? 1 : cout; p->sm = flip ? cout : 1;
  p->nb = nb; p->nchunks = nchunks; p->m = M;
  p->total8 = (long long)groups * nchunks * 9 * nb * 2 * 32;
}

int32_t k_x3_weights(unet_ctx* ctx, const float* w, void* img, int cin, int cout, int flip, hipStream_t s) {
  unet_wimg_prep p;
  x3_prep_item(w, static_cast<unet_bf16*>(img), cin, cout, flip, &p);
  hipLaunchKernelGGL(x3_wimg_kernel, dim3((unsigned)std::min<long long>((p.total8 + 255) / 256, 1024)), dim3(256), 0, s, p.w, p.img, p.nb, p.nchunks, p.tap_stride, p.flip, p.sk,
                     p.sm, p.total8, p.m);
  UNET_CHECK_LAUNCH(ctx, "x3_weights");
  return UNET_OK;
}

int32_t k_x3_weights_multi(unet_ctx* ctx, const float* const* w, void* const* img, const int* cin, const int* cout, const int* flip, int count, hipStream_t s) {
  if (count < 1) return UNET_OK;
  if (count > UNET_WINO_PREP_MAX) UNET_FAIL(ctx, UNET_E_ARG, "x3_weights_multi: too many layers");
  unet_wimg_prep_list L; L.n = count;
  long long most = 1;
  for (int k = 0; k < count; ++k) {
    x3_prep_item(w[k], static_cast<unet_bf16*>(img[k]), cin[k], cout[k], flip[k], &L.item[k]);
    most = std::max(most, L.item[k].total8);
  }
  hipLaunchKernelGGL(x3_wimg_multi_kernel, dim3((unsigned)std::min<long long>((most + 255) / 256, 256), (unsigned)count), dim3(256), 0, s, L);
  UNET_CHECK_LAUNCH(ctx, "x3_weights_multi");
  return UNET_OK;
}

// x [n,h,wd,K] dense NHWC fp32, wimg from k_x3_weights (K contraction channels, M output channels), y [n,h,wd,M] fp32
int32_t k_conv3x3_x3_fwd(unet_ctx* ctx, const float* x, const void* wimg, const float* bias, const float* mask, int mask_mode, float* y, int n, int h, int wd, int K,
                         int M, int act, float rate, uint64_t seed, hipStream_t s) {
  if (K < 16 || (K % 16) || M < 32 || (M % 32)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 x3: K=%d (multiple of 16) M=%d (multiple of 32)", K, M);
  if ((long long)h * wd * std::max(K, M) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 x3: one image must stay below 1 GiB (32-bit buffer offsets)");
  const unet_bf16* img = static_cast<const unet_bf16*>(wimg);
  // tile choice (UNET_X3_TILE for measurements: 1 = the 8-row two-workgroups-per-CU tile everywhere, 2 = the 16-row double-buffered tile everywhere):
  //   short contractions (K <= 64: 2-4 chunks per tile, prologue + epilogue weigh as much as the loop) -> 8-row tiles, 33 KB of LDS, <= 256 registers,
  //   two workgroups per CU cover each other's ends; long contractions -> 16-row tiles, two patch buffers (117 KB), one workgroup per CU, 512 registers
  static const int tile = [] { const char* e = getenv("UNET_X3_TILE"); return e ? atoi(e) : 0; }();
  const bool small = tile == 1 || (tile != 2 && (K <= 64 || h <= 8));
  if (x3_nb(M) == 1) {
    if (small) return launch_x3<1, 2, 2, false>(ctx, x, img, bias, mask, mask_mode, y, n, h, wd, K, M, act, rate, seed, s);
    return launch_x3<1, 4, 1, true>(ctx, x, img, bias, mask, mask_mode, y, n, h, wd, K, M, act, rate, seed, s);
  }
  if (small) return launch_x3<2, 2, 2, false>(ctx, x, img, bias, mask, mask_mode, y, n, h, wd, K, M, act, rate, seed, s);
  return launch_x3<2, 4, 1, true>(ctx, x, img, bias, mask, mask_mode, y, n, h, wd, K, M, act, rate, seed, s);
}
