// Mixed-precision convolutions of the U-Net hot path: activations and activation gradients are stored as bf16 in HBM (half the
// traffic of the fp32 path), parameters / parameter gradients / accumulators stay fp32, products run on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense: 16x the fp32 MFMA rate, so these layers are HBM- or LDS-bound, not MFMA-bound).
//
// Forward / data gradient / transposed conv share one implicit-GEMM kernel (conv_bf16_kernel):
//   D[m][pixel] = sum_k Wimg[m][k] * X[pixel][k]        A operand (32 x 16) = weights, B operand (16 x 32) = 32 pixels of a row
//   so a lane ends up with 16 CONSECUTIVE output channels of ONE pixel (the A rows are stored permuted, cperm()) and the
//   tile leaves as 32-byte runs per lane.  The weights are re-laid out once per launch into the exact LDS image the kernel
//   reads ([group][chunk][k-step][tap][n-block][k-half][row][8], bf16) by wimg_kernel.
//   workgroup = 4 waves, output tile 16 rows x 32 columns x (32*NB) channels, wave = 4 rows: per staged 16-channel chunk a
//   wave issues 9 taps x 4 rows x NB MFMAs from 18 + 9*NB ds_read_b128 (the three ky taps share their pixel rows).
//   LDS is double buffered (global -> registers under the MFMAs of the previous chunk -> LDS), one barrier per chunk.
// Weight gradient (wgrad_bf16_kernel): K = pixels, both operands need 8 consecutive PIXELS of one channel per lane while NHWC
//   keeps channels contiguous: the rows are staged as they come (16-byte pieces) and read back through the gfx950 LDS
//   transpose read (ds_read_b64_tr_b16), two reads per operand.
#include <stdlib.h>

#include "common.h"

namespace {

typedef unet_bf16x8 bf16x8;
typedef unet_f32x16 f32x16;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// A-tile row m <-> output channel: lane (hi, register r) of the MFMA result holds row (r&3) + 8*(r>>2) + 4*hi; storing channel
// hi*16 + r in that row gives every lane 16 consecutive channels
__host__ __device__ inline int cperm(int m) { return ((m >> 2) & 1) * 16 + (m & 3) + 4 * (m >> 3); }

// ---------------------------------------------------------------------------------------------------------------------
// weight image: img[(((((g*nchunks + chunk)*KS + ks)*T + tap)*NB + nb)*2 + half)*32 + m][j] = W(tap, k, mm)
//   k = (chunk*KS + ks)*16 + half*8 + j,  mm = (g*NB + nb)*32 + cperm(m),  W(tap,k,mm) = w[tap_base(tap) + k*sk + mm*sm]
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wimg_body(const float* __restrict__ w, unet_bf16* __restrict__ img, int T, int KS, int NB, int nchunks,
                                          long long tap_stride, int tap_flip, long long sk, long long sm, long long total8, int M) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total8; e += (long long)gridDim.x * 256) {
    long long r = e;
    const int m = (int)(r & 31); r >>= 5;
    const int half = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % NB); r /= NB;
    const int tap = (int)(r % T); r /= T;
    const int ks = (int)(r % KS); r /= KS;
    const int chunk = (int)(r % nchunks); const int g = (int)(r / nchunks);
    const long long k0 = ((long long)chunk * KS + ks) * 16 + half * 8;
    const long long mm = ((long long)g * NB + nb) * 32 + cperm(m);
    const float* src = w + (long long)(tap_flip ? T - 1 - tap : tap) * tap_stride + k0 * sk + mm * sm;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = mm < M ? src[j * sk] : 0.0f;          // rows past M (a 32-row tile of a 16-channel layer) are zero
    *reinterpret_cast<uint4*>(img + e * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  }
}
__global__ __launch_bounds__(256) void wimg_kernel(const float* __restrict__ w, unet_bf16* __restrict__ img, int T, int KS, int NB, int nchunks,
                                                   long long tap_stride, int tap_flip, long long sk, long long sm, long long total8, int M) {
  wimg_body(w, img, T, KS, NB, nchunks, tap_stride, tap_flip, sk, sm, total8, M);
}
// the weight images of all conv3x3 layers of a program in ONE launch (blockIdx.y = layer)
__global__ __launch_bounds__(256) void wimg_multi_kernel(unet_wimg_prep_list L) {
  const unet_wimg_prep& p = L.item[blockIdx.y];
  wimg_body(p.w, p.img, 9, 1, p.nb, p.nchunks, p.tap_stride, p.flip, p.sk, p.sm, p.total8, p.m);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }

// MODE 0: conv3x3 'same' (forward; data gradient with the flipped/transposed weight image)
// MODE 1: convT2x2s2 forward  = per-pixel GEMM [pixels, Cin] x [Cin, 4*Cout] with a scatter epilogue into the (2i+a, 2j+b)
//         positions of a channel slice (pixel stride ldy) of the concat buffer
// MODE 2: convT2x2s2 data gradient = per-pixel GEMM over the virtual channels k = (ab, o): chunk -> (ab, o0) selects the
//         parity plane (2i+a, 2j+b) of dU (pixel stride ldx) that is staged
template <int MODE, int NB, bool GEN, int RW>
__global__ __launch_bounds__(256, 2) void conv_bf16_kernel(const unet_bf16* __restrict__ x, int ldx, const unet_bf16* __restrict__ wimg,
                                                           const float* __restrict__ bias, const unet_bf16* __restrict__ mask,
                                                           unet_bf16* __restrict__ y, int ldy, int N, int H, int W, int K, int M, int act,
                                                           int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y,
                                                           int groups, int total_blocks, double* __restrict__ stats, int stats_c) {
  constexpr int T = MODE == 0 ? 9 : 1, KS = MODE == 0 ? 1 : 2;
  constexpr int TH = 4 * RW;                             // tile rows: RW per wave
  constexpr int PR = MODE == 0 ? TH + 2 : TH, PWD = MODE == 0 ? 34 : 32, NPIX = PR * PWD;
  constexpr int PLANE = NPIX * 32;                       // bytes of one 16-channel plane of the pixel patch
  constexpr int IN_BYTES = KS * PLANE, W_BYTES = KS * T * NB * 2 * 32 * 16;
  constexpr int PPIECES = KS * NPIX * 2, WPIECES = W_BYTES / 16;
  constexpr int PL = (PPIECES + 255) / 256, WL = (WPIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];            // [2][IN_BYTES] then [2][W_BYTES]
  char* const s_in = smem; char* const s_w = smem + 2 * IN_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block map: workgroup b runs on XCD b % 8; give every XCD a contiguous range of work items (channel groups of
  // one spatial tile, then the neighbouring tiles) so that a patch is fetched into ONE L2
  const int per = gridDim.x >> 3;
  const int wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wi >= total_blocks) return;
  const int g = wi % groups; int t = wi / groups;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; const int n = t / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int HI = MODE == 2 ? 2 * H : H, WI = MODE == 2 ? 2 * W : W;
  const int nchunks = K / (16 * KS);

  f32x16 acc[RW][NB];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(x + (long long)n * HI * WI * ldx), 0, (int)((long long)HI * WI * ldx * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(wimg), 0, (int)((long long)groups * nchunks * W_BYTES), 0x00020000);
  int poff[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = tid + k * 256;
    const int half = idx & 1, pp = idx >> 1, ks = pp / NPIX, pix = pp - ks * NPIX;
    int gy, gx;
    if (MODE == 0) { const int r = pix / PWD, c = pix - r * PWD; gy = y0 + r - 1; gx = x0 + c - 1; }
    else { gy = y0 + (pix >> 5); gx = x0 + (pix & 31); }
    const bool ok = idx < PPIECES && gy >= 0 && gy < H && gx >= 0 && gx < W;
    if (MODE == 2) poff[k] = ok ? (((2 * gy) * WI + 2 * gx) * ldx + ks * 16 + half * 8) * 2 : UNET_OOB;
    else poff[k] = ok ? ((gy * WI + gx) * ldx + ks * 16 + half * 8) * 2 : UNET_OOB;
  }
  unet_u32x4 preg[PL], wreg[WL];
  auto issue_loads = [&](int chunk) __attribute__((always_inline)) {
    int soff;
    if (MODE == 2) { const int ct = K >> 2, k0 = chunk * 32, ab = k0 / ct, o0 = k0 - ab * ct; soff = (((ab >> 1) * WI + (ab & 1)) * ldx + o0) * 2; }
    else soff = chunk * 16 * KS * 2;
#pragma unroll
    for (int k = 0; k < PL; ++k) preg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, poff[k], soff, 0);
    const int wsoff = (g * nchunks + chunk) * W_BYTES;
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const int idx = tid + k * 256;
      wreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, idx < WPIECES ? idx * 16 : UNET_OOB, wsoff, 0);
    }
  };
  auto store_lds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int idx = tid + k * 256;
      if (idx < PPIECES) *reinterpret_cast<unet_u32x4*>(s_in + buf * IN_BYTES + idx * 16) = preg[k];     // piece order == LDS order: [ks][pix][half]
    }
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const int idx = tid + k * 256;
      if (idx < WPIECES) *reinterpret_cast<unet_u32x4*>(s_w + buf * W_BYTES + idx * 16) = wreg[k];
    }
  };

  issue_loads(0);
  store_lds(0);
  __syncthreads();
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cur = chunk & 1;
    if (chunk + 1 < nchunks) issue_loads(chunk + 1);
    const char* in = s_in + cur * IN_BYTES; const char* wt = s_w + cur * W_BYTES;
    if (MODE == 0) {
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        bf16x8 px[RW + 2];
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) px[rr] = lds_frag(in + (((wave * RW + rr) * PWD + l31 + kx) * 32 + hi * 16));
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          bf16x8 wf[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) wf[nb] = lds_frag(wt + ((((ky * 3 + kx) * NB + nb) * 2 + hi) * 512 + l31 * 16));
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], px[r + ky], acc[r][nb], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 px[RW], wf[NB];
#pragma unroll
        for (int r = 0; r < RW; ++r) px[r] = lds_frag(in + ks * PLANE + (((wave * RW + r) * 32 + l31) * 32 + hi * 16));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wf[nb] = lds_frag(wt + (((ks * NB + nb) * 2 + hi) * 512 + l31 * 16));
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb], px[r], acc[r][nb], 0, 0, 0);
      }
    }
    if (chunk + 1 < nchunks) store_lds(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane (l31, hi) holds, for pixel column l31 of each of its 4 rows, channels mb + 0..15
  const int px_ = x0 + l31;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int mb0 = (g * NB + nb) * 32;
    if (mb0 >= M) continue;                                 // (wave-uniform) zero-padded block of a tile that overhangs M
    // M % 32 == 16: the hi half of the last block is padding -- those lanes (and the lanes past the image edge below) compute on a valid neighbour's
    // addresses and only take part in the LDS transpose of the stores
    const int mb = mb0 + hi * 16 < M ? mb0 + hi * 16 : mb0;
    int ab = 0, oc = mb;
    if (MODE == 1) { const int ct = M >> 2; ab = mb / ct; oc = mb - ab * ct; }
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = bias ? *reinterpret_cast<const float4*>(bias + oc + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      bv[q * 4] = b4.x; bv[q * 4 + 1] = b4.y; bv[q * 4 + 2] = b4.z; bv[q * 4 + 3] = b4.w;
    }
    float st1[8], st2[8];                                   // BatchNorm statistics of the bf16 values this lane stores (channel oct lane & 3): unet_request_bn_stats
#pragma unroll
    for (int i = 0; i < 8; ++i) { st1[i] = 0.f; st2[i] = 0.f; }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int py = y0 + wave * RW + r;
      if (py >= H) continue;                                // (wave-uniform)
      const int pxs = px_ < W ? px_ : W - 1;
      long long o;
      if (MODE == 1) o = (((long long)n * 2 * H + 2 * py + (ab >> 1)) * (2 * W) + 2 * pxs + (ab & 1)) * ldy + oc;
      else o = (((long long)n * H + py) * W + pxs) * ldy + mb;
      float v[16];
      if (MODE == 0 && mask_mode >= MASK_BN_BWD) {
        // data gradient of a conv whose input BatchNorm is folded: dx = K0 dz + K1 x + K2, x read where a ReLU layer reads its mask
        const uint4 m0 = *reinterpret_cast<const uint4*>(mask + o), m1 = *reinterpret_cast<const uint4*>(mask + o + 8);
        const unsigned mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 k1 = *reinterpret_cast<const float4*>(bias + M + mb + q * 4), k2 = *reinterpret_cast<const float4*>(bias + 2 * M + mb + q * 4);
          v[q * 4] = fmaf(bv[q * 4], acc[r][nb][q * 4], fmaf(k1.x, bf16_lo(mw[2 * q]), k2.x));
          v[q * 4 + 1] = fmaf(bv[q * 4 + 1], acc[r][nb][q * 4 + 1], fmaf(k1.y, bf16_hi(mw[2 * q]), k2.y));
          v[q * 4 + 2] = fmaf(bv[q * 4 + 2], acc[r][nb][q * 4 + 2], fmaf(k1.z, bf16_lo(mw[2 * q + 1]), k2.z));
          v[q * 4 + 3] = fmaf(bv[q * 4 + 3], acc[r][nb][q * 4 + 3], fmaf(k1.w, bf16_hi(mw[2 * q + 1]), k2.w));
          if (mask_mode == MASK_BN_BWD_RELU) {                // x = relu(conv): the gradient stops where it was clipped
            v[q * 4] = bf16_lo(mw[2 * q]) > 0.f ? v[q * 4] : 0.f; v[q * 4 + 1] = bf16_hi(mw[2 * q]) > 0.f ? v[q * 4 + 1] : 0.f;
            v[q * 4 + 2] = bf16_lo(mw[2 * q + 1]) > 0.f ? v[q * 4 + 2] : 0.f; v[q * 4 + 3] = bf16_hi(mw[2 * q + 1]) > 0.f ? v[q * 4 + 3] : 0.f;
          }
        }
      } else if (MODE == 0 && mask_mode == MASK_BIAS_TAB && (py == 0 || py == H - 1 || px_ == 0 || px_ == W - 1)) {
        // forward of such a conv: border pixels see fewer taps of the BatchNorm shift -- the bias vector of their border class (`mask` = float table [16][M])
        const int cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px_ == 0) | ((px_ == W - 1) << 1));
        const float* tb = reinterpret_cast<const float*>(mask) + (long long)cls * M + mb;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(tb + q * 4);
          v[q * 4] = acc[r][nb][q * 4] + b4.x; v[q * 4 + 1] = acc[r][nb][q * 4 + 1] + b4.y; v[q * 4 + 2] = acc[r][nb][q * 4 + 2] + b4.z; v[q * 4 + 3] = acc[r][nb][q * 4 + 3] + b4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[r][nb][i] + bv[i];
      }
      if (MODE == 0 || MODE == 2) {
        if (!GEN) {
          if (act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
          }
          if (mask_mode == MASK_RELU) {
            const uint4 m0 = *reinterpret_cast<const uint4*>(mask + o), m1 = *reinterpret_cast<const uint4*>(mask + o + 8);
            const unsigned mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {            // bf16 > 0  <=>  sign clear and magnitude non-zero
              v[2 * i] = ((mw[i] & 0x8000u) == 0 && (mw[i] & 0x7FFFu) != 0) ? v[2 * i] : 0.f;
              v[2 * i + 1] = ((mw[i] & 0x80000000u) == 0 && (mw[i] & 0x7FFF0000u) != 0) ? v[2 * i + 1] : 0.f;
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = apply_act(v[i], act);
          if (mask_mode >= MASK_BN_BWD) {                      // (v already holds K0 dz + K1 x + K2) then the ELU (+ dropout) derivative of x's producer
            const int mm = mask_mode == MASK_BN_BWD_ELU_DROP ? MASK_ELU_DROP : MASK_ELU;
            const uint4 m0 = *reinterpret_cast<const uint4*>(mask + o), m1 = *reinterpret_cast<const uint4*>(mask + o + 8);
            const unsigned mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 ks4 = make_float4(1.f, 1.f, 1.f, 1.f);
              if (mm == MASK_ELU_DROP) ks4 = keep_scale((o >> 2) + q, rate, seed);
              v[q * 4] *= mask_factor(bf16_lo(mw[2 * q]), mm, ks4.x, rate); v[q * 4 + 1] *= mask_factor(bf16_hi(mw[2 * q]), mm, ks4.y, rate);
              v[q * 4 + 2] *= mask_factor(bf16_lo(mw[2 * q + 1]), mm, ks4.z, rate); v[q * 4 + 3] *= mask_factor(bf16_hi(mw[2 * q + 1]), mm, ks4.w, rate);
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
            const uint4 m0 = *reinterpret_cast<const uint4*>(mask + o), m1 = *reinterpret_cast<const uint4*>(mask + o + 8);
            const unsigned mw[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float4 ks4 = make_float4(1.f, 1.f, 1.f, 1.f);
              if (mask_mode == MASK_ELU_DROP) ks4 = keep_scale((o >> 2) + q, rate, seed);
              v[q * 4] *= mask_factor(bf16_lo(mw[2 * q]), mask_mode, ks4.x, rate); v[q * 4 + 1] *= mask_factor(bf16_hi(mw[2 * q]), mask_mode, ks4.y, rate);
              v[q * 4 + 2] *= mask_factor(bf16_lo(mw[2 * q + 1]), mask_mode, ks4.z, rate); v[q * 4 + 3] *= mask_factor(bf16_hi(mw[2 * q + 1]), mask_mode, ks4.w, rate);
            }
          }
        }
      }
      // the row leaves through a per-wave LDS staging row (both tile buffers are dead behind the K loop's last barrier): a lane holds 32 B of ONE pixel,
      // so its two 16-B stores landed 64+ B from its neighbours'; transposed, four consecutive lanes write the 64 B of a pixel's block (kernels_conv_h2.hip)
      char* const s_out = smem + wave * (32 * 80);
      *reinterpret_cast<uint4*>(s_out + l31 * 80 + hi * 32) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      *reinterpret_cast<uint4*>(s_out + l31 * 80 + hi * 32 + 16) = make_uint4(pack_bf16x2(v[8], v[9]), pack_bf16x2(v[10], v[11]), pack_bf16x2(v[12], v[13]), pack_bf16x2(v[14], v[15]));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pj = j * 16 + (lane >> 2), cj = lane & 3, pxj = x0 + pj;
        const uint4 t4 = *reinterpret_cast<const uint4*>(s_out + pj * 80 + cj * 16);
        int abj = 0, ocj = mb0;
        if (MODE == 1) { const int ct = M >> 2; abj = mb0 / ct; ocj = mb0 - abj * ct; }
        long long oj;
        if (MODE == 1) oj = (((long long)n * 2 * H + 2 * py + (abj >> 1)) * (2 * W) + 2 * pxj + (abj & 1)) * ldy + ocj;
        else oj = (((long long)n * H + py) * W + pxj) * ldy + mb0;
        if (pxj < W && mb0 + cj * 8 < M) {
          *reinterpret_cast<uint4*>(y + oj + cj * 8) = t4;
          if (MODE != 2 && stats) {                          // (wave-uniform) sums of what is STORED: the statistics pass would read these bf16 values
            const unsigned tw[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float a = bf16_lo(tw[i]), b = bf16_hi(tw[i]);
              st1[2 * i] += a; st1[2 * i + 1] += b; st2[2 * i] = fmaf(a, a, st2[2 * i]); st2[2 * i + 1] = fmaf(b, b, st2[2 * i + 1]);
            }
          }
        }
      }
    }
    if (MODE != 2 && stats) {                               // lanes L, L + 4, ... hold the same channel oct: fold them, lanes 0-3 post the wave's sums
      float* const s_stat = reinterpret_cast<float*>(smem + 4 * 32 * 80);          // [wave][nb][sum | sum of squares][32 channels], behind the staging rows
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        st1[i] += __shfl_xor(st1[i], 4); st1[i] += __shfl_xor(st1[i], 8); st1[i] += __shfl_xor(st1[i], 16); st1[i] += __shfl_xor(st1[i], 32);
        st2[i] += __shfl_xor(st2[i], 4); st2[i] += __shfl_xor(st2[i], 8); st2[i] += __shfl_xor(st2[i], 16); st2[i] += __shfl_xor(st2[i], 32);
      }
      if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { s_stat[((wave * NB + nb) * 2 + 0) * 32 + lane * 8 + i] = st1[i]; s_stat[((wave * NB + nb) * 2 + 1) * 32 + lane * 8 + i] = st2[i]; }
      }
    }
  }
  if (MODE != 2 && stats) {
    float* const s_stat = reinterpret_cast<float*>(smem + 4 * 32 * 80);
    __syncthreads();
    if (tid < NB * 64) {
      const int nb = tid >> 6, kind = (tid >> 5) & 1, c32 = tid & 31;
      const int mb = (g * NB + nb) * 32;
      if (mb + c32 < M) {
        float t = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += s_stat[((wv * NB + nb) * 2 + kind) * 32 + c32];
        int ch = mb + c32;
        if (MODE == 1) ch = ch % (M >> 2);                 // ConvT: the four (a, b) planes of a channel
        atomicAdd(stats + (size_t)(blockIdx.x % UNET_BN_SLOTS) * UNET_BN_SLOT_DOUBLES + (kind ? stats_c : 0) + ch, (double)t);
      }
    }
  }
}

template <int MODE, int NB, int RW = 4>
int32_t launch_conv_bf16(unet_ctx* ctx, const unet_bf16* x, int ldx, const unet_bf16* wimg, const float* bias, const unet_bf16* mask, int mask_mode,
                         unet_bf16* y, int ldy, int n, int h, int wd, int K, int M, int act, float rate, unsigned long long seed, hipStream_t s) {
  constexpr int T = MODE == 0 ? 9 : 1, KS = MODE == 0 ? 1 : 2;
  constexpr int TH = 4 * RW;
  constexpr int NPIX = MODE == 0 ? (TH + 2) * 34 : TH * 32;
  constexpr int IN_BYTES = KS * NPIX * 32, W_BYTES = KS * T * NB * 2 * 32 * 16;
  constexpr size_t smem = 2 * (size_t)(IN_BYTES + W_BYTES);
  if (!mask) mask_mode = MASK_NONE;
  if ((long long)(MODE == 2 ? 4 : 1) * h * wd * ldx * 2 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv bf16: one image must stay below 1 GiB (32-bit buffer offsets)");
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH, groups = (M + 32 * NB - 1) / (32 * NB);
  const long long total = (long long)tiles_x * tiles_y * n * groups;
  if (total >= (1LL << 28)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv bf16: too many tiles");
  const unsigned grid = (unsigned)(8 * ((total + 7) / 8));
  const bool gen = MODE == 0 && (act == ACT_ELU || rate > 0.0f || mask_mode == MASK_ELU || mask_mode == MASK_ELU_DROP || mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP);
  // an armed statistics request (common.h: unet_ctx::stats_req_c), exactly as the fp32 launcher honours it (kernels_conv_h2.hip: launch_h2)
  double* stats = nullptr; int stats_c = 0;
  if (MODE != 2 && ctx->stats_req_c > 0) {
    const int c = ctx->stats_req_c; ctx->stats_req_c = 0;
    if ((mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) && 2 * c <= UNET_BN_SLOT_DOUBLES && c == (MODE == 1 ? M / 4 : M) && ctx->bn_slots && smem >= 4 * 32 * 80 + 4 * NB * 64 * 4 &&
        (M % 32) == 0 && !ctx->opt_deterministic) {          // (deterministic mode: the bf16 kernels leave the statistics to their own fixed-order pass)          // (a half-padded 16-channel block pays more in this epilogue than its statistics pass costs: classifier c1b 0.28 -> 0.39 ms)
      stats = ctx->bn_slots; stats_c = c; ctx->stats_in_slots = y; ctx->stats_in_slots_c = c; ctx->stats_in_slots_xs = false;
    }
  }
  auto go = [&](auto kern) -> int32_t {
    if (smem > 65536) UNET_BIG_LDS(ctx, kern, smem, "conv_bf16");
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, x, ldx, wimg, bias, mask, y, ldy, n, h, wd, K, M, act, mask_mode, rate, seed, tiles_x, tiles_y, groups, (int)total, stats,
                       stats_c);
    return UNET_OK;
  };
  int32_t r;
  if (gen) r = go(conv_bf16_kernel<MODE, NB, (MODE == 0), RW>); else r = go(conv_bf16_kernel<MODE, NB, false, RW>);
  if (r) return r;
  UNET_CHECK_LAUNCH(ctx, "conv_bf16");
  return UNET_OK;
}

// weight image for (T taps, K contraction channels, M output channels); returns bytes written
int32_t make_wimg(unet_ctx* ctx, const float* w, unet_bf16* img, int T, int KS, int NB, int K, int M, long long tap_stride, int tap_flip, long long sk,
                  long long sm, hipStream_t s) {
  const int nchunks = K / (16 * KS), groups = (M + 32 * NB - 1) / (32 * NB);
  const long long total8 = (long long)groups * nchunks * KS * T * NB * 2 * 32;
  const unsigned grid = (unsigned)std::min<long long>((total8 + 255) / 256, 2048);
  hipLaunchKernelGGL(wimg_kernel, dim3(grid), dim3(256), 0, s, w, img, T, KS, NB, nchunks, tap_stride, tap_flip, sk, sm, total8, M);
  UNET_CHECK_LAUNCH(ctx, "wimg");
  return UNET_OK;
}

}  // namespace

bool bf16_conv3x3_supported(int cin, int cout) { return cin >= 16 && (cin % 16) == 0 && cout >= 16 && (cout % 16) == 0; }
bool bf16_convT_supported(int cin, int cout) { return cin >= 32 && (cin % 32) == 0 && cout >= 32 && (cout % 32) == 0; }

static void bf16_conv_tile(int cout, int* NB, int* RW) {       // see the tile-choice note in k_conv3x3_bf16_fwd
  constexpr int narrow_max = 32;
  *NB = (cout % 64) == 0 ? 2 : 1; *RW = 4;
  if (cout <= narrow_max) { *NB = 1; *RW = 2; }
}

// item k: weights w (the layer's forward kernel), image scratch, (cin, cout) as k_conv3x3_bf16_fwd takes them (already swapped for flip = 1)
int32_t k_wimg_multi(unet_ctx* ctx, unet_wimg_prep_list* L, const int* cin, const int* cout, hipStream_t s) {
  if (L->n < 1) return UNET_OK;
  long long most = 1;
  for (int k = 0; k < L->n; ++k) {
    unet_wimg_prep& p = L->item[k];
    int NB, RW; bf16_conv_tile(cout[k], &NB, &RW);
    p.nb = NB; p.nchunks = cin[k] / 16; p.m = cout[k];
    const int groups = (cout[k] + 32 * NB - 1) / (32 * NB);
    p.total8 = (long long)groups * p.nchunks * 9 * NB * 2 * 32;
    p.tap_stride = (long long)cin[k] * cout[k];
    if (!p.flip) { p.sk = cout[k]; p.sm = 1; } else { p.sk = 1; p.sm = cin[k]; }
    most = std::max(most, p.total8);
  }
  hipLaunchKernelGGL(wimg_multi_kernel, dim3((unsigned)std::min<long long>((most + 255) / 256, 512), (unsigned)L->n), dim3(256), 0, s, *L);
  UNET_CHECK_LAUNCH(ctx, "wimg_multi");
  return UNET_OK;
}

// forward (flip = 0, w = [3][3][cin][cout]) or data gradient (flip = 1: w = the layer's forward weights [3][3][cout][cin], x = dy with
// `cin` channels, y = dx with `cout` channels).  wimg: scratch of >= 9*cin*cout bf16.
int32_t k_conv3x3_bf16_fwd(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, const unet_bf16* mask, int mask_mode, unet_bf16* y,
                           int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, unet_bf16* wimg, int flip, hipStream_t s,
                           const unet_bf16* prepared) {
  if (!bf16_conv3x3_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 bf16: cin=%d cout=%d unsupported (multiples of 16)", cin, cout);
  // tile choice.  Wide layers (>= 128 output channels) are MFMA / LDS bound: 64-channel x 16-row tiles (most operand reuse, 2 workgroups
  // per CU).  32-channel layers are short-K and HBM bound: 32-channel x 8-row tiles need 40 KB of LDS and 89 registers, so 4 workgroups
  // per CU hide the load latency of each other's prologues (c1b / c9b: -15 %; 64-channel layers measured neutral to worse).  UNET_BF16_TILE = "<nb><rw>" overrides (measurements).
  int NB, RW; bf16_conv_tile(cout, &NB, &RW);
  if (!prepared) {
    int32_t r;
    if (!flip) r = make_wimg(ctx, w, wimg, 9, 1, NB, cin, cout, (long long)cin * cout, 0, cout, 1, s);
    else r = make_wimg(ctx, w, wimg, 9, 1, NB, cin, cout, (long long)cin * cout, 1, 1, cin, s);      // W(tap,k,m) = w_f[8-tap][m][k], row length = cout_f = cin here
    if (r) return r;
  } else wimg = const_cast<unet_bf16*>(prepared);          // laid out by k_wimg_multi at the start of the program
  if (NB == 2 && RW == 4) return launch_conv_bf16<0, 2, 4>(ctx, x, cin, wimg, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
  if (NB == 2) return launch_conv_bf16<0, 2, 2>(ctx, x, cin, wimg, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
  if (RW == 4) return launch_conv_bf16<0, 1, 4>(ctx, x, cin, wimg, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
  return launch_conv_bf16<0, 1, 2>(ctx, x, cin, wimg, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
}

// u[n,2i+a,2j+b,o] = bias[o] + sum_c x[n,i,j,c] * K[a,b,o,c]   (Keras ConvT kernel [2][2][cout][cin])
int32_t k_convT_bf16_fwd(unet_ctx* ctx, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int ldy, int n, int h, int wd, int cin,
                         int cout, unet_bf16* wimg, hipStream_t s) {
  if (!bf16_convT_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT bf16: cin=%d cout=%d unsupported", cin, cout);
  int32_t r = make_wimg(ctx, w, wimg, 1, 2, 2, cin, 4 * cout, 0, 0, 1, cin, s);             // W(k = c, m = ab*cout + o) = K[m*cin + c]
  if (r) return r;
  return launch_conv_bf16<1, 2>(ctx, x, cin, wimg, bias, nullptr, MASK_NONE, y, ldy, n, h, wd, cin, 4 * cout, ACT_NONE, 0.0f, 0, s);
}

// dx[n,i,j,c] = sum_{ab,o} dU[n,2i+a,2j+b,o] * K[ab,o,c]; mask: ReLU of the producer of x
int32_t k_convT_bf16_dgrad(unet_ctx* ctx, const unet_bf16* dy, int lddy, const float* w, const unet_bf16* mask, unet_bf16* dx, int n, int h, int wd,
                           int cin, int cout, unet_bf16* wimg, hipStream_t s) {
  if (!bf16_convT_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT bf16: cin=%d cout=%d unsupported", cin, cout);
  const int NB = (cin % 64) == 0 ? 2 : 1;
  int32_t r = make_wimg(ctx, w, wimg, 1, 2, NB, 4 * cout, cin, 0, 0, cin, 1, s);            // W(k = ab*cout + o, m = c) = K[k*cin + c]
  if (r) return r;
  const int mm = mask ? MASK_RELU : MASK_NONE;
  if (NB == 2) return launch_conv_bf16<2, 2>(ctx, dy, lddy, wimg, nullptr, mask, mm, dx, cin, n, h, wd, 4 * cout, cin, ACT_NONE, 0.0f, 0, s);
  return launch_conv_bf16<2, 1>(ctx, dy, lddy, wimg, nullptr, mask, mm, dx, cin, n, h, wd, 4 * cout, cin, ACT_NONE, 0.0f, 0, s);
}

namespace {

// =====================================================================================================================
// Weight gradient.  dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co]   (conv3x3, TAPS = 9, X = layer input, halo 1)
//                   dK[ab][o][c]    = sum_p dU[2p + ab][o] * X[p][c]     (convT2x2, TAPS = 4: A = dU parity planes, B = X)
// GEMM per tap: D[a_ch][b_ch] += A^T B with K = pixels; split-K over (image, 32-column strip, row chunk), partial slabs
// [split][tap][CA][CB] (+ bias sums) reduced in a fixed order by the fp32 path's reduce kernels (deterministic).
// workgroup = 4 waves = WA x WB channel tiles of 32 x 32 (all taps: 144 accumulator registers per wave) x WR row phases; a step
// stages 4 B rows (+ the 6 A rows they touch) in LDS as they come from HBM ([32-channel plane][row][pixel][32 ch]) and each
// operand (8 consecutive pixels of one channel per lane) is fetched with two ds_read_b64_tr_b16.
// =====================================================================================================================
__device__ __forceinline__ bf16x8 lds_tr_frag(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int MODE, int WA, int WB, int WR>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const unet_bf16* __restrict__ A, int ldA, const unet_bf16* __restrict__ B, int ldB,
                                                            float* __restrict__ part, int N, int H, int W, int CA, int CB, int tiles_b,
                                                            int strips, int rows_per_chunk, int chunks_per_strip, int nsplit, int npairs,
                                                            long long pstride, int units, int upb) {
  static_assert(WA * WB * WR == 4, "4 waves");
  constexpr int TAPS = MODE == 0 ? 9 : 4;
  constexpr int R = MODE == 0 ? 4 : 2;                            // B rows per step
  static_assert(WR <= R, "row phases");
  constexpr int AROWS = MODE == 0 ? R + 2 : 2 * R, APX = MODE == 0 ? 34 : 64;      // A rows / pixels per row staged per step
  constexpr int ASUB = AROWS * APX * 64, BSUB = R * 32 * 64;      // bytes of one 32-channel plane
  constexpr int APIECES = WA * AROWS * APX * 4, BPIECES = WB * R * 32 * 4;
  constexpr int AL = (APIECES + 255) / 256, BL = (BPIECES + 255) / 256;
  constexpr int RED = WR > 1 ? 2 * TAPS * 16 * 64 * 4 : 0;        // two accumulator images for the row-phase reduction
  constexpr int STAGE = WA * ASUB + WB * BSUB;
  __shared__ __attribute__((aligned(16))) char smem[STAGE > RED ? STAGE : RED];
  __shared__ float s_bs[4][64];
  char* const s_a = smem; char* const s_b = smem + WA * ASUB;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave % WR, wb = (wave / WR) % WB, wa = wave / (WR * WB);
  // XCD-aware block map (as the fp32 kernels): all channel-tile pairs of one pixel split run on the same XCD back to back
  const int sq = blockIdx.x >> 3;
  const int pair = sq % npairs, split = (sq / npairs) * 8 + (blockIdx.x & 7);
  if (split >= nsplit) return;
  const int ta = pair / tiles_b, tb = pair % tiles_b;
  const int a0 = ta * 32 * WA, b0 = tb * 32 * WB;
  // a split = one row chunk of `upb` consecutive (image, 32-column strip) units
  const int chunk = split % chunks_per_strip; const int ublk = split / chunks_per_strip;
  const int ya = chunk * rows_per_chunk;
  const int yb = ya + rows_per_chunk < H ? ya + rows_per_chunk : H;
  const int HA = MODE == 0 ? H : 2 * H, WA_ = MODE == 0 ? W : 2 * W;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum = 0.0f;

  // transpose-read addressing: 16-lane group g4 reads [4 pixels][16 channels]; lane i -> pixel i>>2, channel quad i&3
  const int i16 = lane & 15, g4 = lane >> 4;
  const int tr_px = (g4 >> 1) * 8 + (i16 >> 2), tr_ch = ((g4 & 1) * 16 + (i16 & 3) * 4) * 2;
  const char* const pa = s_a + wa * ASUB + tr_ch;
  const char* const pb = s_b + wb * BSUB + tr_ch;

  const int u1 = (ublk + 1) * upb < units ? (ublk + 1) * upb : units;
  for (int unit = ublk * upb; unit < u1; ++unit) {
    const int cs = unit % strips, n = unit / strips;
    const int x0 = cs * 32;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(A + (long long)n * HA * WA_ * ldA), 0, (int)((long long)HA * WA_ * ldA * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(B + (long long)n * H * W * ldB), 0, (int)((long long)H * W * ldB * 2), 0x00020000);
    // staging plan, one register per piece: byte offset of the piece for staged row 0 (a multiple of 16) | staged row (bits 0-2) |
    // invalid column / channel (bit 3)
    int aoff[AL], boff[BL];
#pragma unroll
    for (int k = 0; k < AL; ++k) {
      const int idx = tid + k * 256;
      const int q = idx & 3; int r = idx >> 2;
      const int px = r % APX; r /= APX;
      const int row = r % AROWS, sub = r / AROWS;
      const int gx = MODE == 0 ? x0 + px - 1 : 2 * x0 + px;
      const int ch = a0 + sub * 32 + q * 8;
      const bool ok = idx < APIECES && gx >= 0 && gx < WA_ && ch < CA;
      aoff[k] = ok ? (((row * WA_ + gx) * ldA * 2 + ch * 2) | row) : 8;
    }
#pragma unroll
    for (int k = 0; k < BL; ++k) {
      const int idx = tid + k * 256;
      const int q = idx & 3; int r = idx >> 2;
      const int px = r & 31; r >>= 5;
      const int row = r % R, sub = r / R;
      const int gx = x0 + px, ch = b0 + sub * 32 + q * 8;
      const bool ok = idx < BPIECES && gx < W && ch < CB;
      boff[k] = ok ? (((row * W + gx) * ldB * 2 + ch * 2) | row) : 8;
    }
    unet_u32x4 areg[AL], breg[BL];
    auto issue_loads = [&](int ys) __attribute__((always_inline)) {        // ys = first B row of the step
      const int ysa = MODE == 0 ? ys - 1 : 2 * ys;                       // image row of staged A row 0 (conv3x3: the halo row above)
      const int abase = ysa * WA_ * ldA * 2, bbase = ys * W * ldB * 2;
#pragma unroll
      for (int k = 0; k < AL; ++k) {
        const int v = aoff[k], gy = ysa + (v & 7);
        const bool ok = !(v & 8) && gy >= 0 && gy < HA;
        areg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? (v & ~15) + abase : UNET_OOB, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < BL; ++k) {
        const int v = boff[k], gy = ys + (v & 7);
        breg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (!(v & 8) && gy < yb) ? (v & ~15) + bbase : UNET_OOB, 0, 0);   // rows past the chunk contribute 0
      }
    };
    auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < AL; ++k) { const int idx = tid + k * 256; if (idx < APIECES) *reinterpret_cast<unet_u32x4*>(s_a + idx * 16) = areg[k]; }    // piece order == LDS order
#pragma unroll
      for (int k = 0; k < BL; ++k) { const int idx = tid + k * 256; if (idx < BPIECES) *reinterpret_cast<unet_u32x4*>(s_b + idx * 16) = breg[k]; }
    };

    issue_loads(ya);
    store_lds();
    __syncthreads();
    for (int ys = ya; ys < yb; ys += R) {
      const bool more = ys + R < yb;
      if (more) issue_loads(ys + R);
#pragma unroll
      for (int r = wr; r < R; r += WR) {
#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
          const char* bp = pb + (r * 32 + kst * 16 + tr_px) * 64;
          const bf16x8 bf = lds_tr_frag(bp, bp + 4 * 64);
          if (MODE == 0 && wa == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += (float)bf[j];
          }
          if (MODE == 0) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
              for (int kx = 0; kx < 3; ++kx) {
                const char* ap = pa + ((r + ky) * APX + kst * 16 + tr_px + kx) * 64;
                const bf16x8 af = lds_tr_frag(ap, ap + 4 * 64);
                acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[ky * 3 + kx], 0, 0, 0);
              }
          } else {
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {          // A = dU: pixel (2y + a, 2x + b); consecutive k = consecutive x -> stride 2 pixels
              const char* ap = pa + ((2 * r + (ab >> 1)) * APX + 2 * (kst * 16 + tr_px) + (ab & 1)) * 64;
              const bf16x8 af = lds_tr_frag(ap, ap + 8 * 64);
              if (wb == 0) {                          // bias gradient of the ConvT = sum of dU over all four parity planes
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum += (float)af[j];
              }
              acc[ab] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[ab], 0, 0, 0);
            }
          }
        }
      }
      __syncthreads();
      if (more) { store_lds(); __syncthreads(); }
    }
  }

  // ---- row phases of one channel tile are summed inside the workgroup (fixed order, through LDS): one slab per split
  bsum += __shfl_xor(bsum, 32, 64);
  if (WR > 1) {
    float* red = reinterpret_cast<float*>(smem);
    const int grp = wave / WR;                                      // (wa, wb) tile of this wave; WR/2 * (4/WR) = 2 images at most
    s_bs[wave][lane] = bsum;
#pragma unroll
    for (int stride = WR / 2; stride >= 1; stride >>= 1) {
      float* img = red + (size_t)(grp * stride + (wr % stride)) * (TAPS * 16 * 64);
      if (wr >= stride && wr < 2 * stride) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) img[(t * 16 + r) * 64 + lane] = acc[t][r];
      }
      __syncthreads();
      if (wr < stride) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += img[(t * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (wr == 0) { bsum = s_bs[wave][lane]; for (int k = 1; k < WR; ++k) bsum += s_bs[wave + k][lane]; }
  }
  if (wr != 0) return;
  float* P = part + (long long)split * pstride;
  const int ar = a0 + wa * 32, bc = b0 + wb * 32 + l31;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (ar + m < CA && bc < CB) P[((long long)t * CA + ar + m) * CB + bc] = acc[t][r];
    }
  if (MODE == 0) { if (wa == 0 && ta == 0 && lane < 32 && bc < CB) P[(long long)TAPS * CA * CB + bc] = bsum; }
  else if (wb == 0 && tb == 0 && lane < 32 && ar + l31 < CA) P[(long long)TAPS * CA * CB + ar + l31] = bsum;
}

struct WgPlan { int WA, WB, WR, tiles_a, tiles_b, strips, rows_per_chunk, chunks_per_strip, nsplit, nslabs, units, upb; size_t floats; };

WgPlan plan_wgrad_bf16(int mode, int n, int h, int w, int ca, int cb) {
  WgPlan p;
  const int R = mode == 0 ? 4 : 2, taps = mode == 0 ? 9 : 4, cbias = mode == 0 ? cb : ca;
  p.WA = (ca % 64) == 0 ? 2 : 1; p.WB = (cb % 64) == 0 ? 2 : 1;
  p.WR = 4 / (p.WA * p.WB); if (p.WR > R) { p.WB = 2; p.WR = 4 / (p.WA * p.WB); }      // (a 64-wide B tile may overhang cb: masked)
  p.tiles_a = (ca + 32 * p.WA - 1) / (32 * p.WA); p.tiles_b = (cb + 32 * p.WB - 1) / (32 * p.WB); p.strips = (w + 31) / 32;
  const long long pairs = (long long)p.tiles_a * p.tiles_b, per = (long long)taps * ca * cb;
  const long long units = (long long)n * p.strips;
  constexpr long long target = 512;   // 256 CUs x 2 resident workgroups (1024 measured 15 % slower: twice the partial-slab traffic)
  long long want = std::max<long long>(1, target / pairs);          // pixel splits wanted
  const long long cap = std::max<long long>(1, (64LL << 20) / per);
  want = std::min(want, cap);
  // fewer splits than units: a workgroup walks `upb` consecutive units (halves / quarters the partial-slab traffic of the deep layers,
  // whose slabs are megabytes); more: the rows of a unit are cut into chunks
  p.units = (int)units; p.upb = (int)std::max<long long>(1, units / want);
  const long long ublocks = (units + p.upb - 1) / p.upb;
  long long cps = std::max<long long>(1, want / ublocks);
  cps = std::min<long long>(cps, std::max<long long>(1, h / 8));
  int rpc = (int)((h + cps - 1) / cps); rpc = (rpc + R - 1) / R * R;                // whole steps
  p.rows_per_chunk = rpc; p.chunks_per_strip = (h + rpc - 1) / rpc;
  p.nsplit = (int)(ublocks * p.chunks_per_strip); p.nslabs = p.nsplit;
  p.floats = (size_t)p.nslabs * (per + cbias) + wgrad_reduce_scratch_floats(taps, ca, cb, cbias, p.nslabs);
  return p;
}

template <int MODE>
int32_t run_wgrad_bf16(unet_ctx* ctx, const unet_bf16* A, int ldA, const unet_bf16* B, int ldB, float* dw, float* db, void* ws, size_t ws_bytes, int n,
                       int h, int w, int ca, int cb, hipStream_t s) {
  const int taps = MODE == 0 ? 9 : 4, cbias = MODE == 0 ? cb : ca;
  if ((long long)(MODE == 1 ? 4 : 1) * h * w * ldA * 2 >= (1LL << 30) || (long long)h * w * ldB * 2 >= (1LL << 30))
    UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad bf16: one image must stay below 1 GiB (32-bit buffer offsets)");
  const WgPlan p = plan_wgrad_bf16(MODE, n, h, w, ca, cb);
  if (!ws || ws_bytes < p.floats * sizeof(float)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad bf16: workspace %zu < %zu bytes", ws_bytes, p.floats * sizeof(float));
  float* part = static_cast<float*>(ws);
  const long long S = (long long)taps * ca * cb + cbias;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));
#define UNET_WG(WA_, WB_, WR_) hipLaunchKernelGGL((wgrad_bf16_kernel<MODE, WA_, WB_, WR_>), grid, dim3(256), 0, s, A, ldA, B, ldB, part, n, h, w, ca, cb, \
                                                  p.tiles_b, p.strips, p.rows_per_chunk, p.chunks_per_strip, p.nsplit, npairs, S, p.units, p.upb)
  if (p.WA == 2 && p.WB == 2) UNET_WG(2, 2, 1);
  else if (p.WA == 2) UNET_WG(2, 1, 2);
  else if (p.WB == 2) UNET_WG(1, 2, 2);
  else { if constexpr (MODE == 0) UNET_WG(1, 1, 4); else UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad bf16: internal plan error"); }
#undef UNET_WG
  UNET_CHECK_LAUNCH(ctx, "wgrad_bf16");
  // a wave's bias lanes are only written by the (ta == 0 / tb == 0) tiles; every slab position is written by exactly one wave
  return k_wgrad_reduce(ctx, part, p.nslabs, taps, ca, cb, cbias, dw, db, s);
}

}  // namespace

bool bf16_wgrad_supported(int ca, int cb) { return ca >= 16 && (ca % 8) == 0 && cb >= 16 && (cb % 8) == 0; }      // 32-wide tiles, overhang masked per 8-channel piece
// 16 -> 16 channels with an even width: the 32 -> 32 problem on pixel pairs (kernels_wgrad_h2.hip: k_conv3x3_h2_wgrad_c16), its 9 x 32 x 32 + 32 results behind the slabs
static bool bf16_wgrad_c16(int wd, int cin, int cout) { return cin == 16 && cout == 16 && wd >= 2 && (wd & 1) == 0; }
static size_t bf16_wgrad_c16_inner(int n, int h, int wd) { return plan_wgrad_bf16(0, n, h, wd / 2, 32, 32).floats * sizeof(float); }
size_t bf16_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) {
  if (!bf16_wgrad_supported(cin, cout)) return 0;
  const size_t plain = plan_wgrad_bf16(0, n, h, wd, cin, cout).floats * sizeof(float);
  return bf16_wgrad_c16(wd, cin, cout) ? std::max(plain, bf16_wgrad_c16_inner(n, h, wd) + (9 * 32 * 32 + 32) * sizeof(float)) : plain;
}
size_t bf16_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) { return bf16_convT_supported(cin, cout) ? plan_wgrad_bf16(1, n, h, wd, cout, cin).floats * sizeof(float) : 0; }

int32_t k_conv3x3_bf16_wgrad(unet_ctx* ctx, const unet_bf16* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd,
                             int cin, int cout, hipStream_t s) {
  if (!bf16_wgrad_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad bf16: cin=%d cout=%d unsupported (multiples of 8, >= 16)", cin, cout);
  if (bf16_wgrad_c16(wd, cin, cout) && ws && ws_bytes >= bf16_wgrad_c16_inner(n, h, wd) + (9 * 32 * 32 + 32) * sizeof(float)) {
    const size_t inner = bf16_wgrad_c16_inner(n, h, wd);
    float* G = reinterpret_cast<float*>(static_cast<char*>(ws) + inner);
    int32_t r = run_wgrad_bf16<0>(ctx, x, 32, dy, 32, G, G + 9 * 32 * 32, ws, inner, n, h, wd / 2, 32, 32, s);
    return r ? r : k_wgrad_c16_gather(ctx, G, dw, db, s);
  }
  return run_wgrad_bf16<0>(ctx, x, cin, dy, cout, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);
}

// convT: A = dU (channels = cout, pixel stride lddy, 2h x 2w), B = x (channels = cin, h x w); dK is [4][cout][cin]
int32_t k_convT_bf16_wgrad(unet_ctx* ctx, const unet_bf16* x, const unet_bf16* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                           int wd, int cin, int cout, hipStream_t s) {
  if (!bf16_convT_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT wgrad bf16: cin=%d cout=%d unsupported", cin, cout);
  return run_wgrad_bf16<1>(ctx, dy, lddy, x, cin, dw, db, ws, ws_bytes, n, h, wd, cout, cin, s);
}
