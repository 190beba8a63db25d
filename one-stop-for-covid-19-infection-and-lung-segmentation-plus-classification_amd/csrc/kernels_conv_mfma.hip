// The STRICT fp32 kernel family (UNET_ALGO_MFMA; also what UNET_ALGO_AUTO falls back to for channel counts the fp16-split h2
// kernels do not take): fp32 implicit-GEMM 3x3 convolution / ConvT / weight gradients on the CDNA4 matrix cores with
// v_mfma_f32_32x32x2_f32 -- exact fp32 multiply-add (bitwise an fmaf chain), 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 peak; there is
// no TF32 on gfx950.  The out-of-domain fallback of kernels_conv_h2.hip and the on-device cross-check of its accuracy claims.
//
// Forward / data-gradient (same kernel, the data gradient passes flipped+transposed weights):
//   GEMM view  M = output pixels, N = Cout, K = 9*Cin.
//   block  = 256 threads = 4 waves, output tile = TH rows x 32 columns x TN channels.
//   an MFMA M-tile = 32 consecutive pixels of one image row, so the A operand of tap (dr,dc) is
//   the same LDS patch read at a shifted pixel offset: each input element is fetched from
//   global once per block and reused 9x (taps) x TN/32 (N tiles) from LDS.
//   K loop: chunks of CK=8 input channels.  LDS holds the (TH+2)x34 halo patch of the chunk
//   ([pixel][12] floats: 8 channels + 4 pad -> conflict-free ds_read_b128) and the 9x8xTN weight
//   slab ([tap][cin][TN]: conflict-free ds_read_b32 along cout).
//   k-pairing trick: MFMA k=0 lanes (0-31) carry channel j, k=1 lanes (32-63) carry channel 4+j,
//   so ONE ds_read_b128 per lane feeds the A operand of 4 consecutive MFMAs.
//   Epilogue: +bias, ReLU, optional ReLU-mask multiply (backward), coalesced 128-B row stores.
//
// Weight-gradient: see wgrad_mfma_kernel below (split-K over pixels, deterministic 2-stage).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 8;     // input channels per K chunk
constexpr int CKP = 12;   // padded pixel stride in LDS (floats): 4*odd -> ds_read_b128 conflict-free
constexpr int PW = 34;    // patch width: 32 + halo

// MODE 0: conv3x3 'same' (forward, and data-gradient with flipped/transposed weights)
// MODE 1: convT2x2s2 forward  = 1x1 GEMM [pixels,Cin] x [Cin, 4*Cout] with a scatter epilogue into the
//         (2i+a, 2j+b) positions of a channel slice (pixel stride ldy) of the concat buffer; the
//         Keras [2,2,Cout,Cin] kernel is transposed on the fly while staging the weight slab
// MODE 2: convT2x2s2 data-gradient = 2x2 stride-2 'valid' convolution of dU (pixel stride ldx):
//         the 2x-upsampled patch is de-interleaved by column parity in LDS so that the A operand of
//         tap (a,b) is again 32 consecutive pixels (conflict-free ds_read_b128)
// GEN = general epilogue (ELU, fused dropout, ELU-derivative masks: the U-Net++ graph); the plain one (ReLU / ReLU mask) keeps
// the U-Net instances free of the Philox + expm1 code
template <int MODE, int TN, int TH, int WR, int WC, bool PF, bool GEN, int ABL = 0>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ mask,
                                                        float* __restrict__ y, int ldy, int N, int H, int W, int Cin, int Cout,
                                                        int act, int mask_mode, float rate, unsigned long long seed, int tiles_x,
                                                        int tiles_y) {
  static_assert(WR * WC == 4, "4 waves");
  constexpr int TAPS = MODE == 0 ? 9 : (MODE == 1 ? 1 : 4);
  constexpr int RW = TH / WR;          // image rows (M tiles) per wave
  constexpr int NW = TN / 32 / WC;     // N tiles per wave
  constexpr int NPIX = MODE == 0 ? (TH + 2) * PW : (MODE == 1 ? TH * 32 : 2 * TH * 64);
  constexpr int PTOT = NPIX * 2;                                        // float4 per patch chunk
  constexpr int WTOT = MODE == 1 ? TN * 2 : TAPS * CK * (TN / 4);       // float4 per weight slab
  constexpr int PL = (PTOT + 255) / 256, WL = (WTOT + 255) / 256;       // per-thread staging registers
  static_assert(RW >= 1 && NW >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float s_in[NPIX * CKP];
  __shared__ __attribute__((aligned(16))) float s_w[TAPS * CK * TN];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform, and the compiler is told so (scalar address math)
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; const int n = b / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int nbase = blockIdx.y * TN;
  const int HI = MODE == 2 ? 2 * H : H, WI = MODE == 2 ? 2 * W : W;     // input pixel grid

  f32x16 acc[RW][NW];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- per-thread staging plan, hoisted out of the K loop: element offsets (-1 = zero fill) and LDS slots
  const float* xn = x + (long long)n * HI * WI * ldx;
  int poff[PL], plds[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = tid + k * 256;
    const int q = idx & 1, pix = idx >> 1;
    int gy, gx, lp;
    if (MODE == 0) { const int r = pix / PW, c = pix - r * PW; gy = y0 + r - 1; gx = x0 + c - 1; lp = pix; }
    else if (MODE == 1) { gy = y0 + (pix >> 5); gx = x0 + (pix & 31); lp = pix; }
    else { const int rr = pix >> 6, cc = pix & 63; gy = 2 * y0 + rr; gx = 2 * x0 + cc; lp = (rr * 2 + (cc & 1)) * 32 + (cc >> 1); }
    const bool ok = idx < PTOT && gy >= 0 && gy < HI && gx >= 0 && gx < WI;
    poff[k] = ok ? ((gy * WI + gx) * ldx + q * 4) * 4 : UNET_OOB;           // byte offset into image n; halo -> out of range -> 0
    plds[k] = idx < PTOT ? lp * CKP + q * 4 : -1;
  }
  int woff[WL], wlds[WL];
#pragma unroll
  for (int k = 0; k < WL; ++k) {
    const int idx = tid + k * 256;
    if (MODE == 1) {
      const int half = idx & 1, nn = idx >> 1;
      woff[k] = (idx < WTOT && nbase + nn < Cout) ? ((nbase + nn) * Cin + half * 4) * 4 : UNET_OOB;     // Keras ConvT kernel [ab][o][c]: row nn holds Cin floats
      wlds[k] = half * 4 * TN + nn;
    } else {
      const int q = idx % (TN / 4), row = idx / (TN / 4);
      const int tap = row >> 3, ci = row & 7;
      woff[k] = (idx < WTOT && nbase + q * 4 < Cout) ? ((tap * Cin + ci) * Cout + nbase + q * 4) * 4 : UNET_OOB;   // tile may overhang Cout (zero fill)
      wlds[k] = row * TN + q * 4;
    }
  }
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xn, (long long)HI * WI * ldx * 4);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(w, (long long)TAPS * Cin * Cout * 4);      // MODE 1: Cout = 4 x ConvT channels, TAPS = 1
  f32x4 preg[PL], wreg[WL];
  auto issue_loads = [&](int c0) __attribute__((always_inline)) {                 // all global loads of a chunk in flight before any wait;
#pragma unroll                                                                    // branch-free: invalid lanes read out of range -> 0
    for (int k = 0; k < PL; ++k) preg[k] = buf_ld4(rs_x, poff[k], c0 * 4);
#pragma unroll
    for (int k = 0; k < WL; ++k) wreg[k] = buf_ld4(rs_w, woff[k], (MODE == 1 ? c0 : c0 * Cout) * 4);
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k)
      if (plds[k] >= 0) *reinterpret_cast<f32x4*>(&s_in[plds[k]]) = preg[k];
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      if (tid + k * 256 >= WTOT) continue;
      if (MODE == 1) {            // transpose while staging: s_w[ci][nn]
        s_w[wlds[k]] = wreg[k][0]; s_w[wlds[k] + TN] = wreg[k][1]; s_w[wlds[k] + 2 * TN] = wreg[k][2]; s_w[wlds[k] + 3 * TN] = wreg[k][3];
      } else {
        *reinterpret_cast<f32x4*>(&s_w[wlds[k]]) = wreg[k];
      }
    }
  };

  // ABL (timing ablations only, results are wrong): bit0 = no LDS operand reads, bit1 = stage only the first chunk
  if (PF) issue_loads(0);
  f32x4 a_fix[RW]; float b_fix[NW];
  if (ABL & 1) {
#pragma unroll
    for (int i = 0; i < RW; ++i) a_fix[i] = (f32x4){1.f + tid, 2.f, 3.f, 4.f};
#pragma unroll
    for (int jn = 0; jn < NW; ++jn) b_fix[jn] = 0.5f + lane;
  }
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    if (!(ABL & 2) || c0 == 0) {
      if (!PF) issue_loads(c0);
      store_lds();
    }
    __syncthreads();
    if (PF && c0 + CK < Cin && !(ABL & 2)) issue_loads(c0 + CK);      // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      f32x4 a[RW];
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        int lp;
        if (MODE == 0) lp = (wr * RW + i + tap / 3) * PW + l31 + tap % 3;
        else if (MODE == 1) lp = (wr * RW + i) * 32 + l31;
        else lp = ((2 * (wr * RW + i) + (tap >> 1)) * 2 + (tap & 1)) * 32 + l31;
        if (ABL & 1) { a[i] = a_fix[i]; asm volatile("" : "+v"(a[i])); }
        else a[i] = *reinterpret_cast<const f32x4*>(&s_in[lp * CKP + hi * 4]);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bv[NW];
#pragma unroll
        for (int jn = 0; jn < NW; ++jn) {
          if (ABL & 1) { bv[jn] = b_fix[jn]; asm volatile("" : "+v"(bv[jn])); }
          else bv[jn] = s_w[(tap * CK + j + 4 * hi) * TN + (wc * NW + jn) * 32 + l31];
        }
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
          for (int jn = 0; jn < NW; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], bv[jn], acc[i][jn], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue.  MFMA D layout: lane (l31, hi), register r holds D[pixel (r&3)+8*(r>>2)+4*hi][cout l31].  A 4x4 transpose
  // inside each lane quad (two xor-shuffle butterfly stages) gives every lane 4 CONSECUTIVE couts of ONE pixel, so the tile
  // leaves as 16-byte stores (1 KiB per wave instruction): 4x fewer store (and ReLU-mask load) instructions than the natural
  // one-dword-per-lane form -- the store tail of a tile is instruction-issue bound (cdna guide T21), not bandwidth bound.
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
#pragma unroll
  for (int jn = 0; jn < NW; ++jn) {
    const int co = nbase + (wc * NW + jn) * 32 + q4;         // first of this lane's 4 output channels
    const int cq = Cout >> 2;                                // MODE 1: Cout = 4 * (ConvT output channels)
    const int ab = MODE == 1 ? co / cq : 0, oc = MODE == 1 ? co - ab * cq : co;
    const float4 bb = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + oc) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int py = y0 + wr * RW + i;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = acc[i][jn][4 * g + 0], v1 = acc[i][jn][4 * g + 1], v2 = acc[i][jn][4 * g + 2], v3 = acc[i][jn][4 * g + 3];
        {   // stage 1: partner lane ^ 1, register pairs (v0,v1) (v2,v3)
          const float s01 = odd1 ? v0 : v1, s23 = odd1 ? v2 : v3;
          const float r01 = __shfl_xor(s01, 1, 64), r23 = __shfl_xor(s23, 1, 64);
          if (odd1) { v0 = r01; v2 = r23; } else { v1 = r01; v3 = r23; }
        }
        {   // stage 2: partner lane ^ 2, register pairs (v0,v2) (v1,v3)
          const float s02 = odd2 ? v0 : v2, s13 = odd2 ? v1 : v3;
          const float r02 = __shfl_xor(s02, 2, 64), r13 = __shfl_xor(s13, 2, 64);
          if (odd2) { v0 = r02; v1 = r13; } else { v2 = r02; v3 = r13; }
        }
        // now (v0..v3) = D[pixel e + 8g + 4hi][couts q4 .. q4+3]
        const int px = x0 + e + 8 * g + 4 * hi;
        if (py >= H || px >= W || co >= Cout) continue;
        float4 v = make_float4(v0 + bb.x, v1 + bb.y, v2 + bb.z, v3 + bb.w);
        if (MODE == 1) {
          *reinterpret_cast<float4*>(y + (((long long)n * 2 * H + 2 * py + (ab >> 1)) * (2 * W) + 2 * px + (ab & 1)) * ldy + oc) = v;
        } else {
          const long long o = (((long long)n * H + py) * W + px) * Cout + co;
          if (!GEN) {
            if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (mask_mode == MASK_RELU) {
              const float4 m = *reinterpret_cast<const float4*>(mask + o);
              v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
            }
            *reinterpret_cast<float4*>(y + o) = v;
            continue;
          }
          v.x = apply_act(v.x, act); v.y = apply_act(v.y, act); v.z = apply_act(v.z, act); v.w = apply_act(v.w, act);
          if (mask_mode == MASK_NONE) {
            if (rate > 0.0f) {                        // fused inverted dropout on the output (forward, Keras Dropout after the conv)
              const float4 ks = keep_scale(o >> 2, rate, seed);
              v.x *= ks.x; v.y *= ks.y; v.z *= ks.z; v.w *= ks.w;
            }
          } else {                                    // backward: derivative of the activation (+dropout) that produced `mask`
            const float4 m = *reinterpret_cast<const float4*>(mask + o);
            float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
            if (mask_mode == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
            v.x *= mask_factor(m.x, mask_mode, ks.x, rate); v.y *= mask_factor(m.y, mask_mode, ks.y, rate);
            v.z *= mask_factor(m.z, mask_mode, ks.z, rate); v.w *= mask_factor(m.w, mask_mode, ks.w, rate);
          }
          *reinterpret_cast<float4*>(y + o) = v;
        }
      }
    }
  }
}


template <int MODE, int TN, int TH, int WR, int WC>
int32_t launch_conv(unet_ctx* ctx, const float* x, int ldx, const float* w, const float* bias, const float* mask, int mask_mode, float* y,
                    int ldy, int n, int h, int wd, int cin, int cout, int act, float rate, unsigned long long seed, hipStream_t s) {
  if (!mask) mask_mode = MASK_NONE;
  if ((long long)(MODE == 2 ? 4 : 1) * h * wd * ldx * 4 >= (1LL << 30) || (long long)(MODE == 1 ? 1 : 9) * cin * cout * 4 >= (1LL << 30))
    UNET_FAIL(ctx, UNET_E_SHAPE, "conv mfma: one image / the weight tensor must stay below 1 GiB (32-bit buffer offsets); use UNET_ALGO_NAIVE");
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH;
  dim3 grid((unsigned)(tiles_x * tiles_y * n), (unsigned)((cout + TN - 1) / TN));
  {
    // register prefetch of the next chunk: always (in isolation +7 % on the 128-wide tile and -5 % on the 32-wide one, where it costs a wave of
    // occupancy; inside the training step "always" measured +0.7 % over "128-wide only")
    const bool pf = true;
    const bool gen = MODE == 0 && (act == ACT_ELU || rate > 0.0f || mask_mode >= MASK_ELU);
#define UNET_LAUNCH_CONV(PF_, GEN_)                                                                                                  \
  hipLaunchKernelGGL((conv_mfma_kernel<MODE, TN, TH, WR, WC, PF_, GEN_>), grid, dim3(256), 0, s, x, ldx, w, bias, mask, y, ldy, n, h, wd, \
                     cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y)
    if (MODE == 0 && gen) { if (pf) UNET_LAUNCH_CONV(true, (MODE == 0)); else UNET_LAUNCH_CONV(false, (MODE == 0)); }
    else { if (pf) UNET_LAUNCH_CONV(true, false); else UNET_LAUNCH_CONV(false, false); }
#undef UNET_LAUNCH_CONV
  }
  UNET_CHECK_LAUNCH(ctx, "conv_mfma");
  return UNET_OK;
}

// =========================================================================================
// Weight gradient on the matrix cores.
//   conv3x3 (MODE 0): dW[tap][ci][co] = sum_p X[p+tap][ci] * dY[p][co]       A = X,  B = dY
//   convT2x2 (MODE 1): dK[ab][o][c]   = sum_p dU[2i+a,2j+b][o] * X[i,j][c]   A = dU, B = X
// GEMM view per tap: D[a_ch][b_ch] += A^T B with K = pixels (millions) and a tiny M x N, so the
// work is split over K: ONE WAVE per workgroup owns one 32x32 (a_ch, b_ch) tile for ALL taps
// (9 x 16 = 144 accumulator registers -> 3 waves/SIMD) and walks a range of image rows of a
// 32-pixel-wide column strip.  MFMA k = 2 consecutive pixels of the row (lanes 0-31 / 32-63), lanes
// run along channels, so every LDS read is 32 consecutive floats (conflict-free) and every global
// read is a full 128-B line of an NHWC pixel.  conv3x3 keeps a 3-row ring of X rows in LDS: each
// row is fetched once per strip (no halo re-reads along y).  The B operand (dY) is shared by the
// 9 taps.  Partials go to a workspace [split][tap][a_ch][b_ch] and a second kernel sums the
// splits in a fixed order (deterministic, no float atomics).  The bias gradient rides along as
// per-lane column sums of the dY / dU values the wave already holds.
// =========================================================================================
template <int MODE>
__global__ __launch_bounds__(64, 2) void wgrad_mfma_kernel(const float* __restrict__ A, int ldA, const float* __restrict__ B, int ldB,
                                                        float* __restrict__ part, float* __restrict__ part_b, int N, int H, int W,
                                                        int CA, int CB, int tiles_b, int strips, int rows_per_chunk,
                                                        int chunks_per_strip, int nsplit, int tiles_a_x_b, long long pstride) {
  static_assert(MODE == 0 || MODE == 1, "conv3x3 / ConvT");
  constexpr int TAPS = MODE == 0 ? 9 : 4;
  constexpr int ROWF = 34 * 32;                    // floats per ring row (conv3x3)
  constexpr int AL = MODE != 1 ? 5 : 16, BL = 4;   // float4 staging registers per lane (A rows, B row)
  __shared__ __attribute__((aligned(16))) float s_a[MODE != 1 ? 3 * ROWF : 4 * 32 * 32];
  __shared__ __attribute__((aligned(16))) float s_b[32 * 32];
  const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
  // XCD-aware block map (workgroup L runs on XCD L % 8): all (a, b) channel-tile pairs of one pixel split go to the SAME XCD back to
  // back, so the X / dY rows of that split are fetched into one L2 once instead of once per pair on eight different L2s.
  const int npairs = tiles_a_x_b, sq = blockIdx.x >> 3;
  const int pair = sq % npairs, split = (sq / npairs) * 8 + (blockIdx.x & 7);
  if (split >= nsplit) return;                                  // grid padded to 8 * ceil(nsplit / 8) * npairs workgroups
  const int ta = pair / tiles_b, tb = pair % tiles_b;
  const int a0 = ta * 32, b0 = tb * 32;
  const int chunk = split % chunks_per_strip; const int t2 = split / chunks_per_strip;
  const int cs = t2 % strips, n = t2 / strips;
  const int x0 = cs * 32;
  const int ya = chunk * rows_per_chunk;
  const int yb = ya + rows_per_chunk < H ? ya + rows_per_chunk : H;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum = 0.0f;

  const int HA = MODE != 1 ? H : 2 * H, WA = MODE != 1 ? W : 2 * W;
  const float* An = A + (long long)n * HA * WA * ldA;
  const float* Bn = B + (long long)n * H * W * ldB;

  // per-lane staging plan (column part of the address; -1 = zero fill) hoisted out of the row loop
  int aoff[AL], alds[AL], boff[BL], blds[BL];
#pragma unroll
  for (int k = 0; k < AL; ++k) {
    const int idx = lane + 64 * k;
    if (MODE != 1) {
      const int pix = idx >> 3, q = idx & 7; const int gx = x0 - 1 + pix;
      aoff[k] = (idx < 34 * 8 && gx >= 0 && gx < W && a0 + q * 4 < CA) ? (gx * ldA + a0 + q * 4) * 4 : UNET_COL_OOB;
      alds[k] = idx < 34 * 8 ? pix * 32 + q * 4 : -1;
    } else {
      const int q = idx & 7; const int cc = (idx >> 3) & 63; const int a = idx >> 9;   // a = row parity of dU
      const int gx = 2 * x0 + cc;
      aoff[k] = (gx < WA && a0 + q * 4 < CA) ? ((a * WA + gx) * ldA + a0 + q * 4) * 4 : UNET_COL_OOB;
      alds[k] = (((a * 2 + (cc & 1)) * 32) + (cc >> 1)) * 32 + q * 4;
    }
  }
#pragma unroll
  for (int k = 0; k < BL; ++k) {
    const int idx = lane + 64 * k; const int pix = idx >> 3, q = idx & 7; const int gx = x0 + pix;
    boff[k] = (gx < W && b0 + q * 4 < CB) ? (gx * ldB + b0 + q * 4) * 4 : UNET_COL_OOB;
    blds[k] = pix * 32 + q * 4;
  }
  f32x4 areg[AL], breg[BL];            // ext-vector type: stays in registers (a float4 struct select goes through scratch)
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(An, (long long)HA * WA * ldA * 4), rs_b = make_rsrc(Bn, (long long)H * W * ldB * 4);
  // MODE 0 step t handles ring row yy = ya-1+t (X row yy and dY row yy-1); MODE 1 step t handles row y = ya+t.
  // Branch-free: rows outside the image / the chunk and columns outside the strip carry out-of-range byte offsets -> 0.
  const int nsteps = MODE != 1 ? (yb - ya + 2) : (yb - ya);
  auto issue = [&](int t) __attribute__((always_inline)) {
    if (MODE != 1) {
      const int yy = ya - 1 + t, yd = yy - 1;
      const int ra = (yy >= 0 && yy < H) ? yy * W * ldA * 4 : UNET_OOB, rb = (yd >= ya && yd < yb) ? yd * W * ldB * 4 : UNET_OOB;
#pragma unroll
      for (int k = 0; k < AL; ++k) areg[k] = buf_ld4(rs_a, aoff[k] + ra);
#pragma unroll
      for (int k = 0; k < BL; ++k) breg[k] = buf_ld4(rs_b, boff[k] + rb);
    } else {
      const int yrow = ya + t;
      const int ra = 2 * yrow * WA * ldA * 4, rb = yrow * W * ldB * 4;
#pragma unroll
      for (int k = 0; k < AL; ++k) areg[k] = buf_ld4(rs_a, aoff[k] + ra);
#pragma unroll
      for (int k = 0; k < BL; ++k) breg[k] = buf_ld4(rs_b, boff[k] + rb);
    }
  };

  issue(0);
  for (int t = 0; t < nsteps; ++t) {
    const int yy = ya - 1 + t;                                  // conv3x3 only
    const int slot_w = MODE != 1 ? ((yy + 3) % 3) * ROWF : 0;
#pragma unroll
    for (int k = 0; k < AL; ++k)
      if (alds[k] >= 0) *reinterpret_cast<f32x4*>(&s_a[slot_w + alds[k]]) = areg[k];
#pragma unroll
    for (int k = 0; k < BL; ++k) *reinterpret_cast<f32x4*>(&s_b[blds[k]]) = breg[k];
    __syncthreads();
    if (t + 1 < nsteps) issue(t + 1);                            // next row's loads fly under this row's MFMAs
    if (MODE == 1 || t >= 2) {
      int slot_off[3];
#pragma unroll
      for (int dr = 0; dr < 3; ++dr) slot_off[dr] = ((yy - 2 + dr + 3) % 3) * ROWF;   // rows y-1, y, y+1 with y = yy-1
#pragma unroll 2
      for (int pp = 0; pp < 16; ++pp) {
        const int c = 2 * pp + hi;
        const float bv = s_b[c * 32 + l31];
        if (MODE == 0) bsum += bv;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          float av;
          if (MODE == 0) av = s_a[slot_off[tp / 3] + (c + tp % 3) * 32 + l31];
          else { av = s_a[(tp * 32 + c) * 32 + l31]; bsum += av; }
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tp], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // partial tile: rows = a channel (r&3)+8*(r>>2)+4*hi, cols = b channel l31.  Same quad transpose as the conv epilogue:
  // each lane ends up with 4 consecutive b channels of one a channel -> 16-byte stores (36 instead of 144 per wave).
  float* P = part + (long long)split * pstride;                 // partials of one split: [TAPS][CA][CB] weights, then the bias sums
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v0 = acc[t][4 * g + 0], v1 = acc[t][4 * g + 1], v2 = acc[t][4 * g + 2], v3 = acc[t][4 * g + 3];
      {
        const float s01 = odd1 ? v0 : v1, s23 = odd1 ? v2 : v3;
        const float r01 = __shfl_xor(s01, 1, 64), r23 = __shfl_xor(s23, 1, 64);
        if (odd1) { v0 = r01; v2 = r23; } else { v1 = r01; v3 = r23; }
      }
      {
        const float s02 = odd2 ? v0 : v2, s13 = odd2 ? v1 : v3;
        const float r02 = __shfl_xor(s02, 2, 64), r13 = __shfl_xor(s13, 2, 64);
        if (odd2) { v0 = r02; v1 = r13; } else { v2 = r02; v3 = r13; }
      }
      const int i = e + 8 * g + 4 * hi;
      if (a0 + i < CA && b0 + q4 < CB) *reinterpret_cast<float4*>(&P[((long long)t * CA + a0 + i) * CB + b0 + q4]) = make_float4(v0, v1, v2, v3);
    }
  bsum += __shfl_xor(bsum, 32, 64);
  if (MODE != 1) { if (ta == 0 && lane < 32 && b0 + l31 < CB) part_b[(long long)split * pstride + b0 + l31] = bsum; }
  else { if (tb == 0 && lane < 32 && a0 + l31 < CA) part_b[(long long)split * pstride + a0 + l31] = bsum; }
}

// Sum the split-K partials in a fixed order.  Two levels so that a tiny output (e.g. 9x32x32) with
// thousands of splits still spreads over the chip: level 1 reduces groups of splits (grid.y = groups),
// level 2 reduces the group sums.  4 independent accumulators keep 4 loads in flight per thread.
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, long long n4,
                                                            long long stride, int nsplit, int per_group, long long out_stride) {
  const int k0 = blockIdx.y * per_group;
  int k1 = k0 + per_group; if (k1 > nsplit) k1 = nsplit;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    int k = k0;
    for (; k + 3 < k1; k += 4) {
      const float4 a = *reinterpret_cast<const float4*>(part + (long long)k * stride + i * 4);
      const float4 b = *reinterpret_cast<const float4*>(part + (long long)(k + 1) * stride + i * 4);
      const float4 c = *reinterpret_cast<const float4*>(part + (long long)(k + 2) * stride + i * 4);
      const float4 d = *reinterpret_cast<const float4*>(part + (long long)(k + 3) * stride + i * 4);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w; s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
    }
    for (; k < k1; ++k) {
      const float4 a = *reinterpret_cast<const float4*>(part + (long long)k * stride + i * 4);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
    s0.x += s1.x + (s2.x + s3.x); s0.y += s1.y + (s2.y + s3.y); s0.z += s1.z + (s2.z + s3.z); s0.w += s1.w + (s2.w + s3.w);
    *reinterpret_cast<float4*>(out + (long long)blockIdx.y * out_stride + i * 4) = s0;
  }
}

// (round 4: both levels in ONE launch -- the last workgroup of a column, by ticket, sums the group results -- measured +1.5 ms per step: the device-scope fences each
//  workgroup needs around its ticket write back / invalidate the XCD's L2 on this 8-XCD part.  Two launches it stays.)
// Last reduction level, weights and bias in ONE launch.  src = `count` slabs of `stride` floats, each [taps][ca*cb] weight partials
// followed by the bias partials.
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ src, long long stride, int count, int n4 /* ca*cb/4 */, int taps,
                                                           int nb4 /* bias floats / 4 */, float* __restrict__ dw, float* __restrict__ db) {
  const long long st = (long long)n4 * 4;
  const int nw = taps * n4;
  auto sum = [&](long long off) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    int c = 0;
    for (; c + 1 < count; c += 2) {
      const float4 u = *reinterpret_cast<const float4*>(src + (long long)c * stride + off), v = *reinterpret_cast<const float4*>(src + (long long)(c + 1) * stride + off);
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    if (c < count) { const float4 u = *reinterpret_cast<const float4*>(src + (long long)c * stride + off); a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  };
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nw + nb4; i += gridDim.x * 256) {
    if (i >= nw) *reinterpret_cast<float4*>(db + (long long)(i - nw) * 4) = sum((long long)taps * st + (long long)(i - nw) * 4);
    else *reinterpret_cast<float4*>(dw + (long long)i * 4) = sum((long long)i * 4);
  }
}

struct WgradPlan { int tiles_a, tiles_b, strips, nsplit, rows_per_chunk, chunks_per_strip, groups, per_group; size_t part_floats, bias_floats, part2_floats; };

WgradPlan plan_wgrad(int taps, int n, int h, int w, int ca, int cb, int cbias) {
  WgradPlan p;
  p.tiles_a = (ca + 31) / 32; p.tiles_b = (cb + 31) / 32; p.strips = (w + 31) / 32;      // a tile may overhang (channels % 32 != 0)
  const long long pairs = (long long)p.tiles_a * p.tiles_b;
  const long long per = (long long)taps * ca * cb;
  // K-split: every (image, 32-column strip) is cut into row chunks; aim at ~4096 waves (2 waves/SIMD x 256 CUs x 2 rounds),
  // at least 8 rows per chunk (the 3x3 ring re-reads 2 halo rows per chunk), at most 192 MiB of partials
  const long long units = (long long)n * p.strips;
  const long long target = 1536;                      // one resident round: 256 CUs x 6 one-wave workgroups (measured best of 1536..6144)
  long long want = (target + pairs - 1) / pairs;
  const long long cap = std::max<long long>(1, (48LL << 20) / per);
  want = std::min(want, cap);
  long long cps = std::max<long long>(1, (want + units - 1) / units);
  cps = std::min<long long>(cps, std::max<long long>(1, h / 8));
  p.rows_per_chunk = (int)((h + cps - 1) / cps);
  p.chunks_per_strip = (h + p.rows_per_chunk - 1) / p.rows_per_chunk;
  p.nsplit = (int)(units * p.chunks_per_strip);
  p.part_floats = (size_t)p.nsplit * per; p.bias_floats = (size_t)p.nsplit * cbias;
  // second-level groups: enough blocks to fill the chip when the output tile is tiny
  const long long out_blocks = std::max<long long>(1, per / 1024);
  long long g = std::min<long long>(std::min<long long>(64, p.nsplit / 8), 1024 / out_blocks);
  p.groups = (int)std::max<long long>(1, g);
  p.per_group = (p.nsplit + p.groups - 1) / p.groups;
  p.groups = (p.nsplit + p.per_group - 1) / p.per_group;
  p.part2_floats = p.groups > 1 ? (size_t)p.groups * (per + cbias) : 0;
  return p;
}

template <int MODE>
int32_t run_wgrad(unet_ctx* ctx, const float* A, int ldA, const float* B, int ldB, float* dw, float* db, void* ws, size_t ws_bytes, int n,
                  int h, int w, int ca, int cb, hipStream_t s) {
  const int taps = MODE == 0 ? 9 : 4; const int cbias = MODE != 1 ? cb : ca;
  if ((long long)(MODE == 1 ? 4 : 1) * h * w * ldA * 4 >= (1LL << 30) || (long long)h * w * ldB * 4 >= (1LL << 30))
    UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad mfma: one image must stay below 1 GiB (32-bit buffer offsets); use UNET_ALGO_NAIVE");
  const WgradPlan p = plan_wgrad(taps, n, h, w, ca, cb, cbias);
  const long long per = (long long)taps * ca * cb, S = per + cbias;      // one split's partial slab: weights, then bias sums
  const size_t need = (p.part_floats + p.bias_floats + p.part2_floats) * sizeof(float);
  if (!ws || ws_bytes < need) UNET_FAIL(ctx, UNET_E_ARG, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  float* part = static_cast<float*>(ws); float* part_b = part + per;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));            // see the block map in the kernels
  hipLaunchKernelGGL(wgrad_mfma_kernel<MODE>, grid, dim3(64), 0, s, A, ldA, B, ldB, part, part_b, n, h, w, ca, cb, p.tiles_b, p.strips,
                       p.rows_per_chunk, p.chunks_per_strip, p.nsplit, npairs, S);
  UNET_CHECK_LAUNCH(ctx, "wgrad_mfma");
  // deterministic two-level reduction of the split slabs: level 1 sums groups of splits (skipped when there are few), the final
  // level also writes the bias gradient
  const float* src = part; int count = p.nsplit;
  if (p.groups > 1) {
    float* part2 = part + (size_t)p.nsplit * S;
    const unsigned gx = (unsigned)std::min<long long>((S / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(gx, p.groups), dim3(256), 0, s, part, part2, S / 4, S, p.nsplit, p.per_group, S);
    src = part2; count = p.groups;
  }
  const int n4 = ca * cb / 4, nb4 = cbias / 4;
  const int items = taps * n4 + nb4;
  hipLaunchKernelGGL(reduce_final_kernel, dim3((unsigned)std::min(2048, (items + 255) / 256)), dim3(256), 0, s, src, S, count, n4, taps, nb4, dw, db);
  UNET_CHECK_LAUNCH(ctx, "wgrad_reduce");
  return UNET_OK;
}

}  // namespace

// Shared with the bf16 weight-gradient kernels (kernels_bf16.hip): fixed-order reduction of `nslabs` partial slabs of S = taps*ca*cb
// + cbias floats each (scratch for the group sums directly behind the slabs: wgrad_reduce_scratch_floats)
static void reduce_groups(long long per, int nslabs, int* groups, int* per_group) {
  const long long out_blocks = std::max<long long>(1, per / 1024);
  long long g = std::min<long long>(std::min<long long>(64, nslabs / 8), 1024 / out_blocks);
  *groups = (int)std::max<long long>(1, g);
  *per_group = (nslabs + *groups - 1) / *groups;
  *groups = (nslabs + *per_group - 1) / *per_group;
}
size_t wgrad_reduce_scratch_floats(int taps, int ca, int cb, int cbias, int nslabs) {
  int groups, per_group; reduce_groups((long long)taps * ca * cb, nslabs, &groups, &per_group);
  return groups > 1 ? (size_t)groups * ((size_t)taps * ca * cb + cbias) : 0;
}
int32_t k_wgrad_reduce(unet_ctx* ctx, float* part, int nslabs, int taps, int ca, int cb, int cbias, float* dw, float* db, hipStream_t s) {
  const long long per = (long long)taps * ca * cb, S = per + cbias;
  int groups, per_group; reduce_groups(per, nslabs, &groups, &per_group);
  const float* src = part; int count = nslabs;
  if (groups > 1) {
    float* part2 = part + (size_t)nslabs * S;
    const unsigned gx = (unsigned)std::min<long long>((S / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(gx, groups), dim3(256), 0, s, part, part2, S / 4, S, nslabs, per_group, S);
    src = part2; count = groups;
  }
  const int n4 = ca * cb / 4, nb4 = cbias / 4;
  const int items = taps * n4 + nb4;
  hipLaunchKernelGGL(reduce_final_kernel, dim3((unsigned)std::min(2048, (items + 255) / 256)), dim3(256), 0, s, src, S, count, n4, taps, nb4, dw, db);
  UNET_CHECK_LAUNCH(ctx, "wgrad_reduce");
  return UNET_OK;
}

bool mfma_conv3x3_supported(int cin, int cout) { return cin >= CK && (cin % CK) == 0 && cout >= 4 && (cout % 4) == 0; }
bool mfma_wgrad_supported(int ca, int cb) { return ca >= 8 && (ca % 4) == 0 && cb >= 8 && (cb % 4) == 0; }

int32_t k_conv3x3_mfma_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, const float* mask, int mask_mode, float* y,
                           int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, hipStream_t s) {
  if (!mfma_conv3x3_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 mfma: cin=%d cout=%d unsupported", cin, cout);
  if (cout % 128 == 0) return launch_conv<0, 128, 4, 2, 2>(ctx, x, cin, w, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
  if (cout % 64 == 0) return launch_conv<0, 64, 8, 4, 1>(ctx, x, cin, w, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
  // (a 16-row tile for this config measured 3 % slower end to end: fewer, longer blocks)
  return launch_conv<0, 32, 8, 4, 1>(ctx, x, cin, w, bias, mask, mask_mode, y, cout, n, h, wd, cin, cout, act, rate, seed, s);
}

bool mfma_convT_supported(int cin, int cout) { return cin >= 32 && (cin % 32) == 0 && cout >= 32 && (cout % 32) == 0; }

// u[n,2i+a,2j+b,o] = bias[o] + sum_c x[n,i,j,c] * K[a,b,o,c]; GEMM N dimension = 4*cout (ab,o)
int32_t k_convT_mfma_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int ldy, int n, int h, int wd,
                         int cin, int cout, hipStream_t s) {
  if (!mfma_convT_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT mfma: cin=%d cout=%d unsupported", cin, cout);
  return launch_conv<1, 128, 4, 2, 2>(ctx, x, cin, w, bias, nullptr, MASK_NONE, y, ldy, n, h, wd, cin, 4 * cout, ACT_NONE, 0.0f, 0, s);
}

// dx[n,i,j,c] = sum_{ab,o} dU[n,2i+a,2j+b,o] * K[ab,o,c]: K is already [tap][o][c] = [tap][Cin'][Cout']; mask: ReLU of the producer of x
int32_t k_convT_mfma_dgrad(unet_ctx* ctx, const float* dy, int lddy, const float* w, const float* mask, float* dx, int n, int h, int wd,
                           int cin, int cout, hipStream_t s) {
  if (!mfma_convT_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT mfma: cin=%d cout=%d unsupported", cin, cout);
  const int mm = mask ? MASK_RELU : MASK_NONE;
  if (cin % 128 == 0) return launch_conv<2, 128, 4, 2, 2>(ctx, dy, lddy, w, nullptr, mask, mm, dx, cin, n, h, wd, cout, cin, ACT_NONE, 0.0f, 0, s);
  if (cin % 64 == 0) return launch_conv<2, 64, 4, 2, 2>(ctx, dy, lddy, w, nullptr, mask, mm, dx, cin, n, h, wd, cout, cin, ACT_NONE, 0.0f, 0, s);
  return launch_conv<2, 32, 4, 4, 1>(ctx, dy, lddy, w, nullptr, mask, mm, dx, cin, n, h, wd, cout, cin, ACT_NONE, 0.0f, 0, s);
}

size_t mfma_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) {        // large enough for this family AND the h2 form
  if (!mfma_wgrad_supported(cin, cout)) return 0;
  const WgradPlan p = plan_wgrad(9, n, h, wd, cin, cout, cout);
  return std::max(std::max((p.part_floats + p.bias_floats + p.part2_floats) * sizeof(float), h2_wgrad_ws_bytes(n, h, wd, cin, cout)),
                  h2_wgrad_c16_selected(UNET_ALGO_AUTO, wd, cin, cout) ? h2_wgrad_c16_ws_bytes(n, h, wd) : (size_t)0);
}

int32_t k_conv3x3_mfma_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                             int wd, int cin, int cout, hipStream_t s) {
  if (!mfma_wgrad_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad mfma: cin=%d cout=%d unsupported", cin, cout);
  return run_wgrad<0>(ctx, x, cin, dy, cout, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);
}

size_t mfma_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) {
  if (!mfma_wgrad_supported(cout, cin)) return 0;
  const WgradPlan p = plan_wgrad(4, n, h, wd, cout, cin, cout);
  return (p.part_floats + p.bias_floats + p.part2_floats) * sizeof(float);
}

// convT: A = dU (channels = cout, pixel stride lddy, 2h x 2w), B = x (channels = cin, h x w); dK is [4][cout][cin]
int32_t k_convT_mfma_wgrad(unet_ctx* ctx, const float* x, const float* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n,
                           int h, int wd, int cin, int cout, hipStream_t s) {
  if (!mfma_wgrad_supported(cout, cin)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT wgrad mfma: cin=%d cout=%d unsupported", cin, cout);
  return run_wgrad<1>(ctx, dy, lddy, x, cin, dw, db, ws, ws_bytes, n, h, wd, cout, cin, s);
}
