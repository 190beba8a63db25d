// fp32 implicit-GEMM 3x3 convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact
// fp32, 64 FLOP/clk/SIMD = the chip's 157 TFLOP/s fp32 peak; there is no TF32 on gfx950).
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

inline int conv_ablation() {
  static const int v = [] { const char* e = getenv("UNET_CONV_ABL"); return e ? atoi(e) : 0; }();
  return v;
}
inline int conv_prefetch_enabled() {        // UNET_CONV_PF: 0 = never, 1 = 128-wide tiles only, 2 = always (default)
  static const int v = [] { const char* e = getenv("UNET_CONV_PF"); return e ? atoi(e) : 2; }();
  return v;
}

template <int MODE, int TN, int TH, int WR, int WC>
int32_t launch_conv(unet_ctx* ctx, const float* x, int ldx, const float* w, const float* bias, const float* mask, int mask_mode, float* y,
                    int ldy, int n, int h, int wd, int cin, int cout, int act, float rate, unsigned long long seed, hipStream_t s) {
  if (!mask) mask_mode = MASK_NONE;
  if ((long long)(MODE == 2 ? 4 : 1) * h * wd * ldx * 4 >= (1LL << 30) || (long long)(MODE == 1 ? 1 : 9) * cin * cout * 4 >= (1LL << 30))
    UNET_FAIL(ctx, UNET_E_SHAPE, "conv mfma: one image / the weight tensor must stay below 1 GiB (32-bit buffer offsets); use UNET_ALGO_NAIVE");
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH;
  dim3 grid((unsigned)(tiles_x * tiles_y * n), (unsigned)((cout + TN - 1) / TN));
  if (MODE == 0 && conv_ablation()) {          // timing experiments (tools/conv_ablate.py); never set in production
    const int a = conv_ablation();
    if (a == 1) hipLaunchKernelGGL((conv_mfma_kernel<MODE, TN, TH, WR, WC, false, false, 1>), grid, dim3(256), 0, s, x, ldx, w, bias, mask, y, ldy, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
    else if (a == 2) hipLaunchKernelGGL((conv_mfma_kernel<MODE, TN, TH, WR, WC, false, false, 2>), grid, dim3(256), 0, s, x, ldx, w, bias, mask, y, ldy, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_mfma_kernel<MODE, TN, TH, WR, WC, false, false, 3>), grid, dim3(256), 0, s, x, ldx, w, bias, mask, y, ldy, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
  } else {
    // default prefetch: always.  (In isolation the register prefetch is +7 % on the 128-wide tile and -5 % on the 32-wide one,
    // where it costs a wave of occupancy; inside the training step "always" measured +0.7 % over "128-wide only".)
    const bool pf = conv_prefetch_enabled() && (TN >= 128 || conv_prefetch_enabled() > 1);
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
  static_assert(MODE == 0 || MODE == 1, "the Winograd-domain gradient (run_wgrad<2>) has its own kernel: wgrad_wino_kernel");
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

// Winograd-domain conv3x3 weight gradient, two waves per workgroup (see MODE 2 above for the math).  The 12 accumulator tiles
// dU[ky][k] are split by k between the waves (wave 0: k = 0,1; wave 1: k = 3,2 -> 96 accumulator registers each, so the next-row
// register prefetch fits again and 3 waves share a SIMD); both read the same raw LDS rows.  With e0,e1,e2 = columns (w, w+1, w+2)
// of the tile's 4-pixel input window both waves form slot 0 = e0-e2 (V0 resp. V3) and slot 1 = e1 + sgn*(w ? e0 : e2) (V1 resp.
// V2); the dY factors are slot 0 = (w ? -dy1 : dy0), slot 1 = dy0 + sgn*dy1: no divergent code, one basic block per row step.
__global__ __launch_bounds__(128, 2) void wgrad_wino_kernel(const float* __restrict__ A, int ldA, const float* __restrict__ B, int ldB,
                                                            float* __restrict__ part, float* __restrict__ part_b, int N, int H, int W,
                                                            int CA, int CB, int tiles_b, int strips, int rows_per_chunk,
                                                            int chunks_per_strip, int nsplit, int tiles_a_x_b, long long pstride) {
  constexpr int ROWF = 34 * 32;
  constexpr int AL = 3, BL = 2;                       // float4 staging registers per lane (272 / 256 items over 128 lanes)
  __shared__ __attribute__((aligned(16))) float s_a[ROWF];                 // newest X row only (older rows live in the register window)
  __shared__ __attribute__((aligned(16))) float s_b[32 * 32];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float bsum = 0.0f;
  const float* An = A + (long long)n * H * W * ldA;
  const float* Bn = B + (long long)n * H * W * ldB;

  int aoff[AL], alds[AL], boff[BL], blds[BL];
#pragma unroll
  for (int k = 0; k < AL; ++k) {
    const int idx = min(tid + 128 * k, 34 * 8 - 1);               // surplus lanes redo the last item: no branch in the loop
    const int pix = idx >> 3, q = idx & 7; const int gx = x0 - 1 + pix;
    aoff[k] = (gx >= 0 && gx < W && a0 + q * 4 < CA) ? (gx * ldA + a0 + q * 4) * 4 : UNET_COL_OOB;
    alds[k] = pix * 32 + q * 4;
  }
#pragma unroll
  for (int k = 0; k < BL; ++k) {
    const int idx = tid + 128 * k; const int pix = idx >> 3, q = idx & 7; const int gx = x0 + pix;
    boff[k] = (gx < W && b0 + q * 4 < CB) ? (gx * ldB + b0 + q * 4) * 4 : UNET_COL_OOB;
    blds[k] = pix * 32 + q * 4;
  }
  f32x4 areg[AL], breg[BL];
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(An, (long long)H * W * ldA * 4), rs_b = make_rsrc(Bn, (long long)H * W * ldB * 4);
  const int nsteps = yb - ya + 2;                        // step t: ring row yy = ya-1+t (X row yy, dY row yy-1)
  auto issue = [&](int t) __attribute__((always_inline)) {
    const int yy = ya - 1 + t, yd = yy - 1;
    const int ra = (yy >= 0 && yy < H) ? yy * W * ldA * 4 : UNET_OOB, rb = (yd >= ya && yd < yb) ? yd * W * ldB * 4 : UNET_OOB;
#pragma unroll
    for (int k = 0; k < AL; ++k) areg[k] = buf_ld4(rs_a, aoff[k] + ra);
#pragma unroll
    for (int k = 0; k < BL; ++k) breg[k] = buf_ld4(rs_b, boff[k] + rb);
  };
  const float sgn = w ? -1.0f : 1.0f;

  // Register-resident row window: the two transformed A operands of every tile pair (8 pairs x 2 slots) of the LAST THREE X rows
  // stay in registers (48 VGPRs), so a row is read from LDS and transformed once instead of three times (as ky = 2, 1, 0 of three
  // consecutive output rows): 5 instead of 11 LDS reads per 6 MFMAs.  LDS then only holds the newest X row and the dY row.
  // The row loop is unrolled by 3 so the window slot of a step is a compile-time constant.
  float win[3][8][2];
  auto step = [&](int t, auto phc) __attribute__((always_inline)) {
    constexpr int PH = decltype(phc)::value;
#pragma unroll
    for (int k = 0; k < AL; ++k) *reinterpret_cast<f32x4*>(&s_a[alds[k]]) = areg[k];
#pragma unroll
    for (int k = 0; k < BL; ++k) *reinterpret_cast<f32x4*>(&s_b[blds[k]]) = breg[k];
    __syncthreads();
    if (t + 1 < nsteps) issue(t + 1);                      // next row's loads fly under this row's MFMAs (two rows ahead measured slower)
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {                       // newest X row (= ky 2 of this step's output row) -> window slot PH
      const float* r = &s_a[(2 * (2 * pp + hi) + w) * 32 + l31];          // px 0 <-> column x0-1: window px 2tt .. 2tt+3
      const float e0 = r[0], e1 = r[32], e2 = r[64];
      const float z = w ? e0 : e2;
      win[PH][pp][0] = e0 - e2; win[PH][pp][1] = e1 + sgn * z;
    }
    if (t >= 2) {
#pragma unroll
      for (int pp = 0; pp < 8; ++pp) {
        const int tt = 2 * pp + hi;                        // MFMA k-pair = 2 consecutive Winograd tiles of the strip
        const float dy0 = s_b[(2 * tt) * 32 + l31], dy1 = s_b[(2 * tt + 1) * 32 + l31];
        if (w == 0) bsum += dy0 + dy1;
        const float bm0 = w ? -dy1 : dy0, bm1 = dy0 + sgn * dy1;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {                   // X row of tap ky was staged (2 - ky) steps ago
          const int sl = (PH + 1 + ky) % 3;
          acc[ky][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(win[sl][pp][0], bm0, acc[ky][0], 0, 0, 0);
          acc[ky][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(win[sl][pp][1], bm1, acc[ky][1], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  };
  issue(0);
  for (int t = 0; t < nsteps; t += 3) {                     // window slot = t % 3: static inside the 3-step body
    step(t, std::integral_constant<int, 0>{});
    if (t + 1 < nsteps) step(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < nsteps) step(t + 2, std::integral_constant<int, 2>{});
  }

  // partial tiles: slot j of wave w is Winograd index k = w ? 3 - j : j; quad transpose -> 16-byte stores (see wgrad_mfma_kernel)
  float* P = part + (long long)split * pstride;
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int tap = ky * 4 + (w ? 3 - j : j);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = acc[ky][j][4 * g + 0], v1 = acc[ky][j][4 * g + 1], v2 = acc[ky][j][4 * g + 2], v3 = acc[ky][j][4 * g + 3];
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
        if (a0 + i < CA && b0 + q4 < CB) *reinterpret_cast<float4*>(&P[((long long)tap * CA + a0 + i) * CB + b0 + q4]) = make_float4(v0, v1, v2, v3);
      }
    }
  bsum += __shfl_xor(bsum, 32, 64);
  if (w == 0 && ta == 0 && lane < 32 && b0 + l31 < CB) part_b[(long long)split * pstride + b0 + l31] = bsum;
}

// Weight gradient in the 2-D Winograd domain F(2x2,3x3): 16 instead of 24 (F(2,3) along x) or 36 (direct) MFMAs per 2x2 output tile.
//   dU[xi][nu][ci][co] = sum over 2x2 output tiles of V[xi][nu][ci] * dM[xi][nu][co],   V = B^T d B (4x4 input tile d),  dM = A dY A^T,
//   dW = G^T dU G (16 -> 9, applied by reduce_final_wino2d_kernel).
// Workgroup = 4 waves, wave = xi (the y index): it forms ITS row combination of the 4-row input window (R = rowP + sgn * rowQ: 0-2, 1+2,
// 2-1, 1-3) and of the dY row pair (S = c0 * dy_r0 + c1 * dy_r1: dy0, dy0+dy1, dy0-dy1, -dy1), then the x transforms in registers, and
// owns the four nu accumulator tiles (64 registers).  A step = one output ROW PAIR of a 32-column strip = 16 tiles = 8 MFMA k-pairs per
// (xi, nu).  LDS keeps the rows TRANSPOSED ([channel][pixel], pitch 34: conflict-free ds_read_b64), so a lane fetches the 4 columns of
// its tile with two 8-byte reads per row: 6 reads per 4 MFMAs.  X rows live in a 4-slot ring (two new rows per step), the next step's
// rows are register-prefetched under the MFMAs.  (A two-wave form -- wave 0: xi = 1, 2 from rows 1, 2; wave 1: xi = 0, 3 -- reads a third
// less from LDS but needs 128 accumulator registers per wave: 2 instead of 3 waves per SIMD, measured 10 % slower.)
__global__ __launch_bounds__(256, 4) void wgrad_wino2d_kernel(const float* __restrict__ A, int ldA, const float* __restrict__ B, int ldB,
                                                              float* __restrict__ part, float* __restrict__ part_b, int N, int H, int W,
                                                              int CA, int CB, int tiles_b, int strips, int rows_per_chunk,
                                                              int chunks_per_strip, int nsplit, int tiles_a_x_b, long long pstride) {
  constexpr int PIT = 34, ROW = 32 * PIT;
  constexpr int XP = 2 * 34 * 8, YP = 2 * 32 * 8;          // 16-byte pieces per batch: two X rows (34 px), two dY rows (32 px)
  constexpr int XL = (XP + 255) / 256, YL = YP / 256;
  __shared__ __attribute__((aligned(16))) float s_x[4 * ROW];
  __shared__ __attribute__((aligned(16))) float s_y[4 * ROW];                  // the four dY row combinations dy0, dy0+dy1, dy0-dy1, dy1, formed while staging
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = tiles_a_x_b, sq = blockIdx.x >> 3;
  const int pair = sq % npairs, split = (sq / npairs) * 8 + (blockIdx.x & 7);      // XCD-aware block map, as wgrad_wino_kernel
  if (split >= nsplit) return;
  const int ta = pair / tiles_b, tb = pair % tiles_b;
  const int a0 = ta * 32, b0 = tb * 32;
  const int chunk = split % chunks_per_strip; const int t2 = split / chunks_per_strip;
  const int cs = t2 % strips, n = t2 / strips;
  const int x0 = cs * 32;
  const int ya = chunk * rows_per_chunk;                                          // even (plan_wgrad rounds the chunk height up)
  const int yb = ya + rows_per_chunk < H ? ya + rows_per_chunk : H;
  const int npr = (yb - ya + 1) >> 1;                                             // row pairs of this chunk

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float bsum = 0.0f;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + (long long)n * H * W * ldA, (long long)H * W * ldA * 4);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(B + (long long)n * H * W * ldB, (long long)H * W * ldB * 4);

  int xoff[XL], xlds[XL], yoff[YL], ylds[YL];               // column part of the global offset; LDS float index | row-in-batch << 20
#pragma unroll
  for (int k = 0; k < XL; ++k) {
    const int idx = min(tid + 256 * k, XP - 1);             // surplus lanes redo the last piece
    const int i = idx / 272, rem = idx - i * 272, pix = rem >> 3, q = rem & 7, gx = x0 - 1 + pix;
    xoff[k] = (gx >= 0 && gx < W && a0 + q * 4 < CA) ? (gx * ldA + a0 + q * 4) * 4 : UNET_COL_OOB;
    xlds[k] = (q * 4 * PIT + pix) | (i << 20);
  }
#pragma unroll
  for (int k = 0; k < YL; ++k) {
    const int idx = tid + 256 * k;
    const int i = idx >> 8, rem = idx & 255, pix = rem >> 3, q = rem & 7, gx = x0 + pix;
    yoff[k] = (gx < W && b0 + q * 4 < CB) ? (gx * ldB + b0 + q * 4) * 4 : UNET_COL_OOB;
    ylds[k] = (q * 4 * PIT + pix) | (i << 20);
  }
  f32x4 xreg[XL], yreg[YL];
  // batch b = X rows ya-1+2b, ya+2b  and (b >= 1) the dY rows of pair b-1
  auto issue = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XL; ++k) {
      const int yr = ya - 1 + 2 * b + (xlds[k] >> 20);
      xreg[k] = buf_ld4(rs_a, xoff[k] + ((yr >= 0 && yr < H) ? yr * W * ldA * 4 : UNET_OOB));
    }
#pragma unroll
    for (int k = 0; k < YL; ++k) {
      const int yd = ya + 2 * (b - 1) + (ylds[k] >> 20);
      yreg[k] = buf_ld4(rs_b, yoff[k] + ((b >= 1 && yd < yb) ? yd * W * ldB * 4 : UNET_OOB));
    }
  };
  auto store = [&](int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < XL; ++k) {
      float* d = s_x + ((2 * b + (xlds[k] >> 20)) & 3) * ROW + (xlds[k] & 0xFFFFF);
      d[0] = xreg[k][0]; d[PIT] = xreg[k][1]; d[2 * PIT] = xreg[k][2]; d[3 * PIT] = xreg[k][3];
    }
    if (b >= 1) {                                          // piece k = row k of the pair, same (pixel, channel quad) for both: combine here, once,
      float* d = s_y + (ylds[0] & 0xFFFFF);                // instead of in every wave of the main loop
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = yreg[0][j], c = yreg[1][j];
        d[j * PIT] = a; d[ROW + j * PIT] = a + c; d[2 * ROW + j * PIT] = a - c; d[3 * ROW + j * PIT] = c;
      }
    }
  };
  // wave-uniform row combinations: X window rows (P, Q) with sign, dY rows with (c0, c1)
  const int rowP = (w == 0) ? 0 : (w == 2 ? 2 : 1), rowQ = (w == 0 || w == 1) ? 2 : (w == 2 ? 1 : 3);
  const float sgx = (w == 1) ? 1.0f : -1.0f;

  issue(0); store(0); issue(1);
  for (int s = 0; s < npr; ++s) {
    store(s + 1);
    __syncthreads();
    if (s + 2 <= npr) issue(s + 2);
    const float* rp = s_x + ((2 * s + rowP) & 3) * ROW + l31 * PIT + 2 * hi;
    const float* rq = s_x + ((2 * s + rowQ) & 3) * ROW + l31 * PIT + 2 * hi;
    const float* ys = s_y + w * ROW + l31 * PIT + 2 * hi;    // this wave's dY combination (xi = 3 reads +dy1: its sign is applied by the final reduction)
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {                       // MFMA k-pair = tiles 2pp (lanes 0-31) and 2pp+1 (lanes 32-63)
      const float2 p01 = *reinterpret_cast<const float2*>(rp + 4 * pp), p23 = *reinterpret_cast<const float2*>(rp + 4 * pp + 2);
      const float2 q01 = *reinterpret_cast<const float2*>(rq + 4 * pp), q23 = *reinterpret_cast<const float2*>(rq + 4 * pp + 2);
      const float2 sv = *reinterpret_cast<const float2*>(ys + 4 * pp);
      const float e0 = fmaf(sgx, q01.x, p01.x), e1 = fmaf(sgx, q01.y, p01.y), e2 = fmaf(sgx, q23.x, p23.x), e3 = fmaf(sgx, q23.y, p23.y);
      const float s0 = sv.x, s1 = sv.y;
      if (w == 1) bsum += s0 + s1;                         // wave 1 sees dy_r0 + dy_r1: the bias gradient rides along
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0 - e2, s0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1 + e2, s0 + s1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e2 - e1, s0 - s1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1 - e3, s1, acc[3], 0, 0, 0);       // dM3 = -s1: sign applied by the final reduction
    }
    __syncthreads();
  }

  // partial tiles dU[xi = w][nu]; quad transpose -> 16-byte stores (see wgrad_mfma_kernel)
  float* P = part + (long long)split * pstride;
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    const int tap = w * 4 + nu;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v0 = acc[nu][4 * g + 0], v1 = acc[nu][4 * g + 1], v2 = acc[nu][4 * g + 2], v3 = acc[nu][4 * g + 3];
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
      if (a0 + i < CA && b0 + q4 < CB) *reinterpret_cast<float4*>(&P[((long long)tap * CA + a0 + i) * CB + b0 + q4]) = make_float4(v0, v1, v2, v3);
    }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (w == 1 && ta == 0 && lane < 32 && b0 + l31 < CB) part_b[(long long)split * pstride + b0 + l31] = bsum;
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

// Last reduction level, weights and bias in ONE launch.  src = `count` slabs of `stride` floats, each [taps][ca*cb] weight partials
// followed by the bias partials.  WINO: the slab is in the Winograd domain dU[ky][k] (12 taps); the 12 -> 9 transform (transpose of
// the weight transform G: dg0 = dU0 + (dU1+dU2)/2, dg1 = (dU1-dU2)/2, dg2 = (dU1+dU2)/2 + dU3) is applied on the fly.
template <int WINO>       // 0: plain, 1: 12 -> 9 (F(2,3) along x); the 16 -> 9 form of F(2x2,3x3) is reduce_final_wino2d_kernel
__global__ __launch_bounds__(256) void reduce_final_kernel(const float* __restrict__ src, long long stride, int count, int n4 /* ca*cb/4 */, int taps,
                                                           int nb4 /* bias floats / 4 */, float* __restrict__ dw, float* __restrict__ db) {
  const long long st = (long long)n4 * 4;
  const int nw = (WINO == 1 ? 3 : taps) * n4;
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
    if (i >= nw) { *reinterpret_cast<float4*>(db + (long long)(i - nw) * 4) = sum((long long)taps * st + (long long)(i - nw) * 4); continue; }
    if (WINO == 0) { *reinterpret_cast<float4*>(dw + (long long)i * 4) = sum((long long)i * 4); continue; }
    const int ky = i / n4, j = i - ky * n4;
    const long long o = (long long)ky * 4 * st + (long long)j * 4;
    const float4 u0 = sum(o), u1 = sum(o + st), u2 = sum(o + 2 * st), u3 = sum(o + 3 * st);
    float* q = dw + (long long)ky * 3 * st + (long long)j * 4;
    const float4 hs = make_float4(0.5f * (u1.x + u2.x), 0.5f * (u1.y + u2.y), 0.5f * (u1.z + u2.z), 0.5f * (u1.w + u2.w));
    *reinterpret_cast<float4*>(q) = make_float4(u0.x + hs.x, u0.y + hs.y, u0.z + hs.z, u0.w + hs.w);
    *reinterpret_cast<float4*>(q + st) = make_float4(0.5f * (u1.x - u2.x), 0.5f * (u1.y - u2.y), 0.5f * (u1.z - u2.z), 0.5f * (u1.w - u2.w));
    *reinterpret_cast<float4*>(q + 2 * st) = make_float4(hs.x + u3.x, hs.y + u3.y, hs.z + u3.z, hs.w + u3.w);
  }
}

// Final level for the F(2x2,3x3) weight gradient: 16 Winograd taps -> 9 kernel taps, dW = G^T dU G.  A workgroup = 16 (ci, co) quads x
// 16 taps: every thread sums ITS (tap, quad) over the slabs (16x the parallelism of one thread per quad: the 32 x 32-channel layers
// have only 256 quads), the 16 x 16 sums meet in LDS and 144 threads apply the two 4 -> 3 transforms.  Extra workgroups sum the bias.
__global__ __launch_bounds__(256) void reduce_final_wino2d_kernel(const float* __restrict__ src, long long stride, int count, int n4, int nb4,
                                                                  int wblocks, float* __restrict__ dw, float* __restrict__ db) {
  const long long st = (long long)n4 * 4;
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
  if ((int)blockIdx.x >= wblocks) {                      // bias gradient: plain sums
    const int i = ((int)blockIdx.x - wblocks) * 256 + threadIdx.x;
    if (i < nb4) *reinterpret_cast<float4*>(db + (long long)i * 4) = sum(16 * st + (long long)i * 4);
    return;
  }
  __shared__ float4 s_u[16][16];
  const int tq = threadIdx.x & 15, tap = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + tq;
  s_u[tap][tq] = j < n4 ? sum((long long)tap * st + (long long)j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  if (threadIdx.x >= 144) return;
  const int q = threadIdx.x & 15, kk = threadIdx.x >> 4, ky = kk / 3, kx = kk - ky * 3;
  const int jq = blockIdx.x * 16 + q;
  if (jq >= n4) return;
  // rows of G^T: (1, 1/2, 1/2, 0), (0, 1/2, -1/2, 0), (0, 1/2, 1/2, 1)
  // (the kernel accumulates xi = 3 and nu = 3 with the opposite sign -- it skips the negations of -dy1 and -s1 --: folded in here)
  const float gy[4] = {ky == 0 ? 1.f : 0.f, 0.5f, ky == 1 ? -0.5f : 0.5f, ky == 2 ? -1.f : 0.f};
  const float gx[4] = {kx == 0 ? 1.f : 0.f, 0.5f, kx == 1 ? -0.5f : 0.5f, kx == 2 ? -1.f : 0.f};
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int xi = 0; xi < 4; ++xi)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float c = gy[xi] * gx[nu];
      const float4 u = s_u[xi * 4 + nu][q];
      r.x = fmaf(c, u.x, r.x); r.y = fmaf(c, u.y, r.y); r.z = fmaf(c, u.z, r.z); r.w = fmaf(c, u.w, r.w);
    }
  *reinterpret_cast<float4*>(dw + (long long)kk * st + (long long)jq * 4) = r;
}

struct WgradPlan { int tiles_a, tiles_b, strips, nsplit, rows_per_chunk, chunks_per_strip, groups, per_group; size_t part_floats, bias_floats, part2_floats; };

WgradPlan plan_wgrad(int taps, int n, int h, int w, int ca, int cb, int cbias, bool even_rows = false) {
  WgradPlan p;
  p.tiles_a = (ca + 31) / 32; p.tiles_b = (cb + 31) / 32; p.strips = (w + 31) / 32;      // a tile may overhang (channels % 32 != 0)
  const long long pairs = (long long)p.tiles_a * p.tiles_b;
  const long long per = (long long)taps * ca * cb;
  // K-split: every (image, 32-column strip) is cut into row chunks; aim at ~4096 waves (2 waves/SIMD x 256 CUs x 2 rounds),
  // at least 8 rows per chunk (the 3x3 ring re-reads 2 halo rows per chunk), at most 192 MiB of partials
  const long long units = (long long)n * p.strips;
  static const long long target_env = [] { const char* e = getenv("UNET_WGRAD_BLOCKS"); return e ? atoll(e) : 0LL; }();
  // one resident round: 256 CUs x 6 two-wave workgroups (1-D Winograd / direct kernels; measured best of 1536..6144), x 4 four-wave
  // workgroups of the F(2x2,3x3) kernel (1024: best of 768..3072)
  const long long target = target_env ? target_env : (even_rows ? 1024LL : 1536LL);
  long long want = (target + pairs - 1) / pairs;
  const long long cap = std::max<long long>(1, (48LL << 20) / per);
  want = std::min(want, cap);
  long long cps = std::max<long long>(1, (want + units - 1) / units);
  cps = std::min<long long>(cps, std::max<long long>(1, h / 8));
  p.rows_per_chunk = (int)((h + cps - 1) / cps);
  if (even_rows) p.rows_per_chunk += p.rows_per_chunk & 1;                 // the F(2x2,3x3) kernel walks row pairs
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
  const int taps = MODE == 0 ? 9 : (MODE == 1 ? 4 : (MODE == 2 ? 12 : 16)); const int cbias = MODE != 1 ? cb : ca;
  if ((long long)(MODE == 1 ? 4 : 1) * h * w * ldA * 4 >= (1LL << 30) || (long long)h * w * ldB * 4 >= (1LL << 30))
    UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad mfma: one image must stay below 1 GiB (32-bit buffer offsets); use UNET_ALGO_NAIVE");
  const WgradPlan p = plan_wgrad(taps, n, h, w, ca, cb, cbias, MODE == 3);
  const long long per = (long long)taps * ca * cb, S = per + cbias;      // one split's partial slab: weights, then bias sums
  const size_t need = (p.part_floats + p.bias_floats + p.part2_floats) * sizeof(float);
  if (!ws || ws_bytes < need) UNET_FAIL(ctx, UNET_E_ARG, "wgrad: workspace %zu < %zu bytes", ws_bytes, need);
  float* part = static_cast<float*>(ws); float* part_b = part + per;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));            // see the block map in the kernels
  if constexpr (MODE == 3)
    hipLaunchKernelGGL(wgrad_wino2d_kernel, grid, dim3(256), 0, s, A, ldA, B, ldB, part, part_b, n, h, w, ca, cb, p.tiles_b, p.strips, p.rows_per_chunk,
                       p.chunks_per_strip, p.nsplit, npairs, S);
  else if constexpr (MODE == 2)
    hipLaunchKernelGGL(wgrad_wino_kernel, grid, dim3(128), 0, s, A, ldA, B, ldB, part, part_b, n, h, w, ca, cb, p.tiles_b, p.strips, p.rows_per_chunk,
                       p.chunks_per_strip, p.nsplit, npairs, S);
  else
    hipLaunchKernelGGL(wgrad_mfma_kernel<MODE>, grid, dim3(64), 0, s, A, ldA, B, ldB, part, part_b, n, h, w, ca, cb, p.tiles_b, p.strips,
                       p.rows_per_chunk, p.chunks_per_strip, p.nsplit, npairs, S);
  UNET_CHECK_LAUNCH(ctx, "wgrad_mfma");
  // deterministic two-level reduction of the split slabs: level 1 sums groups of splits (skipped when there are few), the final
  // level also writes the bias gradient and (MODE 2) applies the 12 -> 9 Winograd transform: 2 launches instead of 5
  const float* src = part; int count = p.nsplit;
  if (p.groups > 1) {
    float* part2 = part + (size_t)p.nsplit * S;
    const unsigned gx = (unsigned)std::min<long long>((S / 4 + 255) / 256, 2048);
    hipLaunchKernelGGL(reduce_splits_kernel, dim3(gx, p.groups), dim3(256), 0, s, part, part2, S / 4, S, p.nsplit, p.per_group, S);
    src = part2; count = p.groups;
  }
  const int n4 = ca * cb / 4, nb4 = cbias / 4;
  const int items = (MODE == 3 ? 1 : MODE == 2 ? 3 : taps) * n4 + nb4;
  const dim3 gf((unsigned)std::min(2048, (items + 255) / 256));
  if (MODE == 3) {
    const int wblocks = (n4 + 15) / 16;
    hipLaunchKernelGGL(reduce_final_wino2d_kernel, dim3((unsigned)(wblocks + (nb4 + 255) / 256)), dim3(256), 0, s, src, S, count, n4, nb4, wblocks, dw, db);
  }
  else if (MODE == 2) hipLaunchKernelGGL(reduce_final_kernel<1>, gf, dim3(256), 0, s, src, S, count, n4, taps, nb4, dw, db);
  else hipLaunchKernelGGL(reduce_final_kernel<0>, gf, dim3(256), 0, s, src, S, count, n4, taps, nb4, dw, db);
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
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3((unsigned)std::min(2048, (items + 255) / 256)), dim3(256), 0, s, src, S, count, n4, taps, nb4, dw, db);
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

size_t mfma_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) {        // large enough for the direct AND the Winograd form
  if (!mfma_wgrad_supported(cin, cout)) return 0;
  const WgradPlan p = plan_wgrad(9, n, h, wd, cin, cout, cout), q = plan_wgrad(12, n, h, wd, cin, cout, cout), r = plan_wgrad(16, n, h, wd, cin, cout, cout, true);
  return std::max(std::max(std::max(p.part_floats + p.bias_floats + p.part2_floats, q.part_floats + q.bias_floats + q.part2_floats), r.part_floats + r.bias_floats + r.part2_floats) * sizeof(float),
                  h2_wgrad_ws_bytes(n, h, wd, cin, cout));
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
__global__ __launch_bounds__(256) void fold_bn_bwd_sums_kernel(const float* __restrict__ w, const float* __restrict__ dw_raw, const float* __restrict__ S,
                                                               const float* __restrict__ mean, const float* __restrict__ istd, double* __restrict__ sums, int cin, int cout) {
  __shared__ double s_a[256], s_b[256];
  const int c = blockIdx.x;
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < 9 * cout; i += 256) {
    const int tap = i / cout, o = i - tap * cout;
    const long long j = ((long long)tap * cin + c) * cout + o;
    const float wv = w[j];
    a = fmaf(wv, dw_raw[j], a); b = fmaf(wv, S[i], b);
  }
  s_a[threadIdx.x] = (double)a; s_b[threadIdx.x] = (double)b;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) { s_a[threadIdx.x] += s_a[threadIdx.x + st]; s_b[threadIdx.x] += s_b[threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[c] += s_b[0]; sums[cin + c] += (double)istd[c] * (s_a[0] - (double)mean[c] * s_b[0]); }
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
                                      float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums) {
  if (!dy || !scale || !shift || !dw || !db || !scratch || !wgrad_bn_fold_supported(cout)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad_bn_fold_fix: bad args (cout=%d)", cout);
  float* border = scratch; float* S = scratch + (size_t)n * BORDER_SEG * 8 * cout;
  hipLaunchKernelGGL(border_sums_kernel<T>, dim3(8 * BORDER_SEG, (unsigned)n), dim3(256), 0, s, dy, border, h, wd, cout);
  hipLaunchKernelGGL(fold_tap_sums_kernel, dim3(9, (unsigned)((cout + 63) / 64)), dim3(1024), 0, s, border, db, S, n * BORDER_SEG, cout);
  if (bn_bwd_sums) {
    if (!w || !mean || !istd) UNET_FAIL(ctx, UNET_E_ARG, "wgrad_bn_fold_fix: the BatchNorm backward sums need w, mean, istd");
    hipLaunchKernelGGL(fold_bn_bwd_sums_kernel, dim3((unsigned)cin), dim3(256), 0, s, w, dw, S, mean, istd, bn_bwd_sums, cin, cout);
  }
  const long long total4 = 9LL * cin * cout / 4;
  hipLaunchKernelGGL(fold_fix_kernel, dim3((unsigned)std::min<long long>((total4 + 255) / 256, 2048)), dim3(256), 0, s, dw, scale, shift, S, cin, cout / 4, total4);
  UNET_CHECK_LAUNCH(ctx, "wgrad_bn_fold_fix");
  return UNET_OK;
}
int32_t k_wgrad_bn_fold_fix(unet_ctx* ctx, const float* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                            float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums) {
  return wgrad_bn_fold_fix_impl(ctx, dy, n, h, wd, cin, cout, scale, shift, dw, db, scratch, s, w, mean, istd, bn_bwd_sums);
}
int32_t k_wgrad_bn_fold_fix_bf16(unet_ctx* ctx, const unet_bf16* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                                 float* scratch, hipStream_t s, const float* w, const float* mean, const float* istd, double* bn_bwd_sums) {
  return wgrad_bn_fold_fix_impl(ctx, dy, n, h, wd, cin, cout, scale, shift, dw, db, scratch, s, w, mean, istd, bn_bwd_sums);
}

static int wgrad_wino_form() {
  static const int form = [] { const char* e = getenv("UNET_WINO_WGRAD_2D"); return e ? atoi(e) : 1; }();      // A/B switch: 0 = F(2,3) along x
  return form;
}
// executed / algorithmic multiplies of the Winograd-domain weight gradient of this shape: 4/9 (F(2x2,3x3)) or 2/3 (F(2,3) along x)
double wino_wgrad_exec_ratio(int h) { return (wgrad_wino_form() && h >= 2) ? 4.0 / 9.0 : 2.0 / 3.0; }

int32_t k_conv3x3_wino_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                             int wd, int cin, int cout, hipStream_t s) {
  if (!mfma_wgrad_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad winograd: cin=%d cout=%d unsupported", cin, cout);
  const int form = wgrad_wino_form();
  if (form && h >= 2) return run_wgrad<3>(ctx, x, cin, dy, cout, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);
  return run_wgrad<2>(ctx, x, cin, dy, cout, dw, db, ws, ws_bytes, n, h, wd, cin, cout, s);
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
