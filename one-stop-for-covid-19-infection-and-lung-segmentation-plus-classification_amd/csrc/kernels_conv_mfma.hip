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
// Weight-gradient: see conv3x3_wgrad_mfma_kernel below (split-K over pixels, deterministic 2-stage).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CK = 8;     // input channels per K chunk
constexpr int CKP = 12;   // padded pixel stride in LDS (floats): 4*odd -> ds_read_b128 conflict-free
constexpr int PW = 34;    // patch width: 32 + halo

template <int TN, int TH, int WR, int WC>
__global__ __launch_bounds__(256) void conv3x3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ mask,
                                                           float* __restrict__ y, int N, int H, int W, int Cin, int Cout,
                                                           int relu, int tiles_x, int tiles_y) {
  static_assert(WR * WC == 4, "4 waves");
  constexpr int RW = TH / WR;          // image rows (M tiles) per wave
  constexpr int NW = TN / 32 / WC;     // N tiles per wave
  constexpr int PR = TH + 2;
  static_assert(RW >= 1 && NW >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float s_in[PR * PW * CKP];
  __shared__ __attribute__((aligned(16))) float s_w[9 * CK * TN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  int b = blockIdx.x;
  const int tx = b % tiles_x; b /= tiles_x;
  const int ty = b % tiles_y; const int n = b / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int nbase = blockIdx.y * TN;

  f32x16 acc[RW][NW];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float* xn = x + (long long)n * H * W * Cin;
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    // ---- stage the halo patch of this channel chunk (zero fill = 'same' padding + ragged edges)
    for (int idx = tid; idx < PR * PW * 2; idx += 256) {
      const int q = idx & 1, pix = idx >> 1;
      const int r = pix / PW, c = pix - r * PW;
      const int gy = y0 + r - 1, gx = x0 + c - 1;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < H && gx >= 0 && gx < W)
        v = *reinterpret_cast<const float4*>(xn + ((long long)gy * W + gx) * Cin + c0 + q * 4);
      *reinterpret_cast<float4*>(&s_in[pix * CKP + q * 4]) = v;
    }
    // ---- stage the weight slab [tap][ci][TN]
    for (int idx = tid; idx < 9 * CK * (TN / 4); idx += 256) {
      const int q = idx % (TN / 4), row = idx / (TN / 4);
      const int tap = row >> 3, ci = row & 7;
      const float4 v = *reinterpret_cast<const float4*>(w + ((long long)(tap * Cin + c0 + ci)) * Cout + nbase + q * 4);
      *reinterpret_cast<float4*>(&s_w[row * TN + q * 4]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dr = tap / 3, dc = tap % 3;
      f32x4 a[RW];
#pragma unroll
      for (int i = 0; i < RW; ++i)
        a[i] = *reinterpret_cast<const f32x4*>(&s_in[((wr * RW + i + dr) * PW + l31 + dc) * CKP + hi * 4]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float bv[NW];
#pragma unroll
        for (int jn = 0; jn < NW; ++jn) bv[jn] = s_w[(tap * CK + j + 4 * hi) * TN + (wc * NW + jn) * 32 + l31];
#pragma unroll
        for (int i = 0; i < RW; ++i)
#pragma unroll
          for (int jn = 0; jn < NW; ++jn)
            acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], bv[jn], acc[i][jn], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: D[row = pixel][col = cout]; lane holds col = l31, rows (r&3)+8*(r>>2)+4*hi
#pragma unroll
  for (int jn = 0; jn < NW; ++jn) {
    const int co = nbase + (wc * NW + jn) * 32 + l31;
    const float bb = bias ? bias[co] : 0.0f;
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int py = y0 + wr * RW + i;
      if (py >= H) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (px >= W) continue;
        const long long o = (((long long)n * H + py) * W + px) * Cout + co;
        float v = acc[i][jn][r] + bb;
        if (relu) v = fmaxf(v, 0.0f);
        if (mask) v = mask[o] > 0.0f ? v : 0.0f;
        y[o] = v;
      }
    }
  }
}

template <int TN, int TH, int WR, int WC>
int32_t launch_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, const float* mask, float* y, int n, int h,
                   int wd, int cin, int cout, int relu, hipStream_t s) {
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH;
  dim3 grid((unsigned)(tiles_x * tiles_y * n), (unsigned)(cout / TN));
  hipLaunchKernelGGL((conv3x3_mfma_kernel<TN, TH, WR, WC>), grid, dim3(256), 0, s, x, w, bias, mask, y, n, h, wd, cin, cout, relu,
                     tiles_x, tiles_y);
  UNET_CHECK_LAUNCH(ctx, "conv3x3_mfma_fwd");
  return UNET_OK;
}

}  // namespace

bool mfma_conv3x3_supported(int cin, int cout) { return cin >= CK && (cin % CK) == 0 && (cout % 32) == 0; }

int32_t k_conv3x3_mfma_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, const float* mask, float* y, int n,
                           int h, int wd, int cin, int cout, int relu, hipStream_t s) {
  if (!mfma_conv3x3_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 mfma: cin=%d cout=%d unsupported", cin, cout);
  if (cout % 128 == 0) return launch_fwd<128, 4, 2, 2>(ctx, x, w, bias, mask, y, n, h, wd, cin, cout, relu, s);
  if (cout % 64 == 0) return launch_fwd<64, 8, 4, 1>(ctx, x, w, bias, mask, y, n, h, wd, cin, cout, relu, s);
  return launch_fwd<32, 8, 4, 1>(ctx, x, w, bias, mask, y, n, h, wd, cin, cout, relu, s);
}

size_t mfma_wgrad_ws_bytes(int, int, int, int, int) { return 0; }
int32_t k_conv3x3_mfma_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void*, size_t, int n, int h,
                             int wd, int cin, int cout, hipStream_t s) {
  return k_conv3x3_naive_wgrad(ctx, x, dy, dw, db, n, h, wd, cin, cout, s);
}
