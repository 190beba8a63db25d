// 3x3 convolution with the Winograd F(2,3) transform along the image row, on the fp32 matrix cores.
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector rate (157 TFLOP/s), so the 3x3 convolutions of the U-Net are
// MFMA-bound: the only way below the direct-convolution floor is fewer multiplies.  F(2,3) along x:
// two adjacent outputs of a row need 4 multiplies per kernel row instead of 6 -> 12 instead of 18 MFMAs per pair of output
// pixels, 1.5x less matrix work, with +-1 / 0.5 transform coefficients only (fp32 round-off stays ~1e-7 relative; the 2-D
// F(2x2,3x3) would save 2.25x but needs 16 live accumulator tiles per output tile and leaves no room for register blocking).
//
//   tile t of a row = output columns (2t, 2t+1), input columns d0..d3 = 2t-1 .. 2t+2
//   V_k[t]  : d0-d2, d1+d2, d2-d1, d1-d3                      (input transform, done while staging the LDS patch)
//   U_k[ky] : g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2              (weight transform, wino_weight_kernel, once per launch)
//   M_k[t]  = sum_{ky, ci} V_k[row+ky][t][ci] * U_k[ky][ci][co]   -> 12 "taps" (ky, k), each an MFMA GEMM like the direct kernel
//   out(2t) = M0+M1+M2, out(2t+1) = M1-M2-M3                  (output transform, in registers, in the epilogue)
//
// Same skeleton as conv_mfma_kernel: 256 threads = 4 waves, MFMA M-tile = 32 consecutive TILES of one image row (64 output
// columns), K loop over 8-channel chunks, transformed patch [(TH+2) rows][4][32 tiles][12] and weight slab [12][8][TN] in LDS,
// k-pairing (lanes 0-31 channel j, 32-63 channel 4+j), register prefetch of the next chunk, quad-transpose 16-byte epilogue.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = UNET_OOB;
constexpr int CK = 8;     // input channels per K chunk
constexpr int CKP = 12;   // padded channel stride in LDS (floats): conflict-free ds_read_b128
constexpr int WT = 32;    // Winograd tiles per MFMA M-tile

// u[ky][k][ci][co] (12 x Cin' x Cout') from the Keras kernel w[ky][kx][cin][cout].
// flip = 0: forward (Cin' = cin, Cout' = cout).  flip = 1: data gradient, the convolution of dy with the flipped/transposed
// kernel g[ky][kx][c'][o'] = w[2-ky][2-kx][o'][c']  (Cin' = cout, Cout' = cin).
__device__ __forceinline__ void wino_weight_body(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int flip, int two_d) {
  const int ci2 = flip ? cout : cin, co2 = flip ? cin : cout;
  const int total = ci2 * co2;
  const long long st = (long long)ci2 * co2;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i % co2; const int c = i / co2;
    float u1[3][4];                                    // x-transformed rows ky = 0..2
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      float g0, g1, g2;
      if (!flip) {
        const float* p = w + ((long long)(ky * 3) * cin + c) * cout + o;
        g0 = p[0]; g1 = p[(long long)cin * cout]; g2 = p[2LL * cin * cout];
      } else {
        const float* p = w + ((long long)((2 - ky) * 3) * cin + o) * cout + c;
        g0 = p[2LL * cin * cout]; g1 = p[(long long)cin * cout]; g2 = p[0];
      }
      u1[ky][0] = g0; u1[ky][1] = 0.5f * (g0 + g1 + g2); u1[ky][2] = 0.5f * (g0 - g1 + g2); u1[ky][3] = g2;
    }
    float* q = u + (long long)c * co2 + o;
    if (!two_d) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int k = 0; k < 4; ++k) q[(ky * 4 + k) * st] = u1[ky][k];
    } else {                                           // the same transform once more along ky: 4 x 4 taps
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        q[(0 * 4 + k) * st] = u1[0][k];
        q[(1 * 4 + k) * st] = 0.5f * (u1[0][k] + u1[1][k] + u1[2][k]);
        q[(2 * 4 + k) * st] = 0.5f * (u1[0][k] - u1[1][k] + u1[2][k]);
        q[(3 * 4 + k) * st] = u1[2][k];
      }
    }
  }
}
__global__ void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ u, int cin, int cout, int flip, int two_d) {
  wino_weight_body(w, u, cin, cout, flip, two_d);
}
// all layers of a program in ONE launch (blockIdx.y = layer): the per-layer launches were 34 x 6 us of launch latency per step
__global__ void wino_weight_multi_kernel(unet_wino_prep_list L) {
  const unet_wino_prep& p = L.item[blockIdx.y];
  wino_weight_body(p.w, p.u, p.cin, p.cout, p.flip, p.two_d);
}

template <int TN, int TH, int WR, int WC, bool GEN>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                           const float* __restrict__ bias, const float* __restrict__ mask,
                                                           float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int act,
                                                           int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y) {
  static_assert(WR * WC == 4, "4 waves");
  constexpr int TAPS = 12;
  constexpr int RW = TH / WR, NW = TN / 32 / WC;
  constexpr int ROWF = 4 * WT * CKP;                        // floats per transformed patch row
  constexpr int ITEMS = (TH + 2) * WT * 2;                  // (row, tile, channel quad) staging items
  constexpr int WTOT = TAPS * CK * (TN / 4);
  constexpr int PL = (ITEMS + 255) / 256, WL = (WTOT + 255) / 256;
  static_assert(RW >= 1 && NW >= 1, "tile");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem;                                         // [(TH+2)][4][WT][CKP]
  float* s_u = smem + (TH + 2) * ROWF;                       // [12][CK][TN]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform, and the compiler is told so (scalar address math)
  const int l31 = lane & 31, hi = lane >> 5;
  const int wr = wave / WC, wc = wave % WC;
  // XCD-aware block -> tile map (workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2): an XCD walks a contiguous run of
  // pixel tiles ordered with ty fastest (vertical neighbours share 2 of their 4 patch rows) and does ALL cout groups of a tile
  // back to back, so the re-reads of an input patch (once per cout group, plus the halo rows) hit that XCD's L2.
  int n, tx, ty, nbase;
  {
    const int G = (Cout + TN - 1) / TN, P = tiles_x * tiles_y * N, per = (P + 7) >> 3;
    const int L = blockIdx.x, xcd = L & 7, sq = L >> 3;
    const int lt = sq / G, g = sq - lt * G;
    const int pt = xcd * per + lt;
    if (pt >= P) return;                               // grid is padded to 8 * per * G workgroups
    ty = pt % tiles_y; const int r = pt / tiles_y;
    tx = r % tiles_x; n = r / tiles_x;
    nbase = g * TN;
  }
  const int x0 = tx * 2 * WT, y0 = ty * TH;

  f32x16 acc[RW][NW][4];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.0f;

  // ---- staging plan: item = (patch row r, tile t, channel quad q): 4 input pixels d0..d3 -> 4 transformed values.
  // Byte offsets into the image / the weight tensor; out-of-image pixels and overhanging channels are OOB (-> 0).
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (long long)n * H * W * Cin, (long long)H * W * Cin * 4);
  const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(u, 12LL * Cin * Cout * 4);
  int poff[PL][4], plds[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = min(tid + k * 256, ITEMS - 1);      // surplus threads redo the last item (same data, same slot): no branch in the loop
    const int q = idx & 1, t = (idx >> 1) & (WT - 1), r = idx >> 6;
    const int gy = y0 + r - 1, gx = x0 + 2 * t - 1;
    const bool rok = gy >= 0 && gy < H;
#pragma unroll
    for (int d = 0; d < 4; ++d) poff[k][d] = (rok && gx + d >= 0 && gx + d < W) ? ((gy * W + gx + d) * Cin + q * 4) * 4 : OOB;
    plds[k] = (r * 4 * WT + t) * CKP + q * 4;
  }
  int woff[WL], wlds[WL];
#pragma unroll
  for (int k = 0; k < WL; ++k) {
    const int idx = min(tid + k * 256, WTOT - 1);
    const int q = idx % (TN / 4), row = idx / (TN / 4);
    const int tap = row >> 3, ci = row & 7;
    woff[k] = (nbase + q * 4 < Cout) ? ((tap * Cin + ci) * Cout + nbase + q * 4) * 4 : OOB;
    wlds[k] = row * TN + q * 4;
  }
  f32x4 preg[PL][4], wreg[WL];
  auto issue_loads = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k)
#pragma unroll
      for (int d = 0; d < 4; ++d) preg[k][d] = buf_ld4(rs_x, poff[k][d], c0 * 4);
#pragma unroll
    for (int k = 0; k < WL; ++k) wreg[k] = buf_ld4(rs_u, woff[k], c0 * Cout * 4);
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      float* p = s_v + plds[k];
      // (the empty asm pins each transformed quad in a register tuple: without it the backend scalarises the four stores and
      //  re-pairs them ACROSS the k planes into 8 ds_write2st64_b32 instead of 4 ds_write_b128)
      f32x4 t0 = preg[k][0] - preg[k][2], t1 = preg[k][1] + preg[k][2], t2 = preg[k][2] - preg[k][1], t3 = preg[k][1] - preg[k][3];
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      *reinterpret_cast<f32x4*>(p) = t0;
      *reinterpret_cast<f32x4*>(p + WT * CKP) = t1;
      *reinterpret_cast<f32x4*>(p + 2 * WT * CKP) = t2;
      *reinterpret_cast<f32x4*>(p + 3 * WT * CKP) = t3;
    }
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      *reinterpret_cast<f32x4*>(&s_u[wlds[k]]) = wreg[k];
    }
  };

  issue_loads(0);
  for (int c0 = 0; c0 < Cin; c0 += CK) {
    store_lds();
    __syncthreads();
    if (c0 + CK < Cin) issue_loads(c0 + CK);             // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 a[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i)
          a[i] = *reinterpret_cast<const f32x4*>(&s_v[(((wr * RW + i + ky) * 4 + k) * WT + l31) * CKP + hi * 4]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float bv[NW];
#pragma unroll
          for (int jn = 0; jn < NW; ++jn) bv[jn] = s_u[((ky * 4 + k) * CK + j + 4 * hi) * TN + (wc * NW + jn) * 32 + l31];
#pragma unroll
          for (int i = 0; i < RW; ++i)
#pragma unroll
            for (int jn = 0; jn < NW; ++jn)
              acc[i][jn][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][j], bv[jn], acc[i][jn][k], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: output transform in registers, then the quad transpose of conv_mfma_kernel (MFMA D layout: lane (l31, hi),
  // register r holds D[tile (r&3)+8*(r>>2)+4*hi][cout l31]) -> every lane owns 4 consecutive couts of one tile -> 16-byte stores
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
#pragma unroll
  for (int jn = 0; jn < NW; ++jn) {
    const int co = nbase + (wc * NW + jn) * 32 + q4;
    const float4 bb = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RW; ++i) {
      const int py = y0 + wr * RW + i;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float m0 = acc[i][jn][0][4 * g + r], m1 = acc[i][jn][1][4 * g + r], m2 = acc[i][jn][2][4 * g + r], m3 = acc[i][jn][3][4 * g + r];
            v[r] = half == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
          }
          float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
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
          const int px = x0 + 2 * (e + 8 * g + 4 * hi) + half;
          if (py >= H || px >= W || co >= Cout) continue;
          float4 o4 = make_float4(v0 + bb.x, v1 + bb.y, v2 + bb.z, v3 + bb.w);
          const long long o = (((long long)n * H + py) * W + px) * Cout + co;
          if (!GEN) {
            if (act == ACT_RELU) { o4.x = fmaxf(o4.x, 0.f); o4.y = fmaxf(o4.y, 0.f); o4.z = fmaxf(o4.z, 0.f); o4.w = fmaxf(o4.w, 0.f); }
            if (mask_mode == MASK_RELU) {
              const float4 m = *reinterpret_cast<const float4*>(mask + o);
              o4.x = m.x > 0.f ? o4.x : 0.f; o4.y = m.y > 0.f ? o4.y : 0.f; o4.z = m.z > 0.f ? o4.z : 0.f; o4.w = m.w > 0.f ? o4.w : 0.f;
            }
          } else {
            o4.x = apply_act(o4.x, act); o4.y = apply_act(o4.y, act); o4.z = apply_act(o4.z, act); o4.w = apply_act(o4.w, act);
            if (mask_mode == MASK_NONE) {
              if (rate > 0.0f) { const float4 ks = keep_scale(o >> 2, rate, seed); o4.x *= ks.x; o4.y *= ks.y; o4.z *= ks.z; o4.w *= ks.w; }
            } else {
              const float4 m = *reinterpret_cast<const float4*>(mask + o);
              float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
              if (mask_mode == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
              o4.x *= mask_factor(m.x, mask_mode, ks.x, rate); o4.y *= mask_factor(m.y, mask_mode, ks.y, rate);
              o4.z *= mask_factor(m.z, mask_mode, ks.z, rate); o4.w *= mask_factor(m.w, mask_mode, ks.w, rate);
            }
          }
          *reinterpret_cast<float4*>(y + o) = o4;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// F(2x2,3x3): the same transform along y as well -> 16 instead of 24 MFMAs per 2x2 output tile (direct: 36).  The 16
// accumulator tiles M[ky'][k] of a (row pair x 32 tiles x 32 couts) tile are split between TWO waves by ky' (kh = 0: ky' 0,1;
// kh = 1: ky' 2,3 -> 128 accumulator registers each); the y-transform of the input is done at operand-read time from the
// x-transformed LDS rows (R0 = V0-V2, R1 = V1+V2, R2 = V2-V1, R3 = V1-V3: one extra ds_read_b128 + 4 subtractions per operand);
// the output transform needs both halves: out(row 0) = M0+M1+M2, out(row 1) = M1-M2-M3, so the two waves swap one partial
// each through LDS in the epilogue (kh = 0 finishes row 0, kh = 1 row 1).  Waves = 2 (kh) x WM (row pairs) x WC (cout tiles).
// ---------------------------------------------------------------------------------------------------------------------------
// BREG: every weight element of the slab is used by exactly ONE wave (its 8 taps x its 32 couts), so the B operands go straight
// from L2 into registers in MFMA operand layout (lane = cout, 128-B coalesced rows) and never touch LDS: no weight slab, no LDS
// writes/reads for it, and the next chunk's value is loaded into the same register right behind the MFMA that consumed it.
// WTT = Winograd tiles per row of the MFMA M-tile: 32 (one row pair x 64 columns) or 16 (two row pairs x 32 columns, for images
// narrower than 64 columns: lane l31 -> row pair l31 / 16, tile l31 % 16).
// CKV = input channels per staged chunk: 8, or 16 (BREG only, Cin % 16 == 0): half as many barriers per MFMA.
template <int WTT, int WM, int WC, bool GEN, bool BREG, int CKV>
__global__ __launch_bounds__(256, 2) void conv_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                             const float* __restrict__ bias, const float* __restrict__ mask,
                                                             float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int act,
                                                             int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y) {
  static_assert(WM * WC == 2, "4 waves = 2 x WM x WC");
  constexpr int WT = WTT, RPW = 32 / WTT;                    // (shadows the file-level WT) row pairs per wave tile
  constexpr int TAPS = 16, TH = 2 * RPW * WM, TN = 32 * WC;
  static_assert(CKV == 8 || (CKV == 16 && BREG), "16-channel chunks only with register-resident weights");
  constexpr int CKP = CKV + 4, QPI = CKV / 4;                // (shadows the file-level CKP) padded channel stride, channel quads per pixel
  constexpr int ROWF = 4 * WT * CKP;
  constexpr int ITEMS = (TH + 2) * WT * QPI;
  constexpr int WTOT = TAPS * CK * (TN / 4);
  constexpr int PL = (ITEMS + 255) / 256, WL = BREG ? 0 : (WTOT + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem;                                         // [(TH+2)][4][WT][CKP]  x-transformed patch rows
  float* s_u = smem + (TH + 2) * ROWF;                       // [16][CK][TN]   (!BREG only)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: the kh branches below are scalar branches
  const int l31 = lane & 31, hi = lane >> 5;
  const int kh = wave & 1, wm = (wave >> 1) / WC, wc = (wave >> 1) % WC;
  // XCD-aware block -> tile map (workgroup b runs on XCD b % 8, each XCD has its own 4 MiB L2): an XCD walks a contiguous run of
  // pixel tiles ordered with ty fastest (vertical neighbours share 2 of their 4 patch rows) and does ALL cout groups of a tile
  // back to back, so the re-reads of an input patch (once per cout group, plus the halo rows) hit that XCD's L2.
  int n, tx, ty, nbase;
  {
    const int G = (Cout + TN - 1) / TN, P = tiles_x * tiles_y * N, per = (P + 7) >> 3;
    const int L = blockIdx.x, xcd = L & 7, sq = L >> 3;
    const int lt = sq / G, g = sq - lt * G;
    const int pt = xcd * per + lt;
    if (pt >= P) return;                               // grid is padded to 8 * per * G workgroups
    ty = pt % tiles_y; const int r = pt / tiles_y;
    tx = r % tiles_x; n = r / tiles_x;
    nbase = g * TN;
  }
  const int x0 = tx * 2 * WT, y0 = ty * TH;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (long long)n * H * W * Cin, (long long)H * W * Cin * 4);
  const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(u, 16LL * Cin * Cout * 4);
  int poff[PL][4], plds[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = min(tid + k * 256, ITEMS - 1);      // surplus threads redo the last item (same data, same slot): no branch in the loop
    const int q = idx % QPI, t = (idx / QPI) % WT, r = idx / (QPI * WT);
    const int gy = y0 + r - 1, gx = x0 + 2 * t - 1;
    const bool rok = gy >= 0 && gy < H;
#pragma unroll
    for (int d = 0; d < 4; ++d) poff[k][d] = (rok && gx + d >= 0 && gx + d < W) ? ((gy * W + gx + d) * Cin + q * 4) * 4 : OOB;
    plds[k] = (r * 4 * WT + t) * CKP + q * 4;
  }
  int woff[WL + 1], wlds[WL + 1];
#pragma unroll
  for (int k = 0; k < WL; ++k) {
    const int idx = min(tid + k * 256, WTOT - 1);
    const int q = idx % (TN / 4), row = idx / (TN / 4);
    const int tap = row >> 3, ci = row & 7;
    woff[k] = (nbase + q * 4 < Cout) ? ((tap * Cin + ci) * Cout + nbase + q * 4) * 4 : OOB;
    wlds[k] = row * TN + q * 4;
  }
  // BREG: per-lane byte offset of U[tap(kk,k)][ci = 4*hi][cout of this lane]; + (c0 + j) * Cout * 4 (uniform, SGPR operand)
  int boff[2][4];
  float breg[2][4][4];                                     // [kk][k][j]: B operand of the MFMA (kk, k, j) of the current chunk
  if (BREG) {
    const int co = nbase + wc * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int k = 0; k < 4; ++k) boff[kk][k] = co < Cout ? (((((kh ? 3 - kk : kk) * 4 + k) * Cin + 4 * hi) * Cout) + co) * 4 : OOB;
  }
  f32x4 preg[PL][4], wreg[WL + 1];
  auto issue_loads = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k)
#pragma unroll
      for (int d = 0; d < 4; ++d) preg[k][d] = buf_ld4(rs_x, poff[k][d], c0 * 4);
#pragma unroll
    for (int k = 0; k < WL; ++k) wreg[k] = buf_ld4(rs_u, woff[k], c0 * Cout * 4);
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      float* p = s_v + plds[k];
      // (the empty asm pins each transformed quad in a register tuple: without it the backend scalarises the four stores and
      //  re-pairs them ACROSS the k planes into 8 ds_write2st64_b32 instead of 4 ds_write_b128)
      f32x4 t0 = preg[k][0] - preg[k][2], t1 = preg[k][1] + preg[k][2], t2 = preg[k][2] - preg[k][1], t3 = preg[k][1] - preg[k][3];
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      *reinterpret_cast<f32x4*>(p) = t0;
      *reinterpret_cast<f32x4*>(p + WT * CKP) = t1;
      *reinterpret_cast<f32x4*>(p + 2 * WT * CKP) = t2;
      *reinterpret_cast<f32x4*>(p + 3 * WT * CKP) = t3;
    }
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      *reinterpret_cast<f32x4*>(&s_u[wlds[k]]) = wreg[k];
    }
  };

  issue_loads(0);
  if (BREG) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) breg[kk][k][j] = buf_ld1(rs_u, boff[kk][k], j * Cout * 4);
  }
  for (int c0 = 0; c0 < Cin; c0 += CKV) {
    store_lds();
    __syncthreads();
    if (c0 + CKV < Cin) issue_loads(c0 + CKV);
    __builtin_amdgcn_s_setprio(2);                         // MFMA phase outranks the other wave's staging phase at the issue arbiter
    __builtin_amdgcn_iglp_opt(0);                          // scheduler hint "small GEMM": LDS operand reads interleaved with the MFMAs of the block (-5 % on the conv launches)
#pragma unroll
    for (int sub = 0; sub < CKV / 8; ++sub) {
    const int cb = c0 + sub * 8;
    const int cn = cb + 8 < Cin ? cb + 8 : cb;              // BREG: 8-channel group whose weights are fetched behind the MFMAs (the last one re-reads itself)
    // Rows A, B, C = patch rows kh, kh+1, kh+2 of this wave's row pair.  kh = 0 (ky' 0,1): R0 = A-C, R1 = B+C.  kh = 1 (ky' 2,3):
    // R3 = A-C, R2 = B-A.  So both halves compute d = A-C and e = B + sgn*Z (Z = kh ? A : C) -- no branch, the loop body stays one
    // basic block -- and kh = 1 simply walks its two taps in the order (3, 2): tap of slot kk = kh ? 3-kk : kk.
    // k in pairs: 4 accumulators (kk, k) in rotation -> no back-to-back MFMAs on one accumulator (16-pass dependency stall).
    const float sgn = kh ? -1.0f : 1.0f;
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      f32x4 a[2][2];                                            // [kk][k - 2kp]
#pragma unroll
      for (int kq = 0; kq < 2; ++kq) {
        const float* vb = &s_v[(((2 * (wm * RPW + l31 / WT) + kh) * 4 + 2 * kp + kq) * WT + l31 % WT) * CKP + sub * 8 + hi * 4];
        const f32x4 A = *reinterpret_cast<const f32x4*>(vb), B = *reinterpret_cast<const f32x4*>(vb + ROWF), C = *reinterpret_cast<const f32x4*>(vb + 2 * ROWF);
        const f32x4 Z = kh ? A : C;
        a[0][kq] = A - C;
        a[1][kq] = B + sgn * Z;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int kq = 0; kq < 2; ++kq) {
            const int k = 2 * kp + kq;
            if (BREG) {
              acc[kk][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][kq][j], breg[kk][k][j], acc[kk][k], 0, 0, 0);
              breg[kk][k][j] = buf_ld1(rs_u, boff[kk][k], (cn + j) * Cout * 4);
            } else {
              const float bv = s_u[(((kh ? 3 - kk : kk) * 4 + k) * CK + j + 4 * hi) * TN + wc * 32 + l31];
              acc[kk][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk][kq][j], bv, acc[kk][k], 0, 0, 0);
            }
          }
        }
      }
    }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- epilogue.  x-direction output transform per accumulator pair, then the y-direction combination across the two waves.
  //   T[kk][half] = half 0: m0+m1+m2, half 1: m1-m2-m3 (over k).   kh=0 holds M0,M1: row0 partial T0+T1, row1 partial T1.
  //   kh=1 holds M3,M2 (slots 0,1): row0 partial T1 (= M2), row1 partial -(T0+T1).   Each wave keeps the partial of ITS output row (row kh)
  //   and sends the other one to its partner (wave ^ 1) through LDS (the operand buffers are dead after the last barrier).
  float* xb = smem + wave * (32 * 64);                       // this wave's outbox: [32 values][64 lanes]
  const float* pb = smem + (wave ^ 1) * (32 * 64);
  f32x16 mine[2];                                            // [half]
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t[2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const float m0 = acc[kk][0][r], m1 = acc[kk][1][r], m2 = acc[kk][2][r], m3 = acc[kk][3][r];
        t[kk] = half == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
      }
      float keep, send;
      if (kh == 0) { keep = t[0] + t[1]; send = t[1]; } else { keep = -(t[0] + t[1]); send = t[1]; }      // kh = 1: t[0] = M3, t[1] = M2
      mine[half][r] = keep;
      xb[(half * 16 + r) * 64 + lane] = send;
    }
  }
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
  const int co = nbase + wc * 32 + q4;
  // the per-element epilogue input (ReLU mask of a data gradient / x of a folded BatchNorm backward) is requested BEFORE the exchange barrier: its HBM
  // latency overlaps the exchange and the transposes instead of ending every workgroup (the accumulators are dead, registers are free)
  float4 mpre[4][2];
  const bool want_m = mask_mode != MASK_NONE && mask_mode != MASK_BIAS_TAB;
  if (want_m) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int mt = e + 8 * g + 4 * hi;
        const int px = x0 + 2 * (mt % WT) + half, py = y0 + 2 * (wm * RPW + mt / WT) + kh;
        mpre[g][half] = (py < H && px < W && co < Cout) ? *reinterpret_cast<const float4*>(mask + (((long long)n * H + py) * W + px) * Cout + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
  }
  __syncthreads();
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[half][r] += pb[(half * 16 + r) * 64 + lane];

  const float4 bb = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 kb1 = bb, kb2 = bb;                                 // MASK_BN_BWD: bb = K0, these K1, K2
  if (mask_mode >= MASK_BN_BWD && co < Cout) { kb1 = *reinterpret_cast<const float4*>(bias + Cout + co); kb2 = *reinterpret_cast<const float4*>(bias + 2 * Cout + co); }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float v0 = mine[half][4 * g + 0], v1 = mine[half][4 * g + 1], v2 = mine[half][4 * g + 2], v3 = mine[half][4 * g + 3];
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
      const int mt = e + 8 * g + 4 * hi;                      // M index of this value: row pair mt / WT, tile mt % WT
      const int px = x0 + 2 * (mt % WT) + half, py = y0 + 2 * (wm * RPW + mt / WT) + kh;
      if (py >= H || px >= W || co >= Cout) continue;
      float4 bs = bb;
      if (mask_mode == MASK_BIAS_TAB) {                       // folded input BatchNorm: border pixels see fewer taps of the shift
        const int cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px == 0) | ((px == W - 1) << 1));
        if (cls) bs = *reinterpret_cast<const float4*>(mask + (long long)cls * Cout + co);
      }
      float4 o4 = make_float4(v0 + bs.x, v1 + bs.y, v2 + bs.z, v3 + bs.w);
      const long long o = (((long long)n * H + py) * W + px) * Cout + co;
      if (!GEN) {
        if (mask_mode == MASK_BN_BWD || mask_mode == MASK_BN_BWD_RELU) {      // the folded BatchNorm's backward: dx = K0 * dz + K1 * x + K2 (x = relu(.): masked)
          const float4 m = mpre[g][half];
          o4.x = fmaf(bb.x, v0, fmaf(kb1.x, m.x, kb2.x)); o4.y = fmaf(bb.y, v1, fmaf(kb1.y, m.y, kb2.y));
          o4.z = fmaf(bb.z, v2, fmaf(kb1.z, m.z, kb2.z)); o4.w = fmaf(bb.w, v3, fmaf(kb1.w, m.w, kb2.w));
          if (mask_mode == MASK_BN_BWD_RELU) { o4.x = m.x > 0.f ? o4.x : 0.f; o4.y = m.y > 0.f ? o4.y : 0.f; o4.z = m.z > 0.f ? o4.z : 0.f; o4.w = m.w > 0.f ? o4.w : 0.f; }
        }
        if (act == ACT_RELU) { o4.x = fmaxf(o4.x, 0.f); o4.y = fmaxf(o4.y, 0.f); o4.z = fmaxf(o4.z, 0.f); o4.w = fmaxf(o4.w, 0.f); }
        if (mask_mode == MASK_RELU) {
          const float4 m = mpre[g][half];
          o4.x = m.x > 0.f ? o4.x : 0.f; o4.y = m.y > 0.f ? o4.y : 0.f; o4.z = m.z > 0.f ? o4.z : 0.f; o4.w = m.w > 0.f ? o4.w : 0.f;
        }
      } else {
        o4.x = apply_act(o4.x, act); o4.y = apply_act(o4.y, act); o4.z = apply_act(o4.z, act); o4.w = apply_act(o4.w, act);
        if (mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) {
          if (rate > 0.0f) { const float4 ks = keep_scale(o >> 2, rate, seed); o4.x *= ks.x; o4.y *= ks.y; o4.z *= ks.z; o4.w *= ks.w; }
        } else if (mask_mode >= MASK_BN_BWD) {                // folded BatchNorm backward, then the ELU (+ dropout) derivative of x's producer
          const float4 m = mpre[g][half];
          const int mm = mask_mode == MASK_BN_BWD_ELU_DROP ? MASK_ELU_DROP : MASK_ELU;
          float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
          if (mm == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
          o4.x = fmaf(bb.x, v0, fmaf(kb1.x, m.x, kb2.x)) * mask_factor(m.x, mm, ks.x, rate); o4.y = fmaf(bb.y, v1, fmaf(kb1.y, m.y, kb2.y)) * mask_factor(m.y, mm, ks.y, rate);
          o4.z = fmaf(bb.z, v2, fmaf(kb1.z, m.z, kb2.z)) * mask_factor(m.z, mm, ks.z, rate); o4.w = fmaf(bb.w, v3, fmaf(kb1.w, m.w, kb2.w)) * mask_factor(m.w, mm, ks.w, rate);
        } else {
          const float4 m = mpre[g][half];
          float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
          if (mask_mode == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
          o4.x *= mask_factor(m.x, mask_mode, ks.x, rate); o4.y *= mask_factor(m.y, mask_mode, ks.y, rate);
          o4.z *= mask_factor(m.z, mask_mode, ks.z, rate); o4.w *= mask_factor(m.w, mask_mode, ks.w, rate);
        }
      }
      *reinterpret_cast<float4*>(y + o) = o4;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Same F(2x2,3x3) math, accumulators split FOUR ways: wave ky' (0..3) of a workgroup owns the 4 taps (ky', k = 0..3) of one
// (row pair x 32 tiles x 32 couts) tile -> 64 accumulator registers, ~150 VGPRs -> three workgroups per CU instead of two.
// Every weight of the slab belongs to exactly one wave (register-resident B operands, as BREG above).  A operand of wave ky':
// R = row[ra] + sg * row[rb] with (ra, rb, sg) = (0,2,-) (1,2,+) (2,1,-) (1,3,-): two ds_read_b128 + one fma per operand quad.
// Epilogue: each wave x-transforms its M_ky' and posts both halves in LDS; wave w then finishes (output row w & 1, column parity
// w >> 1) = sum over ky' of c[row][ky'] * T_ky' with c[0] = (1,1,1,0), c[1] = (0,1,-1,-1).
// ---------------------------------------------------------------------------------------------------------------------------
template <int WTT, bool GEN, int CKV>
__global__ __launch_bounds__(256, 3) void conv_wino2d4_kernel(const float* __restrict__ x, const float* __restrict__ u,
                                                              const float* __restrict__ bias, const float* __restrict__ mask,
                                                              float* __restrict__ y, int N, int H, int W, int Cin, int Cout, int act,
                                                              int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y) {
  constexpr int WT = WTT, RPW = 32 / WTT, TH = 2 * RPW, TN = 32;
  constexpr int CKP = CKV + 4, QPI = CKV / 4;
  constexpr int ROWF = 4 * WT * CKP;
  constexpr int ITEMS = (TH + 2) * WT * QPI;
  constexpr int PL = (ITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_v = smem;                                         // [(TH+2)][4][WT][CKP]  x-transformed patch rows

  const int tid = threadIdx.x, lane = tid & 63;
  const int kq = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's ky'
  const int l31 = lane & 31, hi = lane >> 5;
  int n, tx, ty, nbase;
  {
    const int G = (Cout + TN - 1) / TN, P = tiles_x * tiles_y * N, per = (P + 7) >> 3;
    const int L = blockIdx.x, xcd = L & 7, sq = L >> 3;
    const int lt = sq / G, g = sq - lt * G;
    const int pt = xcd * per + lt;
    if (pt >= P) return;
    ty = pt % tiles_y; const int r = pt / tiles_y;
    tx = r % tiles_x; n = r / tiles_x;
    nbase = g * TN;
  }
  const int x0 = tx * 2 * WT, y0 = ty * TH;

  f32x16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (long long)n * H * W * Cin, (long long)H * W * Cin * 4);
  const __amdgpu_buffer_rsrc_t rs_u = make_rsrc(u, 16LL * Cin * Cout * 4);
  int poff[PL][4], plds[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = min(tid + k * 256, ITEMS - 1);
    const int q = idx % QPI, t = (idx / QPI) % WT, r = idx / (QPI * WT);
    const int gy = y0 + r - 1, gx = x0 + 2 * t - 1;
    const bool rok = gy >= 0 && gy < H;
#pragma unroll
    for (int d = 0; d < 4; ++d) poff[k][d] = (rok && gx + d >= 0 && gx + d < W) ? ((gy * W + gx + d) * Cin + q * 4) * 4 : OOB;
    plds[k] = (r * 4 * WT + t) * CKP + q * 4;
  }
  int boff[4];
  float breg[4][4];                                          // [k][j]
  {
    const int co = nbase + l31;
#pragma unroll
    for (int k = 0; k < 4; ++k) boff[k] = co < Cout ? ((((kq * 4 + k) * Cin + 4 * hi) * Cout) + co) * 4 : OOB;
  }
  f32x4 preg[PL][4];
  auto issue_loads = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k)
#pragma unroll
      for (int d = 0; d < 4; ++d) preg[k][d] = buf_ld4(rs_x, poff[k][d], c0 * 4);
  };
  auto store_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      float* p = s_v + plds[k];
      f32x4 t0 = preg[k][0] - preg[k][2], t1 = preg[k][1] + preg[k][2], t2 = preg[k][2] - preg[k][1], t3 = preg[k][1] - preg[k][3];
      asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      *reinterpret_cast<f32x4*>(p) = t0;
      *reinterpret_cast<f32x4*>(p + WT * CKP) = t1;
      *reinterpret_cast<f32x4*>(p + 2 * WT * CKP) = t2;
      *reinterpret_cast<f32x4*>(p + 3 * WT * CKP) = t3;
    }
  };
  // patch rows this wave combines: R_ky' = row[ra] + sg * row[rb]
  const int ra = kq == 0 ? 0 : (kq == 2 ? 2 : 1), rb = kq == 3 ? 3 : (kq == 2 ? 1 : 2);
  const float sg = kq == 1 ? 1.0f : -1.0f;

  issue_loads(0);
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) breg[k][j] = buf_ld1(rs_u, boff[k], j * Cout * 4);
  for (int c0 = 0; c0 < Cin; c0 += CKV) {
    store_lds();
    __syncthreads();
    if (c0 + CKV < Cin) issue_loads(c0 + CKV);
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int sub = 0; sub < CKV / 8; ++sub) {
      const int cb = c0 + sub * 8;
      const int cn = cb + 8 < Cin ? cb + 8 : cb;
      f32x4 a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float* vb = &s_v[((2 * (l31 / WT) * 4 + k) * WT + l31 % WT) * CKP + sub * 8 + hi * 4];
        const f32x4 A = *reinterpret_cast<const f32x4*>(vb + ra * ROWF), B = *reinterpret_cast<const f32x4*>(vb + rb * ROWF);
        a[k] = A + sg * B;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k][j], breg[k][j], acc[k], 0, 0, 0);
          breg[k][j] = buf_ld1(rs_u, boff[k], (cn + j) * Cout * 4);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // ---- epilogue: post T[half] = x-output-transform of this wave's M_ky', then wave w finishes (row w & 1, half w >> 1)
  float* xb = smem + kq * (32 * 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
    xb[r * 64 + lane] = (m0 + m1) + m2;
    xb[(16 + r) * 64 + lane] = (m1 - m2) - m3;
  }
  const int orow = kq & 1, half = kq >> 1;
  // the per-element epilogue input (ReLU mask / x of a folded BatchNorm backward) is requested before the exchange barrier (see conv_wino2d_kernel)
  float4 mpre[4];
  const bool want_m = mask_mode != MASK_NONE && mask_mode != MASK_BIAS_TAB;
  if (want_m) {
    const int co_ = nbase + (l31 & ~3);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int mt = (l31 & 3) + 8 * g + 4 * hi;
      const int px = x0 + 2 * (mt % WT) + half, py = y0 + 2 * (mt / WT) + orow;
      mpre[g] = (py < H && px < W && co_ < Cout) ? *reinterpret_cast<const float4*>(mask + (((long long)n * H + py) * W + px) * Cout + co_) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  f32x16 mine;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* q = smem + (half * 16 + r) * 64 + lane;
    const float t0 = q[0], t1 = q[32 * 64], t2 = q[2 * 32 * 64], t3 = q[3 * 32 * 64];
    mine[r] = orow == 0 ? (t0 + t1) + t2 : (t1 - t2) - t3;
  }
  const int e = l31 & 3, q4 = l31 & ~3;
  const bool odd1 = e & 1, odd2 = e & 2;
  const int co = nbase + q4;
  const float4 bb = (bias && co < Cout) ? *reinterpret_cast<const float4*>(bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
  float4 kb1 = bb, kb2 = bb;                                 // MASK_BN_BWD: bb = K0, these K1, K2
  if (mask_mode >= MASK_BN_BWD && co < Cout) { kb1 = *reinterpret_cast<const float4*>(bias + Cout + co); kb2 = *reinterpret_cast<const float4*>(bias + 2 * Cout + co); }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v0 = mine[4 * g + 0], v1 = mine[4 * g + 1], v2 = mine[4 * g + 2], v3 = mine[4 * g + 3];
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
    const int mt = e + 8 * g + 4 * hi;
    const int px = x0 + 2 * (mt % WT) + half, py = y0 + 2 * (mt / WT) + orow;
    if (py >= H || px >= W || co >= Cout) continue;
    float4 bs = bb;
    if (mask_mode == MASK_BIAS_TAB) {                         // folded input BatchNorm: border pixels see fewer taps of the shift
      const int cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px == 0) | ((px == W - 1) << 1));
      if (cls) bs = *reinterpret_cast<const float4*>(mask + (long long)cls * Cout + co);
    }
    float4 o4 = make_float4(v0 + bs.x, v1 + bs.y, v2 + bs.z, v3 + bs.w);
    const long long o = (((long long)n * H + py) * W + px) * Cout + co;
    if (!GEN) {
      if (mask_mode == MASK_BN_BWD || mask_mode == MASK_BN_BWD_RELU) {        // the folded BatchNorm's backward: dx = K0 * dz + K1 * x + K2 (x = relu(.): masked)
        const float4 m = mpre[g];
        o4.x = fmaf(bb.x, v0, fmaf(kb1.x, m.x, kb2.x)); o4.y = fmaf(bb.y, v1, fmaf(kb1.y, m.y, kb2.y));
        o4.z = fmaf(bb.z, v2, fmaf(kb1.z, m.z, kb2.z)); o4.w = fmaf(bb.w, v3, fmaf(kb1.w, m.w, kb2.w));
        if (mask_mode == MASK_BN_BWD_RELU) { o4.x = m.x > 0.f ? o4.x : 0.f; o4.y = m.y > 0.f ? o4.y : 0.f; o4.z = m.z > 0.f ? o4.z : 0.f; o4.w = m.w > 0.f ? o4.w : 0.f; }
      }
      if (act == ACT_RELU) { o4.x = fmaxf(o4.x, 0.f); o4.y = fmaxf(o4.y, 0.f); o4.z = fmaxf(o4.z, 0.f); o4.w = fmaxf(o4.w, 0.f); }
      if (mask_mode == MASK_RELU) {
        const float4 m = mpre[g];
        o4.x = m.x > 0.f ? o4.x : 0.f; o4.y = m.y > 0.f ? o4.y : 0.f; o4.z = m.z > 0.f ? o4.z : 0.f; o4.w = m.w > 0.f ? o4.w : 0.f;
      }
    } else {
      o4.x = apply_act(o4.x, act); o4.y = apply_act(o4.y, act); o4.z = apply_act(o4.z, act); o4.w = apply_act(o4.w, act);
      if (mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) {
        if (rate > 0.0f) { const float4 ks = keep_scale(o >> 2, rate, seed); o4.x *= ks.x; o4.y *= ks.y; o4.z *= ks.z; o4.w *= ks.w; }
      } else if (mask_mode >= MASK_BN_BWD) {                  // folded BatchNorm backward, then the ELU (+ dropout) derivative of x's producer
        const float4 m = mpre[g];
        const int mm = mask_mode == MASK_BN_BWD_ELU_DROP ? MASK_ELU_DROP : MASK_ELU;
        float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mm == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
        o4.x = fmaf(bb.x, v0, fmaf(kb1.x, m.x, kb2.x)) * mask_factor(m.x, mm, ks.x, rate); o4.y = fmaf(bb.y, v1, fmaf(kb1.y, m.y, kb2.y)) * mask_factor(m.y, mm, ks.y, rate);
        o4.z = fmaf(bb.z, v2, fmaf(kb1.z, m.z, kb2.z)) * mask_factor(m.z, mm, ks.z, rate); o4.w = fmaf(bb.w, v3, fmaf(kb1.w, m.w, kb2.w)) * mask_factor(m.w, mm, ks.w, rate);
      } else {
        const float4 m = mpre[g];
        float4 ks = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask_mode == MASK_ELU_DROP) ks = keep_scale(o >> 2, rate, seed);
        o4.x *= mask_factor(m.x, mask_mode, ks.x, rate); o4.y *= mask_factor(m.y, mask_mode, ks.y, rate);
        o4.z *= mask_factor(m.z, mask_mode, ks.z, rate); o4.w *= mask_factor(m.w, mask_mode, ks.w, rate);
      }
    }
    *reinterpret_cast<float4*>(y + o) = o4;
  }
}

template <int WTT, int CKV>
int32_t launch_wino2d4(unet_ctx* ctx, const float* x, const float* u, const float* bias, const float* mask, int mask_mode, float* y, int n, int h,
                       int wd, int cin, int cout, int act, float rate, unsigned long long seed, hipStream_t s) {
  if (!mask) mask_mode = MASK_NONE;
  constexpr int WT = WTT, TH = 2 * (32 / WTT), TN = 32;
  const int tiles_x = (wd + 2 * WT - 1) / (2 * WT), tiles_y = (h + TH - 1) / TH;
  const dim3 grid((unsigned)(8 * ((tiles_x * tiles_y * n + 7) / 8) * ((cout + TN - 1) / TN)));
  const size_t lds = std::max((size_t)((TH + 2) * 4 * WT * (CKV + 4)) * sizeof(float), (size_t)4 * 32 * 64 * sizeof(float));
  const bool gen = act == ACT_ELU || rate > 0.0f || mask_mode == MASK_ELU || mask_mode == MASK_ELU_DROP || mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP;
  UNET_BIG_LDS(ctx, (conv_wino2d4_kernel<WTT, false, CKV>), lds, "conv_wino2d4");
  UNET_BIG_LDS(ctx, (conv_wino2d4_kernel<WTT, true, CKV>), lds, "conv_wino2d4");
  if (gen) hipLaunchKernelGGL((conv_wino2d4_kernel<WTT, true, CKV>), grid, dim3(256), lds, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
  else hipLaunchKernelGGL((conv_wino2d4_kernel<WTT, false, CKV>), grid, dim3(256), lds, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
  UNET_CHECK_LAUNCH(ctx, "conv_wino2d4");
  return UNET_OK;
}

template <int WTT, int WM, int WC>
int32_t launch_wino2d(unet_ctx* ctx, const float* x, const float* u, const float* bias, const float* mask, int mask_mode, float* y, int n, int h,
                      int wd, int cin, int cout, int act, float rate, unsigned long long seed, hipStream_t s) {
  if (!mask) mask_mode = MASK_NONE;
  constexpr int WT = WTT, TH = 2 * (32 / WTT) * WM, TN = 32 * WC;
  const int tiles_x = (wd + 2 * WT - 1) / (2 * WT), tiles_y = (h + TH - 1) / TH;
  const dim3 grid((unsigned)(8 * ((tiles_x * tiles_y * n + 7) / 8) * ((cout + TN - 1) / TN)));       // see the block -> tile map in the kernel
  // weights via registers (BREG) or through LDS: registers win 3-8 % on the 64-wide tile (two cout tiles share the patch), the
  // 32-wide tile (two row pairs, 1.5x the patch traffic per wave) measured 2-5 % slower with it.  UNET_WINO_BREG=0/1 forces one.
  static const int breg_env = [] { const char* e = getenv("UNET_WINO_BREG"); return e ? atoi(e) : -1; }();
  const int breg = breg_env >= 0 ? breg_env : (WC == 2 ? 1 : 0);
  const size_t lds = std::max((size_t)((TH + 2) * 4 * WT * CKP + (breg ? 0 : 16 * CK * TN)) * sizeof(float), (size_t)4 * 32 * 64 * sizeof(float));   // >= the epilogue exchange
  const bool gen = act == ACT_ELU || rate > 0.0f || mask_mode == MASK_ELU || mask_mode == MASK_ELU_DROP || mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP;
  {
    const size_t big = (size_t)((TH + 2) * 4 * WT * CKP + 16 * CK * TN) * sizeof(float);
    UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, false, false, 8>), big, "conv_wino2d"); UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, true, false, 8>), big, "conv_wino2d");
    UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, false, true, 8>), big, "conv_wino2d"); UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, true, true, 8>), big, "conv_wino2d");
  }
  static const int ck16_env = [] { const char* e = getenv("UNET_WINO_CK16"); return e ? atoi(e) : 1; }();
  const bool ck16 = ck16_env && breg && (cin % 16) == 0;
  if (ck16) {
    const size_t lds16 = std::max((size_t)((TH + 2) * 4 * WT * 20) * sizeof(float), (size_t)4 * 32 * 64 * sizeof(float));
    UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, false, true, 16>), lds16, "conv_wino2d"); UNET_BIG_LDS(ctx, (conv_wino2d_kernel<WTT, WM, WC, true, true, 16>), lds16, "conv_wino2d");
    if (gen) hipLaunchKernelGGL((conv_wino2d_kernel<WTT, WM, WC, true, true, 16>), grid, dim3(256), lds16, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
    else hipLaunchKernelGGL((conv_wino2d_kernel<WTT, WM, WC, false, true, 16>), grid, dim3(256), lds16, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
    UNET_CHECK_LAUNCH(ctx, "conv_wino2d");
    return UNET_OK;
  }
#define UNET_LAUNCH_W2D(G_, B_) hipLaunchKernelGGL((conv_wino2d_kernel<WTT, WM, WC, G_, B_, 8>), grid, dim3(256), lds, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y)
  if (gen) { if (breg) UNET_LAUNCH_W2D(true, true); else UNET_LAUNCH_W2D(true, false); }
  else { if (breg) UNET_LAUNCH_W2D(false, true); else UNET_LAUNCH_W2D(false, false); }
#undef UNET_LAUNCH_W2D
  UNET_CHECK_LAUNCH(ctx, "conv_wino2d");
  return UNET_OK;
}

template <int TN, int TH, int WR, int WC>
int32_t launch_wino(unet_ctx* ctx, const float* x, const float* u, const float* bias, const float* mask, int mask_mode, float* y, int n, int h,
                    int wd, int cin, int cout, int act, float rate, unsigned long long seed, hipStream_t s) {
  if (!mask) mask_mode = MASK_NONE;
  const int tiles_x = (wd + 2 * WT - 1) / (2 * WT), tiles_y = (h + TH - 1) / TH;
  const dim3 grid((unsigned)(8 * ((tiles_x * tiles_y * n + 7) / 8) * ((cout + TN - 1) / TN)));       // see the block -> tile map in the kernel
  constexpr size_t lds = (size_t)((TH + 2) * 4 * WT * CKP + 12 * CK * TN) * sizeof(float);
  const bool gen = act == ACT_ELU || rate > 0.0f || mask_mode >= MASK_ELU;
  UNET_BIG_LDS(ctx, (conv_wino_kernel<TN, TH, WR, WC, false>), lds, "conv_wino"); UNET_BIG_LDS(ctx, (conv_wino_kernel<TN, TH, WR, WC, true>), lds, "conv_wino");
  if (gen) hipLaunchKernelGGL((conv_wino_kernel<TN, TH, WR, WC, true>), grid, dim3(256), lds, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
  else hipLaunchKernelGGL((conv_wino_kernel<TN, TH, WR, WC, false>), grid, dim3(256), lds, s, x, u, bias, mask, y, n, h, wd, cin, cout, act, mask_mode, rate, seed, tiles_x, tiles_y);
  UNET_CHECK_LAUNCH(ctx, "conv_wino");
  return UNET_OK;
}

}  // namespace

bool wino_conv3x3_supported(int cin, int cout) { return cin >= CK && (cin % CK) == 0 && cout >= 4 && (cout % 4) == 0; }
size_t wino_u_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

inline int wino_2d_mode() {          // UNET_WINO2D: 0 = F(2,3) along x only, 1 (default) = F(2x2,3x3) where the shape allows
  static const int v = [] { const char* e = getenv("UNET_WINO2D"); return e ? atoi(e) : 1; }();
  return v;
}
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

static bool use_2d(int h, int cout) { return wino_2d_mode() && h >= 2 && cout % 32 == 0; }
// columns per row tile the 2-D kernel uses for an image of width wd: 64, or 32 when that fills the last tile much better
int wino_tile_cols(int wd) {
  const double u64 = (double)wd / (64.0 * ((wd + 63) / 64)), u32 = (double)wd / (32.0 * ((wd + 31) / 32));
  return u32 > 1.1 * u64 ? 32 : 64;
}
bool wino_uses_2d(int h, int cout) { return use_2d(h, cout); }

int32_t k_bn_bwd_coef(unet_ctx* ctx, const float* bnp, const double* sums, double count, float* coef, int c, hipStream_t s) {
  if (!bnp || !sums || !coef || c < 1 || count < 1) UNET_FAIL(ctx, UNET_E_ARG, "bn_bwd_coef: bad args");
  hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((unsigned)((c + 127) / 128)), dim3(128), 0, s, bnp, sums, 1.0 / count, coef, c);
  UNET_CHECK_LAUNCH(ctx, "bn_bwd_coef");
  return UNET_OK;
}

size_t bn_fold_scratch_floats(int cin, int cout) { return (size_t)9 * cin * cout + 16 * (size_t)cout + (size_t)((cin + 31) / 32) * 9 * cout; }
// scratch = [w_scaled 9*cin*cout][table 16*cout][tap partials]: bn_fold_scratch_floats(cin, cout)
int32_t k_bn_fold_prepare(unet_ctx* ctx, const float* w, const float* bias, const float* scale, const float* shift, int cin, int cout, float* scratch, hipStream_t s) {
  if (!w || !scale || !shift || !scratch || cin < 1 || cout < 4 || (cout & 3)) UNET_FAIL(ctx, UNET_E_ARG, "bn_fold_prepare: bad args");
  float* w_scaled = scratch; float* table = scratch + (size_t)9 * cin * cout; float* part = table + 16 * (size_t)cout;
  const long long total4 = 9LL * cin * cout / 4;
  const int slices = (cin + 31) / 32;
  hipLaunchKernelGGL(bn_fold_scale_kernel, dim3((unsigned)std::min<long long>((total4 + 255) / 256, 2048)), dim3(256), 0, s, w, scale, w_scaled, cin, cout / 4, total4);
  hipLaunchKernelGGL(bn_fold_taps_kernel, dim3((unsigned)((cout + 63) / 64), 9, (unsigned)slices), dim3(256), 0, s, w, shift, part, cin, cout);
  hipLaunchKernelGGL(bn_fold_table_kernel, dim3((unsigned)((cout + 63) / 64)), dim3(576), 0, s, part, bias, table, slices, cout);
  UNET_CHECK_LAUNCH(ctx, "bn_fold_prepare");
  return UNET_OK;
}

// one launch for several layers: item k = (weights, scratch, cin, cout, flip, image rows) exactly as k_wino_weights takes them
int32_t k_wino_weights_multi(unet_ctx* ctx, unet_wino_prep_list* L, const int* h, hipStream_t s) {
  if (L->n < 1) return UNET_OK;
  // launches that run on the x3 kernels (bf16 matrix cores, exact 3-term split) get their split weight image instead -- same scratch, same call sites
  int hk[UNET_WINO_PREP_MAX];
  {
    const float* xw[UNET_WINO_PREP_MAX]; void* xi[UNET_WINO_PREP_MAX]; int xci[UNET_WINO_PREP_MAX], xco[UNET_WINO_PREP_MAX], xf[UNET_WINO_PREP_MAX], nx = 0, keep = 0;
    for (int k = 0; k < L->n; ++k) {
      const unet_wino_prep p = L->item[k];
      if (h2_conv3x3_selected(p.flip ? p.cout : p.cin, p.flip ? p.cin : p.cout) || x3_conv3x3_selected(p.flip ? p.cout : p.cin, p.flip ? p.cin : p.cout)) { xw[nx] = p.w; xi[nx] = p.u; xci[nx] = p.cin; xco[nx] = p.cout; xf[nx] = p.flip; ++nx; }
      else { L->item[keep] = p; hk[keep] = h[k]; ++keep; }
    }
    if (nx) {
      // (one predicate for the whole list: h2 and x3 take the same shapes, h2 is asked first)
      int32_t r = h2_conv3x3_selected(xf[0] ? xco[0] : xci[0], xf[0] ? xci[0] : xco[0]) ? k_h2_weights_multi(ctx, xw, xi, xci, xco, xf, nx, s)
                                                                                        : k_x3_weights_multi(ctx, xw, xi, xci, xco, xf, nx, s);
      if (r) return r;
    }
    L->n = keep; h = hk;
    if (keep < 1) return UNET_OK;
  }
  long long most = 1;
  for (int k = 0; k < L->n; ++k) {
    unet_wino_prep& p = L->item[k];
    p.two_d = use_2d(h[k], p.flip ? p.cin : p.cout) ? 1 : 0;
    most = std::max(most, (long long)p.cin * p.cout);
  }
  hipLaunchKernelGGL(wino_weight_multi_kernel, dim3((unsigned)std::min<long long>((most + 255) / 256, 256), (unsigned)L->n), dim3(256), 0, s, *L);
  UNET_CHECK_LAUNCH(ctx, "wino_weights_multi");
  return UNET_OK;
}

// transformed weights for k_conv3x3_wino_fwd on an image of `h` rows (the 2-D form is picked per shape, both sides must agree)
int32_t k_wino_weights(unet_ctx* ctx, const float* w, float* u, int cin, int cout, int flip, int h, hipStream_t s) {
  if (h2_conv3x3_selected(flip ? cout : cin, flip ? cin : cout)) return k_h2_weights(ctx, w, u, cin, cout, flip, s);
  if (x3_conv3x3_selected(flip ? cout : cin, flip ? cin : cout)) return k_x3_weights(ctx, w, u, cin, cout, flip, s);
  const long long total = (long long)cin * cout;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 2048)), dim3(256), 0, s, w, u, cin, cout, flip,
                     use_2d(h, flip ? cin : cout) ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "wino_weights");
  return UNET_OK;
}

// x [n,h,wd,cin] dense NHWC, u = transformed weights [12 or 16][cin][cout] (k_wino_weights with the same h), y [n,h,wd,cout]
int32_t k_conv3x3_wino_fwd(unet_ctx* ctx, const float* x, const float* u, const float* bias, const float* mask, int mask_mode, float* y, int n,
                           int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, hipStream_t s) {
  if (!wino_conv3x3_supported(cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 winograd: cin=%d cout=%d unsupported", cin, cout);
  if ((long long)h * wd * std::max(cin, cout) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 winograd: one image must stay below 1 GiB (32-bit buffer offsets); use UNET_ALGO_NAIVE");
  if (h2_conv3x3_selected(cin, cout)) return k_conv3x3_h2_fwd(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);      // u = the split fp16 weight image
  if (x3_conv3x3_selected(cin, cout)) return k_conv3x3_x3_fwd(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);      // u = the split weight image
  if (use_2d(h, cout)) {
    // UNET_WINO_4WAY: 0 never, 1 (default) the four-way accumulator split for 32-wide cout groups (c1b 0.54 -> 0.52 ms, c9a 0.89 -> 0.86;
    // on the 64-wide layers it stages the patch twice as often and measured 3-10 % slower), 2 everywhere
    static const int fourway = [] { const char* e = getenv("UNET_WINO_4WAY"); return e ? atoi(e) : 1; }();
    if (fourway == 2 || (fourway == 1 && cout % 64 != 0)) {
      const bool c16 = (cin % 16) == 0;
      if (wino_tile_cols(wd) == 32) return c16 ? launch_wino2d4<16, 16>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s)
                                               : launch_wino2d4<16, 8>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
      return c16 ? launch_wino2d4<32, 16>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s)
                 : launch_wino2d4<32, 8>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
    }
    if (wino_tile_cols(wd) == 32) {                        // narrow images: two row pairs x 16 tiles per MFMA M-tile
      if (cout % 64 == 0) return launch_wino2d<16, 1, 2>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
      return launch_wino2d<16, 2, 1>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
    }
    if (cout % 64 == 0) return launch_wino2d<32, 1, 2>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
    return launch_wino2d<32, 2, 1>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
  }
  if (mask_mode >= MASK_BIAS_TAB) UNET_FAIL(ctx, UNET_E_STATE, "conv3x3 winograd: the folded-BatchNorm epilogues need the F(2x2,3x3) kernels (h=%d cout=%d)", h, cout);
  // One image row per wave (64 accumulator registers) -> 3 workgroups per CU: occupancy pays more than sharing the weight operand
  // between two rows did (the <64,4,2,2> / <32,8,4,1> tiles measured 1-8 % slower on every U-Net layer).
  if (cout % 64 == 0) return launch_wino<64, 2, 2, 2>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
  return launch_wino<32, 4, 4, 1>(ctx, x, u, bias, mask, mask_mode, y, n, h, wd, cin, cout, act, rate, seed, s);
}
