// fp32 weight gradient of conv3x3 on the 16-bit matrix cores with fp32 accuracy: the "h2" form (see kernels_conv_h2.hip for the arithmetic).
//   dW[tap][ci][co] = sum_p X[p + tap][ci] * dY[p][co],   db[co] = sum_p dY[p][co]
// GEMM per tap with K = pixels: A = X (32 input channels x 16 pixels), B = dY (16 pixels x 32 output channels); both operands are ACTIVATIONS here, so
// both are block-scaled while they are staged: the workgroup tracks one running exponent per operand (max |x|, max |dy| of every staged row block:
// registers -> wave shuffle -> LDS words, no extra barrier; when a block needs a lower exponent the accumulators are multiplied by the exact
// power-of-two ratio), splits x 2^e into two fp16 planes and accumulates xh dym + xm dyh + xh dyh in fp32.  The partial slabs leave multiplied
// by 2^-(e_x + e_dy).  Everything else is the structure of wgrad_bf16_kernel (kernels_bf16.hip): split-K over (image, 32-column strip, row chunk),
// rows staged as they come from HBM ([plane][32-channel sub-plane][row][pixel][32 ch] fp16) and read back with the gfx950 LDS transpose read
// ds_read_b64_tr_b16 (8 consecutive pixels of one channel per lane), 4 waves = WA x WB channel tiles x WR row phases, 9 taps = 144 accumulator
// registers per wave, fixed-order slab reduction (k_wgrad_reduce: deterministic).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unet_f32x16 f32x16;

__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m) {
  const f16x2 hh = __builtin_convertvector((unet_f32x2){a, b}, f16x2);
  const f16x2 mm = __builtin_convertvector((unet_f32x2){a - (float)hh[0], b - (float)hh[1]}, f16x2);
  h = __builtin_bit_cast(unsigned, hh); m = __builtin_bit_cast(unsigned, mm);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }          // -126 <= e <= 127

__device__ __forceinline__ f16x8 lds_tr_frag(const char* p0, const char* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}

constexpr int APX = 34;                                       // X pixels per staged row (32 + halo)

// R = dY rows per step.  R = 2 (default): the staged pieces beside the 144 accumulators fit 256 registers -> two workgroups per CU cover each other's
// staging phases (R = 4 at one workgroup per CU measured 25 % slower); R = 4 only for the single 32 x 32 channel tile (four row phases).
// The X rows live in a RING of R + 2 LDS row slots (image row y of the current unit sits in slot (y + 1 - ya) mod (R + 2)): a step fetches and stages only its
// R NEW rows -- the two rows it shares with the step before stay where they are -- so the staging arithmetic (scale, two-term split: ~20 VALU per 16-byte
// piece; the kernel is instruction-issue bound: 4 - 5.6 VALU + ~2 LDS instructions per MFMA before the ring) falls by a third.  If a new row forces the
// running exponent of the X operand down, the two kept rows are fetched again and re-split at the new scale (rare: the maximum has to grow 8x).
// The bias gradient (column sums of the dY operand) is taken by the workgroups of the FIRST X channel tile only.
// VDY: the dY operand is VIRTUAL -- the gradient of the network's last conv3x3 output, dy[p][c] = dz_p w_c [y_pc > 0] (T1:911-913 backwards), staged from the
// 8-byte-per-pixel stream {dz_p, 32 mask bits} of head_dzm_kernel (B = that stream, CB = 32): one value is scaled and split per staged piece, the mask bits pick the
// channels; w_c (colscale) multiplies the finished columns.
template <int WA, int WB, int WR, int R, bool VDY = false>
__global__ __launch_bounds__(256, (R == 4 && WA * WB > 1) ? 1 : 2) void wgrad_h2_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ part, int N, int H, int W,
                                                          int CA, int CB, int tiles_b, int strips, int rows_per_chunk, int chunks_per_strip, int nsplit,
                                                          int npairs, long long pstride, int units, int upb, const float* __restrict__ colscale) {
  static_assert(WA * WB * WR == 4 && WR <= R, "4 waves");
  static_assert(!VDY || WB == 1, "the head stream carries one 32-channel tile");
  constexpr int TAPS = 9, AROWS = R + 2;
  constexpr int ASUB = AROWS * APX * 64, BSUB = R * 32 * 64;      // bytes of one 32-channel sub-plane of one fp16 plane
  constexpr int STAGE1 = WA * ASUB + WB * BSUB;                   // one fp16 plane of everything that is staged per step
  constexpr int RED = WR > 1 ? 2 * TAPS * 16 * 64 * 4 : 0;        // two accumulator images for the row-phase reduction
  constexpr int STAGE = 2 * STAGE1;
  __shared__ __attribute__((aligned(16))) char smem[STAGE > RED ? STAGE : RED];
  __shared__ float s_bs[4][64];
  __shared__ float s_amax[2][4];
  char* const s_a = smem; char* const s_b = smem + WA * ASUB;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave % WR, wb = (wave / WR) % WB, wa = wave / (WR * WB);
  // XCD-aware block map: all channel-tile pairs of one pixel split run on the same XCD back to back
  const int sq = blockIdx.x >> 3;
  const int pair = sq % npairs, split = (sq / npairs) * 8 + (blockIdx.x & 7);
  if (split >= nsplit) return;
  const int ta = pair / tiles_b, tb = pair % tiles_b;
  const int a0 = ta * 32 * WA, b0 = tb * 32 * WB;
  const int chunk = split % chunks_per_strip; const int ublk = split / chunks_per_strip;
  const int ya = chunk * rows_per_chunk;
  const int yb = ya + rows_per_chunk < H ? ya + rows_per_chunk : H;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  float bsum = 0.0f;
  int e_a = 120, e_b = 120;                                      // running exponents: staged values are x * 2^e_a, dy * 2^e_b

  // transpose-read addressing: 16-lane group g4 reads [4 pixels][16 channels]; lane i -> pixel i>>2, channel quad i&3
  const int i16 = lane & 15, g4 = lane >> 4;
  const int tr_px = (g4 >> 1) * 8 + (i16 >> 2), tr_ch = ((g4 & 1) * 16 + (i16 & 3) * 4) * 2;
  const char* const pa = s_a + wa * ASUB + tr_ch;
  const char* const pb = s_b + wb * BSUB + tr_ch;

  const int u1 = (ublk + 1) * upb < units ? (ublk + 1) * upb : units;
  for (int unit = ublk * upb; unit < u1; ++unit) {
    const int cs = unit % strips, n = unit / strips;
    const int x0 = cs * 32;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + (long long)n * H * W * CA, (long long)H * W * CA * 4);
    const __amdgpu_buffer_rsrc_t rs_b = VDY ? make_rsrc(B + (long long)n * H * W * 2, (long long)H * W * 8) : make_rsrc(B + (long long)n * H * W * CB, (long long)H * W * CB * 4);
    // staging plan.  A thread always fetches channel quad q = tid & 7 of pixel column pc = tid >> 3 (0..31) of a staged row, so one byte offset per
    // operand (and per 32-channel sub-plane: its validity differs) is all it keeps; rows and sub-planes are wave-uniform immediates.  The X rows have
    // 34 pixels: columns 0..31 (image column x0 - 1 + pc) go with the main pieces, columns 32 / 33 with one extra piece of the first 16 * WA * AROWS threads.
    const int q8 = tid & 7, pc = tid >> 3;
    int abase_t[WA], bbase_t[WB];
#pragma unroll
    for (int sub = 0; sub < WA; ++sub) {
      const int gx = x0 - 1 + pc, ch = a0 + sub * 32 + q8 * 4;
      abase_t[sub] = (gx >= 0 && gx < W && ch < CA) ? (gx * CA + ch) * 4 : UNET_OOB;
    }
#pragma unroll
    for (int sub = 0; sub < WB; ++sub) {
      const int gx = x0 + pc, ch = b0 + sub * 32 + q8 * 4;
      if (VDY) bbase_t[sub] = gx < W ? gx * 8 : UNET_OOB;          // (the eight quads of a pixel read the same 8 bytes)
      else bbase_t[sub] = (gx < W && ch < CB) ? (gx * CB + ch) * 4 : UNET_OOB;
    }
    // halo piece: thread t < 16 * WA * AROWS -> column 32 + ((t >> 3) & 1), (sub, row) = t >> 4
    constexpr int HALO_T = 16 * WA * AROWS;
    const int h_rs = tid >> 4, h_row = h_rs % AROWS, h_sub = h_rs / AROWS, h_px = 32 + ((tid >> 3) & 1);
    int hbase_t;
    {
      const int gx = x0 - 1 + h_px, ch = a0 + h_sub * 32 + q8 * 4;
      hbase_t = (tid < HALO_T && gx < W && ch < CA) ? (gx * CA + ch) * 4 : UNET_OOB;
    }
    constexpr int NA = WA * AROWS, NB_ = WB * R;                 // main pieces per thread
    unet_u32x4 areg[NA + 1], breg[NB_];
    auto issue_kept = [&](int ys) __attribute__((always_inline)) {          // the two rows a ring step keeps (rare path of store_lds)
#pragma unroll
      for (int sub = 0; sub < WA; ++sub)
#pragma unroll
        for (int row = 0; row < 2; ++row) {
          const int gy = ys - 1 + row;
          const bool ok = gy >= 0 && gy < H;
          areg[sub * AROWS + row] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? abase_t[sub] : UNET_OOB, ok ? gy * W * CA * 4 : 0, 0);
        }
      const int gy = ys - 1 + h_row;
      areg[NA] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (h_row < 2 && gy >= 0 && gy < H) ? hbase_t + gy * W * CA * 4 : UNET_OOB, 0, 0);
    };
    auto issue_loads = [&](int ys, int j0) __attribute__((always_inline)) {        // ys = first dY row of the step
      const int ysa = ys - 1;                                            // image row of staged X row 0 (the halo row above)
#pragma unroll
      for (int sub = 0; sub < WA; ++sub)
#pragma unroll
        for (int row = 0; row < AROWS; ++row) {
          if (row < j0) continue;
          const int gy = ysa + row;                                      // wave-uniform
          const bool ok = gy >= 0 && gy < H;
          areg[sub * AROWS + row] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? abase_t[sub] : UNET_OOB, ok ? gy * W * CA * 4 : 0, 0);
        }
      {
        const int gy = ysa + h_row;
        areg[NA] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, (h_row >= j0 && gy >= 0 && gy < H) ? hbase_t + gy * W * CA * 4 : UNET_OOB, 0, 0);
      }
#pragma unroll
      for (int sub = 0; sub < WB; ++sub)
#pragma unroll
        for (int row = 0; row < R; ++row) {
          const int gy = ys + row;
          const bool ok = gy < yb;                                       // rows past the chunk contribute 0
          if (VDY) { const unet_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_b, ok ? bbase_t[sub] : UNET_OOB, ok ? gy * W * 8 : 0, 0); breg[sub * R + row][0] = v[0]; breg[sub * R + row][1] = v[1]; }
          else breg[sub * R + row] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, ok ? bbase_t[sub] : UNET_OOB, ok ? gy * W * CB * 4 : 0, 0);
        }
    };
    auto post_amax = [&]() __attribute__((always_inline)) {
      float ma = 0.f, mb = 0.f;
#pragma unroll
      for (int k = 0; k < NA + 1; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) ma = fmaxf(ma, fabsf(__uint_as_float(areg[k][j])));
#pragma unroll
      for (int k = 0; k < NB_; ++k) {
        if (VDY) { mb = fmaxf(mb, breg[k][1] ? fabsf(__uint_as_float(breg[k][0])) : 0.f); continue; }
#pragma unroll
        for (int j = 0; j < 4; ++j) mb = fmaxf(mb, fabsf(__uint_as_float(breg[k][j])));
      }
      ma = wave_max_nonneg(ma); mb = wave_max_nonneg(mb);
      if (lane == 0) { s_amax[0][wave] = ma; s_amax[1][wave] = mb; }
    };
    auto store_lds = [&](int ys, int base, int j0) __attribute__((always_inline)) {
      const float ma = fmaxf(fmaxf(s_amax[0][0], s_amax[0][1]), fmaxf(s_amax[0][2], s_amax[0][3]));
      const float mb = fmaxf(fmaxf(s_amax[1][0], s_amax[1][1]), fmaxf(s_amax[1][2], s_amax[1][3]));
      const int eba = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(ma) >> 23) & 0xFF));
      const int ebb = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(mb) >> 23) & 0xFF));
      int d = 0; bool a_moved = false;
      if (eba >= 11 && eba - 127 + e_a >= 15) { d += 138 - eba - e_a; e_a = 138 - eba; a_moved = true; }      // re-centre the block maximum at [2^11, 2^12)
      if (ebb >= 11 && ebb - 127 + e_b >= 15) {
        const int db_ = 138 - ebb - e_b;
        bsum *= pow2f(max(db_, -126));
        d += db_; e_b = 138 - ebb;
      }
      if (d != 0) {
        const float f = pow2f(max(d, -126));
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] *= f;
      }
      const float sa = pow2f(e_a), sb = pow2f(e_b);
      auto put = [&](char* dst, const unet_u32x4& v, float sc) __attribute__((always_inline)) {
        unsigned h0, m0, h1, m1;
        split2_scaled(__uint_as_float(v[0]), __uint_as_float(v[1]), sc, h0, m0);
        split2_scaled(__uint_as_float(v[2]), __uint_as_float(v[3]), sc, h1, m1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + STAGE1) = make_uint2(m0, m1);
      };
      // LDS layout of a plane: X [sub][row][34 px][64 B], then dY [sub][row][32 px][64 B]; image column x0 - 1 + p sits at pixel slot p
#pragma unroll
      for (int sub = 0; sub < WA; ++sub)
#pragma unroll
        for (int row = 0; row < AROWS; ++row) { if (row < j0) continue; int sl = base + row; sl = sl >= AROWS ? sl - AROWS : sl; put(s_a + ((sub * AROWS + sl) * APX + pc) * 64 + q8 * 8, areg[sub * AROWS + row], sa); }
      if (tid < HALO_T && h_row >= j0) { int sl = base + h_row; sl = sl >= AROWS ? sl - AROWS : sl; put(s_a + ((h_sub * AROWS + sl) * APX + h_px) * 64 + q8 * 8, areg[NA], sa); }
      if (a_moved && j0 > 0) {                                   // (workgroup-uniform, rare) the two kept rows sit in LDS at the old scale: fetch and split them again
        issue_kept(ys);
#pragma unroll
        for (int sub = 0; sub < WA; ++sub)
#pragma unroll
          for (int row = 0; row < 2; ++row) { int sl = base + row; sl = sl >= AROWS ? sl - AROWS : sl; put(s_a + ((sub * AROWS + sl) * APX + pc) * 64 + q8 * 8, areg[sub * AROWS + row], sa); }
        if (tid < HALO_T && h_row < 2) { int sl = base + h_row; sl = sl >= AROWS ? sl - AROWS : sl; put(s_a + ((h_sub * AROWS + sl) * APX + h_px) * 64 + q8 * 8, areg[NA], sa); }
      }
#pragma unroll
      for (int sub = 0; sub < WB; ++sub)
#pragma unroll
        for (int row = 0; row < R; ++row) {
          char* dst = s_b + ((sub * R + row) * 32 + pc) * 64 + q8 * 8;
          if (VDY) {
            const unet_u32x4& v = breg[sub * R + row];
            const float x = __uint_as_float(v[0]) * sb;
            unsigned hh, mm;
            split2(x, x, hh, mm);
            const unsigned nib = v[1] >> (q8 * 4);                   // bits 0..3: this piece's four channels
            const unsigned k01 = h2_pair_mask(nib, 0), k23 = h2_pair_mask(nib, 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(hh & k01, hh & k23);
            *reinterpret_cast<uint2*>(dst + STAGE1) = make_uint2(mm & k01, mm & k23);
          } else put(dst, breg[sub * R + row], sb);
        }
    };

    issue_loads(ya, 0);
    post_amax();
    __syncthreads();
    store_lds(ya, 0, 0);
    __syncthreads();
    int base = 0;
    for (int ys = ya; ys < yb; ys += R) {
      const bool more = ys + R < yb;
      int nbase = base + R; nbase = nbase >= AROWS ? nbase - AROWS : nbase;
      int roff[AROWS];
#pragma unroll
      for (int j = 0; j < AROWS; ++j) { int sl = base + j; sl = sl >= AROWS ? sl - AROWS : sl; roff[j] = sl * APX * 64; }
      if (more) issue_loads(ys + R, 2);
#pragma unroll
      for (int r = wr; r < R; r += WR) {
#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
          const char* bp = pb + (r * 32 + kst * 16 + tr_px) * 64;
          const f16x8 bh = lds_tr_frag(bp, bp + 4 * 64), bm = lds_tr_frag(bp + STAGE1, bp + STAGE1 + 4 * 64);
          if (wa == 0 && ta == 0) {                                  // (wave-uniform) the bias gradient: only the first X channel tile's workgroups sum their dY operand
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += (float)bh[j] + (float)bm[j];
          }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            // the three taps of a kernel row together: 3 products x 3 taps, consecutive MFMAs on different accumulators (a dependent 32x32x16 MFMA right
            // behind its producer stalls the matrix pipe)
            f16x8 ah[3], am[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const char* ap = pa + roff[r + ky] + (kst * 16 + tr_px + kx) * 64;
              ah[kx] = lds_tr_frag(ap, ap + 4 * 64); am[kx] = lds_tr_frag(ap + STAGE1, ap + STAGE1 + 4 * 64);
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kx], bm, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[kx], bh, acc[ky * 3 + kx], 0, 0, 0);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kx], bh, acc[ky * 3 + kx], 0, 0, 0);
          }
        }
      }
      if (more) post_amax();
      __syncthreads();
      if (more) { store_lds(ys + R, nbase, 2); __syncthreads(); }
      base = nbase;
    }
  }
  const float un_b = pow2f(max(-e_b, -126));
  const float un = pow2f(max(-e_a, -126)) * un_b;                 // 2^-(e_x + e_dy)
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] *= un;
  bsum *= un_b;

  // ---- row phases of one channel tile are summed inside the workgroup (fixed order, through LDS): one slab per split
  bsum += __shfl_xor(bsum, 32, 64);
  if (WR > 1) {
    float* red = reinterpret_cast<float*>(smem);
    const int grp = wave / WR;
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
  const int ar = a0 + wa * 32, bc = b0 + wb * 32 + l31;
  const float cscale = VDY ? (bc < CB ? colscale[bc] : 0.f) : 1.0f;          // (VDY: the operand was dz [y > 0]; the column's head weight comes last)
  float* P = part + (long long)split * pstride;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (ar + m < CA && bc < CB) P[((long long)t * CA + ar + m) * CB + bc] = VDY ? acc[t][r] * cscale : acc[t][r];
    }
  if (wa == 0 && ta == 0 && lane < 32 && bc < CB) P[(long long)TAPS * CA * CB + bc] = VDY ? bsum * cscale : bsum;
}

// ---- ConvT 2x2 stride 2 (T1:886 ...): dK[ab][o][c] = sum_{n,i,j} dU[n, 2i+a, 2j+b, o] * x[n, i, j, c],  db[o] = sum dU.
// The same machine with A = dU (cout channels, pixel stride ldA inside the concat gradient, 2H x 2W: 2R rows x 64 pixels staged per step) and B = x
// (cin channels, H x W: R rows x 32 pixels); the four taps are the four parity planes of the staged dU rows -- a lane's eight consecutive K pixels sit two
// pixel slots apart -- and the bias gradient is the sum over A.
// BW: x channel tiles per wave (the dU fragments of a k-step feed BW x 12 MFMAs: half the staging, LDS reads and barriers per MFMA at BW = 2)
template <int WA, int WB, int WR, int R, int BW = 1>
__global__ __launch_bounds__(256, 2) void wgradT_h2_kernel(const float* __restrict__ A, int ldA, const float* __restrict__ B, float* __restrict__ part, int N, int H, int W,
                                                           int CA, int CB, int tiles_b, int strips, int rows_per_chunk, int chunks_per_strip, int nsplit,
                                                           int npairs, long long pstride, int units, int upb) {
  static_assert(WA * WB * WR == 4 && WR <= R, "4 waves");
  constexpr int TAPS = 4, AROWS = 2 * R;
  // a staged dU row: its 64 pixels de-interleaved by column parity (a tap reads ONE parity plane: consecutive 64-B slots, conflict-free transpose reads -- interleaved,
  // the 128-B stride put two of a read group's four pixels on the same banks: 21-23 % of the LDS cycles in round 4's counters), 64 B of padding between the planes
  constexpr int PPL = 32 * 64 + 64, ROWB = 2 * PPL;
  constexpr int ASUB = AROWS * ROWB, BSUB = R * 32 * 64;       // bytes of one 32-channel sub-plane of one fp16 plane
  constexpr int STAGE1 = WA * ASUB + WB * BW * BSUB;
  constexpr int RED = WR > 1 ? 2 * TAPS * 16 * 64 * 4 : 0;
  constexpr int STAGE = 2 * STAGE1;
  __shared__ __attribute__((aligned(16))) char smem[STAGE > RED ? STAGE : RED];
  __shared__ float s_bs[4][64];
  __shared__ float s_amax[2][4];
  char* const s_a = smem; char* const s_b = smem + WA * ASUB;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave % WR, wb = (wave / WR) % WB, wa = wave / (WR * WB);
  const int sq = blockIdx.x >> 3;
  const int pair = sq % npairs, split = (sq / npairs) * 8 + (blockIdx.x & 7);
  if (split >= nsplit) return;
  const int ta = pair / tiles_b, tb = pair % tiles_b;
  const int a0 = ta * 32 * WA, b0 = tb * 32 * WB * BW;
  const int chunk = split % chunks_per_strip; const int ublk = split / chunks_per_strip;
  const int ya = chunk * rows_per_chunk;
  const int yb = ya + rows_per_chunk < H ? ya + rows_per_chunk : H;
  const int HA = 2 * H, WAI = 2 * W;

  f32x16 acc[BW][TAPS];
#pragma unroll
  for (int j = 0; j < BW; ++j)
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.0f;
  float bsum = 0.0f;
  int e_a = 120, e_b = 120;

  const int i16 = lane & 15, g4 = lane >> 4;
  const int tr_px = (g4 >> 1) * 8 + (i16 >> 2), tr_ch = ((g4 & 1) * 16 + (i16 & 3) * 4) * 2;
  const char* const pa = s_a + wa * ASUB + tr_ch;
  const char* const pb = s_b + wb * BW * BSUB + tr_ch;

  const int u1 = (ublk + 1) * upb < units ? (ublk + 1) * upb : units;
  for (int unit = ublk * upb; unit < u1; ++unit) {
    const int cs = unit % strips, n = unit / strips;
    const int x0 = cs * 32;
    const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(A + (long long)n * HA * WAI * ldA, (long long)HA * WAI * ldA * 4);
    const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(B + (long long)n * H * W * CB, (long long)H * W * CB * 4);
    // a thread fetches channel quad q8 of pixel column pc of a staged B row, and of pixel columns pc and 32 + pc of a staged A row
    const int q8 = tid & 7, pc = tid >> 3;
    int abase_t[WA][2], bbase_t[WB * BW];
#pragma unroll
    for (int sub = 0; sub < WA; ++sub)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int gx = 2 * x0 + half * 32 + pc, ch = a0 + sub * 32 + q8 * 4;
        abase_t[sub][half] = (gx < WAI && ch < CA) ? (gx * ldA + ch) * 4 : UNET_OOB;
      }
#pragma unroll
    for (int sub = 0; sub < WB * BW; ++sub) {
      const int gx = x0 + pc, ch = b0 + sub * 32 + q8 * 4;
      bbase_t[sub] = (gx < W && ch < CB) ? (gx * CB + ch) * 4 : UNET_OOB;
    }
    constexpr int NA = WA * AROWS * 2, NB_ = WB * BW * R;
    unet_u32x4 areg[NA], breg[NB_];
    auto issue_loads = [&](int ys) __attribute__((always_inline)) {        // ys = first x row of the step
#pragma unroll
      for (int sub = 0; sub < WA; ++sub)
#pragma unroll
        for (int row = 0; row < AROWS; ++row) {
          const int gy = 2 * ys + row;
          const bool ok = ys + (row >> 1) < yb;                            // (rows past the chunk: not part of this split's bias sum either)
#pragma unroll
          for (int half = 0; half < 2; ++half)
            areg[(sub * AROWS + row) * 2 + half] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, ok ? abase_t[sub][half] : UNET_OOB, ok ? gy * WAI * ldA * 4 : 0, 0);
        }
#pragma unroll
      for (int sub = 0; sub < WB * BW; ++sub)
#pragma unroll
        for (int row = 0; row < R; ++row) {
          const int gy = ys + row;
          const bool ok = gy < yb;
          breg[sub * R + row] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, ok ? bbase_t[sub] : UNET_OOB, ok ? gy * W * CB * 4 : 0, 0);
        }
    };
    auto post_amax = [&]() __attribute__((always_inline)) {
      float ma = 0.f, mb = 0.f;
#pragma unroll
      for (int k = 0; k < NA; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) ma = fmaxf(ma, fabsf(__uint_as_float(areg[k][j])));
#pragma unroll
      for (int k = 0; k < NB_; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) mb = fmaxf(mb, fabsf(__uint_as_float(breg[k][j])));
      ma = wave_max_nonneg(ma); mb = wave_max_nonneg(mb);
      if (lane == 0) { s_amax[0][wave] = ma; s_amax[1][wave] = mb; }
    };
    auto store_lds = [&]() __attribute__((always_inline)) {
      const float ma = fmaxf(fmaxf(s_amax[0][0], s_amax[0][1]), fmaxf(s_amax[0][2], s_amax[0][3]));
      const float mb = fmaxf(fmaxf(s_amax[1][0], s_amax[1][1]), fmaxf(s_amax[1][2], s_amax[1][3]));
      const int eba = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(ma) >> 23) & 0xFF));
      const int ebb = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(mb) >> 23) & 0xFF));
      int d = 0;
      if (eba >= 11 && eba - 127 + e_a >= 15) {
        const int da_ = 138 - eba - e_a;
        bsum *= pow2f(max(da_, -126));                                     // (the bias sum rides on the A operand here)
        d += da_; e_a = 138 - eba;
      }
      if (ebb >= 11 && ebb - 127 + e_b >= 15) { d += 138 - ebb - e_b; e_b = 138 - ebb; }
      if (d != 0) {
        const float f = pow2f(max(d, -126));
#pragma unroll
        for (int j = 0; j < BW; ++j)
#pragma unroll
          for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] *= f;
      }
      const float sa = pow2f(e_a), sb = pow2f(e_b);
      auto put = [&](char* dst, const unet_u32x4& v, float sc) __attribute__((always_inline)) {
        unsigned h0, m0, h1, m1;
        split2_scaled(__uint_as_float(v[0]), __uint_as_float(v[1]), sc, h0, m0);
        split2_scaled(__uint_as_float(v[2]), __uint_as_float(v[3]), sc, h1, m1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + STAGE1) = make_uint2(m0, m1);
      };
#pragma unroll
      for (int sub = 0; sub < WA; ++sub)
#pragma unroll
        for (int row = 0; row < AROWS; ++row)
#pragma unroll
          for (int half = 0; half < 2; ++half) put(s_a + (sub * AROWS + row) * ROWB + (pc & 1) * PPL + (half * 16 + (pc >> 1)) * 64 + q8 * 8, areg[(sub * AROWS + row) * 2 + half], sa);
#pragma unroll
      for (int sub = 0; sub < WB * BW; ++sub)
#pragma unroll
        for (int row = 0; row < R; ++row) put(s_b + ((sub * R + row) * 32 + pc) * 64 + q8 * 8, breg[sub * R + row], sb);
    };

    issue_loads(ya);
    post_amax();
    __syncthreads();
    store_lds();
    __syncthreads();
    for (int ys = ya; ys < yb; ys += R) {
      const bool more = ys + R < yb;
      if (more) issue_loads(ys + R);
#pragma unroll
      for (int r = wr; r < R; r += WR) {
#pragma unroll
        for (int kst = 0; kst < 2; ++kst) {
          f16x8 ah[4], am[4];
#pragma unroll
          for (int ab = 0; ab < 4; ++ab) {
            const char* ap = pa + (2 * r + (ab >> 1)) * ROWB + (ab & 1) * PPL + (kst * 16 + tr_px) * 64;
            ah[ab] = lds_tr_frag(ap, ap + 4 * 64); am[ab] = lds_tr_frag(ap + STAGE1, ap + STAGE1 + 4 * 64);
            if (wb == 0 && tb == 0) {                                // (wave-uniform) the bias gradient rides on dU: the first x channel tile's workgroups only
#pragma unroll
              for (int j = 0; j < 8; ++j) bsum += (float)ah[ab][j] + (float)am[ab][j];
            }
          }
          // product-major: a dependent MFMA never directly follows its producer
#pragma unroll
          for (int j = 0; j < BW; ++j) {
            const char* bp = pb + j * BSUB + (r * 32 + kst * 16 + tr_px) * 64;
            const f16x8 bh = lds_tr_frag(bp, bp + 4 * 64), bm = lds_tr_frag(bp + STAGE1, bp + STAGE1 + 4 * 64);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) acc[j][ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ab], bm, acc[j][ab], 0, 0, 0);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) acc[j][ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[ab], bh, acc[j][ab], 0, 0, 0);
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) acc[j][ab] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ab], bh, acc[j][ab], 0, 0, 0);
          }
        }
      }
      if (more) post_amax();
      __syncthreads();
      if (more) { store_lds(); __syncthreads(); }
    }
  }
  const float un_a = pow2f(max(-e_a, -126));
  const float un = un_a * pow2f(max(-e_b, -126));
#pragma unroll
  for (int j = 0; j < BW; ++j)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] *= un;
  bsum *= un_a;

  bsum += __shfl_xor(bsum, 32, 64);
  static_assert(BW == 1 || WR == 1, "several x tiles per wave go with one row phase");
  if (WR > 1) {
    float* red = reinterpret_cast<float*>(smem);
    const int grp = wave / WR;
    s_bs[wave][lane] = bsum;
#pragma unroll
    for (int stride = WR / 2; stride >= 1; stride >>= 1) {
      float* img = red + (size_t)(grp * stride + (wr % stride)) * (TAPS * 16 * 64);
      if (wr >= stride && wr < 2 * stride) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) img[(t * 16 + r) * 64 + lane] = acc[0][t][r];
      }
      __syncthreads();
      if (wr < stride) {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[0][t][r] += img[(t * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (wr == 0) { bsum = s_bs[wave][lane]; for (int k = 1; k < WR; ++k) bsum += s_bs[wave + k][lane]; }
  }
  if (wr != 0) return;
  float* P = part + (long long)split * pstride;
  const int ar = a0 + wa * 32;
#pragma unroll
  for (int j = 0; j < BW; ++j) {
    const int bc = b0 + (wb * BW + j) * 32 + l31;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (ar + m < CA && bc < CB) P[((long long)t * CA + ar + m) * CB + bc] = acc[j][t][r];
      }
  }
  if (wb == 0 && tb == 0 && lane < 32 && ar + l31 < CA) P[(long long)TAPS * CA * CB + ar + l31] = bsum;
}

struct WgPlanT { int WA, WB, WR, R, BW, tiles_a, tiles_b, strips, rows_per_chunk, chunks_per_strip, nsplit, nslabs, units, upb; size_t floats; };

// ca = cout (dU channels), cb = cin (x channels); h, w = the ConvT's INPUT size
WgPlanT plan_wgradT_h2(int n, int h, int w, int ca, int cb) {
  WgPlanT p;
  p.WA = (ca % 64) == 0 ? 2 : 1; p.WB = 2;                      // (cb is a multiple of 64: h2_convT_wgrad_selected)
  p.WR = 4 / (p.WA * p.WB);
  p.R = p.WA == 2 ? 1 : 2;                                       // 64 dU channels: one x row per step keeps the two fp16 planes at 40 KB (two rows: 80 KB, one workgroup per CU)
  // two x channel tiles per wave (64 x 128 workgroup tile, 48 KB) where the tile pairs still fill the chip: u6 / u7 / u8 of the U-Net (cin 512 / 256 / 128)
  p.BW = (p.WA == 2 && (cb % 128) == 0 && (long long)(ca / 64) * (cb / 128) * n * ((w + 31) / 32) * std::max(1, h / 4) >= 512) ? 2 : 1;
  p.tiles_a = (ca + 32 * p.WA - 1) / (32 * p.WA); p.tiles_b = (cb + 32 * p.WB * p.BW - 1) / (32 * p.WB * p.BW); p.strips = (w + 31) / 32;
  const long long pairs = (long long)p.tiles_a * p.tiles_b, per = 4LL * ca * cb;
  const long long units = (long long)n * p.strips;
  long long want = std::max<long long>(1, 512 / pairs);
  want = std::min(want, std::max<long long>(1, (64LL << 20) / per));
  p.units = (int)units; p.upb = (int)std::max<long long>(1, units / want);
  const long long ublocks = (units + p.upb - 1) / p.upb;
  long long cps = std::max<long long>(1, want / ublocks);
  cps = std::min<long long>(cps, std::max<long long>(1, h / 4));
  int rpc = (int)((h + cps - 1) / cps); rpc = (rpc + p.R - 1) / p.R * p.R;
  p.rows_per_chunk = rpc; p.chunks_per_strip = (h + rpc - 1) / rpc;
  p.nsplit = (int)(ublocks * p.chunks_per_strip); p.nslabs = p.nsplit;
  p.floats = (size_t)p.nslabs * (per + ca) + wgrad_reduce_scratch_floats(4, ca, cb, ca, p.nslabs);
  return p;
}

struct WgPlanH2 { int WA, WB, WR, tiles_a, tiles_b, strips, rows_per_chunk, chunks_per_strip, nsplit, nslabs, units, upb; size_t floats; };

constexpr int wgrad_h2_rows() { return 2; }          // measured: 2 dY rows per step at two workgroups per CU (4 rows at one workgroup per CU: 25 % slower)

WgPlanH2 plan_wgrad_h2(int n, int h, int w, int ca, int cb) {
  WgPlanH2 p;
  int R = wgrad_h2_rows();
  p.WA = (ca % 64) == 0 ? 2 : 1; p.WB = (cb % 64) == 0 ? 2 : 1;
  p.WR = 4 / (p.WA * p.WB);
  // one 32 x 32 channel tile (c1b, c9b: T1:860, 911): four row phases over four dY rows per step -- 144 accumulators + 11 staged pieces still fit 256
  // registers, so two workgroups per CU stay (a 64-wide dY tile instead would run half of its MFMAs on padding)
  if (p.WR > R) R = 4;
  p.tiles_a = (ca + 32 * p.WA - 1) / (32 * p.WA); p.tiles_b = (cb + 32 * p.WB - 1) / (32 * p.WB); p.strips = (w + 31) / 32;
  const long long pairs = (long long)p.tiles_a * p.tiles_b, per = 9LL * ca * cb;
  const long long units = (long long)n * p.strips;
  const long long target = R == 4 && p.WA * p.WB > 1 ? 256LL : 512LL;          // one resident round: 256 CUs x 1 (R = 4) or 2 (R = 2) workgroups
  long long want = std::max<long long>(1, target / pairs);
  const long long cap = std::max<long long>(1, (64LL << 20) / per);
  want = std::min(want, cap);
  p.units = (int)units; p.upb = (int)std::max<long long>(1, units / want);
  const long long ublocks = (units + p.upb - 1) / p.upb;
  long long cps = std::max<long long>(1, want / ublocks);
  cps = std::min<long long>(cps, std::max<long long>(1, h / 8));
  int rpc = (int)((h + cps - 1) / cps); rpc = (rpc + R - 1) / R * R;                // whole steps
  p.rows_per_chunk = rpc; p.chunks_per_strip = (h + rpc - 1) / rpc;
  p.nsplit = (int)(ublocks * p.chunks_per_strip); p.nslabs = p.nsplit;
  p.floats = (size_t)p.nslabs * (per + cb) + wgrad_reduce_scratch_floats(9, ca, cb, cb, p.nslabs);
  return p;
}

}  // namespace

bool h2_wgrad_selected(int algo, int cin, int cout) {
  return algo == UNET_ALGO_AUTO && cin >= 16 && (cin % 16) == 0 && cout >= 16 && (cout % 16) == 0;          // (16- / 48-channel tensors: the last 32-channel tile is masked)
}
size_t h2_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) { return h2_wgrad_selected(UNET_ALGO_AUTO, cin, cout) ? plan_wgrad_h2(n, h, wd, cin, cout).floats * sizeof(float) : 0; }

int32_t k_conv3x3_h2_wgrad(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin, int cout,
                           hipStream_t s) {
  if (cin < 16 || (cin % 16) || cout < 16 || (cout % 16)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad h2: cin=%d cout=%d unsupported (multiples of 16)", cin, cout);
  if ((long long)h * wd * std::max(cin, cout) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  const WgPlanH2 p = plan_wgrad_h2(n, h, wd, cin, cout);
  if (!ws || ws_bytes < p.floats * sizeof(float)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad h2: workspace %zu < %zu bytes", ws_bytes, p.floats * sizeof(float));
  float* part = static_cast<float*>(ws);
  const long long S = 9LL * cin * cout + cout;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));
#define UNET_WG(WA_, WB_, WR_) hipLaunchKernelGGL((wgrad_h2_kernel<WA_, WB_, WR_, 2>), grid, dim3(256), 0, s, x, dy, part, n, h, wd, cin, cout, p.tiles_b, p.strips, \
                                                  p.rows_per_chunk, p.chunks_per_strip, p.nsplit, npairs, S, p.units, p.upb, (const float*)nullptr)
  if (p.WA == 2 && p.WB == 2) UNET_WG(2, 2, 1);
  else if (p.WA == 2) UNET_WG(2, 1, 2);
  else if (p.WB == 2) UNET_WG(1, 2, 2);
  else hipLaunchKernelGGL((wgrad_h2_kernel<1, 1, 4, 4>), grid, dim3(256), 0, s, x, dy, part, n, h, wd, cin, cout, p.tiles_b, p.strips, p.rows_per_chunk, p.chunks_per_strip, p.nsplit, npairs, S,
                          p.units, p.upb, (const float*)nullptr);
#undef UNET_WG
  UNET_CHECK_LAUNCH(ctx, "wgrad_h2");
  return k_wgrad_reduce(ctx, part, p.nslabs, 9, cin, cout, cout, dw, db, s);
}

// The weight gradient of the last conv3x3 (cin -> 32, T1:911) from the head's rank-1 stream dzm[n,h,wd] = {dz, 32 mask bits} (k_head_dzm) and the head's weights w_head[32]:
// dw[3][3][cin][32], db[32] (overwritten); cin = 32 (one channel tile: the kernel form with four row phases)
int32_t k_conv3x3_h2_wgrad_dzm(unet_ctx* ctx, const float* x, const void* dzm, const float* w_head, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin,
                               hipStream_t s) {
  const int cout = 32;
  if (!x || !dzm || !w_head || cin != 32) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad h2 behind the head stream: cin=%d (32)", cin);
  if ((long long)h * wd * 32 * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "wgrad h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  const WgPlanH2 p = plan_wgrad_h2(n, h, wd, cin, cout);
  if (!ws || ws_bytes < p.floats * sizeof(float)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad h2: workspace %zu < %zu bytes", ws_bytes, p.floats * sizeof(float));
  float* part = static_cast<float*>(ws);
  const long long S = 9LL * cin * cout + cout;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));
  hipLaunchKernelGGL((wgrad_h2_kernel<1, 1, 4, 4, true>), grid, dim3(256), 0, s, x, static_cast<const float*>(dzm), part, n, h, wd, cin, cout, p.tiles_b, p.strips, p.rows_per_chunk,
                     p.chunks_per_strip, p.nsplit, npairs, S, p.units, p.upb, w_head);
  UNET_CHECK_LAUNCH(ctx, "wgrad_h2_dzm");
  return k_wgrad_reduce(ctx, part, p.nslabs, 9, cin, cout, cout, dw, db, s);
}

// ---- 16 -> 16 channels (the classifier's first block, T2:748-751 at 224 x 224 x 256): a 32 x 32 MFMA tile is three quarters padding there.  An NHWC tensor
// [n, h, w, 16] with w even IS the tensor [n, h, w / 2, 32] (a pixel pair = 32 channels), and the 32 -> 32 weight gradient G of that half-width problem holds
// every product the 16 -> 16 one needs -- with x column 2 (J + B) + p and dy column 2 J + q:
//     dW[a][b][ci][co] = sum over q in {0, 1} of G[a][B + 1][p * 16 + ci][q * 16 + co],   t = q + b - 1,  B = floor(t / 2),  p = t - 2 B
// (a pair is never half outside the image, so the zero padding agrees).  Half of the tile is useful instead of a quarter: 1.07 -> ~0.55 ms at that size.
namespace {
__global__ void wgrad_c16_gather_kernel(const float* __restrict__ G, const float* __restrict__ gb, float* __restrict__ dw, float* __restrict__ db) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 9 * 256) {
    const int co = i & 15, ci = (i >> 4) & 15, tap = i >> 8, a = tap / 3, b = tap - a * 3;
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = q + b - 1, B = t < 0 ? -1 : (t >> 1), p = t - 2 * B;
      acc += G[((a * 3 + B + 1) * 32 + p * 16 + ci) * 32 + q * 16 + co];
    }
    dw[i] = acc;
  } else if (i < 9 * 256 + 16) { const int co = i - 9 * 256; db[co] = gb[co] + gb[16 + co]; }
}
}  // namespace
int32_t k_wgrad_c16_gather(unet_ctx* ctx, const float* G, float* dw, float* db, hipStream_t s) {          // G: [9][32][32] + 32 bias sums of the pixel-pair problem
  hipLaunchKernelGGL(wgrad_c16_gather_kernel, dim3((9 * 256 + 16 + 255) / 256), dim3(256), 0, s, G, G + 9 * 32 * 32, dw, db);
  UNET_CHECK_LAUNCH(ctx, "wgrad_c16_gather");
  return UNET_OK;
}
bool h2_wgrad_c16_selected(int algo, int wd, int cin, int cout) { return algo == UNET_ALGO_AUTO && cin == 16 && cout == 16 && wd >= 2 && (wd & 1) == 0; }
size_t h2_wgrad_c16_ws_bytes(int n, int h, int wd) { return h2_wgrad_ws_bytes(n, h, wd / 2, 32, 32) + (9 * 32 * 32 + 32) * sizeof(float); }
int32_t k_conv3x3_h2_wgrad_c16(unet_ctx* ctx, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, hipStream_t s) {
  if ((wd & 1) || !ws || ws_bytes < h2_wgrad_c16_ws_bytes(n, h, wd)) UNET_FAIL(ctx, UNET_E_ARG, "wgrad h2 (16 channels as pixel pairs): W=%d, workspace %zu < %zu bytes", wd, ws_bytes, h2_wgrad_c16_ws_bytes(n, h, wd));
  const size_t inner = h2_wgrad_ws_bytes(n, h, wd / 2, 32, 32);
  float* G = reinterpret_cast<float*>(static_cast<char*>(ws) + inner);
  int32_t r = k_conv3x3_h2_wgrad(ctx, x, dy, G, G + 9 * 32 * 32, ws, inner, n, h, wd / 2, 32, 32, s);
  if (r) return r;
  return k_wgrad_c16_gather(ctx, G, dw, db, s);
}

// ---- ConvT weight gradient on the h2 kernels: cout (the dU channels) a multiple of 32, cin a multiple of 64
bool h2_convT_wgrad_selected(int algo, int cin, int cout) {
  return algo == UNET_ALGO_AUTO && cout >= 32 && (cout % 32) == 0 && cin >= 64 && (cin % 64) == 0;
}
size_t h2_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout) { return h2_convT_wgrad_selected(UNET_ALGO_AUTO, cin, cout) ? plan_wgradT_h2(n, h, wd, cout, cin).floats * sizeof(float) : 0; }

// x [n,h,wd,cin] dense, dy = dU channel slice (pixel stride lddy) of [n,2h,2wd,.]; dw [2][2][cout][cin], db [cout]
int32_t k_convT_h2_wgrad(unet_ctx* ctx, const float* x, const float* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin, int cout,
                         hipStream_t s) {
  if (!h2_convT_wgrad_selected(UNET_ALGO_AUTO, cin, cout)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT wgrad h2: cin=%d cout=%d unsupported", cin, cout);
  if (4LL * h * wd * lddy * 4 >= (1LL << 30) || (long long)h * wd * cin * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "convT wgrad h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  const WgPlanT p = plan_wgradT_h2(n, h, wd, cout, cin);
  if (!ws || ws_bytes < p.floats * sizeof(float)) UNET_FAIL(ctx, UNET_E_ARG, "convT wgrad h2: workspace %zu < %zu bytes", ws_bytes, p.floats * sizeof(float));
  float* part = static_cast<float*>(ws);
  const long long S = 4LL * cout * cin + cout;
  const int npairs = p.tiles_a * p.tiles_b;
  const dim3 grid((unsigned)(8 * ((p.nsplit + 7) / 8) * npairs));
  if (p.WA == 2 && p.BW == 2) hipLaunchKernelGGL((wgradT_h2_kernel<2, 2, 1, 1, 2>), grid, dim3(256), 0, s, dy, lddy, x, part, n, h, wd, cout, cin, p.tiles_b, p.strips, p.rows_per_chunk, p.chunks_per_strip,
                                    p.nsplit, npairs, S, p.units, p.upb);
  else if (p.WA == 2) hipLaunchKernelGGL((wgradT_h2_kernel<2, 2, 1, 1>), grid, dim3(256), 0, s, dy, lddy, x, part, n, h, wd, cout, cin, p.tiles_b, p.strips, p.rows_per_chunk, p.chunks_per_strip,
                                    p.nsplit, npairs, S, p.units, p.upb);
  else hipLaunchKernelGGL((wgradT_h2_kernel<1, 2, 2, 2>), grid, dim3(256), 0, s, dy, lddy, x, part, n, h, wd, cout, cin, p.tiles_b, p.strips, p.rows_per_chunk, p.chunks_per_strip, p.nsplit,
                          npairs, S, p.units, p.upb);
  UNET_CHECK_LAUNCH(ctx, "wgradT_h2");
  return k_wgrad_reduce(ctx, part, p.nslabs, 4, cout, cin, cout, dw, db, s);
}
