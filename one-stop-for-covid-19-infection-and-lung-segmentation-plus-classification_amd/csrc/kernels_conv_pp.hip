// fp32 3x3 convolution (forward + data gradient) of the 32 -> 32-channel launches of the 512 x 512 level (T1:860, 911: c1b forward, the data gradients of c1b and c9b):
// the arithmetic of kernels_conv_h2.hip -- three fp16 MFMA products of a block-scaled two-term split -- in a different schedule.
//
// conv_h2_kernel leaves the overlap of one tile's loads / split / stores with another tile's MFMAs to the CU's two to four independent workgroups, re-fetches the layer's
// weight slab from the L2 per chunk and tile, and pays two barriers and ~400 instructions of index arithmetic per 54 MFMAs.  Here:
//   * ONE persistent 512-thread workgroup per CU walks the tiles of its XCD; its two 4-wave halves work on tiles of their own, two barrier intervals apart
//     (half 1 enters the loop two barriers late): every SIMD holds one wave of each half, so a matrix stream always runs beside a staging stream;
//   * the layer's whole split weight image (36 B per weight: 36 KB) is copied into LDS once per workgroup;
//   * a tile's whole contraction (both 16-channel chunks) is staged at once: one exponent per tile from the max |x| of its patch, no accumulator rescaling;
//   * a tile's patch travels global -> registers one full MFMA phase ahead of its use; per-thread offsets are computed once per kernel, a tile is a scalar offset;
//   * the MFMA phase requests the operands of step s + 1 before the MFMAs of step s, and carries the previous tile's epilogue in 18 slices (one store every other step);
//   * BatchNorm statistics are kept per lane across all tiles of the workgroup and folded once at the end of the kernel.
// Barrier intervals of an iteration, per half:   P1 max |x| of the landed patch -> LDS     P2 split -> planes; request the next patch; the previous tile's output values
//                                                 P3 MFMA steps 0 .. 2 (+ slices)            P4 MFMA steps 3 .. 17 (+ slices)
// (half 0 in P1 / P2 while half 1 is in P3 / P4 and vice versa; the P1 / P3 interval exists only to make the four partial maxima of a half visible to its waves).
// What bounds it (measured: tools/pp_timeline.py, profiles/r06_pp_timeline.txt): the CU's vector-memory pipeline takes ~12 B per cycle -- 151 KB per tile pair in 12.3 k cycles.
#include <type_traits>

#include "common.h"

#ifndef PP_TRACE
#define PP_TRACE 0
#endif
#ifndef PP_MID
#define PP_MID 2            // the step behind which the MFMA phase has its middle barrier (P3 = steps 0 .. PP_MID: as long as the other half's P1)
#endif
#ifndef PP_STORE_AUX
#define PP_STORE_AUX 2          // nontemporal
#endif
#ifndef PP_EXP
#define PP_EXP 0          // measurement builds: 1 = no output stores, 2 = the patch is requested only once (no loads in the loop)
#endif
#if PP_TRACE
// measurement build only (tools/pp_timeline.py): wave 0 of either half stamps s_memtime at its phase boundaries
__device__ unsigned long long pp_trace[256 * 2 * 40 * 12];
#define PP_STAMP(i) do { if (lane == 0 && wave == 0 && it < 40 && blockIdx.x < 256) pp_trace[((blockIdx.x * 2 + half) * 40 + it) * 12 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t unet_debug_pp_trace(unsigned long long* host, int n) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(pp_trace), (size_t)n * 8) == hipSuccess ? 0 : 1; }
#else
#define PP_STAMP(i) do { } while (0)
#endif

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unet_f32x16 f32x16;
constexpr int PP_HEADER = 256;                               // bytes in front of a weight image (kernels_conv_h2.hip: H2_HEADER)

__device__ __forceinline__ float pp_pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ f16x8 pp_frag(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
// byte offset of 16-B cell c (0..7) of pixel p (0..31) in a wave's 4-KB epilogue staging row (kernels_conv_h2.hip: out_cell)
__device__ __forceinline__ int pp_cell(int p, int c) { return p * 128 + ((c ^ (p & 7)) << 4); }

// KC = 16-channel chunks of the contraction (K = 16 KC); one 32-channel output block per workgroup (M = 32)
// BITS: the data-gradient instance, `mask` = one-bit ReLU mask of the output (MASK_RELU_BITS).  VDY: the input is VIRTUAL -- the gradient of the last conv3x3's output,
// dy[p][c] = dz_p w_c [y_pc > 0] (T1:911-913 backwards), staged from the 8-byte-per-pixel stream {dz_p, 32 mask bits} of head_dzm_kernel (x = that stream): one value is
// scaled and split per piece, the mask bits pick the channels it goes to; w_c is a per-contraction-channel factor of the weight image (kernels_conv_h2.hip: EPI 3)
template <int KC, bool BITS, bool VDY>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(const float* __restrict__ x, const unet_bf16* __restrict__ wimg_hdr, const float* __restrict__ bias,
                                                         const float* __restrict__ mask, float* __restrict__ y, int ldy, int N, int H, int W, int act, int mask_mode,
                                                         int tiles_x, int tiles_y, int total_tiles, double* __restrict__ stats, int stats_c,
                                                         unsigned long long* __restrict__ signs, int xs) {
  constexpr int RW = 2, TH = 8, PWD = 34, NPIX = (TH + 2) * PWD, K = 16 * KC, M = 32;
  constexpr int HS = NPIX * 16;                            // one channel half (8 channels) of one fp16 plane of one chunk: [pixel][8 channels]
  constexpr int PLANE = 2 * HS, CHUNK = 2 * PLANE + 32;    // (+32 B = 8 banks: the two chunks' staging stores of a 16-lane group land on different banks)
  constexpr int IN_HALF = KC * CHUNK;                      // the planes of one half's tile
  constexpr int W_BYTES = 9 * 2048, W_ALL = KC * W_BYTES;  // weight slab of a chunk: [tap][plane][channel half][32 rows][8 fp16]
  constexpr int PPC = 4 * KC, PPIECES = NPIX * PPC, PL = (PPIECES + 255) / 256;          // 16-B fp32 pieces per pixel / per patch / per thread
  static_assert(256 % PPC == 0, "a thread's pieces are the same 16 bytes of 256 / PPC pixels apart");
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [W_ALL] [2 x IN_HALF] [4 x 4 KB staging rows] [2 x 4 floats] [bias 32] [final statistics fold]
  const int tid = threadIdx.x & 255, lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wave8 >> 2, wave = wave8 & 3;
  char* const s_w = smem;
  char* const s_in = smem + W_ALL + half * IN_HALF;
  char* const s_out = smem + W_ALL + 2 * IN_HALF + wave * 4096;          // (the two halves run their epilogues in different barrier intervals: one row per wave pair)
  float* const s_amax = reinterpret_cast<float*>(smem + W_ALL + 2 * IN_HALF + 4 * 4096) + half * 4;
  float* const s_bias = reinterpret_cast<float*>(smem + W_ALL + 2 * IN_HALF + 4 * 4096) + 8;
  float* const s_stat = s_bias + 32;                       // [8 waves][2][32]
  float* const s_tab = s_stat + 8 * 2 * 32;                // [16][32]: the border-class bias table of a conv whose input BatchNorm is folded into it (MASK_BIAS_TAB)

  // ---- the tiles of this half: XCD b % 8 owns a contiguous range of tiles (one L2 sees a tile's neighbours); its 2 x (gridDim / 8) half-workgroups take them round robin
  const int per = gridDim.x >> 3, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int tpx = (total_tiles + 7) >> 3;
  const int t_beg = xcd * tpx, t_end = min(t_beg + tpx, total_tiles);
  const int slots = 2 * per, iters = (tpx + slots - 1) / slots;
  const int t0 = t_beg + j * 2 + half;                     // tile of iteration i: t0 + i * slots

  // ---- per-thread constants of the staging: piece idx = tid + k * 256 -> (pixel, 16-B piece c of the pixel's K floats): a wave's 64 lanes read 8 pixels x 128 B (K = 32).
  // c is the same for all of a thread's pieces, their pixels are 256 / PPC apart: the LDS address of piece k is the first one + k * 512 (an immediate)
  const int pc = tid % PPC, pix0 = tid / PPC;
  const int dst0 = (pc >> 2) * CHUNK + ((pc & 3) >> 1) * HS + pix0 * 16 + (pc & 1) * 8;
  int rel[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int pix = pix0 + k * (256 / PPC), r = pix / PWD, col = pix - r * PWD;
    rel[k] = pix < NPIX ? (VDY ? (r * W + col) * 8 : ((r * W + col) * K + pc * 4) * 4) : UNET_OOB;          // relative to pixel (y0 - 1, x0 - 1); VDY: the 8 pieces of a pixel read the same 8 bytes
  }
  // the descriptor starts one row + one pixel in front of the tensor (never dereferenced there: halo pieces carry an out-of-range offset and read 0)
  constexpr int XPP = VDY ? 2 : K;                          // floats per pixel of the staged tensor
  const long long lead = (long long)(W + 1) * XPP;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x - lead, ((long long)N * H * W * XPP + lead) * 4);
  // the output rows leave as full 128-B lines: lane -> (pixel lane / 8 of an 8-pixel group, channel quad lane % 8); a (row, group) is a scalar offset
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(y, (((long long)N * H * W - 1) * ldy + M) * 4);
  const int st_lane = ((lane >> 3) * ldy + (lane & 7) * 4) * 4;

  unet_u32x4 pregA[PL];
  // a tile as (image, first row, first column); a half's next tile is `slots` tiles further: the step is decomposed once (no division per tile)
  struct pos { int n, y0, x0; };
  auto tile_pos = [&](int t) __attribute__((always_inline)) {
    pos p; const int tx = t % tiles_x; const int t2 = t / tiles_x;
    const int ty = t2 % tiles_y; p.n = t2 / tiles_y; p.x0 = tx * 32; p.y0 = ty * TH;
    return p;
  };
  const int d_x = (slots % tiles_x) * 32, d_y = ((slots / tiles_x) % tiles_y) * TH, d_n = slots / (tiles_x * tiles_y);
  auto advance = [&](pos p) __attribute__((always_inline)) {
    p.x0 += d_x; if (p.x0 >= tiles_x * 32) { p.x0 -= tiles_x * 32; p.y0 += TH; }
    p.y0 += d_y; if (p.y0 >= tiles_y * TH) { p.y0 -= tiles_y * TH; p.n += 1; }
    p.n += d_n;
    return p;
  };
  auto issue_loads = [&](pos p, unet_u32x4 (&preg)[PL]) __attribute__((always_inline)) {
    const int soff = (((p.n * H + p.y0) * W + p.x0) * XPP) * 4;          // + lead - (W + 1) XPP: the tile's pixel (y0 - 1, x0 - 1)
    // ONE sequence of requests behind an offset select: two arms that both define the registers made the compiler copy all 44 of them behind a vmcnt(0) -- the
    // whole HBM round trip of the patch in front of the epilogue
    int off[PL];
    if (p.y0 > 0 && p.y0 + TH < H && p.x0 > 0 && p.x0 + 32 < W) {          // (wave-uniform) an interior tile: every piece of the patch is inside the image
#pragma unroll
      for (int k = 0; k < PL; ++k) off[k] = rel[k];
    } else {
#pragma unroll
      for (int k = 0; k < PL; ++k) {
        const int pix = pix0 + k * (256 / PPC), r = pix / PWD, col = pix - r * PWD;
        const bool ok = (unsigned)(p.y0 - 1 + r) < (unsigned)H && (unsigned)(p.x0 - 1 + col) < (unsigned)W;
        off[k] = ok ? rel[k] : UNET_OOB;
      }
    }
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      asm volatile("" : "+v"(off[k]));
      if (VDY) { const unet_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, off[k], soff, 0); preg[k][0] = v[0]; preg[k][1] = v[1]; }
      else preg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off[k], soff, 0);
    }
  };

  // ---- once per workgroup: the weight image of the layer and the bias row
  {
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(wimg_hdr), 0, PP_HEADER + W_ALL, 0x00020000);
    constexpr int WP = W_ALL / 16;
#pragma unroll
    for (int k = 0; k < (WP + 511) / 512; ++k) {
      const int idx = (int)threadIdx.x + k * 512;
      const unet_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, idx < WP ? PP_HEADER + idx * 16 : UNET_OOB, 0, 0);
      if (idx < WP) *reinterpret_cast<unet_u32x4*>(s_w + idx * 16) = v;
    }
    if (threadIdx.x < 32) s_bias[threadIdx.x] = bias ? bias[threadIdx.x] : 0.f;
    if (!BITS && mask_mode == MASK_BIAS_TAB) s_tab[threadIdx.x] = mask[threadIdx.x];          // (through LDS: a vector-memory read in the loop would wait behind the patch requests -- vmcnt is in order)
  }
  const float w_unscale = *reinterpret_cast<const float*>(wimg_hdr);          // 2^-e_w of the layer
  pos p_prev = {0, 0, 0}, p_cur = tile_pos(t0), p_next = advance(p_cur);
  if (t0 < t_end) issue_loads(p_cur, pregA);
  __syncthreads();
  if (half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }

  f32x16 acc[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
  float vprev[RW][16];                                     // the previous tile's output values (bias, activation applied) in the accumulator layout, until its epilogue slices have taken them
  float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f), st2 = st1;           // BatchNorm statistics of what this lane stores (channel quad lane & 7), over all tiles of the workgroup
  int e_prev = 0, e_cur = 0;
  constexpr bool bits_mask = BITS;                           // (the data-gradient instance: mask = one-bit ReLU mask; the other instance: mask = border-class bias table or null)
  const bool tab_mask = !BITS && mask_mode == MASK_BIAS_TAB;
  // (BITS) the one-bit ReLU mask of a data gradient (MASK_RELU_BITS: u64 words [n][y][x / 8][4]; word k = (lo, hi) dwords holds, for channel % 4 == k, bit
  // (pixel % 8) * 8 + channel / 4): this lane's pixel sits in dword (pixel % 8) / 4 of each word.  The four dwords of a row are requested when the tile is SPLIT, in front
  // of the next tile's patch (vmcnt is in order), and used one iteration later, where the accumulators become output values
  unsigned mpre[BITS ? RW : 1][4];
  const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(mask, BITS ? (long long)N * H * (W >> 3) * 32 : 0);
  float4 t4s[4];
  int sg_lo = 0, sg_hi = 0;                                  // the sign words of a row: word (cell jj, k) in lane 4 jj + k

  // ---- the epilogue of the previous tile in 18 slices, one behind the MFMAs of each step of the current tile: the CU's store path takes one 1-KB store instruction per
  // ~110 cycles (measured: 32 KB per tile in ~3500 cycles whatever else runs) -- a wave that issues its eight stores back to back sits in front of a full queue for
  // 3000+ cycles; spread over the MFMA phase (one every other step, ~430 cycles apart per wave) they are accepted as they come and nothing waits for them.
  //   row 0: slice 0 mask + transpose writes, 1 line-layout reads, 2 / 4 / 6 / 8 one store (+ statistics, sign bits) each;   row 1: slices 9, 10, 11 / 13 / 15 / 17
  auto eslice = [&](const int sl) __attribute__((always_inline)) {
    const int pn = p_prev.n, py0 = p_prev.y0, px0 = p_prev.x0;
    const int r = sl < 9 ? 0 : 1, q = sl - r * 9;
    const int py = py0 + wave * RW + r;
    if (py >= H) return;                                   // (wave-uniform)
    if (q == 0) {
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<float4*>(s_out + pp_cell(l31, hi * 4 + qq)) = make_float4(vprev[r][qq * 4], vprev[r][qq * 4 + 1], vprev[r][qq * 4 + 2], vprev[r][qq * 4 + 3]);
    } else if (q == 1) {
      // (LDS executes a wave's instructions in order: these line-layout reads see the writes of slice 0, row 1's writes come after them)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) t4s[jj] = *reinterpret_cast<const float4*>(s_out + pp_cell(jj * 8 + (lane >> 3), lane & 7));
    }
    if (q >= 2 && (q & 1) == 0) {
      const int jj = (q - 2) >> 1;
      const int pxj = px0 + jj * 8 + (lane >> 3);
      const float4 t4 = t4s[jj];
      if (signs) {
        // (wave-uniform) sign bits of the stored values (MASK_RELU_BITS layout: u64 words [n][y][x / 8][4], word k bit (pixel % 8) * 8 + channel / 4 for channel % 4 == k):
        // in the line layout a compare's lane mask IS word k of the 8-pixel cell.  The 16 words of a row are collected in lanes 0..15 of a register pair (v_writelane:
        // no select chains, no exec-mask branches) and leave as ONE 128-B store behind the row's last cell (W % 8 == 0: a cell is inside the image or not at all)
        const unsigned long long b[4] = {__builtin_amdgcn_ballot_w64(t4.x > 0.f), __builtin_amdgcn_ballot_w64(t4.y > 0.f), __builtin_amdgcn_ballot_w64(t4.z > 0.f), __builtin_amdgcn_ballot_w64(t4.w > 0.f)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unet_writelane(sg_lo, (unsigned)b[k], jj * 4 + k);
          unet_writelane(sg_hi, (unsigned)(b[k] >> 32), jj * 4 + k);
        }
        if (jj == 3) {
          const int cells = min(4, (W - px0) >> 3);
          if (lane < 4 * cells)
            signs[(((long long)pn * H + py) * (W >> 3) + (px0 >> 3)) * 4 + lane] = ((unsigned long long)(unsigned)sg_hi << 32) | (unsigned)sg_lo;
        }
      }
      // one full 128-B line per 8 lanes, streaming (nontemporal: kernels_conv_h2.hip, profiles/r05_ab_streaming_stores.txt); columns past the image: out-of-range offset
      const bool ok = pxj < W;
      const int row_off = (((pn * H + py) * W + px0 + jj * 8) * ldy) * 4;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(unet_u32x4, t4), rs_y, ok ? st_lane : UNET_OOB, row_off, PP_STORE_AUX);
      if (stats) {
        const float kf = ok ? 1.f : 0.f;
        st1.x = fmaf(kf, t4.x, st1.x); st1.y = fmaf(kf, t4.y, st1.y); st1.z = fmaf(kf, t4.z, st1.z); st1.w = fmaf(kf, t4.w, st1.w);
        const float ux = kf * t4.x, uy = kf * t4.y, uz = kf * t4.z, uw = kf * t4.w;
        st2.x = fmaf(ux, t4.x, st2.x); st2.y = fmaf(uy, t4.y, st2.y); st2.z = fmaf(uz, t4.z, st2.z); st2.w = fmaf(uw, t4.w, st2.w);
      }
    }
  };

  // ---- the MFMA phase (P3 / P4): 9 taps x RW rows x 3 products per 16-channel chunk, as 9 KC steps (chunk, tap column kx, tap row ky) of 6 MFMAs.  The operands of
  // step s + 1 -- the weight pair of its tap and the one or two patch rows it adds to the four-row window -- are requested BEFORE the MFMAs of step s and land under
  // them; slice s of the previous tile's epilogue rides behind them.  HC / HP: this iteration has a tile to multiply / a previous tile to store (compile-time: the three
  // forms are straight-line code the scheduler interleaves)
  auto mphase = [&](auto HC_, auto HP_, const int it) __attribute__((always_inline)) {
    constexpr bool HC = decltype(HC_)::value, HP = decltype(HP_)::value;
    f16x8 px[2][RW + 2], wf[2][2];
    auto ld_row = [&](int ch, int kx, int rr) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 2; ++p) px[p][rr] = pp_frag(s_in + ch * CHUNK + p * PLANE + hi * HS + ((wave * RW + rr) * PWD + l31 + kx) * 16);
    };
    auto ld_w = [&](int s_) __attribute__((always_inline)) {
      const int ch = s_ / 9, kx = (s_ / 3) % 3, ky = s_ % 3;
#pragma unroll
      for (int p = 0; p < 2; ++p) wf[s_ & 1][p] = pp_frag(s_w + ch * W_BYTES + (((ky * 3 + kx) * 2 + p) * 2 + hi) * 512 + l31 * 16);
    };
    if (HC) {
#pragma unroll
      for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
      ld_row(0, 0, 0); ld_row(0, 0, 1); ld_w(0);
    }
#pragma unroll
    for (int s_ = 0; s_ < 9 * KC; ++s_) {
      if (HC) {
        const int ky = s_ % 3;
        if (s_ + 1 < 9 * KC) {
          const int s1 = s_ + 1, ch1 = s1 / 9, kx1 = (s1 / 3) % 3, ky1 = s1 % 3;
          ld_w(s1);
          if (ky1 == 0) { ld_row(ch1, kx1, 0); ld_row(ch1, kx1, 1); }          // (a new tap column: rows 0 and 1 of the window are dead by now)
          else ld_row(ch1, kx1, ky1 + 1);
        }
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};          // (weight plane, pixel plane): wh xm, wm xh, wh xh -- small terms first
#pragma unroll
          for (int r = 0; r < RW; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s_ & 1][PW[pr]], px[PX[pr]][r + ky], acc[r], 0, 0, 0);
        }
      }
      if (HP) eslice(s_ * 18 / (9 * KC));                  // (KC = 2: slice s; more chunks: the 18 slices spread over the steps)
      __builtin_amdgcn_sched_barrier(0);
      if (s_ == PP_MID) {
        PP_STAMP(7);
        __builtin_amdgcn_s_barrier();
        PP_STAMP(8);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  auto body = [&](const int it, unet_u32x4 (&preg)[PL]) __attribute__((always_inline)) {
    const int t_cur = t0 + it * slots, t_next = t_cur + slots, t_prev = t_cur - slots;
    const bool have_cur = it < iters && t_cur < t_end, have_next = it + 1 < iters && t_next < t_end, have_prev = it >= 1 && t_prev < t_end;
    // the previous tile's accumulators -> its output values (bias / border-class table, activation, the one-bit mask of a data gradient), which the epilogue slices of P3 / P4 store
    auto output_values = [&]() __attribute__((always_inline)) {
    {
        // lane (l31, hi) holds, for pixel column l31 of each of its RW rows, channels hi * 16 + 0..15 of the previous tile
        const int py0 = p_prev.y0, px0 = p_prev.x0;
        const float unscale = pp_pow2f(-e_prev) * w_unscale;
        const int px_ = px0 + l31;
        const bool border = tab_mask && (py0 == 0 || py0 + TH >= H || px0 == 0 || px0 + 32 >= W);
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 b4 = *reinterpret_cast<const float4*>(s_bias + hi * 16 + q * 4);
          bv[q * 4] = b4.x; bv[q * 4 + 1] = b4.y; bv[q * 4 + 2] = b4.z; bv[q * 4 + 3] = b4.w;
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int py = py0 + wave * RW + r;
          if (border && (py == 0 || py == H - 1 || px_ == 0 || px_ == W - 1)) {
            // forward of a conv whose input BatchNorm is folded into it: border pixels see fewer taps of the shift -- the bias vector of their border class (`mask` = table [16][M])
            const int cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px_ == 0) | ((px_ == W - 1) << 1));
            const float* tb = s_tab + cls * M + hi * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 b4 = *reinterpret_cast<const float4*>(tb + q * 4);
              vprev[r][q * 4] = fmaf(acc[r][q * 4], unscale, b4.x); vprev[r][q * 4 + 1] = fmaf(acc[r][q * 4 + 1], unscale, b4.y);
              vprev[r][q * 4 + 2] = fmaf(acc[r][q * 4 + 2], unscale, b4.z); vprev[r][q * 4 + 3] = fmaf(acc[r][q * 4 + 3], unscale, b4.w);
            }
          } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) vprev[r][i] = fmaf(acc[r][i], unscale, bv[i]);
          }
          if (act == ACT_RELU) {
#pragma unroll
            for (int i = 0; i < 16; ++i) vprev[r][i] = fmaxf(vprev[r][i], 0.f);
          }
          if (bits_mask) {
            // this lane's bits of word k: ((pixel % 4) * 8 + hi * 4 + q) for channel quad q -- a sign-extended one-bit field IS the AND mask of the value
            const unsigned sh = (l31 & 3) * 8 + hi * 4;
#pragma unroll
            for (int qq = 0; qq < 4; ++qq)
#pragma unroll
              for (int k = 0; k < 4; ++k) vprev[r][qq * 4 + k] = __uint_as_float(__float_as_uint(vprev[r][qq * 4 + k]) & (unsigned)__builtin_amdgcn_sbfe((int)mpre[BITS ? r : 0][k], sh + qq, 1u));
          }
        }
      }
    };
    // ---- P1: the patch of this iteration's tile has landed: its max |x| (this wave's pieces) -> LDS
    PP_STAMP(0);
    if (have_cur) {
      float mx = 0.f;
#pragma unroll
      for (int k = 0; k < PL; ++k) {
        if (VDY) { mx = fmaxf(mx, preg[k][1] ? fabsf(__uint_as_float(preg[k][0])) : 0.f); continue; }
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(preg[k][0]), "v"(preg[k][1]));
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(mx) : "v"(preg[k][2]), "v"(preg[k][3]));
      }
      mx = wave_max_nonneg(mx);
      if (lane == 0) s_amax[wave] = mx;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(1);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(2);
    // ---- P2: scale + split the patch into this half's planes; request the next tile's patch; the previous tile's output values (a data gradient's in front: its mask words
    // are overwritten by this tile's request; otherwise behind the requests, where the staging half has slack and the other half's P3 does not wait for it)
    if (BITS && have_prev) output_values();
    if (have_cur) {
      const float4 m4 = *reinterpret_cast<const float4*>(s_amax);
      const float mx = fmaxf(fmaxf(m4.x, m4.y), fmaxf(m4.z, m4.w));
      const int eb = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(mx) >> 23) & 0xFF));
      e_cur = min(138 - eb, 120);                          // the tile's maximum at [2^11, 2^12) (kernels_conv_h2.hip: scale_exp_for); numerically-zero tiles: 2^120
      const float sc = pp_pow2f(e_cur);
#pragma unroll
      for (int k = 0; k < PL; ++k) {
        if (pix0 + k * (256 / PPC) < NPIX) {               // (only the last k is partial)
          unsigned h0, m0, h1, m1;
          if (VDY) {
            unsigned hh, mm;
            split2_scaled(__uint_as_float(preg[k][0]), __uint_as_float(preg[k][0]), sc, hh, mm);
            const unsigned nib = preg[k][1] >> (pc * 4);   // bits 0..3: this piece's four channels (piece pc of the pixel = channels 4 pc .. 4 pc + 3)
            const unsigned k01 = h2_pair_mask(nib, 0), k23 = h2_pair_mask(nib, 2);
            h0 = hh & k01; h1 = hh & k23; m0 = mm & k01; m1 = mm & k23;
          } else {
          split2_scaled(__uint_as_float(preg[k][0]), __uint_as_float(preg[k][1]), sc, h0, m0);
          split2_scaled(__uint_as_float(preg[k][2]), __uint_as_float(preg[k][3]), sc, h1, m1);
          }
          *reinterpret_cast<uint2*>(s_in + dst0 + k * (256 / PPC) * 16) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(s_in + dst0 + k * (256 / PPC) * 16 + PLANE) = make_uint2(m0, m1);
        }
      }
    }
    PP_STAMP(3);
    if (bits_mask && have_cur) {
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int py = p_cur.y0 + wave * RW + r;
        const bool ok = py < H && p_cur.x0 + (l31 & ~7) < W;
        const int mo = ok ? (((p_cur.n * H + py) * (W >> 3) + (p_cur.x0 >> 3) + (l31 >> 3)) * 32 + ((l31 >> 2) & 1) * 4) : UNET_OOB;
#pragma unroll
        for (int k = 0; k < 4; ++k) mpre[BITS ? r : 0][k] = __builtin_amdgcn_raw_buffer_load_b32(rs_m, mo, k * 8, 0);
      }
    }
    // (requests issued among the split's instructions stall the wave in front of the memory pipeline's queue -- measured +30 % per tile: all of them behind it)
    __builtin_amdgcn_sched_barrier(0);
    if (have_next && (PP_EXP & 2) == 0) issue_loads(p_next, preg);
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(4);
    if (!BITS && have_prev) output_values();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PP_STAMP(5);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(6);
    __builtin_amdgcn_sched_barrier(0);
    if (have_cur && have_prev) mphase(std::true_type{}, std::true_type{}, it);
    else if (have_cur) mphase(std::true_type{}, std::false_type{}, it);
    else if (have_prev) mphase(std::false_type{}, std::true_type{}, it);
    else { PP_STAMP(7); __builtin_amdgcn_s_barrier(); PP_STAMP(8); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(9);
    __builtin_amdgcn_s_barrier();                          // (every wave of the half is past its last fragment read: the planes may be overwritten)
    PP_STAMP(10);
    e_prev = e_cur; p_prev = p_cur; p_cur = p_next; p_next = advance(p_next);
  };
  for (int it = 0; it <= iters; ++it) body(it, pregA);
  if (!half) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }

  // ---- the workgroup's BatchNorm statistics: lanes L, L + 8, ... hold the same channel quad -- fold them, then the eight waves through LDS, one atomic per channel and sum
  if (stats) {
    float sv[8] = {st1.x, st1.y, st1.z, st1.w, st2.x, st2.y, st2.z, st2.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) { sv[i] += __shfl_xor(sv[i], 8); sv[i] += __shfl_xor(sv[i], 16); sv[i] += __shfl_xor(sv[i], 32); }
    if (lane < 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { s_stat[(wave8 * 2 + 0) * 32 + lane * 4 + i] = sv[i]; s_stat[(wave8 * 2 + 1) * 32 + lane * 4 + i] = sv[4 + i]; }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int kind = threadIdx.x >> 5, c32 = threadIdx.x & 31;
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) t += s_stat[(wv * 2 + kind) * 32 + c32];
      double* const row = stats + (size_t)(blockIdx.x % UNET_BN_SLOTS) * UNET_BN_SLOT_DOUBLES;
      if (xs) xsum_add(row, (kind ? stats_c : 0) + c32, t); else atomicAdd(row + (kind ? stats_c : 0) + c32, (double)t);
    }
  }
}

}  // namespace

// The launches this schedule takes: K = 32 contraction channels, M = 32 output channels (c1b / c9b forward, c1b data gradient), plain / border-class-table /
// one-bit-mask epilogues, enough tiles to give every half-workgroup of the chip a few.  Everything else stays with conv_h2_kernel.
bool pp_conv3x3_selected(const unet_ctx* ctx, int K, int M, int n, int h, int wd, const float* mask, int mask_mode, int act, float rate, int ldy) {
  if (!ctx || !ctx->opt_conv_pp) return false;
  if (K != 32 || M != 32 || rate != 0.0f || (act != ACT_NONE && act != ACT_RELU)) return false;
  if (mask && mask_mode != MASK_BIAS_TAB && mask_mode != MASK_RELU_BITS) return false;
  if (mask && mask_mode == MASK_RELU_BITS && (wd & 7)) return false;
  if (ldy < M || (ldy & 3)) return false;
  const long long tiles = (long long)((wd + 31) / 32) * ((h + 7) / 8) * n;
  if (tiles < 4LL * 2 * ctx->num_cu && ctx->opt_conv_pp < 2) return false;          // (fewer than four tiles per half-workgroup: the pipeline's fill and drain would dominate)
  if ((long long)n * h * wd * K * 4 + (long long)(wd + 1) * K * 4 >= (1LL << 31)) return false;          // 32-bit buffer offsets over the whole tensor
  return true;
}

int32_t k_conv3x3_pp_fwd(unet_ctx* ctx, const float* x, const void* wimg, const float* bias, const float* mask, int mask_mode, float* y, int ldy, int n, int h, int wd, int K, int M,
                         int act, hipStream_t s, bool vdy) {
  if (K != 32 || M != 32) UNET_FAIL(ctx, UNET_E_SHAPE, "conv pp: K=%d M=%d", K, M);
  constexpr int KC = 2;
  if (!mask) mask_mode = MASK_NONE;
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + 7) / 8;
  const long long total = (long long)tiles_x * tiles_y * n;
  // an armed statistics / sign-bit request (common.h), as launch_h2 takes them
  double* stats = nullptr; int stats_c = 0;
  if (ctx->stats_req_c > 0) {
    const int c = ctx->stats_req_c; ctx->stats_req_c = 0;
    const int xw = ctx->opt_deterministic ? UNET_XW : 1;
    if ((mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) && 2 * c * xw <= UNET_BN_SLOT_DOUBLES && c == M && ctx->bn_slots) {
      stats = ctx->bn_slots; stats_c = c; ctx->stats_in_slots = y; ctx->stats_in_slots_c = c; ctx->stats_in_slots_xs = ctx->opt_deterministic != 0;
    }
  }
  unsigned long long* signs = nullptr;
  if (ctx->signs_req) {
    unsigned long long* q = ctx->signs_req; ctx->signs_req = nullptr;
    if (act == ACT_RELU && (mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) && !(wd & 7)) { signs = q; ctx->signs_done = q; }
  }
  constexpr int NPIX = 340, CHUNK = 4 * NPIX * 16 + 32;
  constexpr size_t smem = (size_t)KC * 9 * 2048 + 2 * (size_t)KC * CHUNK + 4 * 4096 + 32 + 128 + 8 * 2 * 32 * 4 + 16 * 32 * 4;
  if (vdy && (stats || signs || bias || act != ACT_NONE || mask_mode == MASK_BIAS_TAB)) UNET_FAIL(ctx, UNET_E_ARG, "conv pp behind the head's {dz, mask} stream: a plain data-gradient launch only");
  auto kern = vdy ? (mask_mode == MASK_RELU_BITS ? conv_pp_kernel<KC, true, true> : conv_pp_kernel<KC, false, true>)
                  : (mask_mode == MASK_RELU_BITS ? conv_pp_kernel<KC, true, false> : conv_pp_kernel<KC, false, false>);
  UNET_BIG_LDS(ctx, kern, smem, "conv_pp");
  unet_note_kernel(ctx, reinterpret_cast<const void*>(kern));
  const unsigned grid = (unsigned)(ctx->num_cu & ~7);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, x, static_cast<const unet_bf16*>(wimg), bias, mask, y, ldy, n, h, wd, act, mask_mode, tiles_x, tiles_y, (int)total, stats,
                     stats_c, signs, ctx->opt_deterministic ? 1 : 0);
  UNET_CHECK_LAUNCH(ctx, "conv_pp");
  return UNET_OK;
}
