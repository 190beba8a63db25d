// fp32 3x3 convolution (forward + data gradient) on the 16-bit matrix cores with fp32 accuracy: the "h2" kernels.
//
// gfx950's fp32 MFMA runs at the vector rate (157 TFLOP/s, no TF32); v_mfma_f32_32x32x16_f16 runs 16x faster.  An fp32 number that sits in the
// comfortable part of the fp16 range is, to 22 significant bits, the sum of TWO fp16 numbers: x ~ h + m, h = RN_f16(x), m = RN_f16(x - h)
// (11 + 11 bits; x - h is exact in fp32), so
//     x * w = xh wh + xh wm + xm wh + O(2^-22 |x w|)
// -- three fp16 MFMAs with fp32 accumulation per multiply, 3/16 of the fp32-MFMA time (an exact three-term bf16 split needs six products: measured, not kept).
// What fp16 lacks is RANGE (activation gradients are ~1e-8), so both operands are block-scaled by exact powers of two:
//   * weights: one exponent per layer, from the layer's max |w| (h2_wmax_kernel), folded into the weight image; max |w| 2^e lands in [2^11, 2^12);
//   * activations / gradients: one exponent per workgroup, tracked along the K loop.  While a 16-channel chunk of the input patch is staged, the workgroup
//     takes its max |x| (registers -> wave shuffle -> 4 LDS words, no extra barrier); if the chunk would come within 2x of the fp16 maximum under the
//     running exponent, the exponent is lowered to put that maximum at [2^11, 2^12) and the fp32 accumulators are multiplied by the (exact)
//     power-of-two ratio.  So nothing can overflow, and every element keeps >= 22 bits down to 2^-14 of the largest element the workgroup has seen
//     (absolute precision 2^-36 of that maximum below).  The epilogue multiplies by 2^-(e_x + e_w).
// Measured on the box (tools/probe/split3_probe.hip, K = 16 ... 4608, against float64): relative L2 error 7.1e-8 / 2.2e-7 / 8.5e-7 for the three fp16
// products vs 7.1e-8 / 3.2e-7 / 1.3e-6 for v_mfma_f32_32x32x2_f32: the same accuracy class as the fp32 matrix path it replaces.
//
// Kernel structure = the implicit GEMM of kernels_bf16.hip: A = weights (32 output channels x 16 k), B = 32 pixels of an image row,
// a lane ends with 16 consecutive output channels of one pixel.  Per 16-channel chunk: (TH + 2) x 34 pixel patch as two fp16 planes + the weight slab
// (two planes) in LDS, single-buffered; the next chunk travels global -> registers under the current chunk's MFMAs and is scaled, split and stored
// between two barriers while the CU's other workgroup(s) keep the matrix pipes busy (58 KB and ~200 registers per workgroup: two per CU).
#include <stdlib.h>

#include <algorithm>

#include "common.h"

#ifndef H2_EXPERIMENT
#define H2_EXPERIMENT 0
#endif

#if H2_EXPERIMENT == 5
// measurement build only (tools/h2_timeline.py): wave 0 of every workgroup stamps s_memtime at its phase boundaries
__device__ unsigned long long h2_trace[16384 * 16];
#define H2_STAMP(i) do { if (tid == 0 && blockIdx.x < 16384) h2_trace[blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int32_t unet_debug_h2_trace(unsigned long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(h2_trace), (size_t)n * 8) == hipSuccess ? 0 : 1;
}
#else
#define H2_STAMP(i) do { } while (0)
#endif
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unet_f32x16 f32x16;
constexpr int H2_HEADER = 256;                              // bytes in front of a weight image: [0] = 2^-e_w (float)

__host__ __device__ inline int cperm(int m) { return ((m >> 2) & 1) * 16 + (m & 3) + 4 * (m >> 3); }       // lane -> 16 consecutive channels (kernels_bf16.hip)

// two values -> packed fp16 pairs h and m with a ~ h + m
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m) {
  const f16x2 hh = __builtin_convertvector((unet_f32x2){a, b}, f16x2);
  const f16x2 mm = __builtin_convertvector((unet_f32x2){a - (float)hh[0], b - (float)hh[1]}, f16x2);
  h = __builtin_bit_cast(unsigned, hh); m = __builtin_bit_cast(unsigned, mm);
}
// exponent e (as the float 2^e) that puts a block maximum with biased exponent field `eb` into [2^11, 2^12); eb < 11: the block is numerically zero
__device__ __forceinline__ int scale_exp_for(int eb) { return 138 - eb; }
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }          // -126 <= e <= 127

// ---------------------------------------------------------------------------------------------------------------------
// weights: the split fp16 image of a layer (conv3x3: T = 9 taps, KS = 1 k-step per staged chunk; ConvT: T = 1, KS = 2), TWO launches for any number of layers:
//   img = [header 256 B][(((((((g*nchunks + chunk)*KS + ks)*T + tap)*NB + nb)*2 + plane)*2 + half)*32 + m][8 fp16]
//   k = (chunk*KS + ks)*16 + half*8 + j,  mm = (g*NB + nb)*32 + cperm(m),  W(tap,k,mm) = w[(flip ? T-1-tap : tap)*tap_stride + k*sk + mm*sm] * cs[k] * 2^e_w
//   (cs = optional per-input-channel factor: the scale of a BatchNorm folded into the conv, DESIGN.md 4f -- the scaled weights are never materialised)
// header floats: [0] = 2^-e_w, [1] = 2^e_w (written by block 0 of the image kernel), [8 .. 8 + H2_MAXB) = per-workgroup partial max |w cs| of h2_wmax_kernel.
// Every block of the image kernel folds the partial maxima itself (same values, same order -> same exponent): no atomics, no zeroing, no third launch.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int H2_MAXB = 48;                                  // workgroups per layer of the max pass
struct h2_prep { const float* w; const float* cs; unet_bf16* img; long long tap_stride, sk, sm, total, nw4; int T, KS, nb, nchunks, flip, m, cs_div, cs_mod, max_only, cs_bound; };
struct h2_prep_list { h2_prep item[UNET_PREP_MAX]; int n; };          // passed by value as a kernel argument (3.4 KiB)

constexpr int H2_MAXT = 1024;                                // threads per workgroup of the max pass: 16 waves per partial maximum keep a 2.4 M-weight layer at 3 rounds of loads per thread
__global__ __launch_bounds__(H2_MAXT) void h2_wmax_kernel(h2_prep_list L) {          // grid (H2_MAXB, layers)
  const h2_prep& p = L.item[blockIdx.y];
  float mx = 0.f;
  auto one = [&](long long i) __attribute__((always_inline)) {
    const float4 v = reinterpret_cast<const float4*>(p.w)[i];
    float m4 = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    if (p.cs && !p.cs_bound) m4 *= fabsf(p.cs[(int)((i * 4 / p.cs_div) % p.cs_mod)]);          // (the four elements of a float4 share their input channel: cs_div is a multiple of 4)
    return m4;
  };
  // (four independent loads per thread and round: with one, the 2.4 M weights of a 512 -> 512 layer were 48 dependent round trips per thread -- 23 us per launch)
  const long long st = (long long)gridDim.x * H2_MAXT;
  long long i = (long long)blockIdx.x * H2_MAXT + threadIdx.x;
  for (; i + 3 * st < p.nw4; i += 4 * st) mx = fmaxf(fmaxf(mx, fmaxf(one(i), one(i + st))), fmaxf(one(i + 2 * st), one(i + 3 * st)));
  for (; i < p.nw4; i += st) mx = fmaxf(mx, one(i));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  __shared__ float s_m[H2_MAXT / 64];
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = s_m[0];
#pragma unroll
    for (int k = 1; k < H2_MAXT / 64; ++k) m = fmaxf(m, s_m[k]);
    reinterpret_cast<float*>(p.img)[8 + blockIdx.x] = m;
  }
}

__global__ __launch_bounds__(256) void h2_wimg_kernel(h2_prep_list L, int maxb) {          // blockIdx.y = layer
  const h2_prep& p = L.item[blockIdx.y];
  if (p.max_only) return;                                    // (kind 4: only the partial maxima of the raw weights were wanted -- the image follows once the folded BatchNorm's scale exists)
  const int NB = p.nb, nchunks = p.nchunks, M = p.m, T = p.T, KS = p.KS;
  float* const hdr = reinterpret_cast<float*>(p.img);
  float mx = (int)(threadIdx.x & 63) < maxb ? hdr[8 + (threadIdx.x & 63)] : 0.f;          // (every wave folds the partial maxima: no barrier)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (p.cs_bound) {
    // the partial maxima are those of the RAW weights (taken with the program's batch): max |w cs| <= max |w| max |cs| -- an exponent from the bound costs at most the
    // ratio in dynamic range below the 2^-14 the split keeps, and saves this layer its own pass over the weights
    float mc = 0.f;
    for (int i = threadIdx.x & 63; i < p.cs_mod; i += 64) mc = fmaxf(mc, fabsf(p.cs[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mc = fmaxf(mc, __shfl_xor(mc, o));
    mx *= mc;
  }
  const int eb = (int)((__float_as_uint(mx) >> 23) & 0xFF);
  int e = eb >= 11 ? scale_exp_for(eb) : 0;
  e = min(max(e, -100), 100);
  const float sc = pow2f(e);
  if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = pow2f(-e); hdr[1] = sc; }
  unet_bf16* const img = p.img + H2_HEADER / 2;
  for (long long el = (long long)blockIdx.x * 256 + threadIdx.x; el < p.total; el += (long long)gridDim.x * 256) {          // el indexes (g, chunk, ks, tap, nb, half, m)
    long long r = el;
    const int m = (int)(r & 31); r >>= 5;
    const int half = (int)(r & 1); r >>= 1;
    const int nb = (int)(r % NB); r /= NB;
    const int tap = (int)(r % T); r /= T;
    const int ks = (int)(r % KS); r /= KS;
    const int chunk = (int)(r % nchunks); const int g = (int)(r / nchunks);
    const long long k0 = ((long long)chunk * KS + ks) * 16 + half * 8;
    const long long mm = ((long long)g * NB + nb) * 32 + cperm(m);
    const float* src = p.w + (long long)(p.flip ? T - 1 - tap : tap) * p.tap_stride + k0 * p.sk + mm * p.sm;
    unsigned hh[4], ml[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = mm < M ? src[(2 * j) * p.sk] * sc : 0.0f, b = mm < M ? src[(2 * j + 1) * p.sk] * sc : 0.0f;
      if (p.cs) { a *= p.cs[k0 + 2 * j]; b *= p.cs[k0 + 2 * j + 1]; }
      split2(a, b, hh[j], ml[j]);
    }
    const long long base = ((((((long long)g * nchunks + chunk) * KS + ks) * T + tap) * NB + nb) * 2 * 2 + half) * 32 + m;        // plane 0; plane 1 is 2 * 32 rows further
    *reinterpret_cast<uint4*>(img + (base + 0 * 64) * 8) = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    *reinterpret_cast<uint4*>(img + (base + 1 * 64) * 8) = make_uint4(ml[0], ml[1], ml[2], ml[3]);
  }
}

__device__ __forceinline__ f16x8 lds_frag(const char* p) { return *reinterpret_cast<const f16x8*>(p); }
// byte offset of 16-B cell c (0..7) of pixel p (0..31) in a wave's 4-KB epilogue staging row
__device__ __forceinline__ int out_cell(int p, int c) { return p * 128 + ((c ^ (p & 7)) << 4); }

// MODE 0: conv3x3 'same' (forward; data gradient with the flipped / transposed weight image)
// MODE 1: convT2x2s2 forward = per-pixel GEMM [pixels, Cin] x [Cin, 4 Cout] with a scatter epilogue into the (2i+a, 2j+b) positions of a channel slice (pixel
//         stride ldy) of the concat buffer (T1:886-887)
// MODE 2: convT2x2s2 data gradient = per-pixel GEMM over the virtual channels k = (ab, o): chunk -> (ab, o0) selects the parity plane (2i+a, 2j+b) of dU
//         (pixel stride ldx) that is staged
// HEAD (the network's last conv3x3, T1:911-913): the 1x1 sigmoid head, the four loss sums and the three per-channel sums the head's weight gradient is a
// combination of are taken from the output tile while it is in registers / LDS (h2_head_args; one channel group of 32, ReLU, no mask)
// SPLITK (conv3x3 launches that leave most CUs idle -- the 32 x 32 ... 128 x 128 levels of batch-1 inference, T1:1137): blockIdx.y = a slice of `kcps` staged chunks of the
// contraction; the slice's partial sums (no bias, no activation) go to slab blockIdx.y of `y` (slabs `y_split` floats apart) and h2_splitk_finish_kernel adds the slabs,
// the bias / border-class table and the ReLU.  A tile's K loop is a chain of dependent global loads (~1.7 us per chunk with one workgroup per CU and 27 MFMAs a chunk):
// the slices run it in parallel
template <int MODE, int NB, int RW, bool GEN, int WPS, int EPI = 0, bool SPLITK = false>
__global__ __launch_bounds__(256, WPS) void conv_h2_kernel(const float* __restrict__ x, int ldx, const unet_bf16* __restrict__ wimg_hdr, const float* __restrict__ bias,
                                                           const float* __restrict__ mask, float* __restrict__ y, int ldy, int N, int H, int W, int K, int M, int act,
                                                           int mask_mode, float rate, unsigned long long seed, int tiles_x, int tiles_y, int groups,
                                                           int total_blocks, double* __restrict__ stats, int stats_c, unsigned long long* __restrict__ signs,
                                                           int mask_climit, h2_head_args hd, int img_nb, int xs, int kcps, long long y_split) {
  constexpr bool HEAD = EPI == 1, POOLS = EPI == 2;          // EPI: 0 the general epilogue, 1 + the 1x1 sigmoid head (below), 2 + the pooled-path sums of an encoder tail (MASK_POOL_SUMS)
  // EPI 3 (VDY): the general epilogue behind a VIRTUAL input -- the gradient of the last conv3x3's output, dy[p][c] = dz_p w_c [y_pc > 0] (T1:911-913 backwards), staged from
  // the 8-byte-per-pixel stream {dz_p, 32 mask bits} of head_dzm_kernel (x = that stream, ldx = 2): one value is scaled and split per staged piece, the mask bits pick
  // the channels it goes to; w_c is a per-contraction-channel factor of the weight image (h2_prep::cs).  K = 32.
  constexpr bool VDY = EPI == 3;
  static_assert(!VDY || (MODE == 0 && NB == 1 && RW == 2 && !GEN), "the virtual head gradient feeds the 32-channel data-gradient launch");
  static_assert(!HEAD || (MODE == 0 && NB == 1 && RW == 2 && !GEN), "the fused head rides on the 32-channel forward kernel");
  static_assert(!POOLS || (MODE == 0 && !GEN), "the pooled sums ride on a plain conv3x3 data-gradient launch");
  static_assert(!SPLITK || (MODE == 0 && EPI == 0 && !GEN), "K slices: plain conv3x3 launches");
  if (POOLS) { mask_mode = MASK_POOL_SUMS; act = ACT_NONE; signs = nullptr; }          // (compile-time facts of this instance: the other epilogue forms fall away)
  if (HEAD) { mask_mode = MASK_NONE; act = ACT_RELU; stats = nullptr; }
  constexpr int T = MODE == 0 ? 9 : 1, KS = MODE == 0 ? 1 : 2;          // taps; 16-channel k-steps per staged chunk
  constexpr int TH = 4 * RW;                             // tile rows: RW per wave
  constexpr int PR = MODE == 0 ? TH + 2 : TH, PWD = MODE == 0 ? 34 : 32, NPIX = PR * PWD;
  // one fp16 plane of the pixel patch: [k-step][channel half][pixel][8 channels] -- a fragment read (ds_read_b128, serviced in groups of 16 lanes out of
  // lanes 0-31 / 32-63) then covers 256 consecutive bytes = all 64 banks; with the 16 channels of a pixel side by side (32 B per pixel) the 16 lanes of a
  // group hit only every other 16-B bank quad: every fragment read paid two LDS cycles per group (tools/lds_banks.py; round 3 PMC: 16-25 % conflict cycles).
  // (ConvT patches: NPIX is a multiple of 256 -> the two halves would sit on the same banks for the staging stores: 64 B of padding per half)
  constexpr int HS = NPIX * 16 + (MODE == 0 ? 0 : 64);   // bytes of one channel half
  constexpr int PLANE = KS * 2 * HS;
  constexpr int IN_BYTES = 2 * PLANE, W_BYTES = KS * T * NB * 2 * 2 * 32 * 16;
  constexpr int PPIECES = KS * NPIX * 4, WPIECES = W_BYTES / 16;     // 16-B fp32 pieces of the patch (k-step, pixel, channel quad); 16-B pieces of the weight slab
  constexpr int PL = (PPIECES + 255) / 256, WL = (WPIECES + 255) / 256;
  constexpr int OUT_PS = 128;                            // epilogue staging: bytes per pixel of a (row, 32-channel block); 16-B cell c of pixel p sits at cell c ^ (p & 7):
                                                         // conflict-free for the accumulator-layout side (8 pixels x one cell) AND the line-layout side (8 cells of a pixel)
  // the slab's LDS region is rounded up to whole 256-piece rounds (every thread stores every piece it loaded) -- except for the 32-channel groups, where the
  // exact size is what lets a FOURTH workgroup fit a CU's 160 KB (40.2 KB each): those layers are the 512 x 512 ones, bound by the bytes a CU keeps in flight
  constexpr bool WPAD = NB != 1 || MODE != 0;
  constexpr int W_LDS = WPAD ? WL * 256 * 16 : W_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];        // [IN_BYTES] [W_LDS] [4 floats: per-wave max |x| of the chunk being staged]
  char* const s_in = smem; char* const s_w = smem + IN_BYTES;
  float* const s_amax = reinterpret_cast<float*>(smem + IN_BYTES + W_LDS);
  float* const s_bias = s_amax + 4;                      // [3][NB * 32]: bias (+ the two other coefficient rows of a folded-BatchNorm gradient) of this channel group

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware block map (workgroup b runs on XCD b % 8): an XCD gets a contiguous range of work items -- the channel groups of one spatial tile,
  // then the neighbouring tiles -- so a patch is fetched into ONE L2
  const int per = gridDim.x >> 3;
  const int wi = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (wi >= total_blocks) return;
  const int g = wi % groups; int t = wi / groups;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; const int n = t / tiles_y;
  const int x0 = tx * 32, y0 = ty * TH;
  const int HI = MODE == 2 ? 2 * H : H, WI = MODE == 2 ? 2 * W : W;          // the staged tensor's image size
  const int nchunks = K / (16 * KS);
  const int cbeg = SPLITK ? (int)blockIdx.y * kcps : 0, cend = SPLITK ? min(cbeg + kcps, nchunks) : nchunks;          // the staged chunks this workgroup contracts over
  if (SPLITK) y += (long long)blockIdx.y * y_split;
  const unet_bf16* const wimg = wimg_hdr + H2_HEADER / 2;
  const float w_unscale = *reinterpret_cast<const float*>(wimg_hdr);          // 2^-e_w of the layer

  f32x16 acc[RW][NB];
#pragma unroll
  for (int i = 0; i < RW; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (long long)n * HI * WI * ldx, (long long)HI * WI * ldx * 4);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unet_bf16*>(wimg), 0, (int)((long long)groups * nchunks * W_BYTES), 0x00020000);
  int poff[PL];
#pragma unroll
  for (int k = 0; k < PL; ++k) {
    const int idx = tid + k * 256;
    const int q = idx & 3, pp = idx >> 2, ks = pp / NPIX, pix = pp - ks * NPIX;
    int gy, gx;
    if (MODE == 0) { const int r = pix / PWD, c = pix - r * PWD; gy = y0 + r - 1; gx = x0 + c - 1; }
    else { gy = y0 + (pix >> 5); gx = x0 + (pix & 31); }
    const bool ok = idx < PPIECES && gy >= 0 && gy < H && gx >= 0 && gx < W;
    if (MODE == 2) poff[k] = ok ? (((2 * gy) * WI + 2 * gx) * ldx + ks * 16 + q * 4) * 4 : UNET_OOB;
    else if (VDY) poff[k] = ok ? (gy * WI + gx) * 8 : UNET_OOB;                            // (the four pieces of a pixel read the same 8 bytes)
    else poff[k] = ok ? ((gy * WI + gx) * ldx + ks * 16 + q * 4) * 4 : UNET_OOB;          // halo and overhang pieces read 0 (out-of-range buffer offset)
  }
  unet_u32x4 preg[PL], wreg[WL];
  auto issue_loads = [&](int chunk) __attribute__((always_inline)) {
    if (VDY) {                                             // both chunks expand the same {dz, mask} words: fetched once
      if (chunk == 0) {
#pragma unroll
        for (int k = 0; k < PL; ++k) { const unet_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, poff[k], 0, 0); preg[k][0] = v[0]; preg[k][1] = v[1]; }
      }
      return;
    }
    int soff;
    if (MODE == 2) { const int ct = K >> 2, k0 = chunk * 32, ab = k0 / ct, o0 = k0 - ab * ct; soff = (((ab >> 1) * WI + (ab & 1)) * ldx + o0) * 4; }
    else soff = chunk * 16 * KS * 4;
#pragma unroll
    for (int k = 0; k < PL; ++k) preg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, poff[k], soff, 0);
  };
  // the weight slab of a chunk: 27-36 KB that every workgroup of the channel group reads -- L2 hits (~250 cycles).  It is requested AFTER the chunk's MFMAs
  // (behind the barrier), so its staging registers share the MFMA operand registers, and lands while the patch is scaled and split
  // img_nb > NB (NB = 1 only): the image was built for img_nb 32-channel blocks per workgroup (h2_nb(M) = 2; 4 for a ConvT forward) and this launch runs ONE block per
  // workgroup because the wider grid would leave CUs idle (batch-1 inference, the deep levels of small batches): group g is block g % img_nb of the image's group
  // g / img_nb, whose slab holds [tap | k-step][block][plane][half][32] pieces -- 128 consecutive pieces per (tap, block), the taps img_nb * 128 pieces apart
  const bool wide_img = NB == 1 && img_nb > 1;
  auto issue_w_loads = [&](int chunk) __attribute__((always_inline)) {
    const int wsoff = wide_img ? ((g / img_nb) * nchunks + chunk) * (img_nb * W_BYTES) + (g % img_nb) * 2048 : (g * nchunks + chunk) * W_BYTES;
#pragma unroll
    for (int k = 0; k < WL; ++k) {
      const int idx = tid + k * 256;
      const int po = wide_img ? ((idx >> 7) * img_nb * 128 + (idx & 127)) * 16 : idx * 16;
      wreg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, idx < WPIECES ? po : UNET_OOB, wsoff, 0);
    }
  };
  // max |x| of the pieces this wave holds -> s_amax[wave]
  auto post_amax = [&]() __attribute__((always_inline)) {
    float mx = 0.f;
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      if (VDY) { mx = fmaxf(mx, preg[k][1] ? fabsf(__uint_as_float(preg[k][0])) : 0.f); continue; }
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(__uint_as_float(preg[k][j])));
    }
    mx = wave_max_nonneg(mx);
    if (lane == 0) s_amax[wave] = mx;
  };
  int e_run = 120;                                           // running exponent of the activation operand: staged values are x * 2^e_run
  int st_chunk = 0;                                          // (VDY) the chunk the next store_lds stages
  // after the barrier that follows post_amax: lower the running exponent if this chunk needs it (rescaling the accumulators), then scale, split and store
  auto store_lds = [&]() __attribute__((always_inline)) {
    const float mx = fmaxf(fmaxf(s_amax[0], s_amax[1]), fmaxf(s_amax[2], s_amax[3]));
    const int eb = __builtin_amdgcn_readfirstlane((int)((__float_as_uint(mx) >> 23) & 0xFF));
    if (eb >= 11 && eb - 127 + e_run >= 15) {                // the chunk's maximum would land at >= 2^15 (fp16 overflows at 2^16): re-centre at [2^11, 2^12)
      const int e_new = scale_exp_for(eb);
      const int d = max(e_new - e_run, -126);
      const float f = pow2f(d);
#pragma unroll
      for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
      e_run = e_new;
    }
    const float sc = pow2f(e_run);
#pragma unroll
    for (int k = 0; k < PL; ++k) {
      const int idx = tid + k * 256;
      if (idx < PPIECES) {
        unsigned h0, m0, h1, m1;
        if (VDY) {
          const float v = __uint_as_float(preg[k][0]) * sc;
          unsigned hh, mm;
          split2(v, v, hh, mm);
          const unsigned nib = preg[k][1] >> (st_chunk * 16 + (tid & 3) * 4);          // bits 0..3: this piece's four channels
          const unsigned k01 = h2_pair_mask(nib, 0), k23 = h2_pair_mask(nib, 2);
          h0 = hh & k01; h1 = hh & k23; m0 = mm & k01; m1 = mm & k23;
        } else {
        split2_scaled(__uint_as_float(preg[k][0]), __uint_as_float(preg[k][1]), sc, h0, m0);
        split2_scaled(__uint_as_float(preg[k][2]), __uint_as_float(preg[k][3]), sc, h1, m1);
        }
        const int q = tid & 3, pp = (tid >> 2) + k * 64;                // (k * 64 / NPIX is a compile-time constant in the ConvT modes: NPIX % 64 == 0 there)
        const int ks = MODE == 0 ? 0 : (k * 64) / NPIX;
        char* p = s_in + ((ks * 2 + (q >> 1)) * HS + (pp - ks * NPIX) * 16 + (q & 1) * 8);          // piece (pixel, quad) -> 8 bytes of its channel half
        *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(p + PLANE) = make_uint2(m0, m1);
      }
    }
#pragma unroll
    for (int k = 0; k < WL; ++k)
      if (WPAD || tid + k * 256 < WPIECES) *reinterpret_cast<unet_u32x4*>(s_w + (tid + k * 256) * 16) = wreg[k];          // (pieces past the slab: zeros into the padding)
    ++st_chunk;
  };

  H2_STAMP(0);
  // the epilogue's per-channel operands go through LDS: requested from the epilogue they queue behind every other workgroup's patch loads in the CU's
  // memory pipeline (a 2-4 k-cycle round trip per tile); here they are the first request of the workgroup and land before its first barrier
  if (tid < NB * 32) {
    const int ch = g * NB * 32 + tid;
    const bool ok = ch < M;
    s_bias[tid] = (bias && ok) ? bias[MODE == 1 ? ch % (M >> 2) : ch] : 0.f;
    const bool bn_bwd_mode = mask_mode >= MASK_BN_BWD && mask_mode <= MASK_BN_BWD_RELU;
    const bool coef = bn_bwd_mode && mask && ok;
    if (POOLS) {                                           // row 1 = 1 / gamma (0 where gamma is 0), row 2 = beta of the encoder BatchNorm in front of the pooled tensor
      const float gm = ok ? hd.w[ch] : 0.f;
      s_bias[NB * 32 + tid] = gm != 0.f ? 1.0f / gm : 0.f; s_bias[2 * NB * 32 + tid] = ok ? hd.b[ch] : 0.f;
    } else {
    s_bias[NB * 32 + tid] = HEAD ? hd.w[tid] : coef ? bias[M + ch] : 0.f;          // (HEAD: row 1 = the 32 weights of the 1x1 head, row 2 [0] = its bias)
    s_bias[2 * NB * 32 + tid] = HEAD ? hd.b[0] : coef ? bias[2 * M + ch] : 0.f;
    }
  }
  issue_loads(cbeg);
  issue_w_loads(cbeg);
  post_amax();
  __syncthreads();
  H2_STAMP(1);
  store_lds();
  __syncthreads();
  H2_STAMP(2);
  // one staged chunk: 9 taps x RW rows x NB blocks x 3 products (ConvT modes: KS k-steps x ...)
  auto mfma_chunk = [&]() __attribute__((always_inline)) {
    if (MODE == 0) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      f16x8 px[2][RW + 2];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int rr = 0; rr < RW + 2; ++rr) px[p][rr] = lds_frag(s_in + p * PLANE + hi * HS + ((wave * RW + rr) * PWD + l31 + kx) * 16);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        f16x8 wf[NB][2];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[nb][p] = lds_frag(s_w + (((((ky * 3 + kx) * NB + nb) * 2 + p) * 2 + hi) * 512 + l31 * 16));
        // three products per (row, block), small terms first; consecutive MFMAs go to different accumulators
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};            // (weight plane, pixel plane): wh xm, wm xh, wh xh
          // (block-major: RW consecutive MFMAs keep the weight operand -- measured 0.3 % of the step against row-major, six same-box A/B rounds)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < RW; ++r)
              acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb][PW[pr]], px[PX[pr]][r + ky], acc[r][nb], 0, 0, 0);
        }
      }
    }
    } else {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f16x8 px[2][RW], wf[NB][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < RW; ++r) px[p][r] = lds_frag(s_in + p * PLANE + (ks * 2 + hi) * HS + ((wave * RW + r) * 32 + l31) * 16);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int p = 0; p < 2; ++p) wf[nb][p] = lds_frag(s_w + ((((ks * NB + nb) * 2 + p) * 2 + hi) * 512 + l31 * 16));
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          constexpr int PW[3] = {0, 1, 0}, PX[3] = {1, 0, 0};
#pragma unroll
          for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[r][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[nb][PW[pr]], px[PX[pr]][r], acc[r][nb], 0, 0, 0);
        }
      }
    }
  };
  for (int chunk = cbeg; chunk + 1 < cend; ++chunk) {
    issue_loads(chunk + 1);
    __builtin_amdgcn_s_setprio(2);                           // (the wave feeding the matrix pipe goes ahead of the co-resident workgroups' staging / epilogue: 0.3 % of the step)
    mfma_chunk();
    __builtin_amdgcn_s_setprio(0);
    if (chunk == cbeg) H2_STAMP(3);
    issue_w_loads(chunk + 1);                              // (behind this wave's last MFMA issue: the operand registers are free; requested BEFORE the MFMAs into registers of
                                                           //  its own -- the two-block 8-row kernels have 60 to spare -- it measured +0.10 ms per step: profiles/r05_ab_early_weight_slab.txt)
    post_amax();
    __syncthreads();                                       // every wave is done reading this chunk's planes; the four partial maxima are visible
    if (chunk == cbeg) H2_STAMP(4);
    store_lds();
    __syncthreads();
    if (chunk == cbeg) H2_STAMP(5);
  }
  H2_STAMP(6);
  // ---- last chunk: what the epilogue reads per output element (the ReLU / ELU mask of a data gradient, x of a folded-BatchNorm gradient) is requested
  // BEFORE its MFMAs and lands under them; requested from the epilogue it is a full memory round trip per tile on layers with 2-4 chunks per tile
  const int px_ = x0 + l31;
  const bool want_m = mask_mode != MASK_NONE && mask_mode != MASK_BIAS_TAB && mask_mode != MASK_RELU_BITS;
  constexpr bool MPF = MODE != 1 && RW * NB <= 4;          // (64 registers at most; the 16-row tiles keep the epilogue loads)
  float4 mpre[MPF ? NB : 1][MPF ? RW : 1][4];
  if (MPF && mask_mode == MASK_RELU_BITS) {
    // one bit per element: the 32 B of this lane's (8 pixels x 32 channels) cell -- the same address for the 16 lanes of a cell
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int blk = g * NB + nb;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int py = y0 + wave * RW + r;
        const bool ok = blk * 32 < M && py < H && x0 + (l31 & ~7) < W;
        const unsigned long long* p = reinterpret_cast<const unsigned long long*>(mask) + ((((long long)n * H + py) * (W >> 3) + (x0 >> 3) + (l31 >> 3)) * (M >> 5) + blk) * 4;
        mpre[nb][r][0] = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
        mpre[nb][r][1] = ok ? *reinterpret_cast<const float4*>(p + 2) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  } else if (MPF && want_m) {
    // (requested the way the output rows are stored -- eight consecutive lanes per 128-B line, lane -> (pixel j * 8 + lane / 8, channel quad lane % 8) --
    //  and turned into the accumulator layout through the epilogue's LDS staging row)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int mb0 = (g * NB + nb) * 32;
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int py = y0 + wave * RW + r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pxj = x0 + j * 8 + (lane >> 3);
          // (mask_climit: output-channel blocks from there on do not read `mask` -- a folded-BatchNorm gradient whose K1 x term is added by the consumer)
          const bool ok = mb0 + (lane & 7) * 4 < M && mb0 < mask_climit && py < H && pxj < W;
          const long long o = (((long long)n * H + py) * W + pxj) * ldy + mb0 + (lane & 7) * 4;
          mpre[nb][r][j] = ok ? *reinterpret_cast<const float4*>(mask + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
  // HEAD: lane (l31, hi) does the sigmoid / loss arithmetic of pixel (row hi of its wave's two, column l31); its label travels under the last MFMAs
  const int hpy = y0 + wave * RW + hi;
  const bool hvalid = HEAD && hpy < H && px_ < W;
  float hlabel = 0.f;
  if (HEAD && hd.t && hvalid) hlabel = hd.t[((long long)n * H + hpy) * W + px_];
  __builtin_amdgcn_s_setprio(2);
  mfma_chunk();
  __builtin_amdgcn_s_setprio(0);
  H2_STAMP(7);
  const float unscale = pow2f(max(-e_run, -126)) * w_unscale;            // 2^-(e_x + e_w)
  // Output rows leave through LDS: a lane holds 64 B of ONE pixel, so its four 16-B stores land 128+ B apart from its neighbours' and a store instruction
  // touches 32 cache lines with 32 B each.  Transposed through a 4.5-KB per-wave staging row (the patch planes are dead), eight consecutive lanes write one
  // full 128-B line: 1/4 of the write requests (measured with the placement faked: -5...-20 % per launch).
  __syncthreads();                                         // (every wave is past its last fragment read: the planes may be overwritten)
  H2_STAMP(11);
  char* const s_out = smem + wave * (32 * OUT_PS);
  float* const s_stat = reinterpret_cast<float*>(smem + 4 * 32 * OUT_PS);          // [wave][nb][sum | sum of squares][32 channels]

  if constexpr (HEAD) {
    // LDS (planes and slab are dead): [4 waves][32 pixels][128 B] staging rows | s_rec [4 waves][2 rows][32 pixels][8 floats] | s_hsum [4 waves][104 floats]
    float* const s_rec = reinterpret_cast<float*>(smem + 4 * 32 * OUT_PS) + wave * (2 * 32 * 8);
    float* const s_hsum = reinterpret_cast<float*>(smem + 4 * 32 * OUT_PS + 4 * 2 * 32 * 8 * 4);
    float bv[16], wh[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(s_bias + hi * 16 + q * 4), w4 = *reinterpret_cast<const float4*>(s_bias + 32 + hi * 16 + q * 4);
      bv[q * 4] = b4.x; bv[q * 4 + 1] = b4.y; bv[q * 4 + 2] = b4.z; bv[q * 4 + 3] = b4.w;
      wh[q * 4] = w4.x; wh[q * 4 + 1] = w4.y; wh[q * 4 + 2] = w4.z; wh[q * 4 + 3] = w4.w;
    }
    float dpart[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {                         // y = relu(conv + bias) in place of the accumulators; this lane's 16 channels of the head's dot product
      float d = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float v = fmaxf(fmaf(acc[r][0][i], unscale, bv[i]), 0.f); acc[r][0][i] = v; d = fmaf(v, wh[i], d); }
      dpart[r] = d;
    }
    // the partner lane (other channel half, same pixel column) holds the other 16 channels: each lane hands over the half-sum of the row it does NOT finish
    const float z = (hi ? dpart[1] : dpart[0]) + __shfl_xor(hi ? dpart[0] : dpart[1], 32) + s_bias[64];
    // sigmoid and the Keras loss terms of this pixel (T1:784-799; BCE in logit form on the clipped probability): inside the clip range the
    // logit of the clipped p IS z; outside it is the logit of the fp32 clip bound as an fp32 evaluation has it (head_fwd_kernel, TF): 1 - 1e-7 is
    // 1 - 2^-23 in fp32 -> logit 15.942385 (log term 2^-23), 1e-7 -> logit -16.118095 (log term 1.0000001e-7)
    const float ez = expf(-fabsf(z)), rz = 1.0f / (1.0f + ez);
    const float pr = z >= 0.f ? rz : ez * rz;
    const float lo = 1e-7f, hi_ = 1.0f - 1e-7f;
    const bool inr = pr >= lo && pr <= hi_;
    const float pc = fminf(fmaxf(pr, lo), hi_);
    const float zc = inr ? z : (z > 0.f ? 15.942385f : -16.118095f);
    const float tl = hlabel;
    const float bce = fmaxf(zc, 0.f) - zc * tl + (inr ? log1pf(ez) : (z > 0.f ? 1.1920929e-7f : 1.0000001e-7f));
    const float qq = pr * (1.0f - pr);
    if (hvalid) hd.p[((long long)n * H + hpy) * W + px_] = pr;
    {
      const bool on = hvalid && hd.t != nullptr;            // (inference: probabilities only)
      float* rec = s_rec + (hi * 32 + l31) * 8;             // {bce, t p, t, p | a, t q, q, 0}: a = dBCE/dz inside the clip range, q = p (1 - p)
      *reinterpret_cast<float4*>(rec) = on ? make_float4(bce, tl * pr, tl, pr) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(rec + 4) = on ? make_float4(inr ? pc - tl : 0.f, tl * qq, qq, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f), g2 = g1, g3 = g1; float gs = 0.f;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int py = y0 + wave * RW + r;
      if (py >= H) continue;                               // (wave-uniform)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(s_out + out_cell(l31, hi * 4 + q)) = make_float4(acc[r][0][q * 4], acc[r][0][q * 4 + 1], acc[r][0][q * 4 + 2], acc[r][0][q * 4 + 3]);
      float4 t4s[4];                                       // (the four reads together, as in the general epilogue below)
#pragma unroll
      for (int j = 0; j < 4; ++j) t4s[j] = *reinterpret_cast<const float4*>(s_out + out_cell(j * 8 + (lane >> 3), lane & 7));
      asm volatile("" :: "v"(t4s[0].x), "v"(t4s[1].x), "v"(t4s[2].x), "v"(t4s[3].x));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pj = j * 8 + (lane >> 3), cj = lane & 7, pxj = x0 + pj;
        const float4 t4 = t4s[j];
        const long long oj = (((long long)n * H + py) * W + pxj) * ldy;
        if (signs) {                                       // (wave-uniform) sign bits of the stored values: the mask of the head's own backward
          // (W % 8 == 0: an 8-pixel cell is inside the image or not at all; the four words of a cell go to lanes 0..3 of a register pair by v_writelane -- a select chain
          //  per lane compiled to nested exec-mask branches, ~35 instructions per cell)
          const unsigned long long bb[4] = {__builtin_amdgcn_ballot_w64(t4.x > 0.f), __builtin_amdgcn_ballot_w64(t4.y > 0.f), __builtin_amdgcn_ballot_w64(t4.z > 0.f), __builtin_amdgcn_ballot_w64(t4.w > 0.f)};
          int slo = 0, shi = 0;
          unet_writelane(slo, (unsigned)bb[0], 0); unet_writelane(shi, (unsigned)(bb[0] >> 32), 0); unet_writelane(slo, (unsigned)bb[1], 1); unet_writelane(shi, (unsigned)(bb[1] >> 32), 1);
          unet_writelane(slo, (unsigned)bb[2], 2); unet_writelane(shi, (unsigned)(bb[2] >> 32), 2); unet_writelane(slo, (unsigned)bb[3], 3); unet_writelane(shi, (unsigned)(bb[3] >> 32), 3);
          if (lane < 4 && x0 + j * 8 < W)
            signs[(((long long)n * H + py) * (W >> 3) + (x0 >> 3) + j) * 4 + lane] = ((unsigned long long)(unsigned)shi << 32) | (unsigned)slo;
        }
        if (pxj < W) {
          if (y) *reinterpret_cast<float4*>(y + oj + cj * 4) = t4;          // (y null: nothing downstream reads the tensor -- p, the sums and the sign bits are all the backward takes)
          if (hd.t) {
            const float* rec = s_rec + (r * 32 + pj) * 8;
            const float4 c4 = *reinterpret_cast<const float4*>(rec + 4);
            g1.x = fmaf(c4.x, t4.x, g1.x); g1.y = fmaf(c4.x, t4.y, g1.y); g1.z = fmaf(c4.x, t4.z, g1.z); g1.w = fmaf(c4.x, t4.w, g1.w);
            g2.x = fmaf(c4.y, t4.x, g2.x); g2.y = fmaf(c4.y, t4.y, g2.y); g2.z = fmaf(c4.y, t4.z, g2.z); g2.w = fmaf(c4.y, t4.w, g2.w);
            g3.x = fmaf(c4.z, t4.x, g3.x); g3.y = fmaf(c4.z, t4.y, g3.y); g3.z = fmaf(c4.z, t4.z, g3.z); g3.w = fmaf(c4.z, t4.w, g3.w);
            gs += rec[cj];                                 // scalar sum number cj (0..6) of this pixel; rec[7] = 0
          }
        }
      }
    }
    if (hd.t) {                                            // (uniform) lanes L, L + 8, ... hold the same channel quad / scalar: fold, lanes 0-7 post the wave's sums
      float sv[13] = {g1.x, g1.y, g1.z, g1.w, g2.x, g2.y, g2.z, g2.w, g3.x, g3.y, g3.z, g3.w, gs};
#pragma unroll
      for (int i = 0; i < 13; ++i) { sv[i] += __shfl_xor(sv[i], 8); sv[i] += __shfl_xor(sv[i], 16); sv[i] += __shfl_xor(sv[i], 32); }
      if (lane < 8) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int i = 0; i < 4; ++i) s_hsum[wave * 104 + k * 32 + lane * 4 + i] = sv[k * 4 + i];
        s_hsum[wave * 104 + 96 + lane] = sv[12];
      }
      __syncthreads();
      if (tid < 103) {                                     // [0, 96): the three per-channel sums; 96..99: bce, t p, t, p; 100..102: sum a, sum t q, sum q
        const float t = (s_hsum[tid] + s_hsum[104 + tid]) + (s_hsum[208 + tid] + s_hsum[312 + tid]);
        double* const row = hd.slots + (size_t)(blockIdx.x % UNET_BN_SLOTS) * UNET_BN_SLOT_DOUBLES;
        if (xs) xsum_add(row, tid, t); else atomicAdd(row + tid, (double)t);          // (xs: deterministic mode -- exact window sums, common.h)
      }
    }
    H2_STAMP(8);
    return;
  }
  // ---- epilogue: lane (l31, hi) holds, for pixel column l31 of each of its RW rows, channels mb + 0..15
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int mb0 = (g * NB + nb) * 32, mb = mb0 + hi * 16;
    if (mb0 >= M) continue;                                // (wave-uniform) zero-padded block of a tile that overhangs M; M % 32 == 16: the hi half of the last block is padding
    int ab = 0, oc = mb;
    if (MODE == 1) { const int ct = M >> 2; ab = mb / ct; oc = mb - ab * ct; }
    float bv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 b4 = *reinterpret_cast<const float4*>(s_bias + nb * 32 + hi * 16 + q * 4);
      bv[q * 4] = b4.x; bv[q * 4 + 1] = b4.y; bv[q * 4 + 2] = b4.z; bv[q * 4 + 3] = b4.w;
    }
    float4 st1 = make_float4(0.f, 0.f, 0.f, 0.f), st2 = st1;           // BatchNorm statistics of what this lane stores (channel quad lane & 7)
#pragma unroll
    for (int r = 0; r < RW; ++r) {
      const int py = y0 + wave * RW + r;
      if (py >= H) continue;                               // (wave-uniform)
      const bool live = px_ < W && mb < M;                 // lanes past the image edge / in the padded half block take part in the LDS transpose only
      long long o;
      if (MODE == 1) o = (((long long)n * 2 * H + 2 * py + (ab >> 1)) * (2 * W) + 2 * px_ + (ab & 1)) * ldy + oc;
      else o = (((long long)n * H + py) * W + px_) * ldy + mb;
      float v[16], mv[16], a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = acc[r][nb][i] * unscale;
      if (mask_mode == MASK_RELU_BITS) {
        float4 w01, w23;
        if (MPF) { w01 = mpre[MPF ? nb : 0][MPF ? r : 0][0]; w23 = mpre[MPF ? nb : 0][MPF ? r : 0][1]; }
        else {
          const bool ok = py < H && x0 + (l31 & ~7) < W;
          const unsigned long long* p = reinterpret_cast<const unsigned long long*>(mask) + ((((long long)n * H + py) * (W >> 3) + (x0 >> 3) + (l31 >> 3)) * (M >> 5) + (g * NB + nb)) * 4;
          w01 = ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
          w23 = ok ? *reinterpret_cast<const float4*>(p + 2) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // word k = (lo, hi) dwords; this lane's bits: (pixel % 8) * 8 + hi * 4 + q
        const bool up = (l31 & 4) != 0;
        const unsigned d[4] = {__float_as_uint(up ? w01.y : w01.x), __float_as_uint(up ? w01.w : w01.z), __float_as_uint(up ? w23.y : w23.x), __float_as_uint(up ? w23.w : w23.z)};
        const int sh = (l31 & 3) * 8 + hi * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int k = 0; k < 4; ++k) mv[q * 4 + k] = ((d[k] >> (sh + q)) & 1u) ? 1.0f : 0.0f;
      } else if (want_m && !POOLS) {
        if (MPF) {
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(s_out + out_cell(j * 8 + (lane >> 3), lane & 7)) = mpre[MPF ? nb : 0][MPF ? r : 0][j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float4 m4;
          if (MPF) m4 = *reinterpret_cast<const float4*>(s_out + out_cell(l31, hi * 4 + q));
          else m4 = (live && mb0 < mask_climit) ? *reinterpret_cast<const float4*>(mask + o + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
          mv[q * 4] = m4.x; mv[q * 4 + 1] = m4.y; mv[q * 4 + 2] = m4.z; mv[q * 4 + 3] = m4.w;
        }
      }
      if (mask_mode >= MASK_BN_BWD && mask_mode <= MASK_BN_BWD_RELU) {
        // data gradient of a conv whose input BatchNorm is folded: dx = K0 dz + K1 x + K2, x read where a ReLU layer reads its mask
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 k1 = *reinterpret_cast<const float4*>(s_bias + (NB + nb) * 32 + hi * 16 + q * 4), k2 = *reinterpret_cast<const float4*>(s_bias + (2 * NB + nb) * 32 + hi * 16 + q * 4);
          v[q * 4] = fmaf(bv[q * 4], a[q * 4], fmaf(k1.x, mv[q * 4], k2.x));
          v[q * 4 + 1] = fmaf(bv[q * 4 + 1], a[q * 4 + 1], fmaf(k1.y, mv[q * 4 + 1], k2.y));
          v[q * 4 + 2] = fmaf(bv[q * 4 + 2], a[q * 4 + 2], fmaf(k1.z, mv[q * 4 + 2], k2.z));
          v[q * 4 + 3] = fmaf(bv[q * 4 + 3], a[q * 4 + 3], fmaf(k1.w, mv[q * 4 + 3], k2.w));
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
          v[q * 4] = a[q * 4] + b4.x; v[q * 4 + 1] = a[q * 4 + 1] + b4.y; v[q * 4 + 2] = a[q * 4 + 2] + b4.z; v[q * 4 + 3] = a[q * 4 + 3] + b4.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = a[i] + bv[i];
      }
      if (!GEN) {
        if (act == ACT_RELU) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (mask_mode == MASK_RELU || mask_mode == MASK_RELU_BITS) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = mv[i] > 0.f ? v[i] : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = apply_act(v[i], act);
        if (mask_mode >= MASK_BN_BWD && mask_mode <= MASK_BN_BWD_RELU) {                    // (v already holds K0 dz + K1 x + K2) then the ELU (+ dropout) derivative of x's producer
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
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(s_out + out_cell(l31, hi * 4 + q)) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
      // (LDS executes a wave's instructions in order: the reads below see the writes above, the next row's writes come after these reads)
      // The four line-layout reads are issued TOGETHER, in front of the (branchy) store code: read - wait - store per line meant four dependent LDS round trips per
      // (row, block) behind the other workgroup's fragment reads -- ~650 cycles per store instruction, 23-44 % of a tile's lifetime at the 256 / 512 pixel levels
      // (profiles/r05_h2_timeline.txt)
      // (not in the ELU / dropout instances: U-Net++'s kernels sit at their register limit -- twelve more live registers there cost 72 spilled scalars and 6x the launch time)
      float4 t4s[4];
      if (!GEN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t4s[j] = *reinterpret_cast<const float4*>(s_out + out_cell(j * 8 + (lane >> 3), lane & 7));
        asm volatile("" :: "v"(t4s[0].x), "v"(t4s[1].x), "v"(t4s[2].x), "v"(t4s[3].x));          // (all four have landed here: nothing sinks a read back to its use)
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pj = j * 8 + (lane >> 3), cj = lane & 7, pxj = x0 + pj;
        const float4 t4 = GEN ? *reinterpret_cast<const float4*>(s_out + out_cell(pj, cj)) : t4s[j];
        long long oj;
        if (MODE == 1) oj = (((long long)n * 2 * H + 2 * py + (ab >> 1)) * (2 * W) + 2 * pxj + (ab & 1)) * ldy + (oc - hi * 16);
        else oj = (((long long)n * H + py) * W + pxj) * ldy + (mb - hi * 16);
        if (MODE == 0 && !GEN && signs) {                  // (wave-uniform; never with ELU / dropout) sign bits of the stored values: four ballots per 8 pixels x 32 channels
          // (signs are written only for M % 32 == 0 and W % 8 == 0: every lane of a cell inside the image is valid; the four words -> lanes 0..3 by v_writelane)
          const unsigned long long bb[4] = {__builtin_amdgcn_ballot_w64(t4.x > 0.f), __builtin_amdgcn_ballot_w64(t4.y > 0.f), __builtin_amdgcn_ballot_w64(t4.z > 0.f), __builtin_amdgcn_ballot_w64(t4.w > 0.f)};
          int slo = 0, shi = 0;
          unet_writelane(slo, (unsigned)bb[0], 0); unet_writelane(shi, (unsigned)(bb[0] >> 32), 0); unet_writelane(slo, (unsigned)bb[1], 1); unet_writelane(shi, (unsigned)(bb[1] >> 32), 1);
          unet_writelane(slo, (unsigned)bb[2], 2); unet_writelane(shi, (unsigned)(bb[2] >> 32), 2); unet_writelane(slo, (unsigned)bb[3], 3); unet_writelane(shi, (unsigned)(bb[3] >> 32), 3);
          if (lane < 4 && x0 + j * 8 < W)
            signs[((((long long)n * H + py) * (W >> 3) + (x0 >> 3) + j) * (M >> 5) + (g * NB + nb)) * 4 + lane] = ((unsigned long long)(unsigned)shi << 32) | (unsigned)slo;
        }
        if (pxj < W && mb0 + cj * 4 < M) {
          // The output leaves with the streaming (nontemporal) hint: at the 256 x 256 / 512 x 512 levels it is 0.25-1 GB, far beyond what the 256-MB memory-side cache can
          // keep for the next launch, and written normally it pushes out what that launch is about to read.  Same-box A/B: 15.145 -> 14.986, 15.380 -> 15.260, 15.252 ->
          // 15.163 ms on three boxes (the readers of the 512 x 512 tensors 9-20 % faster; batch-1 inference unchanged).  Only the 8-row instances streaming: half the gain;
          // decided per launch by output size behind a branch: +0.2 ms (profiles/r05_ab_streaming_stores.txt)
          {
            float* const q = y + oj + cj * 4;
            __builtin_nontemporal_store(t4.x, q); __builtin_nontemporal_store(t4.y, q + 1); __builtin_nontemporal_store(t4.z, q + 2); __builtin_nontemporal_store(t4.w, q + 3);
          }
          if (POOLS) {
            // the stored gradient g and the pooled activation p of the same elements (line layout both): st1 += g ks, st2 += g ks (p (1 - rate) - beta) / gamma
            const float4 pv = MPF ? mpre[MPF ? nb : 0][MPF ? r : 0][j] : *reinterpret_cast<const float4*>(mask + oj + cj * 4);
            const float4 ig = *reinterpret_cast<const float4*>(s_bias + (NB + nb) * 32 + cj * 4), be = *reinterpret_cast<const float4*>(s_bias + (2 * NB + nb) * 32 + cj * 4);
            const float unkeep = 1.0f - hd.aux, inv = 1.0f / unkeep;
            const bool drop = hd.aux > 0.0f;
            const float kx = drop && __float_as_uint(pv.x) == 0x80000000u ? 0.f : inv, ky = drop && __float_as_uint(pv.y) == 0x80000000u ? 0.f : inv;
            const float kz = drop && __float_as_uint(pv.z) == 0x80000000u ? 0.f : inv, kw = drop && __float_as_uint(pv.w) == 0x80000000u ? 0.f : inv;
            const float gx = t4.x * kx, gy = t4.y * ky, gz = t4.z * kz, gw = t4.w * kw;
            st1.x += gx; st1.y += gy; st1.z += gz; st1.w += gw;
            st2.x = fmaf(gx, (pv.x * unkeep - be.x) * ig.x, st2.x); st2.y = fmaf(gy, (pv.y * unkeep - be.y) * ig.y, st2.y);
            st2.z = fmaf(gz, (pv.z * unkeep - be.z) * ig.z, st2.z); st2.w = fmaf(gw, (pv.w * unkeep - be.w) * ig.w, st2.w);
          } else
          if (MODE != 2 && stats) {
            st1.x += t4.x; st1.y += t4.y; st1.z += t4.z; st1.w += t4.w;
            st2.x = fmaf(t4.x, t4.x, st2.x); st2.y = fmaf(t4.y, t4.y, st2.y); st2.z = fmaf(t4.z, t4.z, st2.z); st2.w = fmaf(t4.w, t4.w, st2.w);
          }
        }
      }
    }
    if (MODE != 2 && stats) {                              // (wave-uniform) lanes L, L + 8, ... hold the same channel quad: fold them, lanes 0-7 post the wave's sums
      float sv[8] = {st1.x, st1.y, st1.z, st1.w, st2.x, st2.y, st2.z, st2.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sv[i] += __shfl_xor(sv[i], 8); sv[i] += __shfl_xor(sv[i], 16); sv[i] += __shfl_xor(sv[i], 32);
      }
      if (lane < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { s_stat[((wave * NB + nb) * 2 + 0) * 32 + lane * 4 + i] = sv[i]; s_stat[((wave * NB + nb) * 2 + 1) * 32 + lane * 4 + i] = sv[4 + i]; }
      }
    }
  }
  H2_STAMP(12);
  if (MODE != 2 && stats) {
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
        double* const row = stats + (size_t)(blockIdx.x % UNET_BN_SLOTS) * UNET_BN_SLOT_DOUBLES;
        if (xs) xsum_add(row, (kind ? stats_c : 0) + ch, t); else atomicAdd(row + (kind ? stats_c : 0) + ch, (double)t);
      }
    }
  }
  H2_STAMP(8);
#if H2_EXPERIMENT == 5
  if (tid == 0 && blockIdx.x < 16384) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); h2_trace[blockIdx.x * 16 + 9] = hw;
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); h2_trace[blockIdx.x * 16 + 10] = xcc; }
#endif
}

// splits > 1 (MODE 0, EPI 0, no statistics / sign bits / mask / dropout): `splits` slices of the contraction into the slabs y + i * n * h * wd * ldy (SPLITK above)
template <int MODE, int NB, int RW, int WPS, int EPI = 0>
int32_t launch_h2(unet_ctx* ctx, const float* x, int ldx, const unet_bf16* wimg, const float* bias, const float* mask, int mask_mode, float* y, int ldy, int n, int h, int wd,
                  int K, int M, int act, float rate, unsigned long long seed, hipStream_t s, int mask_climit = 1 << 30, h2_head_args hd = h2_head_args(), int img_nb = 0,
                  int splits = 1) {
  constexpr int T = MODE == 0 ? 9 : 1, KS = MODE == 0 ? 1 : 2;
  if (img_nb == 0) img_nb = NB;                              // (the image's blocks per group: NB unless a one-block launch reads a two-block image, conv_h2_kernel)
  if (img_nb != NB && !(MODE != 2 && NB == 1 && (img_nb == 2 || img_nb == 4) && (M % (32 * img_nb)) == 0 && EPI == 0)) UNET_FAIL(ctx, UNET_E_ARG, "conv h2: image of %d blocks per group on a %d-block launch", img_nb, NB);
  constexpr int TH = 4 * RW;
  constexpr int NPIX = MODE == 0 ? (TH + 2) * 34 : TH * 32;
  constexpr size_t smem = (size_t)2 * KS * 2 * (NPIX * 16 + (MODE == 0 ? 0 : 64)) + (NB != 1 || MODE != 0 ? (size_t)((KS * T * NB * 2 * 2 * 32 + 255) / 256) * 256 * 16 : (size_t)KS * T * NB * 2 * 2 * 32 * 16) + 16 + 3 * NB * 32 * 4;
  if (!mask) mask_mode = MASK_NONE;
  const int tiles_x = (wd + 31) / 32, tiles_y = (h + TH - 1) / TH, groups = (M + 32 * NB - 1) / (32 * NB);
  const long long total = (long long)tiles_x * tiles_y * n * groups;
  if (total >= (1LL << 28)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv h2: too many tiles");
  const unsigned grid = (unsigned)(8 * ((total + 7) / 8));
  const bool gen = MODE == 0 && (act == ACT_ELU || rate > 0.0f || mask_mode == MASK_ELU || mask_mode == MASK_ELU_DROP || mask_mode == MASK_BN_BWD_ELU || mask_mode == MASK_BN_BWD_ELU_DROP);
  // an armed statistics request (common.h): forward launches only (the sums are those of the values STORED, after activation and dropout), channel counts the slot copies can hold
  double* stats = nullptr; int stats_c = 0;
  if (MODE != 2 && ctx->stats_req_c > 0) {
    const int c = ctx->stats_req_c; ctx->stats_req_c = 0;
    // (deterministic mode: the sums leave as exact window sums, four words per value -- xsum_add, common.h)
    const int xw = ctx->opt_deterministic ? UNET_XW : 1;
    if ((mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) && 2 * c * xw <= UNET_BN_SLOT_DOUBLES && c == (MODE == 1 ? M / 4 : M) && ctx->bn_slots) {
      stats = ctx->bn_slots; stats_c = c; ctx->stats_in_slots = y; ctx->stats_in_slots_c = c; ctx->stats_in_slots_xs = ctx->opt_deterministic != 0;
    }
  }
  if (mask_mode == MASK_POOL_SUMS) {
    if (MODE != 0 || !ctx->bn_slots || 2 * M * (ctx->opt_deterministic ? UNET_XW : 1) > UNET_BN_SLOT_DOUBLES) UNET_FAIL(ctx, UNET_E_ARG, "conv h2: the pooled-sums epilogue is a conv3x3 data-gradient launch (M = %d)", M);
    stats = ctx->bn_slots; stats_c = M;
  }
  if (mask_mode == MASK_RELU_BITS && ((M & 31) || (wd & 7))) UNET_FAIL(ctx, UNET_E_SHAPE, "conv h2: the bit mask needs M %% 32 == 0 and W %% 8 == 0 (M=%d W=%d)", M, wd);
  unsigned long long* signs = nullptr;
  if (MODE == 0 && ctx->signs_req) {
    unsigned long long* q = ctx->signs_req; ctx->signs_req = nullptr;
    if (act == ACT_RELU && rate == 0.0f && (mask_mode == MASK_NONE || mask_mode == MASK_BIAS_TAB) && !(M & 31) && !(wd & 7)) { signs = q; ctx->signs_done = q; }
  }
  const int nchunks_ = K / (16 * KS), kcps = (nchunks_ + splits - 1) / splits;
  auto go = [&](auto kern) -> int32_t {
    if (smem > 65536) UNET_BIG_LDS(ctx, kern, smem, "conv_h2");
    unet_note_kernel(ctx, reinterpret_cast<const void*>(kern));
    hipLaunchKernelGGL(kern, dim3(grid, (unsigned)((nchunks_ + kcps - 1) / kcps)), dim3(256), smem, s, x, ldx, wimg, bias, mask, y, ldy, n, h, wd, K, M, act, mask_mode, rate, seed, tiles_x,
                       tiles_y, groups, (int)total, stats, stats_c, signs, mask_climit, hd, img_nb, ctx->opt_deterministic ? 1 : 0, kcps, (long long)n * h * wd * ldy);
    return UNET_OK;
  };
  int32_t r;
  if (splits > 1) {
    if constexpr (MODE == 0 && EPI == 0 && NB == 1 && WPS == 4) {
      if (gen || stats || signs || mask_mode != MASK_NONE || bias || act != ACT_NONE || rate != 0.0f) UNET_FAIL(ctx, UNET_E_ARG, "conv h2: K slices go with a plain partial-sum launch");
      r = go(conv_h2_kernel<MODE, NB, RW, false, WPS, EPI, true>);
      if (r) return r;
      UNET_CHECK_LAUNCH(ctx, "conv_h2 (K slices)");
      return UNET_OK;
    } else UNET_FAIL(ctx, UNET_E_ARG, "conv h2: K slices exist for the one-block conv3x3 launches");
  }
  if (EPI == 1) {
    if (gen || mask_mode != MASK_NONE || act != ACT_RELU || M != 32 || stats) UNET_FAIL(ctx, UNET_E_ARG, "conv h2 + head: a plain 32-channel ReLU forward launch only");
    r = go(conv_h2_kernel<MODE, NB, RW, false, WPS, EPI>);
  } else if (EPI == 2) {
    if (gen || mask_mode != MASK_POOL_SUMS || act != ACT_NONE) UNET_FAIL(ctx, UNET_E_ARG, "conv h2 + pooled sums: a plain data-gradient launch only");
    r = go(conv_h2_kernel<MODE, NB, RW, false, WPS, EPI>);
  } else if (EPI == 3) {
    if (gen || K != 32 || ldx != 2 || act != ACT_NONE || mask_mode == MASK_POOL_SUMS) UNET_FAIL(ctx, UNET_E_ARG, "conv h2 behind the head's {dz, mask} stream: a plain 32-channel data-gradient launch only");
    r = go(conv_h2_kernel<MODE, NB, RW, false, WPS, EPI>);
  } else if (mask_mode == MASK_POOL_SUMS) { UNET_FAIL(ctx, UNET_E_ARG, "conv h2: MASK_POOL_SUMS goes with its own kernel instance");
  } else if (gen) r = go(conv_h2_kernel<MODE, NB, RW, (MODE == 0), WPS>); else r = go(conv_h2_kernel<MODE, NB, RW, false, WPS>);
  if (r) return r;
  UNET_CHECK_LAUNCH(ctx, "conv_h2");
  return UNET_OK;
}

// the slabs of a K-sliced launch (SPLITK) -> y = act(sum of the slabs + bias): bias[c], or for a folded-BatchNorm forward (MASK_BIAS_TAB) row `border class of the pixel` of the
// table [16][M]; one float4 per thread, slabs dense [n,h,w,M], y with pixel stride ldy
__global__ __launch_bounds__(256) void h2_splitk_finish_kernel(const float* __restrict__ part, int splits, long long stride, const float* __restrict__ bias, int tab, int act,
                                                               float* __restrict__ y, int ldy, int H, int W, int M, long long total4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int m4 = M >> 2, c4 = (int)(i % m4);
  const long long pix = i / m4;
  float4 a = *reinterpret_cast<const float4*>(part + i * 4);
  for (int k = 1; k < splits; ++k) {
    const float4 b = *reinterpret_cast<const float4*>(part + (long long)k * stride + i * 4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (bias) {
    int cls = 0;
    if (tab) { const int px = (int)(pix % W), py = (int)((pix / W) % H); cls = (((py == 0) | ((py == H - 1) << 1)) << 2) | ((px == 0) | ((px == W - 1) << 1)); }
    const float4 b = *reinterpret_cast<const float4*>(bias + (long long)cls * M + c4 * 4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  if (act == ACT_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
  *reinterpret_cast<float4*>(y + pix * ldy + c4 * 4) = a;
}

}  // namespace

static int h2_nb(int M) { return (M % 64) == 0 ? 2 : 1; }      // n-blocks of 32 output channels per workgroup

// a launch with K contraction channels and M output channels runs on the h2 kernels: the family of UNET_ALGO_AUTO wherever the channel counts allow
// (UNET_ALGO_MFMA = the strict fp32 family, kernels_conv_mfma.hip); both the weight preparation and the launch ask this
bool h2_conv3x3_selected(int algo, int K, int M) { return algo == UNET_ALGO_AUTO && K >= 16 && (K % 16) == 0 && M >= 16 && (M % 16) == 0; }

// bytes of the split weight image: 256-B header + 36 * K * M' (M' = M rounded up to whole 32-channel blocks; fits unet_conv3x3_w_ws_floats)
size_t h2_wimg_bytes(int K, int M) { return (size_t)H2_HEADER + (size_t)36 * K * ((M + 31) / 32 * 32); }

// ConvT: one tap, so a staged pixel patch feeds only NB x 32 output columns per k-step (a conv3x3 patch feeds 9 x that): four blocks per workgroup over 8-row
// tiles instead of two over 16 rows -- the same accumulators and MFMAs per chunk, half the pixels scaled, split and stored per MFMA
// (eight blocks over 4-row tiles measured slower again: u6 / u7 forward 0.072 -> 0.079, 0.085 -> 0.093 ms)
static int h2_nb_convT_fwd() { return 4; }
static int h2_nb_convT_dgrad(int cin) { return (cin % 128) == 0 ? 4 : (cin % 64) == 0 ? 2 : 1; }

// One item of a preparation batch.  kind 0: conv3x3 forward image (K = cin, M = cout), 1: conv3x3 data-gradient image (flipped taps, K = cout, M = cin),
// 2: ConvT forward (K = cin, M = 4 cout; Keras kernel [2][2][cout][cin]), 3: ConvT data gradient (K = 4 cout, M = cin).  cs: per-INPUT-channel factor of a
// kind-0 image (folded BatchNorm scale) or null
static void h2_fill_prep(h2_prep* p, const float* w, const float* cs, void* img, int cin, int cout, int kind) {
  p->w = w; p->cs = cs; p->img = static_cast<unet_bf16*>(img); p->flip = 0; p->cs_div = 4; p->cs_mod = 1; p->max_only = kind == 4; p->cs_bound = 0;
  if (kind == 4) kind = 0;
  int K, M;
  if (kind <= 1) {
    const int flip = kind;
    K = flip ? cout : cin; M = flip ? cin : cout;
    p->T = 9; p->KS = 1; p->nb = h2_nb(M); p->nchunks = K / 16; p->flip = flip;
    p->tap_stride = (long long)cin * cout; p->sk = flip ? 1 : cout; p->sm = flip ? cout : 1;
    p->nw4 = 9LL * cin * cout / 4;
    p->cs_div = cout; p->cs_mod = cin;                       // w[tap][c][o]: input channel of flat element i = (i / cout) % cin  (cout % 4 == 0)
  } else if (kind == 2) {
    K = cin; M = 4 * cout;
    p->T = 1; p->KS = 2; p->nb = h2_nb_convT_fwd(); p->nchunks = K / 32; p->tap_stride = 0; p->sk = 1; p->sm = cin;          // W(k = c, m = ab*cout + o) = K[m*cin + c]
    p->nw4 = (long long)cin * cout;
  } else {
    K = 4 * cout; M = cin;
    p->T = 1; p->KS = 2; p->nb = h2_nb_convT_dgrad(cin); p->nchunks = K / 32; p->tap_stride = 0; p->sk = cin; p->sm = 1;      // W(k = ab*cout + o, m = c) = K[k*cin + c]
    p->nw4 = (long long)cin * cout;
  }
  p->m = M;
  const int groups = (M + 32 * p->nb - 1) / (32 * p->nb);
  p->total = (long long)groups * p->nchunks * p->KS * p->T * p->nb * 2 * 32;
}

// count items (w, cs, img, cin, cout, kind) -> their split images, in TWO launches
int32_t k_h2_prep_multi(unet_ctx* ctx, const float* const* w, const float* const* cs, void* const* img, const int* cin, const int* cout, const int* kind, int count, hipStream_t s) {
  if (count < 1) return UNET_OK;
  if (count > UNET_PREP_MAX) UNET_FAIL(ctx, UNET_E_ARG, "h2_prep_multi: too many layers");
  h2_prep_list L; L.n = count;
  long long most = 1;
  for (int k = 0; k < count; ++k) {
    if (cs && cs[k] && (kind[k] > 1 || (cout[k] & 3))) UNET_FAIL(ctx, UNET_E_ARG, "h2_prep_multi: a channel factor goes with a conv3x3 image");
    if (kind[k] < 0 || kind[k] > 4) UNET_FAIL(ctx, UNET_E_ARG, "h2_prep_multi: kind %d", kind[k]);
    h2_fill_prep(&L.item[k], w[k], cs ? cs[k] : nullptr, img[k], cin[k], cout[k], kind[k]);
    // a data-gradient image contracts over the conv's OUTPUT channels: its factor (the 1x1 head's weights behind the last conv3x3, EPI 3) varies inside a float4 of
    // the weights, so the max pass takes the raw maxima and the exponent comes from the bound max |w| max |cs|
    if (cs && cs[k] && kind[k] == 1) { L.item[k].cs_bound = 1; L.item[k].cs_mod = cout[k]; }
    most = std::max(most, L.item[k].total);
  }
  hipLaunchKernelGGL(h2_wmax_kernel, dim3(H2_MAXB, (unsigned)count), dim3(H2_MAXT), 0, s, L);
  hipLaunchKernelGGL(h2_wimg_kernel, dim3((unsigned)std::min<long long>((most + 255) / 256, 256), (unsigned)count), dim3(256), 0, s, L, (int)H2_MAXB);
  UNET_CHECK_LAUNCH(ctx, "h2_prep_multi");
  return UNET_OK;
}

// the image of a conv3x3 forward layer with a per-input-channel factor (a folded BatchNorm's scale) whose raw-weight maxima a kind-4 item of an earlier batch left in its header: ONE launch
int32_t k_h2_weights_bound(unet_ctx* ctx, const float* w, const float* cs, void* img, int cin, int cout, hipStream_t s) {
  if (!w || !cs || !img || (cout & 3)) UNET_FAIL(ctx, UNET_E_ARG, "h2_weights_bound: bad args");
  h2_prep_list L; L.n = 1;
  h2_fill_prep(&L.item[0], w, cs, img, cin, cout, 0);
  L.item[0].cs_bound = 1;
  hipLaunchKernelGGL(h2_wimg_kernel, dim3((unsigned)std::min<long long>((L.item[0].total + 255) / 256, 256), 1), dim3(256), 0, s, L, (int)H2_MAXB);
  UNET_CHECK_LAUNCH(ctx, "h2_weights_bound");
  return UNET_OK;
}

int32_t k_h2_weights_multi(unet_ctx* ctx, const float* const* w, void* const* img, const int* cin, const int* cout, const int* flip, int count, hipStream_t s) {
  return k_h2_prep_multi(ctx, w, nullptr, img, cin, cout, flip, count, s);          // (flip 0 / 1 = kind 0 / 1)
}

int32_t k_h2_weights(unet_ctx* ctx, const float* w, void* img, int cin, int cout, int flip, hipStream_t s, const float* cs) {
  const float* ws[1] = {w}; const float* css[1] = {cs}; void* is[1] = {img};
  return k_h2_prep_multi(ctx, ws, css, is, &cin, &cout, &flip, 1, s);
}

// The data gradient of the last conv3x3 (32 output channels, T1:911) from the head's rank-1 stream dzm[n,h,wd] = {dz, 32 mask bits} (k_head_dzm): dx[n,h,wd,M];
// wimg = the kind-1 image of that conv with cs = the head's 32 weights; mask / mask_mode as for any data gradient (the ReLU of the conv in front)
bool h2_head_bwd_selected(const unet_ctx* ctx, int algo, int wd, int cin) {
  return ctx && ctx->opt_head_bwd_fused && h2_conv3x3_selected(algo, 32, cin) && h2_nb(cin) == 1 && (wd & 7) == 0;
}
int32_t k_conv3x3_h2_dgrad_dzm(unet_ctx* ctx, const void* dzm, const void* wimg, const float* mask, int mask_mode, float* dx, int n, int h, int wd, int M, hipStream_t s) {
  if (!dzm || !wimg || !dx || h2_nb(M) != 1 || M < 16 || (M % 16)) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 dgrad behind the head stream: bad args (M = %d)", M);
  if ((long long)h * wd * std::max(32, M) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  if (pp_conv3x3_selected(ctx, 32, M, n, h, wd, mask, mask_mode, ACT_NONE, 0.0f, M)) return k_conv3x3_pp_fwd(ctx, static_cast<const float*>(dzm), wimg, nullptr, mask, mask_mode, dx, M, n, h, wd, 32, M, ACT_NONE, s, true);
  return launch_h2<0, 1, 2, 4, 3>(ctx, static_cast<const float*>(dzm), 2, static_cast<const unet_bf16*>(wimg), nullptr, mask, mask_mode, dx, M, n, h, wd, 32, M, ACT_NONE, 0.0f, 0, s);
}

// x [n,h,wd,K] dense NHWC fp32, wimg from k_h2_weights (K contraction channels, M output channels), y [n,h,wd,M] fp32
int32_t k_conv3x3_h2_fwd(unet_ctx* ctx, const float* x, const void* wimg, const float* bias, const float* mask, int mask_mode, float* y, int n, int h, int wd, int K,
                         int M, int act, float rate, uint64_t seed, hipStream_t s, int mask_climit, int ldy) {
  if (ldy == 0) ldy = M;                                   // (ldy > M: the output is a channel slice of a wider NHWC buffer -- an encoder conv writing into its concat's skip half)
  if (ldy < M || (ldy & 3) || (ldy != M && mask && mask_mode != MASK_BIAS_TAB)) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 h2: ldy %d (M = %d; a strided output goes with a plain forward launch)", ldy, M);
  if (K < 16 || (K % 16) || M < 16 || (M % 16)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 h2: K=%d M=%d (multiples of 16)", K, M);
  if ((long long)h * wd * std::max(K, M) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  const unet_bf16* img = static_cast<const unet_bf16*>(wimg);
  if (mask_climit >= M && pp_conv3x3_selected(ctx, K, M, n, h, wd, mask, mask_mode, act, rate, ldy)) return k_conv3x3_pp_fwd(ctx, x, wimg, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, s);
  if (mask_climit < M && ((mask_climit % 32) || mask_mode < MASK_BN_BWD)) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 h2: mask_climit %d must be a whole number of 32-channel blocks of a folded-BatchNorm gradient", mask_climit);
  // Grids that leave CUs idle (batch-1 inference below 256 x 256, the 32 x 32 level of small batches) trade tile size for workgroups, on the SAME weight image:
  //   fewer than two two-block 8-row tiles per CU -> ONE 32-channel block per workgroup (twice the workgroups, half the MFMAs per staged chunk; the patch is staged twice,
  //   which an idle CU does for free); still fewer than one workgroup per CU -> 4-row tiles (one row per wave) on top of that
  const int inb = h2_nb(M);
  const long long t8 = (long long)((wd + 31) / 32) * ((h + 7) / 8) * n;
  // K slices: fewer than 1.5 one-block workgroups per CU and a contraction of sixteen or more staged chunks (batch-1 inference from the 128 x 128 level down): four slices of
  // the K loop side by side, then one pass that adds the slabs, the bias and the ReLU.  Only where the caller allowed it for this launch (unet_allow_k_slices: the inference
  // programs do -- the slicing follows the grid, i.e. the batch, and a data-parallel rank's share of a batch must add up in the order of the whole batch), not for launches
  // that carry statistics / sign bits / a per-element mask / dropout, and not in deterministic mode
  {
    const bool small8 = t8 * ((M + 31) / 32) < ctx->num_cu;          // (the 4-row tiles below)
    const long long wgs = (small8 ? (long long)((wd + 31) / 32) * ((h + 3) / 4) * n : t8) * ((M + 31) / 32);
    const int nchunks = K / 16;
    const bool armed = ctx->k_slices_ok != 0; ctx->k_slices_ok = 0;          // (one-shot: unet_allow_k_slices)
    const bool plain = armed && (!mask || mask_mode == MASK_BIAS_TAB) && rate == 0.0f && (act == ACT_NONE || act == ACT_RELU) && !ctx->stats_req_c && !ctx->signs_req && !ctx->opt_deterministic;
    // (measured per launch at batch 1, 512 x 512: 16 chunks and more gain -- c5b 54 -> 27 us, c5a 29 -> 19, c6a 59 -> 43 with the finish pass --, 8 chunks do not)
    if (plain && ctx->splitk_ws && nchunks >= 16 && 2 * wgs < 3 * ctx->num_cu && (inb == 1 || small8 || t8 * ((M + 63) / 64) < 2 * ctx->num_cu) && (M % 32) == 0) {
      const int splits = std::min(4, nchunks / 4);
      const size_t slab = (size_t)n * h * wd * M;
      if (splits >= 2 && (size_t)splits * slab * sizeof(float) <= ctx->splitk_ws_bytes) {
        float* part = static_cast<float*>(ctx->splitk_ws);
        int32_t r;
        if (small8) r = launch_h2<0, 1, 1, 4>(ctx, x, K, img, nullptr, nullptr, MASK_NONE, part, M, n, h, wd, K, M, ACT_NONE, 0.0f, 0, s, 1 << 30, h2_head_args(), inb, splits);
        else r = launch_h2<0, 1, 2, 4>(ctx, x, K, img, nullptr, nullptr, MASK_NONE, part, M, n, h, wd, K, M, ACT_NONE, 0.0f, 0, s, 1 << 30, h2_head_args(), inb, splits);
        if (r) return r;
        const long long total4 = (long long)slab / 4;
        hipLaunchKernelGGL(h2_splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, part, splits, (long long)slab, bias, mask && mask_mode == MASK_BIAS_TAB ? 1 : 0, act,
                           y, ldy, h, wd, M, total4);
        UNET_CHECK_LAUNCH(ctx, "conv_h2 (K slices: finish)");
        return UNET_OK;
      }
    }
  }
  if (t8 * ((M + 31) / 32) < ctx->num_cu) return launch_h2<0, 1, 1, 4>(ctx, x, K, img, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, rate, seed, s, mask_climit, h2_head_args(), inb);
  if (inb == 1) return launch_h2<0, 1, 2, 4>(ctx, x, K, img, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, rate, seed, s, mask_climit);
  if (t8 * ((M + 63) / 64) < 2 * ctx->num_cu) return launch_h2<0, 1, 2, 4>(ctx, x, K, img, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, rate, seed, s, mask_climit, h2_head_args(), 2);
  // 16-row tiles (twice the MFMAs per staged chunk, 7 spilled registers at two workgroups per CU) measured 3-5 % faster on the 64 x 64 ... 128 x 128 layers when
  // they still fill the 512 resident slots, 2-4 % slower on the 256 / 512 pixel layers (fewer, longer workgroups) and much slower when the grid falls below one round
  const long long wgs16 = (long long)((wd + 31) / 32) * ((h + 15) / 16) * n * ((M + 63) / 64);
  if (h <= 128 && wgs16 >= 512) return launch_h2<0, 2, 4, 2>(ctx, x, K, img, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, rate, seed, s, mask_climit);
  return launch_h2<0, 2, 2, 2>(ctx, x, K, img, bias, mask, mask_mode, y, ldy, n, h, wd, K, M, act, rate, seed, s, mask_climit);
}

bool h2_pool_sums_selected(const unet_ctx* ctx, int algo, int wd, int K, int M) {
  (void)wd;
  return ctx && ctx->opt_pool_sums_fused && ctx->bn_slots && h2_conv3x3_selected(algo, K, M) && (M % 32) == 0 && 2 * M * (ctx->opt_deterministic ? UNET_XW : 1) <= UNET_BN_SLOT_DOUBLES;
}
int32_t k_conv3x3_h2_dgrad_pool_sums(unet_ctx* ctx, const float* dy, const void* wimg, const float* pooled, const float* gamma, const float* beta, float rate, float* dx, double* sums,
                                     int n, int h, int wd, int K, int M, hipStream_t s) {
  if (!dy || !wimg || !pooled || !gamma || !beta || !dx || rate < 0.f || rate >= 1.f) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 dgrad + pooled sums: bad args");          // (sums null: the sums stay in the slot copies for k_enc_tail_finish)
  if ((long long)h * wd * std::max(K, M) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  h2_head_args hd; hd.w = gamma; hd.b = beta; hd.aux = rate;
  const unet_bf16* img = static_cast<const unet_bf16*>(wimg);
  int32_t r;
  const long long wgs16 = (long long)((wd + 31) / 32) * ((h + 15) / 16) * n * ((M + 63) / 64);
  if (h2_nb(M) == 1) r = launch_h2<0, 1, 2, 4, 2>(ctx, dy, K, img, nullptr, pooled, MASK_POOL_SUMS, dx, M, n, h, wd, K, M, ACT_NONE, 0.0f, 0, s, 1 << 30, hd);
  else if (h <= 128 && wgs16 >= 512) r = launch_h2<0, 2, 4, 2, 2>(ctx, dy, K, img, nullptr, pooled, MASK_POOL_SUMS, dx, M, n, h, wd, K, M, ACT_NONE, 0.0f, 0, s, 1 << 30, hd);
  else r = launch_h2<0, 2, 2, 2, 2>(ctx, dy, K, img, nullptr, pooled, MASK_POOL_SUMS, dx, M, n, h, wd, K, M, ACT_NONE, 0.0f, 0, s, 1 << 30, hd);
  if (r || !sums) return r;
  return k_slot_fold(ctx, sums, 2 * M, s, ctx->opt_deterministic != 0);
}

// The network's last conv3x3 + its 1x1 sigmoid head (T1:911-913) in one launch: y = relu(conv(x)) [n,h,wd,32], p = sigmoid(y . wh + bh) [n,h,wd]; with labels t:
// the slot copies `ctx->bn_slots` receive [0,96) sum_p a y_c | sum_p t q y_c | sum_p q y_c, [96,100) sum bce, sum t p, sum t, sum p, [100,103) sum a, sum t q, sum q
// (a = dBCE/dz, q = p (1 - p)): k_head_fold moves them out; the head's weight gradient is a combination of them once the batch-global Dice sums are known
bool h2_conv3x3_head_selected(const unet_ctx* ctx, int algo, int wd, int K, int M) {
  return ctx && ctx->opt_head_fused && ctx->bn_slots && h2_conv3x3_selected(algo, K, M) && M == 32 && (wd & 7) == 0;
}
int32_t k_conv3x3_h2_head_fwd(unet_ctx* ctx, const float* x, const void* wimg, const float* bias, float* y, const float* wh, const float* bh, float* p, const float* t,
                              int n, int h, int wd, int K, hipStream_t s) {
  if (K < 16 || (K % 16) || !wh || !bh || !p) UNET_FAIL(ctx, UNET_E_ARG, "conv3x3 h2 + head: bad args");          // (y may be null: the 32-channel tensor is then not stored)
  if ((long long)h * wd * std::max(K, 32) * 4 >= (1LL << 30)) UNET_FAIL(ctx, UNET_E_SHAPE, "conv3x3 h2: one image must stay below 1 GiB (32-bit buffer offsets)");
  h2_head_args hd; hd.w = wh; hd.b = bh; hd.p = p; hd.t = t; hd.slots = ctx->bn_slots;
  return launch_h2<0, 1, 2, 4, 1>(ctx, x, K, static_cast<const unet_bf16*>(wimg), bias, nullptr, MASK_NONE, y, 32, n, h, wd, K, 32, ACT_RELU, 0.0f, 0, s, 1 << 30, hd);
}

// ---- ConvT 2x2 stride 2 (T1:886 ...) on the same kernels: forward = MODE 1 (K = cin, M = 4 cout), data gradient = MODE 2 (K = 4 cout, M = cin).
// The split weight image (4 cin cout weights -> 16 cin cout bytes + header) is rebuilt per launch into a context-owned scratch (common.h: unet_ctx::convt_img).
bool h2_convT_selected(const unet_ctx* ctx, int algo, int cin, int cout) {
  return algo == UNET_ALGO_AUTO && ctx && ctx->convt_img && cin >= 32 && (cin % 32) == 0 && cout >= 32 && (cout % 32) == 0 &&
         (size_t)H2_HEADER + (size_t)16 * cin * cout <= ctx->convt_img_bytes;
}

// u[n,2i+a,2j+b,o] = bias[o] + sum_c x[n,i,j,c] * K[a,b,o,c]   (Keras ConvT kernel [2][2][cout][cin]); y = channel slice with pixel stride ldy
size_t h2_convT_img_bytes(int cin, int cout) { return (size_t)H2_HEADER + (size_t)16 * cin * cout; }

// `prepared`: the image of kind 2 (forward) / 3 (data gradient) built by k_h2_prep_multi at the start of a program; null: built here into the context's scratch
int32_t k_convT_h2_fwd(unet_ctx* ctx, const float* x, const float* w, const float* bias, float* y, int ldy, int n, int h, int wd, int cin, int cout, hipStream_t s, const void* prepared) {
  if (!prepared) {
    const float* ws[1] = {w}; void* is[1] = {ctx->convt_img}; const int kind = 2;
    int32_t r = k_h2_prep_multi(ctx, ws, nullptr, is, &cin, &cout, &kind, 1, s);
    if (r) return r;
    prepared = ctx->convt_img;
  }
  const unet_bf16* img = static_cast<const unet_bf16*>(prepared);
  // (fewer than one four-block workgroup per CU -- the 32 x 32 and 64 x 64 inputs of batch-1 inference --: one block per workgroup on the same four-block image)
  if ((long long)((wd + 31) / 32) * ((h + 7) / 8) * n * ((4 * cout + 127) / 128) < ctx->num_cu && (cout % 32) == 0)
    return launch_h2<1, 1, 2, 2>(ctx, x, cin, img, bias, nullptr, MASK_NONE, y, ldy, n, h, wd, cin, 4 * cout, ACT_NONE, 0.0f, 0, s, 1 << 30, h2_head_args(), 4);
  return launch_h2<1, 4, 2, 2>(ctx, x, cin, img, bias, nullptr, MASK_NONE, y, ldy, n, h, wd, cin, 4 * cout, ACT_NONE, 0.0f, 0, s);
}

// dx[n,i,j,c] = sum_{ab,o} dU[n,2i+a,2j+b,o] * K[ab,o,c]; dy = channel slice with pixel stride lddy; mask: ReLU of the producer of x
int32_t k_convT_h2_dgrad(unet_ctx* ctx, const float* dy, int lddy, const float* w, const float* mask, float* dx, int n, int h, int wd, int cin, int cout, hipStream_t s, int mask_bits,
                         const void* prepared) {
  const int NB = h2_nb_convT_dgrad(cin);
  if (!prepared) {
    const float* ws[1] = {w}; void* is[1] = {ctx->convt_img}; const int kind = 3;
    int32_t r = k_h2_prep_multi(ctx, ws, nullptr, is, &cin, &cout, &kind, 1, s);
    if (r) return r;
    prepared = ctx->convt_img;
  }
  const unet_bf16* img = static_cast<const unet_bf16*>(prepared);
  const int mm = mask ? (mask_bits ? MASK_RELU_BITS : MASK_RELU) : MASK_NONE;
  if (NB == 4) return launch_h2<2, 4, 2, 2>(ctx, dy, lddy, img, nullptr, mask, mm, dx, cin, n, h, wd, 4 * cout, cin, ACT_NONE, 0.0f, 0, s);
  if (NB == 2) return launch_h2<2, 2, 4, 2>(ctx, dy, lddy, img, nullptr, mask, mm, dx, cin, n, h, wd, 4 * cout, cin, ACT_NONE, 0.0f, 0, s);
  return launch_h2<2, 1, 4, 2>(ctx, dy, lddy, img, nullptr, mask, mm, dx, cin, n, h, wd, 4 * cout, cin, ACT_NONE, 0.0f, 0, s);
}
