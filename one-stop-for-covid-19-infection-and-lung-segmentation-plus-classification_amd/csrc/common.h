// Shared host/device helpers for the gfx950 U-Net engine (internal; the public surface is
// include/unet_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <set>
#include <string>

#include "../../include/unet_hip.h"

// BN_SLOTS copies of a per-channel reduction target (<= 2 * 1024 doubles each): the workgroups of a statistics kernel spread their
// final fp64 atomics over the copies (512 workgroups hammering the same 16 cache lines cost ~40 us per launch), a tiny kernel then
// folds the copies into the caller's sums and clears them.  Owned by the context => statistics ops of ONE context must be issued on
// one stream at a time (use a context per stream otherwise).
// Deterministic mode (UNET_OPT_DETERMINISTIC): UNET_BN_SLOTS_DET copies, every workgroup of a reduction kernel owns ONE copy (grids are capped at the copy
// count), so each atomic lands on a zero it alone writes, and the fold kernel sums the copies in index order: bit-identical reruns.
constexpr int UNET_BN_SLOTS = 64, UNET_BN_SLOTS_DET = 1024, UNET_BN_SLOT_DOUBLES = 2048;
constexpr int UNET_HEAD_SUMS = 100;          // doubles behind the loss sums: the per-channel sums of a fused head's weight gradient (k_head_fold)
struct unet_ctx {
  int device = 0;
  int num_cu = 256;
  int profiling = 0;
  // unet_ctx_set_option (include/unet_hip.h: UNET_OPT_*): graph-level choices a model reads when it is created, op-level choices the launches read
  int opt_relu_bits = 1;            // ReLU masks of the data gradients as one bit per element (0: re-read the fp32 activation)
  int opt_bn_fold = 2;              // decoder BatchNorm folded into the conv behind it: 0 explicit passes, 1 forward + weight gradient + sums, 2 + backward in the dgrad epilogue, 3 + the classifier's 16-channel block
  int opt_enc_bn_fused = 1;         // encoder tail backward without a statistics pass
  int opt_bn_concat_analytic = 1;   // decoder BatchNorm statistics: skip half analytic
  int opt_bn_fuse_stats = 1;        // BatchNorm statistics from the producing conv's epilogue
  int opt_pool_sums_fused = 1;      // fp32 U-Net: the pooled-path sums of the encoder tail's BatchNorm backward from the epilogue of the data gradient that produces the pooled gradient
  int opt_skip_raw = 1;             // fp32 U-Net: an encoder block's second conv writes straight into the skip half of its concat; the encoder BatchNorm is composed into the folded decoder one
  int opt_head_fused = 1;           // the 1x1 sigmoid head + loss sums + the head's weight-gradient sums in the epilogue of the last conv3x3 (fp32 h2 kernels)
  int opt_head_bwd_fused = 1;       // the head's backward as an 8-byte-per-pixel {dz, mask} stream that the last conv's two gradients expand (no fp32 dY tensor)
  int opt_conv_pp = 1;              // shallow conv3x3 forward / data-gradient launches on the persistent two-half schedule (kernels_conv_pp.hip)
  int opt_deterministic = 0;        // fixed-order reductions everywhere (no floating-point atomics): bit-identical reruns
  double* bn_slots = nullptr;       // device, UNET_BN_SLOTS_DET x UNET_BN_SLOT_DOUBLES (16 MB), all zero between launches
  int bn_nslots() const { return opt_deterministic ? UNET_BN_SLOTS_DET : UNET_BN_SLOTS; }
  void* convt_img = nullptr;        // device scratch for the split fp16 weight image of a ConvT launch (kernels_conv_h2.hip); launches on one stream only
  size_t convt_img_bytes = 0;
  int k_slices_ok = 0;              // one-shot (unet_allow_k_slices): the next h2 conv3x3 launch may slice its contraction (inference programs arm it)
  void* splitk_ws = nullptr;        // device scratch for the partial-sum slabs of a K-sliced conv3x3 launch (kernels_conv_h2.hip: SPLITK); launches on one stream only
  size_t splitk_ws_bytes = 0;
  // Conv2D / Conv2DTranspose -> BatchNormalization (T1:860-861, 886-888): unet_request_bn_stats() arms the next forward launch; a kernel that can
  // (kernels_conv_h2.hip) adds the per-channel (sum y, sum y^2) of what it writes to bn_slots and leaves the tensor's address in stats_in_slots,
  // and the unet_bn_stats / unet_bn_stats_concat call on that tensor folds the slots without reading it
  int stats_req_c = 0;              // channels of the BatchNorm (0 = not armed)
  const void* stats_in_slots = nullptr;
  int stats_in_slots_c = 0;
  bool stats_in_slots_xs = false;   // ... as exact window sums (deterministic mode, xsum_add below) instead of doubles
  // one-shot: the next h2 conv3x3 forward with a ReLU epilogue also writes the sign bits of what it stores (MASK_RELU_BITS layout) here and leaves the
  // address in signs_done; the data gradients that would re-read the fp32 tensor as their mask read 1/32 of the bytes
  unsigned long long* signs_req = nullptr;
  const void* signs_done = nullptr;
  std::set<const void*> big_lds_kernels;   // kernels already opted in to > 64 KiB of dynamic LDS on this context's device
  std::set<const void*> noted_kernels;     // h2 kernels whose private-segment size has been looked up (unet_note_kernel)
  int max_scratch_bytes = 0;               // the largest private segment (spill scratch) among the h2 conv kernels this context has launched: unet_ctx_max_kernel_scratch_bytes
  std::string err;
};

#define UNET_FAIL(ctx, code, ...)                         \
  do {                                                    \
    char _b[512];                                         \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                \
    if (ctx) (ctx)->err = _b;                             \
    return (code);                                        \
  } while (0)

#define UNET_CHECK_LAUNCH(ctx, what)                                              \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess) UNET_FAIL(ctx, UNET_E_HIP, "%s: %s", what, hipGetErrorString(_e)); \
  } while (0)

#define UNET_HIP(ctx, expr)                                                       \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) UNET_FAIL(ctx, UNET_E_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

// Opt a kernel in to more than 64 KiB of dynamic LDS, once per (context = device, kernel): the attribute is per device, so a
// process that drives several devices through several contexts sets it on each.
static inline int32_t unet_big_lds(unet_ctx* ctx, const void* kernel, size_t bytes, const char* what) {
  if (ctx->big_lds_kernels.count(kernel)) return UNET_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) UNET_FAIL(ctx, UNET_E_HIP, "%s: cannot reserve %zu bytes of LDS", what, bytes);
  ctx->big_lds_kernels.insert(kernel);
  return UNET_OK;
}
// The register-heavy kernels live a few registers from their spill cliff (the ELU / dropout instances of conv_h2_kernel: 32-40 spilled VGPRs, 36-80 B of scratch per lane as
// built); a change that pushes one over it does not fail any numerics test -- it made U-Net++ 2.6x slower once (round 5: 640 B of scratch).  Every launch site notes its
// kernel's private-segment size once per context; tests/test_gpu_unetpp.py holds the maximum under a bound.
static inline void unet_note_kernel(unet_ctx* ctx, const void* kernel) {
  if (ctx->noted_kernels.count(kernel)) return;
  ctx->noted_kernels.insert(kernel);
  hipFuncAttributes at;
  if (hipFuncGetAttributes(&at, kernel) == hipSuccess && (int)at.localSizeBytes > ctx->max_scratch_bytes) ctx->max_scratch_bytes = (int)at.localSizeBytes;
}
#define UNET_BIG_LDS(ctx, kern, bytes, what) do { int32_t _r = unet_big_lds(ctx, reinterpret_cast<const void*>(kern), bytes, what); if (_r) return _r; } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- deterministic mode: sums taken in a kernel EPILOGUE (thousands of workgroups per launch, far more than there are slot copies) as exact, order-independent
// window sums.  A float t is an integer multiple of 2^(e - 23); four 64-bit integer accumulators per value with least-significant bits 2^-80, 2^-48, 2^-16, 2^16
// take it without rounding: q = t / lsb(w) of the window below its lowest bit (56 bits at most), its low 32 bits into window w, the rest into window w + 1.
// Integer addition is associative, so the atomics may land in any order and on any number of slot copies: the folded value is the same bits on every run -- and
// exact, where the fp64 atomics of the default mode round.  Domain: 2^-57 <= |t| < 2^60 (below: the bits under 2^-80 are dropped, the same ones on every run; from 2^39 on a
// float is a whole multiple of 2^16 and goes into the top window as ONE integer -- the sum of y^2 over a 256-pixel tile stays in the domain for |y| up to ~6e7; above 2^60,
// Inf, NaN: the value is poisoned and folds to NaN, as a floating-point sum would be).  A slot row holds UNET_BN_SLOT_DOUBLES / 4 such values.
constexpr int UNET_XW = 4;
__device__ __forceinline__ void xsum_add(double* row, int idx, float t) {
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(row) + (size_t)UNET_XW * idx;
  const int eb = (int)((__float_as_uint(t) >> 23) & 0xFF);
  if (eb == 0) return;                                       // zero (a denormal: below every window)
  const int e = eb - 127;
  if (e >= 60) { atomicMax(reinterpret_cast<long long*>(acc + 3), 1LL << 62); return; }          // out of the domain / Inf / NaN: poison
  if (e >= 39) {                                             // lsb(t) = 2^(e - 23) >= 2^16: t / 2^16 is an integer below 2^44 -- exact in the top window (1024 of them stay below 2^61)
    atomicAdd(acc + 3, (unsigned long long)__double2ll_rd((double)t * (1.0 / 65536.0)));
    return;
  }
  const int w = min(max((e - 23 + 80) >> 5, 0), 2);
  const double sc = __longlong_as_double((long long)(1023 + 80 - 32 * w) << 52);          // 2^(80 - 32 w)
  const long long q = __double2ll_rd((double)t * sc);
  atomicAdd(acc + w, (unsigned long long)(q & 0xFFFFFFFFll));
  atomicAdd(acc + w + 1, (unsigned long long)(q >> 32));
}
// the value of index idx summed over the first nslots slot copies (row stride UNET_BN_SLOT_DOUBLES); the words are cleared for the next launch
__device__ __forceinline__ double xsum_take(double* slots, int idx, int nslots) {
  long long S0 = 0, S1 = 0, S2 = 0, S3 = 0; bool bad = false;
  for (int k0 = 0; k0 < nslots; k0 += 8) {                    // (nslots is a multiple of 8) eight copies' 32 bytes in flight; integer sums: no order to keep
    longlong2 a[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      longlong2* p = reinterpret_cast<longlong2*>(slots + (size_t)(k0 + k) * UNET_BN_SLOT_DOUBLES + (size_t)UNET_XW * idx);
      a[k] = p[0]; b[k] = p[1];
      p[0] = make_longlong2(0, 0); p[1] = make_longlong2(0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { S0 += a[k].x; S1 += a[k].y; S2 += b[k].x; S3 += b[k].y; bad |= b[k].y >= (1LL << 61); }
  }
  const double r = ((double)S3 * 65536.0 + (double)S2 * (1.0 / 65536.0)) + ((double)S1 * __longlong_as_double((long long)(1023 - 48) << 52) + (double)S0 * __longlong_as_double((long long)(1023 - 80) << 52));
  return bad ? __longlong_as_double(0x7FF8000000000000ll) : r;
}

// (a sc, b sc) as packed fp16 pairs h, m with x sc ~ h + m: h = RN_f16(x sc), m = RN_f16(x sc - h) -- the two-term split of the h2 kernels with the block scale folded in.
// v_fma_mixlo / mixhi_f16 take fp32 and fp16 sources in one fused multiply-add and round ONCE to fp16: h is the product itself (exact in fp32: sc is a power of two),
// m the exact difference -- the same bits as scale, v_cvt_pk_f16_f32, two v_cvt_f32_f16, v_pk_fma, v_cvt_pk (six VALU instructions per pair) in four.
__device__ __forceinline__ void split2_scaled(float a, float b, float sc, unsigned& h, unsigned& m) {
  unsigned hh, mm;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hh) : "v"(a), "v"(sc));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hh) : "v"(b), "v"(sc));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(mm) : "v"(a), "v"(sc), "v"(hh));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(mm) : "v"(b), "v"(sc), "v"(hh));
  h = hh; m = mm;
}

// Maximum over the wave of a NON-NEGATIVE float (a block's max |x|), returned wave-uniform in a scalar register: four DPP steps inside each row of 16 lanes (the
// operands ride on the v_max instructions: no LDS round trip) + one v_readlane per row.  Non-negative floats order like their bit patterns, so the maxima are integer
// ones (no NaN canonicalisation; a NaN input wins and poisons the block, which it does anyway).  Replaces six dependent ds_bpermute exchanges (~600 cycles of LDS
// latency on the critical path of every staged chunk of the h2 kernels).
__device__ __forceinline__ float wave_max_nonneg(float x) {
  unsigned v = __float_as_uint(x);
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));           // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));           // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));          // row_half_mirror: the other quad of the 8
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));          // row_mirror: the other 8 of the 16
  const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16), r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
  return __uint_as_float(max(max(r0, r1), max(r2, r3)));
}

// v[lane `l`] = s (v_writelane_b32: the lane select must be an immediate beside a scalar source -- one constant-bus operand; `l` folds after unrolling)
__device__ __forceinline__ void unet_writelane(int& v, unsigned s, int l) {
#define UNET_WL(L) case L: asm("s_nop 1\n\tv_writelane_b32 %0, %1, " #L : "+v"(v) : "s"(s)); break;          // (s_nop: a VALU-written scalar needs two wait states in front of a VALU read on gfx950 -- nothing pads an asm statement)
  switch (l) { UNET_WL(0) UNET_WL(1) UNET_WL(2) UNET_WL(3) UNET_WL(4) UNET_WL(5) UNET_WL(6) UNET_WL(7) UNET_WL(8) UNET_WL(9) UNET_WL(10) UNET_WL(11) UNET_WL(12) UNET_WL(13) UNET_WL(14) UNET_WL(15) default: break; }
#undef UNET_WL
}

// Branch-free guarded 16-byte load: the hardware range check of a buffer descriptor returns 0 for byte offsets >= the record
// count, so halo / overhang lanes simply carry an out-of-range offset (no exec-mask branch, no select).  Tensors addressed this
// way must be smaller than 1 GiB (COL_OOB + a valid row offset must still be out of range).
typedef float unet_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned unet_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned unet_u32x2 __attribute__((ext_vector_type(2)));
#ifdef __HIPCC__
// bits (b, b + 1) of v -> 0xFFFF in the low / high half of a packed fp16 pair: two sign-extending one-bit field extracts and one bit-field insert
__device__ __forceinline__ unsigned h2_pair_mask(unsigned v, int b) {
  // (as instructions: written with & / | / ?: the compiler turns each bit into and + compare + select + or -- 11 operations per pair of masks instead of 3)
  unsigned lo, hi, r;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(lo) : "v"(v), "n"(b));
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(hi) : "v"(v), "n"(b + 1));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0xFFFFu), "v"(lo), "v"(hi));
  return r;
}
#endif
constexpr int UNET_OOB = (int)0x80000000u;        // invalid element (or row)
constexpr int UNET_COL_OOB = 0x40000000;          // invalid column part, may be added to a valid or invalid row part
// `soff` = wave-uniform byte offset (an SGPR operand of the instruction: no per-lane add; it does not bring an out-of-range lane
// back in range, the per-lane offset alone is already >= the record count)
__device__ __forceinline__ unet_f32x4 buf_ld4(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff = 0) {
  return __builtin_bit_cast(unet_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, soff, 0));
}
__device__ __forceinline__ float buf_ld1(__amdgpu_buffer_rsrc_t rs, int byte_off, int soff = 0) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// ---- bf16 storage (activations / activation gradients of the mixed-precision path; arithmetic stays fp32) -----------------
// unet_bf16 (include/unet_hip.h) = raw 16-bit pattern.  Conversions are round-to-nearest-even (v_cvt_pk_bf16_f32), the same
// rounding torch's .bfloat16() applies, so the oracle can quantise at the same points.
typedef __bf16 unet_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 unet_bf16x2 __attribute__((ext_vector_type(2)));
typedef float unet_f32x2 __attribute__((ext_vector_type(2)));
typedef float unet_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((unet_f32x2){lo, hi}, unet_bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// 4 consecutive channels of an NHWC tensor as fp32, whatever the storage type (16-byte / 8-byte lane accesses)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 ld4(const unet_bf16* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
__device__ __forceinline__ void st4(unet_bf16* p, float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const unet_bf16* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(unet_bf16* p, float v) { *p = (unet_bf16)(pack_bf16x2(v, 0.f) & 0xFFFFu); }

// Philox-4x32-10 counter RNG (dropout keep-mask): counter = element-quad index, key = seed.
__device__ __forceinline__ uint4 philox4x32(uint64_t ctr, uint64_t seed) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// inverted-dropout factors (0 or 1/(1-rate)) of element quad `quad_idx` (4 consecutive channels of a dense NHWC tensor)
__device__ __forceinline__ float4 keep_scale(long long quad_idx, float rate, uint64_t seed) {
  uint4 r = philox4x32((uint64_t)quad_idx, seed);
  float s = 1.0f / (1.0f - rate);
  return make_float4(u32_to_unit(r.x) >= rate ? s : 0.f, u32_to_unit(r.y) >= rate ? s : 0.f,
                     u32_to_unit(r.z) >= rate ? s : 0.f, u32_to_unit(r.w) >= rate ? s : 0.f);
}

// activations of the conv epilogues (Keras 'relu' T1:859 / 'elu' task1_unet_plus_plus.py:876)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2 };
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_ELU) return v > 0.0f ? v : expm1f(v);
  return v;
}
// backward factor of the activation (+ dropout) that PRODUCED a stored tensor value m:
//   MASK_RELU: 1[m>0]      MASK_ELU: m>0 ? 1 : m+1  (elu' expressed through its output)
//   MASK_ELU_DROP: m = dropout(elu(z)): keep ? s * elu'(m/s) : 0, keep/s recomputed from the Philox stream
enum { MASK_NONE = 0, MASK_RELU = 1, MASK_ELU = 2, MASK_ELU_DROP = 3 };
// internal to the h2 / bf16 forward kernels: `mask` is not a mask but the border-class bias table of a conv whose input BatchNorm was
// folded into its weights (k_bn_fold_prepare): [16][Cout], class = 4 * (row == 0 | (row == H-1) << 1) + (col == 0 | (col == W-1) << 1)
enum { MASK_BIAS_TAB = 4 };
// internal to the same kernels, data-gradient launches: the BatchNorm backward of the conv's (folded) input BatchNorm applied in the epilogue --
// `mask` = the BatchNorm's input x (same shape as the output), `bias` = coefficients [3][Cout]: out = K0 * dz + K1 * x + K2 (k_bn_bwd_coef)
enum { MASK_BN_BWD = 5 };
// ... followed by the derivative of what produced x (U-Net++ conv_block: x = dropout(elu(conv)), keep mask recomputed from rate / seed): out = mask_factor(x) * (K0 dz + K1 x + K2)
enum { MASK_BN_BWD_ELU = 6, MASK_BN_BWD_ELU_DROP = 7 };
// ... or by the ReLU mask of x's producer (classifier: Conv(relu) -> BN -> Conv, T2:748-751): out = x > 0 ? K0 dz + K1 x + K2 : 0
enum { MASK_BN_BWD_RELU = 8 };
// h2 kernels only: the ReLU mask as ONE BIT per element, written by the forward launch that produced the tensor (unet_ctx::signs_req):
// u64 words [n][y][x / 8][c / 32][4]; word k of an (8 pixels x 32 channels) cell holds bit (pixel % 8) * 8 + (channel % 32) / 4 for channel % 4 == k
enum { MASK_RELU_BITS = 9 };
// h2 kernels only, data-gradient launches whose output is the gradient of a max-pool + dropout output p (U-Net encoder, T1:862-863): `mask` = p; besides the plain data
// gradient g the epilogue adds the two per-channel sums of the encoder tail's BatchNorm backward to the context's slot copies (pool_bwd_sums_kernel's definition):
//   sum g ks,  sum g ks (p (1 - rate) - beta) / gamma,   ks = 1 / (1 - rate) where the element was kept, 0 where dropout removed it
// -- a removed element is stored as -0.0f by the forward (bn_pool_fwd_kernel), a kept zero as +0.0f, so no random stream is replayed here.  h2_head_args carries gamma / beta / rate
enum { MASK_POOL_SUMS = 10 };
int32_t k_bn_bwd_coef(unet_ctx*, const float* bnp, const double* sums, double count, float* coef, int c, hipStream_t s);
__device__ __forceinline__ float mask_factor(float m, int mode, float ks /* keep_scale component */, float rate) {
  if (mode == MASK_RELU) return m > 0.0f ? 1.0f : 0.0f;
  if (mode == MASK_ELU) return m > 0.0f ? 1.0f : m + 1.0f;
  if (mode == MASK_ELU_DROP) { const float a = m * (1.0f - rate); return ks * (a > 0.0f ? 1.0f : a + 1.0f); }
  return 1.0f;
}

// internal launchers shared between the op-level ABI and the model programs -------------
// (definitions in the .hip files; all return a unet status code)
// act: ACT_*; mask_mode: MASK_* (data-gradient epilogue); rate/seed: the fused output dropout (forward) or the dropout
// of the mask tensor's producer (MASK_ELU_DROP)
// conv3x3(BN-affine(x)) with the affine folded into the conv (kernels_bnfold.hip): w_scaled[tap][c][o] = w * scale[c], table[16][cout] = the bias
// per border class (bias + the shift's contribution through the taps that stay inside the image); the forward then runs on the RAW x with
// (bias = table, mask = table, mask_mode = MASK_BIAS_TAB) on the h2 kernels, and the weight gradient on raw x is
// corrected afterwards: dw = scale[c] * dw_raw + shift[c] * S[tap][o] (k_wgrad_bn_fold_fix; S from db and the border sums of dy).
size_t bn_fold_scratch_floats(int cin, int cout);      // [w_scaled 9*cin*cout][table 16*cout][partials]
int32_t k_bn_fold_prepare(unet_ctx*, const float* w, const float* bias, const float* scale, const float* shift, int cin, int cout, float* scratch, hipStream_t s, bool want_scaled = true);
size_t wgrad_bn_fold_scratch_floats(int n, int cout);
bool wgrad_bn_fold_supported(int cout);
int32_t k_wgrad_bn_fold_fix(unet_ctx*, const float* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db, float* scratch,
                            hipStream_t s, const float* w = nullptr, const float* mean = nullptr, const float* istd = nullptr, double* bn_bwd_sums = nullptr,
                            const float* pre_s = nullptr, const float* pre_t = nullptr,           // (pre: the BatchNorm saw pre_s x + pre_t of the tensor the weight gradient ran on)
                            float* dgamma = nullptr, float* dbeta = nullptr);                     // (with bn_bwd_sums: also the BatchNorm's parameter gradients = the local sums)
int32_t k_wgrad_bn_fold_fix_bf16(unet_ctx*, const unet_bf16* dy, int n, int h, int wd, int cin, int cout, const float* scale, const float* shift, float* dw, const float* db,
                                 float* scratch, hipStream_t s, const float* w = nullptr, const float* mean = nullptr, const float* istd = nullptr, double* bn_bwd_sums = nullptr);
constexpr int UNET_PREP_MAX = 36;          // layers per batched weight-preparation launch (the list travels as a kernel argument: < 4 KiB)
// fp32 conv3x3 / ConvT on the fp16 matrix cores through the block-scaled 2-term fp16 split (kernels_conv_h2.hip, kernels_wgrad_h2.hip): three fp16 MFMA products per
// multiply.  The family of UNET_ALGO_AUTO wherever the channel counts allow; K = contraction channels, M = output channels of a launch
bool h2_conv3x3_selected(int algo, int K, int M);
// the persistent two-half schedule of the shallow levels (kernels_conv_pp.hip): same weight image, same epilogue contract as k_conv3x3_h2_fwd for the launches it takes
bool pp_conv3x3_selected(const unet_ctx* ctx, int K, int M, int n, int h, int wd, const float* mask, int mask_mode, int act, float rate, int ldy);
int32_t k_conv3x3_pp_fwd(unet_ctx*, const float* x, const void* wimg, const float* bias, const float* mask, int mask_mode, float* y, int ldy, int n, int h, int wd, int K, int M,
                         int act, hipStream_t s, bool vdy = false);          // (vdy: x = the head's {dz, mask} stream [n,h,wd] of 8 bytes, K = 32: k_conv3x3_h2_dgrad_dzm)
size_t h2_wimg_bytes(int K, int M);
// (cs: optional per-input-channel factor of a forward image -- the scale of a BatchNorm folded into the conv; the scaled weights are never materialised)
int32_t k_h2_weights(unet_ctx*, const float* w, void* img, int cin, int cout, int flip, hipStream_t s, const float* cs = nullptr);
int32_t k_h2_weights_multi(unet_ctx*, const float* const* w, void* const* img, const int* cin, const int* cout, const int* flip, int count, hipStream_t s);
// any mix of layers in TWO launches; kind 0 / 1 = conv3x3 forward / data-gradient image, 2 / 3 = ConvT forward / data-gradient image (h2_convT_img_bytes each)
// (kind 4 = a kind-0 layer of which only the raw-weight maxima are taken: its image is built later by k_h2_weights_bound, once the per-channel factor exists)
int32_t k_h2_weights_bound(unet_ctx*, const float* w, const float* cs, void* img, int cin, int cout, hipStream_t s);
int32_t k_h2_prep_multi(unet_ctx*, const float* const* w, const float* const* cs, void* const* img, const int* cin, const int* cout, const int* kind, int count, hipStream_t s);
size_t h2_convT_img_bytes(int cin, int cout);
// mask_climit (folded-BatchNorm data gradients, MASK_BN_BWD*): output channels >= mask_climit do not read x -- they get K0 dz + K2, the K1 x term is added by their consumer
int32_t k_conv3x3_h2_fwd(unet_ctx*, const float* x, const void* wimg, const float* bias, const float* mask, int mask_mode, float* y, int n, int h, int wd, int K, int M,
                         int act, float rate, uint64_t seed, hipStream_t s, int mask_climit = 1 << 30, int ldy = 0);
// Encoder BatchNorm whose output is the skip half of a decoder concat that is itself normalised and folded into the conv behind it (T1:861 -> 908-910): two affine maps
// in a row are one.  comp = [scale' 2C][shift' 2C][pre_s 2C][pre_t 2C] with, for the skip channels j = C..2C-1, scale' = s_dec s_enc, shift' = s_dec t_enc + t_dec,
// pre = (s_enc, t_enc) (x_dec = pre_s x_raw + pre_t: what the decoder BatchNorm's backward sums are taken over); the upsampled half keeps (s_dec, t_dec), pre = (1, 0)
// the data gradient of the conv behind an encoder tail with the tail's BatchNorm-backward sums in its epilogue (MASK_POOL_SUMS): dy [n,h,wd,K] -> dx [n,h,wd,M], pooled = the
// forward's max-pool + dropout output [n,h,wd,M]; sums[2 M] += (folded out of the slot copies behind the launch)
int32_t k_slot_fold(unet_ctx*, double* sums, int count, hipStream_t s, bool xs = false);          // xs: the slots hold exact window sums (xsum_add), UNET_BN_SLOTS copies          // sums[i] += the slot copies' entries i (cleared), in index order (kernels_pointwise.hip)
// the pooled sums a MASK_POOL_SUMS launch left in the slot copies -> sums[2 c] (+=), + the closed-form skip term (unet_bn_bwd_skip_term) + the BatchNorm's parameter gradients: one launch
int32_t k_enc_tail_finish(unet_ctx*, double* sums, const double* dec_sum_dyxhat, const float* dec_invstd, const float* dec_gamma, const float* gamma, float* dgamma, float* dbeta, int c,
                          double frac, hipStream_t s);
bool h2_pool_sums_selected(const unet_ctx* ctx, int algo, int wd, int K, int M);
int32_t k_conv3x3_h2_dgrad_pool_sums(unet_ctx*, const float* dy, const void* wimg, const float* pooled, const float* gamma, const float* beta, float rate, float* dx, double* sums,
                                     int n, int h, int wd, int K, int M, hipStream_t s);
// unet_bn_finalize_train / _infer of a decoder BatchNorm (c = 2 C_enc channels) AND k_bn_compose in one launch
int32_t k_bn_finalize_compose(unet_ctx*, int training, const double* sums, double count, const float* gamma, const float* beta, float* mm, float* mv, float* bnp, int c,
                              const float* enc_bnp, float* comp, hipStream_t s);
// unet_loss_finalize / unet_cls_loss_finalize with a second destination (unet_model_set_loss_out; nullptr = none)
int32_t k_loss_finalize(unet_ctx*, const double* loss_sums, double count, float* loss_out, float* loss_out2, hipStream_t s);
int32_t k_cls_loss_finalize(unet_ctx*, const double* sums, double count, float* out, float* out2, hipStream_t s);
int32_t k_bn_compose(unet_ctx*, const float* bnp_dec, const float* bnp_enc, float* comp, int c, hipStream_t s);
struct h2_head_args { const float* w = nullptr; const float* b = nullptr; float* p = nullptr; const float* t = nullptr; double* slots = nullptr; float aux = 0.0f; };          // (MASK_POOL_SUMS: w = gamma, b = beta, aux = dropout rate)
bool h2_conv3x3_head_selected(const unet_ctx* ctx, int algo, int wd, int K, int M);
int32_t k_conv3x3_h2_head_fwd(unet_ctx*, const float* x, const void* wimg, const float* bias, float* y, const float* wh, const float* bh, float* p, const float* t,
                              int n, int h, int wd, int K, hipStream_t s);
// the sums k_conv3x3_h2_head_fwd left in the slot copies -> loss_sums[4] (+=) and head_sums[99] (+=): [32 x 3 per-channel sums][sum a, sum t q, sum q]
int32_t k_head_fold(unet_ctx*, double* loss_sums, double* head_sums, hipStream_t s);
// backward of the fused head: dy[p][c] = dz_p w_c [y_pc > 0] (mask from the sign bits `bits` or, if null, from y itself) and -- workgroup 0 -- the head's weight /
// bias gradient out of head_sums and the batch-global loss sums (ACCUMULATED into dw[32], db[1])
int32_t k_head_dy(unet_ctx*, const float* p, const float* t, const double* loss_sums, double count, const double* head_sums, const float* w, const unsigned long long* bits,
                  const float* y, float* dy, float* dw, float* db, int n, int h, int wd, hipStream_t s);
int32_t k_bn_maxpool_bwd_apply_k1(unet_ctx*, const float* x, int ldx, const float* bnp, const double* sums, double count, const float* g_skip, int ldg, const float* skip_k1,
                                  const float* dy_pooled, float* dx, int lddx, int n, int h, int wd, int c, float rate, uint64_t seed, hipStream_t s);
bool h2_convT_selected(const unet_ctx* ctx, int algo, int cin, int cout);
int32_t k_convT_h2_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, int ldy, int n, int h, int wd, int cin, int cout, hipStream_t s, const void* prepared = nullptr);
int32_t k_convT_h2_dgrad(unet_ctx*, const float* dy, int lddy, const float* w, const float* mask, float* dx, int n, int h, int wd, int cin, int cout, hipStream_t s, int mask_bits = 0,
                         const void* prepared = nullptr);
bool h2_wgrad_selected(int algo, int cin, int cout);
bool h2_wgrad_c16_selected(int algo, int wd, int cin, int cout);          // 16 -> 16 channels as pixel pairs (kernels_wgrad_h2.hip)
size_t h2_wgrad_c16_ws_bytes(int n, int h, int wd);
int32_t k_wgrad_c16_gather(unet_ctx*, const float* G, float* dw, float* db, hipStream_t s);
int32_t k_conv3x3_h2_wgrad_c16(unet_ctx*, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, hipStream_t s);
size_t h2_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
int32_t k_conv3x3_h2_wgrad(unet_ctx*, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin, int cout, hipStream_t s);
// the head's backward as a rank-1 stream: k_head_dzm writes dzm[n,h,wd] = {dz, 32 mask bits} (+ the head's own dw / db, accumulated), the two gradients of the
// last conv3x3 expand it while staging
int32_t k_head_dzm(unet_ctx*, const float* p, const float* t, const double* loss_sums, double count, const double* head_sums, const unsigned long long* bits, void* dzm, float* dw,
                   float* db, int n, int h, int wd, hipStream_t s);
bool h2_head_bwd_selected(const unet_ctx* ctx, int algo, int wd, int cin);
int32_t k_conv3x3_h2_dgrad_dzm(unet_ctx*, const void* dzm, const void* wimg, const float* mask, int mask_mode, float* dx, int n, int h, int wd, int M, hipStream_t s);
int32_t k_conv3x3_h2_wgrad_dzm(unet_ctx*, const float* x, const void* dzm, const float* w_head, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin,
                               hipStream_t s);
int32_t k_conv3x3_naive_fwd(unet_ctx*, const float* x, const float* w, const float* bias, const float* mask, int mask_mode,
                            float* y, int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, hipStream_t s);
bool c1_relu_bits_supported(int wd, int cout);
int32_t k_conv3x3_c1_fwd_bits(unet_ctx*, const float* x, const float* w, const float* bias, float* y, unsigned long long* signs, int n, int h, int wd, int cout, hipStream_t s);
int32_t k_conv3x3_c1_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, int n, int h,
                         int wd, int cout, int act, float rate, uint64_t seed, hipStream_t s);
size_t c1_wgrad_ws_bytes(int cout);
int32_t k_conv3x3_c1_wgrad(unet_ctx*, const float* x, const float* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                           int wd, int cout, hipStream_t s);
int32_t k_flip_transpose_w3x3(unet_ctx*, const float* w, float* wt, int cin, int cout, hipStream_t s);
int32_t k_conv3x3_naive_wgrad(unet_ctx*, const float* x, const float* dy, float* dw, float* db, int n, int h,
                              int wd, int cin, int cout, hipStream_t s);
int32_t k_convT_naive_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, int ldy, int n,
                          int h, int wd, int cin, int cout, hipStream_t s);
int32_t k_convT_naive_dgrad(unet_ctx*, const float* dy, int lddy, const float* w, const float* mask, float* dx,
                            int n, int h, int wd, int cin, int cout, hipStream_t s);
int32_t k_convT_naive_wgrad(unet_ctx*, const float* x, const float* dy, int lddy, float* dw, float* db, int n,
                            int h, int wd, int cin, int cout, hipStream_t s);
// MFMA paths (kernels_conv_mfma.hip); return UNET_E_SHAPE if the shape is unsupported
bool mfma_conv3x3_supported(int cin, int cout);
int32_t k_conv3x3_mfma_fwd(unet_ctx*, const float* x, const float* w, const float* bias, const float* mask, int mask_mode,
                           float* y, int n, int h, int wd, int cin, int cout, int act, float rate, uint64_t seed, hipStream_t s);
bool mfma_wgrad_supported(int ca, int cb);
bool mfma_convT_supported(int cin, int cout);
int32_t k_convT_mfma_fwd(unet_ctx*, const float* x, const float* w, const float* bias, float* y, int ldy, int n, int h, int wd,
                         int cin, int cout, hipStream_t s);
int32_t k_convT_mfma_dgrad(unet_ctx*, const float* dy, int lddy, const float* w, const float* mask, float* dx, int n, int h, int wd,
                           int cin, int cout, hipStream_t s);
size_t mfma_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
bool h2_convT_wgrad_selected(int algo, int cin, int cout);
size_t h2_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
int32_t k_convT_h2_wgrad(unet_ctx*, const float* x, const float* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd, int cin, int cout,
                         hipStream_t s);
int32_t k_convT_mfma_wgrad(unet_ctx*, const float* x, const float* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n,
                           int h, int wd, int cin, int cout, hipStream_t s);
size_t mfma_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
int32_t k_conv3x3_mfma_wgrad(unet_ctx*, const float* x, const float* dy, float* dw, float* db, void* ws,
                             size_t ws_bytes, int n, int h, int wd, int cin, int cout, hipStream_t s);
// bf16-storage convolutions (kernels_bf16.hip); wimg = scratch for the re-laid-out bf16 weights (>= 9*cin*cout, resp. 4*cin*cout, elements)
bool bf16_conv3x3_supported(int cin, int cout);
bool bf16_convT_supported(int cin, int cout);
bool bf16_wgrad_supported(int ca, int cb);
int32_t k_conv3x3_bf16_fwd(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, const unet_bf16* mask, int mask_mode, unet_bf16* y, int n, int h,
                           int wd, int cin, int cout, int act, float rate, uint64_t seed, unet_bf16* wimg, int flip, hipStream_t s,
                           const unet_bf16* prepared = nullptr);
struct unet_wimg_prep { const float* w; unet_bf16* img; long long tap_stride, sk, sm, total8; int nb, nchunks, flip, m; };
struct unet_wimg_prep_list { unet_wimg_prep item[UNET_PREP_MAX]; int n; };       // passed by value as a kernel argument (2.2 KiB)
int32_t k_wimg_multi(unet_ctx*, unet_wimg_prep_list* L, const int* cin, const int* cout, hipStream_t s);
int32_t k_convT_bf16_fwd(unet_ctx*, const unet_bf16* x, const float* w, const float* bias, unet_bf16* y, int ldy, int n, int h, int wd, int cin, int cout,
                         unet_bf16* wimg, hipStream_t s);
int32_t k_convT_bf16_dgrad(unet_ctx*, const unet_bf16* dy, int lddy, const float* w, const unet_bf16* mask, unet_bf16* dx, int n, int h, int wd, int cin,
                           int cout, unet_bf16* wimg, hipStream_t s);
size_t bf16_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
size_t bf16_convT_wgrad_ws_bytes(int n, int h, int wd, int cin, int cout);
int32_t k_conv3x3_bf16_wgrad(unet_ctx*, const unet_bf16* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd,
                             int cin, int cout, hipStream_t s);
int32_t k_convT_bf16_wgrad(unet_ctx*, const unet_bf16* x, const unet_bf16* dy, int lddy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h,
                           int wd, int cin, int cout, hipStream_t s);
size_t wgrad_reduce_scratch_floats(int taps, int ca, int cb, int cbias, int nslabs);
int32_t k_wgrad_reduce(unet_ctx*, float* part, int nslabs, int taps, int ca, int cb, int cbias, float* dw, float* db, hipStream_t s);
// first layer (cin = 1): fp32 image in, bf16 activations out / bf16 gradient in (kernels_conv_naive.hip)
int32_t k_conv3x3_c1_fwd_bf16(unet_ctx*, const float* x, const float* w, const float* bias, unet_bf16* y, int n, int h, int wd, int cout, int act,
                              float rate, uint64_t seed, hipStream_t s);
int32_t k_conv3x3_c1_wgrad_bf16(unet_ctx*, const float* x, const unet_bf16* dy, float* dw, float* db, void* ws, size_t ws_bytes, int n, int h, int wd,
                                int cout, hipStream_t s);
