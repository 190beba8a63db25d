// Probe: is fp32 = hi + mid + lo (three bf16 terms) through v_mfma_f32_32x32x16_bf16 as accurate as the fp32 MFMA?
// D[32][32] = sum_k A[m][k] B[k][n], K = 16 * KSTEPS, random normal data.  Compares against a float64 host reference:
//   (a) v_mfma_f32_32x32x2_f32 (the engine's fp32 path), (b) 6-product split (hh, hm, mh, hl, lh, mm), (c) 3-product split (hh, hm, mh),
//   (d) 6-product with the small terms accumulated FIRST into a separate accumulator.  Also times (a) vs (b) per K step.
// build: hipcc --offload-arch=gfx950 -O2 -o split3_probe tools/probe/split3_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_round(float x) {                  // value of RN_bf16(x) as a float
  bf16x2 v = __builtin_convertvector((f32x2){x, 0.f}, bf16x2);
  unsigned u = __builtin_bit_cast(unsigned, v);
  return __uint_as_float(u << 16);
}
__device__ __forceinline__ __bf16 to_bf16(float x) {
  bf16x2 v = __builtin_convertvector((f32x2){x, 0.f}, bf16x2);
  return v[0];
}

// mode 0: fp32 mfma; 1: 6 products; 2: 3 products; 3: 6 products, small terms in their own accumulator
__global__ void gemm_probe(const float* A, const float* B, float* D, int ksteps, int mode, int reps, float wscale) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  const int K = 16 * ksteps;
  f32x16 acc, acc2;
  for (int rep = 0; rep < reps; ++rep) {
    for (int i = 0; i < 16; ++i) { acc[i] = 0; acc2[i] = 0; }
    for (int ks = 0; ks < ksteps; ++ks) {
      float a[8], b[8];
      for (int j = 0; j < 8; ++j) { a[j] = A[r * K + ks * 16 + h * 8 + j]; b[j] = B[(ks * 16 + h * 8 + j) * 32 + r]; }
      if (mode >= 4) {
        // fp16 two-plane split: x = h + m (11 + 11 bits); mode 4: hh + hm + mh; mode 5: + mm
        f16x8 ah, am, bh, bm;
        for (int j = 0; j < 8; ++j) {
          float x = a[j]; _Float16 xh = (_Float16)x; _Float16 xm = (_Float16)(x - (float)xh); ah[j] = xh; am[j] = xm;
          float y = b[j] * wscale; _Float16 yh = (_Float16)y; _Float16 ym = (_Float16)(y - (float)yh); bh[j] = yh; bm[j] = ym;
        }
        if (mode == 5) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
      } else if (mode == 0) {
        // 32x32x2 f32: lane (r, h) supplies k = h for each of 8 sub-steps; remap: sub-step s covers k = 2 s + h of this 16-chunk
        for (int s = 0; s < 8; ++s) {
          const float av = A[r * K + ks * 16 + 2 * s + h], bv = B[(ks * 16 + 2 * s + h) * 32 + r];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
      } else {
        bf16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
          float x = a[j]; float xh = bf16_round(x); float r1 = x - xh; float xm = bf16_round(r1); float xl = r1 - xm;
          ah[j] = to_bf16(xh); am[j] = to_bf16(xm); al[j] = to_bf16(xl);
          float y = b[j]; float yh = bf16_round(y); float s1 = y - yh; float ym = bf16_round(s1); float yl = s1 - ym;
          bh[j] = to_bf16(yh); bm[j] = to_bf16(ym); bl[j] = to_bf16(yl);
        }
        if (mode == 3) {
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        } else {
          if (mode == 1) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
      }
    }
  }
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = (acc[i] + acc2[i]) * (mode >= 4 ? 1.0f / wscale : 1.0f); }
}

// pure issue-rate comparison: back-to-back MFMAs on 4 independent accumulators, 1 wave per SIMD on every CU
__global__ void rate_probe(float* out, int iters, int which) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = threadIdx.x * 1e-9f;
  bf16x8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * threadIdx.x); b[j] = (__bf16)(0.002f); }
  float fa = 0.001f * threadIdx.x, fb = 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (which == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
  }
  float s = 0; for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double randn() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

int main() {
  for (int ksteps : {1, 18, 288}) {                       // K = 16, 288 (= 9 * 32 channels), 4608 (= 9 * 512)
    const int K = 16 * ksteps;
    std::vector<float> A(32 * K), B(K * 32); std::vector<double> ref(1024, 0.0);
    srand(ksteps);
    for (auto& v : A) v = (float)(fabs(randn()));         // activations: non-negative (post-ReLU)
    for (auto& v : B) v = (float)(randn() * 0.02);
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)B[k * 32 + n]; ref[m * 32 + n] = s; }
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 8; ++mode) {
      const float wscale = mode >= 6 ? 64.0f : 1.0f;          // modes 6 / 7 = modes 4 / 5 with the weights pre-scaled by 2^6 (exact)
      gemm_probe<<<1, 64>>>(dA, dB, dD, ksteps, mode >= 6 ? mode - 2 : mode, 1, wscale);
      float D[1024]; hipMemcpy(D, dD, 4096, hipMemcpyDeviceToHost);
      double num = 0, den = 0, mx = 0;
      for (int i = 0; i < 1024; ++i) { double e = D[i] - ref[i]; num += e * e; den += ref[i] * ref[i]; mx = fmax(mx, fabs(e)); }
      printf("K=%5d mode %d (%s): rel L2 err %.3e  max abs err %.3e\n", K, mode, mode == 0 ? "fp32 mfma      " : mode == 1 ? "bf16 x 6 prod  " : mode == 2 ? "bf16 x 3 prod  " : mode == 3 ? "bf16 x 6, split" : mode == 4 ? "fp16 2pl 3 prod " : mode == 5 ? "fp16 2pl 4 prod " : mode == 6 ? "fp16 3 prod w*64" : "fp16 4 prod w*64", sqrt(num / den), mx);
    }
  }
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    rate_probe<<<1024, 256>>>(out, 100, which); hipDeviceSynchronize();
    hipEventRecord(e0); rate_probe<<<1024, 256>>>(out, iters, which); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 1024.0 * 4 * iters * 4 * (which == 0 ? 4096.0 : 32768.0);
    printf("rate %s: %.1f TFLOP/s (%.3f ms)\n", which == 0 ? "f32 32x32x2 " : "bf16 32x32x16", flops / ms / 1e9, ms);
  }
  return 0;
}
