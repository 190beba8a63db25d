// Probe: sustained rate of the two fp16 MFMA shapes under the power cap, with data that looks like the engine's (random fp16 operands, half of one
// operand zero like post-ReLU activations) and with all-zero data.  Every wave keeps NACC independent accumulators and issues MFMAs back to back from
// registers (no memory in the loop), 2 waves per SIMD on every CU, ~1.5 s per case.  Prints TFLOP/s; sample sclk / power with rocm-smi beside it.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/bin/mfma_power_probe tools/probe/mfma_power_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const _Float16* src, float* out, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = src[(threadIdx.x * 4 + i) * 8 + j]; b[i][j] = src[65536 + (threadIdx.x * 4 + i) * 8 + j]; }
  if (SHAPE == 0) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x4 acc[8] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[i & 3], acc[i], 0, 0, 0);
    }
    float s = 0; for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

int main() {
  const int n = 131072;
  _Float16* h = (_Float16*)malloc(n * 2);
  _Float16* d; float* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, 2048 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int data = 0; data < 2; ++data) {
    srand(1);
    for (int i = 0; i < n; ++i) {
      float v = data ? 0.f : ((rand() % 2001) - 1000) / 500.0f;
      if (i < 65536 && (rand() & 1)) v = 0.f;               // "A" operand: half zeros (post-ReLU)
      h[i] = (_Float16)v;
    }
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      const int iters = 2400000;
      const double flops_per_iter_wave = shape == 0 ? 16.0 * 32 * 32 * 16 * 2 : 32.0 * 16 * 16 * 32 * 2;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, d, o, iters); else hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) printf("%s data, %s: %.0f TFLOP/s (%.1f ms)\n", data ? "zero" : "random", shape == 0 ? "32x32x16" : "16x16x32", flops_per_iter_wave * iters * 512 * 4 / ms / 1e9, ms);
      }
    }
  }
  return 0;
}
