// Probe of the three gfx950 facts the bf16 kernels rely on (run once on the box, all reported 0 mismatches):
//   ds_read_b64_tr_b16 lane/element mapping, the v_mfma_f32_32x32x16_bf16 operand layout, v_cvt_pk_bf16_f32 = round-to-nearest-even.
// build: hipcc --offload-arch=gfx950 -O2 -o bf16_probe tools/probe/bf16_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 v = __builtin_convertvector((f32x2){a, b}, bf16x2);
  return __builtin_bit_cast(unsigned, v);
}

__global__ void probe_tr(const unsigned short* img, int ps_bytes, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = img[i];
  __syncthreads();
  const int l = threadIdx.x, i = l & 15, g = l >> 4;
  // group g: lanes 0-15 ch 0-15 half0, 16-31 ch 16-31 half 0, 32-47 ch 0-15 half1, 48-63 ch 16-31 half 1
  const int half = g >> 1, cb = (g & 1) * 16;
  for (int s = 0; s < 2; ++s) {
    const int pixel = half * 8 + s * 4 + (i >> 2);
    const int ch = cb + (i & 3) * 4;
    __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + pixel * ps_bytes + ch * 2);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
    for (int j = 0; j < 4; ++j) out[(l * 2 + s) * 4 + j] = (unsigned short)v[j];
  }
}

// D = A(32x16) * B(16x32): A[m][k], B[k][n] with lane (m|n = l%32, k = (l/32)*8 + j)
__global__ void probe_mfma(const unsigned short* A, const unsigned short* B, float* D) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = __builtin_bit_cast(__bf16, A[r * 16 + h * 8 + j]);
    b[j] = __builtin_bit_cast(__bf16, B[(h * 8 + j) * 32 + r]);
  }
  f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 16; ++i) { int row = (i & 3) + 8 * (i >> 2) + 4 * h; D[row * 32 + r] = acc[i]; }
}
__global__ void probe_cvt(const float* x, unsigned* out) { out[threadIdx.x] = pack2(x[2 * threadIdx.x], x[2 * threadIdx.x + 1]); }

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return u >> 16; }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  unsigned short h_img[4096]; for (int i = 0; i < 4096; ++i) h_img[i] = 0xFFFF;
  const int ps = 64;  // bytes per pixel = 32 channels
  for (int p = 0; p < 16; ++p) for (int c = 0; c < 32; ++c) h_img[p * 32 + c] = (unsigned short)(p * 32 + c);   // raw u16 payload = index
  unsigned short *d_img, *d_out; hipMalloc(&d_img, 8192); hipMalloc(&d_out, 64 * 8 * 2);
  hipMemcpy(d_img, h_img, 8192, hipMemcpyHostToDevice);
  probe_tr<<<1, 64>>>(d_img, ps, d_out);
  unsigned short h_out[512]; hipMemcpy(h_out, d_out, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    int i = l & 15, g = l >> 4, half = g >> 1, cb = (g & 1) * 16;
    printf("lane %2d:", l);
    for (int s = 0; s < 2; ++s) for (int j = 0; j < 4; ++j) {
      int v = h_out[(l * 2 + s) * 4 + j]; int exp = (half * 8 + s * 4 + j) * 32 + cb + i;
      printf(" p%dc%d", v / 32, v % 32); if (v != exp) ++bad;
    }
    printf("\n");
  }
  printf("TR expectation mismatches: %d\n", bad);
  // mfma
  unsigned short hA[512], hB[512]; float ref[1024] = {0}, hD[1024];
  srand(1);
  for (int i = 0; i < 512; ++i) { hA[i] = f2bf((rand() % 17 - 8) / 4.0f); hB[i] = f2bf((rand() % 13 - 6) / 2.0f); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 16; ++k) s += bf2f(hA[m * 16 + k]) * bf2f(hB[k * 32 + n]); ref[m * 32 + n] = s; }
  unsigned short *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  probe_mfma<<<1, 64>>>(dA, dB, dD); hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  int mb = 0; for (int i = 0; i < 1024; ++i) if (hD[i] != ref[i]) ++mb;
  printf("MFMA mismatches: %d\n", mb);
  float hx[128]; unsigned ho[64]; for (int i = 0; i < 128; ++i) hx[i] = (float)(rand() % 100000) / 777.0f - 50.f;
  float* dx; unsigned* dout; hipMalloc(&dx, 512); hipMalloc(&dout, 256); hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
  probe_cvt<<<1, 64>>>(dx, dout); hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
  int cb2 = 0; for (int i = 0; i < 64; ++i) { unsigned e = f2bf(hx[2 * i]) | ((unsigned)f2bf(hx[2 * i + 1]) << 16); if (e != ho[i]) ++cb2; }
  printf("CVT mismatches: %d\n", cb2);
  return 0;
}
