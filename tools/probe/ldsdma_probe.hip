// Probe of gfx950's 16-byte LDS DMA (buffer_load_dwordx4 ... lds): where does lane L's 16 bytes land, what does an out-of-range lane write, does a per-wave base work?
//   hipcc --offload-arch=gfx950 -O2 -o ldsdma_probe ldsdma_probe.hip && ./ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, int n_floats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, n_floats * 4, 0x00020000);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 4096; i += 256) reinterpret_cast<float*>(smem)[i] = -7.f;
  __syncthreads();
  // lane -> a PERMUTED source piece (so that the landing position shows the lane mapping, not the source order); lanes 60..63 of wave 3 out of range
  const int src_piece = (tid * 7) % 256;
  const int voff = (wave == 3 && (tid & 63) >= 60) ? (int)0x80000000 : src_piece * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 512 + wave * 1024), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = tid; i < 4096; i += 256) y[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
  std::vector<float> hx(1024); for (int i = 0; i < 1024; ++i) hx[i] = (float)i;
  float *dx, *dy; hipMalloc(&dx, 4096); hipMalloc(&dy, 16384);
  hipMemcpy(dx, hx.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 16384, 0, dx, dy, 1024);
  std::vector<float> hy(4096); hipMemcpy(hy.data(), dy, 16384, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    const int wave = t >> 6, lane = t & 63, base = (512 + wave * 1024 + lane * 16) / 4;
    const bool oob = wave == 3 && lane >= 60;
    for (int j = 0; j < 4; ++j) { const float want = oob ? 0.f : (float)(((t * 7) % 256) * 4 + j); if (hy[base + j] != want) { if (bad < 8) printf("tid %d j %d: got %g want %g\n", t, j, hy[base + j], want); ++bad; } }
  }
  for (int i = 0; i < 128; ++i) if (hy[i] != -7.f) { ++bad; printf("pad %d touched: %g\n", i, hy[i]); }
  printf("ldsdma probe: %s (%d mismatches): lane L's 16 bytes land at base + 16 L, out-of-range lanes write zeros\n", bad ? "FAILED" : "ok", bad);
  return bad != 0;
}
