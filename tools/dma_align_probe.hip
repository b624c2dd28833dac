// Probe: does global_load_lds_dwordx4 accept a global source that is only 8-byte aligned?  (vsm_raman_quad.hip stages N x N FP64
// blocks whose byte offset is a multiple of 8 N^2.)
// Build: hipcc --offload-arch=gfx950 -O2 tools/dma_align_probe.hip -o tools/dma_align_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const double* g, double* out, int shift) {
  __shared__ __attribute__((aligned(16))) double L[128];
  const int lane = threadIdx.x;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + shift + 2 * lane),
                                   (__attribute__((address_space(3))) void*)L, 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  out[lane] = L[lane];
  out[64 + lane] = L[64 + lane];
}
int main() {
  double *g, *o;
  hipMalloc(&g, 4096);
  hipMalloc(&o, 4096);
  double h[512];
  for (int i = 0; i < 512; ++i) h[i] = i;
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, shift);
    double r[128];
    hipError_t e = hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; ++i) bad += r[i] != shift + i;
    printf("shift %d doubles (%d bytes): %s, %d mismatches (r[0..3] = %g %g %g %g)\n", shift, 8 * shift, hipGetErrorString(e), bad, r[0], r[1], r[2], r[3]);
  }
  return 0;
}
