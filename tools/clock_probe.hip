// Calibrates the units of the kernels' cycle stamps: s_memtime (__builtin_readcyclecounter) against s_memrealtime (the constant
// 100 MHz reference counter) and the wall clock, around a known number of back-to-back f64 MFMAs (64 cycles each at the nominal
// rate).  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/clock_probe.hip -o tools/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(int n, unsigned long long* out, double* sink) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  const double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  unsigned long long* out;
  double* sink;
  hipMalloc(&out, 16);
  hipMalloc(&sink, sizeof(double) * 256 * p.multiProcessorCount * 2);
  for (int rep = 0; rep < 3; ++rep)
    for (int n : {20000, 2000000}) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(p.multiProcessorCount), dim3(256), 0, 0, n, out, sink);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2];
      hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
      const double mf = 4.0 * n;
      printf("n=%8d MFMAs/wave=%.0f: wall %.3f ms; s_memtime %llu ticks (%.2f per MFMA, %.1f MHz by memrealtime@100MHz, %.1f MHz by wall); "
             "memrealtime %llu ticks = %.3f ms; nominal 64 cycles/MFMA -> clock %.0f MHz by wall\n",
             n, mf, ms, h[0], h[0] / mf, h[0] / (h[1] / 100.0), h[0] / (ms * 1e3), h[1], h[1] / 1e5, mf * 64 / (ms * 1e3));
    }
  return 0;
}
