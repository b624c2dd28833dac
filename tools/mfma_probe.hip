// Micro-benchmark: issue rate of the FP64/FP32 MFMA forms and of v_fma_f64 on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k_mfma_f64(int iters, double* out) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_mfma_f64_4x4(int iters, double* out) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = 0;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_mfma_f32(int iters, float* out) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k_fma_f64(int iters, double* out) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  printf("device %s CUs %d clock %d kHz\n", p.name, cus, p.clockRate);
  double* out;
  hipMalloc(&out, sizeof(double) * 1024 * 1024 * 8);
  const int iters = 20000;
  for (int wavesPerSimd = 1; wavesPerSimd <= 2; ++wavesPerSimd) {
    const int threads = 256 * wavesPerSimd;  // 4 or 8 waves per CU
    const int blocks = cus;
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f64<4>, dim3(blocks), dim3(threads), 0, 0, iters, out); });
      double n = (double)iters * 4 * (threads / 64) * blocks;
      printf("mfma_f64_16x16x4  waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz)\n", wavesPerSimd, ms,
             n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / (blocks * 4.0)));
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f64_4x4<8>, dim3(blocks), dim3(threads), 0, 0, iters, out); });
      double n = (double)iters * 8 * (threads / 64) * blocks;
      printf("mfma_f64_4x4x4_4b waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n", wavesPerSimd, ms, n * 512 / ms / 1e9);
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_f32<4>, dim3(blocks), dim3(threads), 0, 0, iters, (float*)out); });
      double n = (double)iters * 4 * (threads / 64) * blocks;
      printf("mfma_f32_16x16x4  waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n", wavesPerSimd, ms, n * 2048 / ms / 1e9);
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(k_fma_f64<16>, dim3(blocks), dim3(threads), 0, 0, iters, out); });
      double n = (double)iters * 16 * threads * blocks;
      printf("v_fma_f64         waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n", wavesPerSimd, ms, n * 2 / ms / 1e9);
    }
  }
  {
    const int threads = 1024, blocks = cus * 2;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_fma_f64<16>, dim3(blocks), dim3(threads), 0, 0, iters, out); });
    double n = (double)iters * 16 * threads * blocks;
    printf("v_fma_f64         8 waves/SIMD : %.3f ms  %.2f TFLOP/s\n", ms, n * 2 / ms / 1e9);
  }
  return 0;
}
