// Probe 2: does the f64 MFMA issue rate depend on (a) VGPR-form vs AGPR-form accumulators,
// (b) fresh A/B operand registers per MFMA, (c) operands coming from LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_inv(int iters, double* out) {   // loop-invariant operands
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
__global__ void k_var(int iters, double* out) {   // distinct operand registers, changing every iteration
  d4 acc[NACC];
  double a[NACC], b[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = d4{0, 0, 0, 0}; a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + i * 1e-6; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] += 1e-9; b[i] -= 1e-9; }
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// 2 A-frags x 2 B-frags from LDS per k-step, 4 MFMAs (the real kernel's shape)
__global__ void k_lds(int iters, double* out) {
  __shared__ double sa[64 * 64], sb[64 * 64];
  for (int e = threadIdx.x; e < 4096; e += blockDim.x) { sa[e] = e * 1e-6; sb[e] = 1.0 - e * 1e-7; }
  __syncthreads();
  d4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    const int k = (it & 15) * 4 + (lane >> 4);
    double a0 = sa[(lane & 15) + 64 * k], a1 = sa[16 + (lane & 15) + 64 * k];
    double b0 = sb[k + 64 * (lane & 15)], b1 = sb[k + 64 * (16 + (lane & 15))];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * 1024 * 1024 * 8);
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    const int threads = 256 * w, blocks = cus;
    auto rep = [&](const char* name, double ms, double nmf) {
      printf("%-28s waves/SIMD=%d: %.3f ms %.2f TFLOP/s  %.1f cycles/MFMA/SIMD@2.4GHz\n", name, w, ms, nmf * 2048 / ms / 1e9,
             ms * 1e-3 * 2.4e9 / (nmf / (blocks * 4.0)));
    };
    rep("invariant operands x4", time_ms([&] { hipLaunchKernelGGL(k_inv<4>, dim3(blocks), dim3(threads), 0, 0, iters, out); }), (double)iters * 4 * (threads / 64) * blocks);
    rep("varying operands x4", time_ms([&] { hipLaunchKernelGGL(k_var<4>, dim3(blocks), dim3(threads), 0, 0, iters, out); }), (double)iters * 4 * (threads / 64) * blocks);
    rep("varying operands x2", time_ms([&] { hipLaunchKernelGGL(k_var<2>, dim3(blocks), dim3(threads), 0, 0, iters, out); }), (double)iters * 2 * (threads / 64) * blocks);
    rep("LDS operands 2x2", time_ms([&] { hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(threads), 0, 0, iters, out); }), (double)iters * 4 * (threads / 64) * blocks);
  }
  return 0;
}
