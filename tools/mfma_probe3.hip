// Probe 3: the f64 MFMA issue rate of the strip kernels' product loop -- per k-step four A fragments from LDS (conflict-free
// column-major image, one ds_read_b64 per MFMA) and the B operand from registers -- against the same loop fed by ds_read_b128
// (k-major image: one read serves two k-steps of a row tile), both software-pipelined by one step, 4 waves per workgroup,
// 1 or 2 workgroups per CU.   Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe3.hip -o tools/mfma_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int KS = 16;   // k-steps per product (N = 64)

// column-major 64 x 64 image, row index XOR-swizzled by the column like vsm_lds.h lidx<64>
__device__ __forceinline__ int lidx(int a, int b) { return (a ^ (((b & 1) << 4) | (((b >> 1) & 7) << 1))) + 64 * b; }

__global__ __launch_bounds__(256, 2) void k_b64(int products, double* out) {
  extern __shared__ double sa[];
  for (int e = threadIdx.x; e < 4096; e += 256) sa[e] = 1e-6 * e;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  d4 acc[4], bs[4];
  for (int i = 0; i < 4; ++i) { acc[i] = d4{0, 0, 0, 0}; bs[i] = d4{1.0 + lane, 2.0, 3.0, 4.0 + i}; }
  for (int pr = 0; pr < products; ++pr) {
    double a[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[0][t] = sa[lidx(16 * t + l15, kq)];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) a[(ks + 1) & 1][t] = sa[lidx(16 * t + l15, 4 * (ks + 1) + kq)];
      }
      const double b = bs[ks >> 2][ks & 3];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks & 1][t], b, acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) bs[t] = acc[t];   // the result is the next product's B operand, like the doubling chain
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// k-major image: row r holds its 64 k-values contiguously (leading dimension 66 doubles: the four kq groups of a b128 read fall
// on distinct bank sets); lane (row, kq) reads k = 16 j + 4 kq + {0,1} and {2,3}: two b128 per four k-steps and row tile
__global__ __launch_bounds__(256, 2) void k_b128(int products, double* out) {
  extern __shared__ double sa[];
  constexpr int LD = 66;
  for (int e = threadIdx.x; e < 64 * LD; e += 256) sa[e] = 1e-6 * e;
  __syncthreads();
  const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
  d4 acc[4], bs[4];
  for (int i = 0; i < 4; ++i) { acc[i] = d4{0, 0, 0, 0}; bs[i] = d4{1.0 + lane, 2.0, 3.0, 4.0 + i}; }
  for (int pr = 0; pr < products; ++pr) {
    d2 a[2][4];   // fragment pairs of two consecutive k-steps
#pragma unroll
    for (int t = 0; t < 4; ++t) a[0][t] = *reinterpret_cast<const d2*>(&sa[(16 * t + l15) * LD + 4 * kq]);
#pragma unroll
    for (int kp = 0; kp < KS / 2; ++kp) {
      if (kp + 1 < KS / 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          a[(kp + 1) & 1][t] = *reinterpret_cast<const d2*>(&sa[(16 * t + l15) * LD + 16 * ((kp + 1) >> 1) + 4 * kq + 2 * ((kp + 1) & 1)]);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ks = 2 * kp + h;
        const double b = bs[ks >> 2][ks & 3];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kp & 1][t][h], b, acc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) bs[t] = acc[t];
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
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
  double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 4);
  const int products = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_b64), hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_b128), hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024);
  for (int w = 1; w <= 2; ++w) {
    const int blocks = cus * w;
    const double nmf = (double)products * KS * 4 * 4 * blocks;
    // 73 KB of LDS per workgroup, as the layer kernel: at most two workgroups per CU
    double ms = time_ms([&] { hipLaunchKernelGGL(k_b64, dim3(blocks), dim3(256), 73 * 1024, 0, products, out); });
    printf("ds_read_b64  per MFMA        workgroups/CU=%d: %.3f ms %.2f TFLOP/s (%.3f of 78.6)\n", w, ms, nmf * 2048 / ms / 1e9, nmf * 2048 / ms / 1e9 / 78.6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_b128, dim3(blocks), dim3(256), 73 * 1024, 0, products, out); });
    printf("ds_read_b128 per two MFMAs   workgroups/CU=%d: %.3f ms %.2f TFLOP/s (%.3f of 78.6)\n", w, ms, nmf * 2048 / ms / 1e9, nmf * 2048 / ms / 1e9 / 78.6);
  }
  return 0;
}
