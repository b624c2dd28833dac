// v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950: which rows of 16 lanes of the two operands change places, and the
// 4 x 4 (register, lane-row) transposition built from them (diagnostic, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/permlane_probe && tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o, int which) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  if (which == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[threadIdx.x + 64] = r[1];
  } else {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[threadIdx.x + 64] = r[1];
  }
}
// X[r][q] = 10 r + q in register r, lane-row q  ->  X[q][r] expected
__global__ void kt(unsigned* o) {
  const unsigned q = threadIdx.x >> 4;
  unsigned r0 = q, r1 = 10 + q, r2 = 20 + q, r3 = 30 + q;
  const auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  o[threadIdx.x] = c[0];
  o[64 + threadIdx.x] = c[1];
  o[128 + threadIdx.x] = d[0];
  o[192 + threadIdx.x] = d[1];
}
int main() {
  unsigned* d;
  unsigned h[256];
  (void)hipMalloc(&d, sizeof(h));
  for (int which = 16; which <= 32; which += 16) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, which);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
      printf("permlane%d_swap result %d (operand %s = %d + lane):", which, w, w ? "b" : "a", w ? 200 : 100);
      for (int row = 0; row < 4; ++row) printf("  row%d: %u..%u", row, h[64 * w + 16 * row], h[64 * w + 16 * row + 15]);
      printf("\n");
    }
  }
  hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r) {
    printf("transposed register %d, lane-rows 0..3 (10 r + q expected as 10 q + r):", r);
    for (int q = 0; q < 4; ++q) printf(" %2u", h[64 * r + 16 * q]);
    printf("\n");
  }
  return 0;
}
