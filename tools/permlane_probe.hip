// v_permlane16_swap_b32 on gfx950: which rows of 16 lanes of the two operands change places (diagnostic, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o tools/permlane_probe && tools/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) {
    printf("result %d (operand %s = %d + lane):", w, w ? "b" : "a", w ? 200 : 100);
    for (int row = 0; row < 4; ++row) printf("  row%d: %u..%u", row, h[64 * w + 16 * row], h[64 * w + 16 * row + 15]);
    printf("\n");
  }
  return 0;
}
