// Layout probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: which lane holds which element of A, B and D.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma44_probe.hip -o tools/mfma44_probe
// B is a one-hot over the lanes, A holds lane + 1: the non-zero lanes of D and their values name the (A lane, B lane) -> D lane map.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_probe(double* out) {
  const int lane = threadIdx.x;
  for (int hot = 0; hot < 64; ++hot) {
    const double a = lane + 1.0, b = (lane == hot) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[hot * 64 + lane] = d;
  }
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 64 * 64);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out);
  static double h[64 * 64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  // expected if A: i = lane & 3, k = (lane >> 2) & 3 ; B: j = lane & 3, k = (lane >> 2) & 3 ; D: j = lane & 3, i = (lane >> 2) & 3 (block = lane >> 4)
  int bad = 0;
  for (int hot = 0; hot < 64; ++hot) {
    const int b = hot >> 4, k = (hot >> 2) & 3, j = hot & 3;
    for (int lane = 0; lane < 64; ++lane) {
      const int lb = lane >> 4, i = (lane >> 2) & 3, lj = lane & 3;
      const double want = (lb == b && lj == j) ? (double)(16 * b + 4 * k + i + 1) : 0.0;
      if (h[hot * 64 + lane] != want) ++bad;
    }
  }
  printf("hypothesis A(i = l & 3, k = (l >> 2) & 3), B(j = l & 3, k = (l >> 2) & 3), D(j = l & 3, i = (l >> 2) & 3): %s (%d mismatches)\n",
         bad ? "WRONG" : "confirmed", bad);
  if (bad) {
    for (int hot = 0; hot < 20; ++hot) {
      printf("B one-hot at lane %2d: ", hot);
      for (int lane = 0; lane < 64; ++lane)
        if (h[hot * 64 + lane] != 0.0) printf("D[%d]=A[%d] ", lane, (int)h[hot * 64 + lane] - 1);
      printf("\n");
    }
  }
  return 0;
}
