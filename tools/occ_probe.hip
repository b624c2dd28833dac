// Residency census on gfx950: how many 6-wave workgroups (168 VGPRs, ~80 KB LDS) does a CU really hold, and on which SIMDs
// do their waves land?  Every workgroup bumps a per-CU counter, spins, records the maximum it saw.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
template <int NV, int NT>
__global__ __launch_bounds__(NT) void k(int* cnt, int* maxres, int* simd_hist, long long spin) {
  extern __shared__ float s[];
  // force the register allocation up to NV VGPRs
  float x[NV - 24];
#pragma unroll
  for (int i = 0; i < NV - 24; ++i) x[i] = s[(threadIdx.x + i) & 1023] + i;
  const unsigned hw = __builtin_amdgcn_s_getreg(0xF804), xcc = __builtin_amdgcn_s_getreg(0xF814) & 15;
  const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, simd = (hw >> 4) & 3;
  const int cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
  if ((threadIdx.x & 63) == 0) atomicAdd(&simd_hist[(blockIdx.x & 1023) * 4 + simd], 1);
  __shared__ int now;
  if (threadIdx.x == 0) now = atomicAdd(&cnt[cuid], 1) + 1;
  __syncthreads();
  long long t0 = clock64();
  float acc = 0;
  while (clock64() - t0 < spin) {
#pragma unroll
    for (int i = 0; i < NV - 24; ++i) acc += x[i] * 1.0001f;
  }
  if (threadIdx.x == 0) {
    int c = atomicAdd(&cnt[cuid], 0);
    atomicMax(&maxres[cuid], c > now ? c : now);
    atomicSub(&cnt[cuid], 1);
  }
  if (acc == 12345.f) s[0] = acc;
}
template <int NV, int NT>
void run(int lds) {
  int *cnt, *mx, *sh;
  hipMalloc(&cnt, 4096 * 4); hipMalloc(&mx, 4096 * 4); hipMalloc(&sh, 4096 * 4);
  hipMemset(cnt, 0, 4096 * 4); hipMemset(mx, 0, 4096 * 4); hipMemset(sh, 0, 4096 * 4);
  (void)hipFuncSetAttribute((const void*)k<NV, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
  hipLaunchKernelGGL((k<NV, NT>), dim3(1024), dim3(NT), lds, 0, cnt, mx, sh, 2000000LL);
  hipEventRecord(e1);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  int h[4096], hs[4096];
  hipMemcpy(h, mx, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(hs, sh, sizeof(hs), hipMemcpyDeviceToHost);
  int hist[8] = {0}, ncu = 0;
  for (int i = 0; i < 4096; ++i) if (h[i] > 0) { ncu++; hist[h[i] < 7 ? h[i] : 7]++; }
  int pat[5][5] = {{0}};
  for (int b = 0; b < 1024; ++b) { int mxs = 0, mns = 9; for (int j = 0; j < 4; ++j) { if (hs[b*4+j] > mxs) mxs = hs[b*4+j]; if (hs[b*4+j] < mns) mns = hs[b*4+j]; } pat[mns][mxs]++; }
  printf("threads %d %.2f ms VGPR~%d lds %d: %s; CUs seen %d; max resident WGs per CU histogram: 1:%d 2:%d 3:%d 4:%d ; waves per SIMD of a WG (min,max): (1,2):%d (0,2):%d (0,3):%d other:%d\n",
         NT, ms, NV, lds, hipGetErrorString(e), ncu, hist[1], hist[2], hist[3], hist[4], pat[1][2], pat[0][2], pat[0][3], 1024 - pat[1][2] - pat[0][2] - pat[0][3]);
  hipFree(cnt); hipFree(mx); hipFree(sh);
}
int main() {
  run<160, 384>(81860);
  run<120, 384>(40000);
  run<120, 256>(40000);
  run<250, 256>(81860);
  run<160, 256>(81860);
  run<80, 384>(40000);
  run<80, 512>(40000);
  run<80, 768>(81860);
  return 0;
}
