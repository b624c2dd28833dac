// Rotational-Raman doubling, inelastic part of one doubling step -- ONE WAVE PER RAMAN LINE (FP64, N <= 30).
//
// doubling_inelastic.jl:62-123 (the two `for dn` loops of doubling_helper!(::RRS, ...)).  Per (recipient point n1,
// line dn; donor n0 = n1 + shift[dn]) the step is ten 32 x 32 x Kend products plus eight mat-vecs:
//     X    = ier r0 + r1 ier              W1 = iet + X gt0          WA = ier + X gr0        W3 = WA t0 + r1 iet
//     iet' = ttg1 W1 + iet gt0            ier' = ier + iet grt0 + ttg1 W3          (+ the source recurrences, riding in
//                                                                                     the spare columns N, N+1)
// k_raman_doubling_lines (vsm_fused.hip) gives the four 16 x 16 output tiles of each product to four waves and needs
// ten workgroup barriers per line around 6-MFMA bursts: measured 31 % of the FP64 MFMA rate.  Here a wave owns a whole
// line: right operands and every intermediate live in registers in the accumulator layout (the FP64 MFMA's B operand
// layout IS its accumulator layout, as in the strip kernels), left operands are wave-private k-major LDS images, and no
// barrier is needed after the two shared images (r1, ttg1) are staged.  Four waves (one per SIMD) walk the in-band lines
// of one recipient point round-robin; the 24 back-to-back MFMAs of a product keep the matrix pipe busy from one wave.
//
// LDS: 2 shared + 4 x 3 private images of 32 x 34 doubles = 119 KB.
#include "vsm_common.h"
#include "vsm_internal.h"

namespace vsm {
namespace {

constexpr int WLD = 34;            // k-major row pitch: conflict-free ds_read_b64 for the (l15, kq) fragment pattern
constexpr int WIMG = 32 * WLD;     // doubles per image
constexpr int RW_WAVES = 4;
constexpr int RW_PRIV = 3;         // private images per wave: ier -> WA, iet, X

struct wmat {
  d4_t v[2][2];  // [row tile][column tile]; element r of a tile: row 16 a + kq + 4 r, column 16 b + l15
};
struct cvec {
  double x[2][4];  // rows 16 a + kq + 4 r (every lane of a kq group holds the same rows)
};
struct wpos {
  int l15, kq, lane;
};

__device__ __forceinline__ void w_zero(wmat& m) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) m.v[a][b] = acc_zero<double>();
}
// N x N column-major global block -> accumulator layout (zero outside)
__device__ __forceinline__ void w_load(wmat& m, const double* __restrict__ g, int N, const wpos& p) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * a + p.kq + 4 * r, col = 16 * b + p.l15;
        m.v[a][b][r] = (row < N && col < N) ? g[row + N * col] : 0.0;
      }
}
__device__ __forceinline__ void w_store(double* __restrict__ g, const wmat& m, int N, const wpos& p) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * a + p.kq + 4 * r, col = 16 * b + p.l15;
        if (row < N && col < N) g[row + N * col] = m.v[a][b][r];
      }
}
// accumulator layout -> k-major LDS image, columns >= N zeroed (the riders never enter a left operand)
__device__ __forceinline__ void w_to_aform(double* L, const wmat& m, int N, const wpos& p) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * a + p.kq + 4 * r, col = 16 * b + p.l15;
        L[col + WLD * row] = (col < N) ? m.v[a][b][r] : 0.0;
      }
}
// acc += A B : A from a k-major image, B from registers; k runs in the order the accumulator layout stores it
template <int KS>
__device__ __forceinline__ void w_mm(wmat& acc, const double* A, const wmat& B, const wpos& p) {
  const double* a0 = A + p.kq + WLD * p.l15;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int a = i >> 2, r = i & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double af = a0[16 * a + 4 * r + WLD * 16 * t];
#pragma unroll
      for (int b = 0; b < 2; ++b) acc.v[t][b] = mfma<double>::mma(af, B.v[a][b][r], acc.v[t][b]);
    }
  }
}
__device__ __forceinline__ void v_load(cvec& x, const double* __restrict__ g, int N, const wpos& p) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * a + p.kq + 4 * r;
      x.x[a][r] = (row < N) ? g[row] : 0.0;
    }
}
// column c of m, valid on the lanes that hold it (l15 == c & 15)
__device__ __forceinline__ cvec w_col(const wmat& m, int c) {
  cvec x;
  const bool hi = c >= 16;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) x.x[a][r] = hi ? m.v[a][1][r] : m.v[a][0][r];
  return x;
}
__device__ __forceinline__ void w_set_col(wmat& m, int c, const cvec& x, const wpos& p) {
  const bool mine = p.l15 == (c & 15), hi = c >= 16;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      m.v[a][0][r] = (mine && !hi) ? x.x[a][r] : m.v[a][0][r];
      m.v[a][1][r] = (mine && hi) ? x.x[a][r] : m.v[a][1][r];
    }
}
// copy the values held by the lanes of column c to every lane of the same kq group
__device__ __forceinline__ cvec v_bcast(const cvec& x, int c, const wpos& p) {
  cvec y;
  const int src = (p.lane & 48) | (c & 15);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) y.x[a][r] = __shfl(x.x[a][r], src, 64);
  return y;
}
// keep the columns < N
__device__ __forceinline__ void w_mask_cols(wmat& m, int N, const wpos& p) {
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const bool keep = 16 * b + p.l15 < N;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) m.v[a][b][r] = keep ? m.v[a][b][r] : 0.0;
  }
}

template <int KS>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void k_raman_doubling_wave(
    int N, int S, int K, const int* __restrict__ shift, const double* __restrict__ r, const double* __restrict__ t,
    const double* __restrict__ ttg, const double* __restrict__ gt, const double* __restrict__ gr,
    const double* __restrict__ grt, const double* __restrict__ jp, const double* __restrict__ j1m,
    const double* __restrict__ tmp1, const double* __restrict__ tmp2, const double* __restrict__ expk, double* ier,
    double* iet, double* ieJp, double* ieJm) {
  extern __shared__ __attribute__((aligned(16))) double rw_smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  wpos p;
  p.lane = tid & 63;
  p.l15 = p.lane & 15;
  p.kq = p.lane >> 4;
  double* R1a = rw_smem;
  double* TTGa = rw_smem + WIMG;
  double* IERa = rw_smem + (2 + RW_PRIV * wave) * WIMG;   // ier, later WA
  double* IETa = IERa + WIMG;
  double* Xa = IETa + WIMG;
  const int n1 = blockIdx.x;
  const int NN = N * N;
  const int cA = N, cB = N + 1;
  {
    const double* g1 = r + (long long)n1 * NN;
    const double* g2 = ttg + (long long)n1 * NN;
    for (int e = tid; e < 1024; e += 64 * RW_WAVES) {
      const int row = e & 31, col = e >> 5;
      const bool in = row < N && col < N;
      R1a[col + WLD * row] = in ? g1[row + N * col] : 0.0;
      TTGa[col + WLD * row] = in ? g2[row + N * col] : 0.0;
    }
  }
  __syncthreads();
  int cnt = 0;
  for (int d = 0; d < K; ++d) {
    const int n0 = n1 + shift[d];
    if (n0 < 0 || n0 >= S) continue;
    if ((cnt++ & (RW_WAVES - 1)) != wave) continue;
    const long long o4 = ((long long)n1 + (long long)S * d) * NN, o4v = ((long long)n1 + (long long)S * d) * N;
    const long long e4 = (long long)n0 * NN, e1 = (long long)n0 * N;
    wmat IERw, IETw, Bw;
    w_load(IERw, ier + o4, N, p);
    w_load(IETw, iet + o4, N, p);
    w_load(Bw, r + e4, N, p);
    cvec cJp, cJm, cx, cy;
    v_load(cJp, ieJp + o4v, N, p);
    v_load(cJm, ieJm + o4v, N, p);
    v_load(cx, j1m + e1, N, p);
    v_load(cy, jp + e1, N, p);
    const double e0 = expk[n0];
    cvec cJ1m;   // iej1- = iej0- expk[n0]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) cJ1m.x[a][q] = cJm.x[a][q] * e0;
    w_to_aform(IERa, IERw, N, p);
    w_to_aform(IETa, IETw, N, p);
    w_set_col(IETw, cA, cJ1m, p);
    w_set_col(IETw, cB, cJp, p);
    w_set_col(Bw, cA, cx, p);
    w_set_col(Bw, cB, cy, p);
    // X = ier r0 + r1 ier (columns N, N+1: ier j1-, ier j0+) ;  r1 iet (columns N, N+1: r1 iej1-, r1 iej0+)
    wmat X, R1IET;
    w_zero(X);
    w_mm<KS>(X, IERa, Bw, p);
    w_mm<KS>(X, R1a, IERw, p);
    w_zero(R1IET);
    w_mm<KS>(R1IET, R1a, IETw, p);
    w_to_aform(Xa, X, N, p);
    // X gt0 (column N: X tmp1), iet gt0 (column N: iet tmp1)
    v_load(cx, tmp1 + e1, N, p);
    v_load(cy, tmp2 + e1, N, p);
    w_load(Bw, gt + e4, N, p);
    w_set_col(Bw, cA, cx, p);
    wmat W1, O1;
    w_zero(W1);
    w_mm<KS>(W1, Xa, Bw, p);
    w_zero(O1);
    w_mm<KS>(O1, IETa, Bw, p);
    // X gr0 (column N: X tmp2)
    w_load(Bw, gr + e4, N, p);
    w_set_col(Bw, cA, cy, p);
    wmat WA;
    w_zero(WA);
    w_mm<KS>(WA, Xa, Bw, p);
    // ier + iet grt0 (column N: iet tmp2)
    w_load(Bw, grt + e4, N, p);
    w_set_col(Bw, cA, cy, p);
    wmat O2 = IERw;
    w_mm<KS>(O2, IETa, Bw, p);
    // W1 = iet + X gt0, column N = a3 = iej0+ + r1 iej1- + ier j1- + X tmp1
    {
      const cvec q1 = w_col(R1IET, cA), q2 = w_col(X, cA), q3 = w_col(W1, cA);
      cvec a3;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) a3.x[a][q] = cJp.x[a][q] + q1.x[a][q] + q2.x[a][q] + q3.x[a][q];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) W1.v[a][b] += IETw.v[a][b];
      w_mask_cols(W1, N, p);
      w_set_col(W1, cA, a3, p);
    }
    // WA = ier + X gr0 ;  a4 = iej1- + ier j0+ + r1 iej0+ + X tmp2
    cvec a4;
    {
      const cvec q1 = w_col(X, cB), q2 = w_col(R1IET, cB), q3 = w_col(WA, cA);
      cvec s;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) s.x[a][q] = q1.x[a][q] + q2.x[a][q];
      s = v_bcast(s, cB, p);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) a4.x[a][q] = cJ1m.x[a][q] + s.x[a][q] + q3.x[a][q];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) WA.v[a][b] += IERw.v[a][b];
    }
    w_to_aform(IERa, WA, N, p);   // ier's image is dead (the wave's own earlier reads retire in order)
    // W3 = WA t0 + r1 iet, column N = a4
    w_load(Bw, t + e4, N, p);
    w_mask_cols(R1IET, N, p);
    w_mm<KS>(R1IET, IERa, Bw, p);
    w_set_col(R1IET, cA, a4, p);
    // iet' = ttg1 W1 + iet gt0 ;  ieJ0+' = iej0+ expk0 + ttg1 a3 + iet tmp1  (column N of the same accumulator)
    w_mm<KS>(O1, TTGa, W1, p);
    // ier' = ier + iet grt0 + ttg1 W3 ;  ieJ0-' = iej0- + ttg1 a4 + iet tmp2
    w_mm<KS>(O2, TTGa, R1IET, p);
    w_store(iet + o4, O1, N, p);
    w_store(ier + o4, O2, N, p);
    {
      const cvec q1 = w_col(O1, cA), q2 = w_col(O2, cA);
      if (p.l15 == (cA & 15)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = 16 * a + p.kq + 4 * q;
            if (row < N) {
              ieJp[o4v + row] = cJp.x[a][q] * e0 + q1.x[a][q];
              ieJm[o4v + row] = cJm.x[a][q] + q2.x[a][q];
            }
          }
      }
    }
  }
}

template <int KS>
int launch_rw(int N, int S, int K, const int* shift, const double* r, const double* t, const double* ttg, const double* gt,
              const double* gr, const double* grt, const double* jp, const double* j1m, const double* tmp1,
              const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp, double* ieJm, hipStream_t st) {
  auto kern = k_raman_doubling_wave<KS>;
  const size_t bytes = (size_t)(2 + RW_PRIV * RW_WAVES) * WIMG * sizeof(double);
  static hipError_t prepared =
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (prepared != hipSuccess) return hip_fail(prepared, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  hipLaunchKernelGGL(kern, dim3(S), dim3(64 * RW_WAVES), bytes, st, N, S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1,
                     tmp2, expk, ier, iet, ieJp, ieJm);
  VSM_LAUNCH_CHECK("k_raman_doubling_wave");
  return VSM_OK;
}

}  // namespace

// FP64, N <= 30; VSM_ERR_UNSUPPORTED otherwise (the caller falls back to k_raman_doubling_lines / the operator chain)
int raman_doubling_wave(int N, int S, int K, const int* shift, const double* r, const double* t, const double* ttg,
                        const double* gt, const double* gr, const double* grt, const double* jp, const double* j1m,
                        const double* tmp1, const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp,
                        double* ieJm, hipStream_t st) {
  static const bool off = getenv("VSM_NO_RAMAN_WAVE") != nullptr;
  if (off || N > 30 || N < 1) return VSM_ERR_UNSUPPORTED;
  if (S <= 0 || K <= 0) return VSM_OK;
#define RW_CASE(KS_)                                                                                                 \
  case KS_:                                                                                                          \
    return launch_rw<KS_>(N, S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm, st)
  switch ((N + 3) >> 2) {
    RW_CASE(1);
    RW_CASE(2);
    RW_CASE(3);
    RW_CASE(4);
    RW_CASE(5);
    RW_CASE(6);
    RW_CASE(7);
    RW_CASE(8);
  }
#undef RW_CASE
  return VSM_ERR_UNSUPPORTED;
}

}  // namespace vsm
