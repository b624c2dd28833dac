// Linearized doubling step, column-strip form (FP64, 32 < N <= 60): ONE launch per doubling step does the forward
// recurrences AND the derivative recurrences of every active parameter of doubling_allparams_helper!
// (doubling_lin.jl:216-339) for one spectral point per workgroup, instead of ~40 batched-operator launches that
// each stream three [N,N,S] arrays through HBM.
//
// Same layout as vsm_strip.hip (strips in MFMA accumulator registers = B operands, A operands in LDS A-form), with
// FOUR A-form buffers (128 KB: one workgroup per CU, up to 512 VGPRs per lane):
//
//     BR = r           BT = t -> tt = t G           BX = rdot_p -> ttdot_p           BY = series scratch, tdot_p -> Y_p
//
//   forward:   E = r r ; G = (I - E)^-1 ; tt = t G ; rt = r t                      (A: BR, BT;  B: r_s, G_s, t_s)
//   per parameter p (rdot = ap_r-+, tdot = ap_t++), with Gdot = G X1 G eliminated (t Gdot = tt X1 G):
//     X1 = rdot r + r rdot ;  Q2 = rdot t + r tdot                                 (A: BX, BR shared by both products)
//     Y  = tdot + tt X1 ;  ttdot = Y G                                             (A: BT ; BY)
//     rdot' = rdot + ttdot rt + tt Q2 ;  tdot' = ttdot t + tt tdot                 (A: BX, BT shared by both products)
//   forward:   r' = r + tt rt ;  t' = tt t                                         (A: BT shared)
//   = 5 products + inverse, + 10 products per parameter (the reference: 12 per parameter).
//
// All 4 + 8 P matrix-vector products of the source recurrences ride in the two spare columns (Kend, Kend+1) of the B
// strips of these products:
//     rt[:,c1] = r j0+ (+ j1-) = A      rt[:,c2] = r j1- (+ j0+) = B           j1+- = j0+- expk
//     X1[:,c1] = rdot j0+ + r aJ+       X1[:,c2] = rdot j1- + r aJ1-           aJ1+- = aJ+- expk + j0+- ekl_p
//     rdot'[:,c1] = aJ- + ttdot A + tt v = aJ-'      rdot'[:,c2] = aJ1+ + ttdot B + tt u = aJ+'
//     r'[:,c1] = j0- + tt A = j0-'                   r'[:,c2] = j1+ + tt B = j0+'
// State (r, t, j, rdot, tdot, aJ, expk, ekl) lives in HBM between steps: 115 KB (1 + P) per point and step against
// (12 + 20 P) N^3 flop -- far above the ridge.
#include <stdlib.h>

#include "vsm_internal.h"
#include "vsm_strip_dev.h"

namespace vsm {
namespace {

struct lsmem {
  double BR[SNP * SNP];
  double BT[SNP * SNP];
  double BX[SNP * SNP];
  double BY[SNP * SNP];
  double vec[4][SNP];   // j0+, j0-, aJ+_p, aJ-_p
  float red[2][4];
  gj_scratch<double, SNP> gj;
};

// spare-column access: lanes of the owning wave with col == c1 (A) / c1 + 1 (B)
struct spare {
  bool own, A, B, AB;
  __device__ __forceinline__ spare(const spos& p, int c1) {
    own = (p.wave == (c1 >> 4));
    A = own && (p.col == c1);
    B = own && (p.col == c1 + 1);
    AB = A || B;
  }
  // x[:, c1] = fa(row, old), x[:, c1+1] = fb(row, old)
  template <typename FA, typename FB>
  __device__ __forceinline__ void put(sstrip& x, const spos& p, FA fa, FB fb) const {
    if (own) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          const double old = x.v[ta][r];
          const double va = fa(row, old), vb = fb(row, old);
          x.v[ta][r] = A ? va : (B ? vb : old);
        }
    }
  }
  // dstA[row] = x[:, c1], dstB[row] = x[:, c1+1]   (global vectors of length N)
  __device__ __forceinline__ void get(const sstrip& x, const spos& p, int N, double* dstA, double* dstB) const {
    if (AB) {
      double* d = A ? dstA : dstB;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          if (row < N) d[row] = x.v[ta][r];
        }
    }
  }
};

template <int KS>
__global__ __launch_bounds__(SNT, 1) void k_dbl_lin_step(int N, int S, int P, double* __restrict__ expk,
                                                         double* __restrict__ ekl, added<double> a, added_lin<double> al) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lsmem& sm = *reinterpret_cast<lsmem*>(smem_raw);
  double* BR = sm.BR;
  double* BT = sm.BT;
  double* BX = sm.BX;
  double* BY = sm.BY;
  double* jp = sm.vec[0];
  double* jm = sm.vec[1];
  double* ajp = sm.vec[2];
  double* ajm = sm.vec[3];
  spos p;
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  const int Kend = ((N + 3) >> 2) << 2;
  const spare sp(p, Kend);
  const double k = expk[s];
  double* g_r = a.r_mp + (long long)s * NN;
  double* g_t = a.t_pp + (long long)s * NN;
  auto keepN = [N](double x, int r, int c) { return (r < N && c < N) ? x : 0.0; };

  stage_aform(BR, g_r, N, p);
  stage_aform(BT, g_t, N, p);
  if (tid < SNP) {
    jp[tid] = (tid < N) ? a.j0_p[(long long)s * N + tid] : 0.0;
    jm[tid] = (tid < N) ? a.j0_m[(long long)s * N + tid] : 0.0;
  }
  __syncthreads();
  sstrip r_s, t_s, G, rt;
  load_strip(r_s, BR, p);
  load_strip(t_s, BT, p);
  int slot = 0;
  {
    sstrip E;
    E.zero();
    mm_ab<KS>(E, BR, r_s, p);
    invert_strip<KS>(E, G, BY, N, sm, slot, p, 0);
  }
  sp.put(t_s, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
  {
    sstrip tt;
    tt.zero();
    mm_ab<KS>(tt, BT, G, p);
    rt.zero();
    mm_ab<KS>(rt, BR, t_s, p);   // r t  (+ r j0+, r j1-)
    __syncthreads();             // BT (t) and BY (series powers) no longer read
    store_strip(BT, tt, p, keepN);
  }
  sp.put(rt, p, [&](int row, double o) { return jm[row] * k + o; }, [&](int row, double o) { return jp[row] + o; });
  sp.put(r_s, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
  __syncthreads();   // tt complete in BT

  for (int pp = 0; pp < P; ++pp) {
    const double kl = ekl[s + (long long)S * pp];
    double* g_ar = al.ap_r_mp + (long long)pp * MS + (long long)s * NN;
    double* g_at = al.ap_t_pp + (long long)pp * MS + (long long)s * NN;
    double* g_ajp = al.ap_J0_p + (long long)pp * VS + (long long)s * N;
    double* g_ajm = al.ap_J0_m + (long long)pp * VS + (long long)s * N;
    stage_aform(BX, g_ar, N, p);
    stage_aform(BY, g_at, N, p);
    if (tid < SNP) {
      ajp[tid] = (tid < N) ? g_ajp[tid] : 0.0;
      ajm[tid] = (tid < N) ? g_ajm[tid] : 0.0;
    }
    __syncthreads();
    sstrip rd, td;
    load_strip(rd, BX, p);
    load_strip(td, BY, p);
    sp.put(rd, p, [&](int row, double) { return ajp[row]; }, [&](int row, double) { return ajm[row] * k + jm[row] * kl; });
    sstrip X1, Q2;
    X1.zero();
    Q2.zero();
    mm_ab2<KS>(X1, Q2, BX, r_s, t_s, p);   // rdot r (+ rdot j0+, rdot j1-) ; rdot t
    mm_ab2<KS>(X1, Q2, BR, rd, td, p);     // + r rdot (+ r aJ+, r aJ1-)   ; + r tdot
    // v = aJ1- + rdot j0+ + r aJ+ ; u = aJ+ + rdot j1- + r aJ1-   -> spare columns of Q2
    if (sp.own) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          const double x = X1.v[ta][r];
          const double v = ajm[row] * k + jm[row] * kl + x, u = ajp[row] + x;
          Q2.v[ta][r] = sp.A ? v : (sp.B ? u : Q2.v[ta][r]);
        }
    }
    {
      sstrip Y = td;
      mm_ab<KS>(Y, BT, X1, p);   // Y = tdot + tt X1
      __syncthreads();           // every wave has its tdot strip (BY) and is done with rdot's A-form (BX)
      store_strip(BY, Y, p, keepN);
    }
    __syncthreads();   // Y complete in BY
    {
      sstrip ttl;
      ttl.zero();
      mm_ab<KS>(ttl, BY, G, p);   // ttdot = Y G
      store_strip(BX, ttl, p, keepN);
    }
    sp.put(rd, p, [&](int row, double) { return ajm[row]; }, [&](int row, double) { return ajp[row] * k + jp[row] * kl; });
    __syncthreads();   // ttdot complete in BX
    sstrip tdn;
    tdn.zero();
    mm_ab2<KS>(rd, tdn, BX, rt, t_s, p);   // rdot += ttdot rt (+ ttdot A, ttdot B) ; tdot' = ttdot t
    mm_ab2<KS>(rd, tdn, BT, Q2, td, p);    // rdot += tt Q2 (+ tt v, tt u)          ; tdot' += tt tdot
    store_strip_global(g_ar, rd, N, p);
    store_strip_global(g_at, tdn, N, p);
    sp.get(rd, p, N, g_ajm, g_ajp);
    if (tid == 0) ekl[s + (long long)S * pp] = 2.0 * k * kl;
    __syncthreads();   // BX, BY, aJ+- free for the next parameter
  }

  // forward update: r' = r + tt rt (+ tt A, tt B on top of j0-, j1+) ; t' = tt t
  sp.put(r_s, p, [&](int row, double) { return jm[row]; }, [&](int row, double) { return jp[row] * k; });
  sstrip tn;
  tn.zero();
  mm_ab2<KS>(r_s, tn, BT, rt, t_s, p);
  store_strip_global(g_r, r_s, N, p);
  store_strip_global(g_t, tn, N, p);
  sp.get(r_s, p, N, a.j0_m + (long long)s * N, a.j0_p + (long long)s * N);
  if (tid == 0) expk[s] = k * k;
}

}  // namespace

#define VSM_CAT2(a, b) a##b
#define VSM_CAT(a, b) VSM_CAT2(a, b)
#define VSM_STRIPLIN_DECL(KS) \
  int VSM_CAT(launch_dbl_lin_step_, KS)(int, int, int, double*, double*, const added<double>&, const added_lin<double>&, hipStream_t);

#ifdef VSM_STRIP_KS
VSM_STRIPLIN_DECL(VSM_STRIP_KS)
int VSM_CAT(launch_dbl_lin_step_, VSM_STRIP_KS)(int N, int S, int P, double* expk, double* ekl, const added<double>& a,
                                                const added_lin<double>& al, hipStream_t st) {
  static int prepared = [] {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dbl_lin_step<VSM_STRIP_KS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(lsmem));
    return e == hipSuccess ? (int)VSM_OK : hip_fail(e, "hipFuncSetAttribute(k_dbl_lin_step)");
  }();
  if (prepared) return prepared;
  hipLaunchKernelGGL(k_dbl_lin_step<VSM_STRIP_KS>, dim3(S), dim3(SNT), sizeof(lsmem), st, N, S, P, expk, ekl, a, al);
  VSM_LAUNCH_CHECK("k_dbl_lin_step");
  return VSM_OK;
}

#else  // ---- dispatcher object -----------------------------------------------------------------------------------------

VSM_STRIPLIN_DECL(9)
VSM_STRIPLIN_DECL(10)
VSM_STRIPLIN_DECL(11)
VSM_STRIPLIN_DECL(12)
VSM_STRIPLIN_DECL(13)
VSM_STRIPLIN_DECL(14)
VSM_STRIPLIN_DECL(15)

// One fused doubling step (forward + P parameters); VSM_ERR_UNSUPPORTED outside 32 < N <= 60 or when the added layer is
// not the plain [N,N,S] layout.
int strip_doubling_lin_step(int N, int S, int P, double* expk, double* ekl, const added<double>& a,
                            const added_lin<double>& al, hipStream_t st) {
  static const bool off = getenv("VSM_NO_STRIP_LIN") != nullptr;
  if (off || !strip_supported(N) || a.mat_stride != (long long)N * N || al.mat_stride != (long long)N * N)
    return VSM_ERR_UNSUPPORTED;
  switch ((N + 3) / 4) {
#define VSM_CASE(KS) \
  case KS:           \
    return VSM_CAT(launch_dbl_lin_step_, KS)(N, S, P, expk, ekl, a, al, st);
    VSM_CASE(9)
    VSM_CASE(10)
    VSM_CASE(11)
    VSM_CASE(12)
    VSM_CASE(13)
    VSM_CASE(14)
    VSM_CASE(15)
#undef VSM_CASE
    default:
      return VSM_ERR_UNSUPPORTED;
  }
}

#endif

}  // namespace vsm
