// Rotational-Raman (RRS) pass of the CoreRT hot path, operator level.
//
//   vsm_elemental_inelastic_rrs   elemental_inelastic!(::RRS)            elemental_inelastic.jl:23-105
//   vsm_doubling_inelastic_rrs    doubling_helper!(::RRS)                doubling_inelastic.jl:13-164
//   vsm_interaction_inelastic_rrs interaction_helper!(::RRS, ::_11)      interaction_inelastic.jl:319-521
//   vsm_copy_added_to_composite_ie, vsm_postprocess_vza_ie               rt_helpers.jl:222-228, postprocessing_vza.jl:117-151
//
// The reference walks the Raman offsets dn one at a time on the host and, per offset, multiplies views
// A[:,:,n1,dn] (x) B[:,:,n0] with n0 = n1 + i_lambda1lambda0[dn].  Here ONE launch covers every
// (n1, dn) pair: k_gemm_rs resolves, per workgroup, where each operand block lives -- the 3-D elastic
// arrays at the recipient point n1 or the donor point n0, or the 4-D inelastic arrays at (n1, dn) --
// and skips pairs whose donor is out of band, so the launch count of a layer step does not depend on
// the number of Raman lines.  Inelastic arrays are [N,N,S,K] column-major like the reference's.
#include "vsm_internal.h"
#include "vsm_gemm_lds.h"
#include <stdlib.h>
#include <type_traits>

namespace vsm {

enum rs_kind { RS_N1 = 0, RS_N0 = 1, RS_4D = 2 };

template <typename T>
struct rs_op {
  const T* p;
  int kind;
  long long stride;  // elements between consecutive spectral blocks (0 = one block shared by all points)
};
template <typename T>
static rs_op<T> at_n1(const T* p, long long stride) { return rs_op<T>{p, RS_N1, stride}; }
template <typename T>
static rs_op<T> at_n0(const T* p, long long stride) { return rs_op<T>{p, RS_N0, stride}; }
template <typename T>
static rs_op<T> at_4d(const T* p, long long stride) { return rs_op<T>{p, RS_4D, stride}; }

template <typename T>
__device__ __forceinline__ const T* rs_block(const rs_op<T>& o, int n1, int n0, int dn, int S) {
  if (o.kind == RS_N1) return o.p + (long long)n1 * o.stride;
  if (o.kind == RS_N0) return o.p + (long long)n0 * o.stride;
  return o.p + ((long long)n1 + (long long)S * dn) * o.stride;
}

// C[:,:,n1,dn] = alpha * A * B + beta * D      for every in-band (n1, dn); out-of-band blocks are left
// untouched (zero_oob = 0) or zeroed (zero_oob = 1).  One wave per 16x16 output tile.
template <typename T>
__global__ __launch_bounds__(256) void k_gemm_rs(int M, int Nc, int K, int S, const int* __restrict__ shift, rs_op<T> A,
                                                 rs_op<T> B, T* C, T alpha, rs_op<T> D, T beta, int zero_oob) {
  const int n1 = blockIdx.y, dn = blockIdx.z;
  const int n0 = n1 + shift[dn];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tilesM = (M + 15) >> 4, tilesN = (Nc + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tilesM * tilesN) return;
  const int ti = tile % tilesM, tj = tile / tilesM;
  T* Cs = C + ((long long)n1 + (long long)S * dn) * ((long long)M * Nc);
  const int col = tj * 16 + (lane & 15);
  if (n0 < 0 || n0 >= S) {
    if (zero_oob) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + mfma<T>::crow(lane, r);
        if (row < M && col < Nc) Cs[row + (long long)M * col] = T(0);
      }
    }
    return;
  }
  const T* As = rs_block(A, n1, n0, dn, S);
  const T* Bs = rs_block(B, n1, n0, dn, S);
  const int ai = ti * 16 + (lane & 15);
  const int kq = lane >> 4;
  typename mfma<T>::acc_t acc = acc_zero<T>();
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + kq;
    const T a = (ai < M && k < K) ? As[ai + (long long)M * k] : T(0);
    const T b = (col < Nc && k < K) ? Bs[k + (long long)K * col] : T(0);
    acc = mfma<T>::mma(a, b, acc);
  }
  const T* Ds = D.p ? rs_block(D, n1, n0, dn, S) : nullptr;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ti * 16 + mfma<T>::crow(lane, r);
    if (row < M && col < Nc) {
      T v = alpha * acc[r];
      if (Ds) v += beta * Ds[row + (long long)M * col];
      Cs[row + (long long)M * col] = v;
    }
  }
}

// the same for 16 < M <= 128, 16 <= Nc <= 128: one workgroup per (n1, dn), operands cross the memory system once
// (vsm_gemm_lds.h)
template <typename T, int MT, int NW>
__global__ __launch_bounds__(64 * NW) void k_gemm_rs_lds(int M, int Nc, int K, int S, const int* __restrict__ shift, rs_op<T> A,
                                                         rs_op<T> B, T* C, T alpha, rs_op<T> D, T beta, int zero_oob) {
  constexpr int KC = 16;
  __shared__ __attribute__((aligned(16))) T As_lds[gemm_lds_cfg<MT, KC>::LDS_ELEMS];
  const int n1 = blockIdx.x, dn = blockIdx.y;
  const int n0 = n1 + shift[dn];
  T* Cs = C + ((long long)n1 + (long long)S * dn) * ((long long)M * Nc);
  if (n0 < 0 || n0 >= S) {   // workgroup-uniform
    if (zero_oob)
      for (int e = threadIdx.x; e < M * Nc; e += 64 * NW) Cs[e] = T(0);
    return;
  }
  gemm_lds_body<T, MT, NW, KC>(M, Nc, K, rs_block(A, n1, n0, dn, S), rs_block(B, n1, n0, dn, S), Cs,
                               D.p ? rs_block(D, n1, n0, dn, S) : nullptr, alpha, beta, T(0), As_lds);
}
template <typename T>
static bool gemm_rs_lds(int M, int Nc, int K, int S, int Kr, const int* shift, rs_op<T> A, rs_op<T> B, T* C, rs_op<T> D,
                        hipStream_t st, int zero_oob) {
  static const bool off = ab_switch("VSM_NO_GEMM_LDS");
  if (off || M <= 16 || M > 128 || Nc < 16 || Nc > 128 || K < 8 || Kr > 65535) return false;
  const dim3 grid(S, Kr);
#define VSM_GL(MT_, NW_) \
  hipLaunchKernelGGL((k_gemm_rs_lds<T, MT_, NW_>), grid, dim3(64 * NW_), 0, st, M, Nc, K, S, shift, A, B, C, T(1), D, T(1), zero_oob)
  VSM_GEMM_LDS_DISPATCH(M, Nc, VSM_GL);
#undef VSM_GL
  return true;
}

template <typename T>
static int gemm_rs(int M, int Nc, int K, int S, int Kr, const int* shift, rs_op<T> A, rs_op<T> B, T* C, rs_op<T> D,
                   hipStream_t st, int zero_oob = 0) {
  if (S <= 0 || Kr <= 0) return VSM_OK;
  if (gemm_rs_lds<T>(M, Nc, K, S, Kr, shift, A, B, C, D, st, zero_oob)) {
    VSM_LAUNCH_CHECK("k_gemm_rs_lds");
    return VSM_OK;
  }
  const int tiles = ((M + 15) / 16) * ((Nc + 15) / 16);
  dim3 grid((tiles + 3) / 4, S, Kr);
  hipLaunchKernelGGL(k_gemm_rs<T>, grid, dim3(256), 0, st, M, Nc, K, S, shift, A, B, C, T(1), D, T(1), zero_oob);
  VSM_LAUNCH_CHECK("k_gemm_rs");
  return VSM_OK;
}

// dst[:, n1, dn] = src[:, n1, dn] (* scale[n0])   for in-band pairs; `per` elements per block
template <typename T>
__global__ void k_copy_rs(long long per, int S, const int* __restrict__ shift, const T* __restrict__ src,
                          const T* __restrict__ scale_n0, T* dst) {
  const int n1 = blockIdx.y, dn = blockIdx.z;
  const int n0 = n1 + shift[dn];
  if (n0 < 0 || n0 >= S) return;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= per) return;
  const long long o = ((long long)n1 + (long long)S * dn) * per + e;
  dst[o] = scale_n0 ? src[o] * scale_n0[n0] : src[o];
}
template <typename T>
static int copy_rs(long long per, int S, int Kr, const int* shift, const T* src, const T* scale_n0, T* dst, hipStream_t st) {
  if (S <= 0 || Kr <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_copy_rs<T>, dim3((unsigned)((per + 255) / 256), S, Kr), dim3(256), 0, st, per, S, shift, src, scale_n0,
                     dst);
  VSM_LAUNCH_CHECK("k_copy_rs");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// elemental_inelastic!(::RRS)
// ---------------------------------------------------------------------------
template <typename T>
struct tol;
template <>
struct tol<double> {
  static __device__ __forceinline__ double weight() { return 1e-8; }  // rt_helpers.jl:56-58
  static __device__ __forceinline__ double close() { return 1e-8; }   // :66-68
  static __device__ __forceinline__ double loose() { return 1e-6; }   // :76
};
template <>
struct tol<float> {
  static __device__ __forceinline__ float weight() { return 1e-8f; }
  static __device__ __forceinline__ float close() { return 1e-8f; }
  static __device__ __forceinline__ float loose() { return 1e-6f; }
};

// get_elem_rt_RRS! (elemental_inelastic.jl:117-206) + apply_D_elemental_RRS! (:619-637)
template <typename T>
__global__ __launch_bounds__(256) void k_elemental_rrs(int N, int ns, int S, int m, int ndoubl, const int* __restrict__ shift,
                                                       const T* __restrict__ varpi_ie, const T* __restrict__ fscatt,
                                                       const T* __restrict__ dtau, const T* __restrict__ Zpp,
                                                       const T* __restrict__ Zmp, const T* __restrict__ mu,
                                                       const T* __restrict__ wt, T* ier_mp, T* iet_pp, T* ier_pm, T* iet_mm) {
  // One workgroup per (recipient n1, line dn).  The exponentials are separable over the two points: x_i = dtau[n1]/mu_i,
  // y_j = dtau[n0]/mu_j, exp(-x), expm1(-x) (and the same for y) are tabulated once per workgroup (2 N values instead of
  // N^2), and  -expm1(-(x + y)) = -(p + q + p q)  with p = expm1(-x), q = expm1(-y)  (both negative: no cancellation).
  // Per element one transcendental is left (the expm1 of the difference inside expdiff_neg) instead of three.
  __shared__ T sx[2][128], se[2][128], sp[2][128], smu[128];
  const int n1 = blockIdx.x, dn = blockIdx.y, tid = threadIdx.x;
  const int n0 = n1 + shift[dn];
  const bool inband = n0 >= 0 && n0 < S;
  {
    const int h = tid >> 7, i = tid & 127;
    if (i < N && (h == 0 || inband)) {
      const T mi = mu[i];
      const T x = dtau[h == 0 ? n1 : n0] / mi;
      sx[h][i] = x;
      se[h][i] = exp(-x);
      sp[h][i] = expm1(-x);
      if (h == 0) smu[i] = mi;
    }
  }
  __syncthreads();
  const T d1 = dtau[n1], d0 = inband ? dtau[n0] : T(1);
  const T w = varpi_ie[dn], f = inband ? fscatt[n0] : T(0);
  const T ratio = d1 / d0;
  const long long ob = ((long long)n1 + (long long)S * dn) * N * N;
  for (int e = tid; e < N * N; e += 256) {
    const int i = e % N, j = e / N;
    T r = T(0), t = T(0);
    const T wct = (m == 0) ? wt[j] / T(2) : wt[j] / T(4);
    if (inband && wct > tol<T>::weight()) {
      const T mi = smu[i], mj = smu[j];
      const T x = sx[0][i], y = sx[1][j];
      const T p = sp[0][i], q = sp[1][j];
      const T ey = se[1][j];
      r = f * w * Zmp[e] * (T(1) / ((mi / mj) + ratio)) * (-(p + q + p * q)) * wct;
      const T ed = (x == y) ? T(0) : ((x < y) ? se[0][i] * (-expm1(-(y - x))) : -ey * (-expm1(-(x - y))));   // expdiff_neg(x, y)
      if (mi == mj) {
        if (fabs(d0 - d1) > tol<T>::loose())
          t = w * f * Zpp[e] * wct * ed / (T(1) - ratio);
        else
          t = (d0 / mi) * w * f * Zpp[e] * wct * ey;
      } else {
        const T den = (mi / mj) - ratio;
        if (fabs(den) < tol<T>::close())
          t = (d0 / mi) * w * f * Zpp[e] * wct * ey;
        else
          t = w * f * Zpp[e] * (T(1) / den) * wct * ed;
      }
    }
    const long long o = ob + e;
    if (ns == 1) {
      ier_mp[o] = r;
      iet_pp[o] = t;
      ier_pm[o] = r;
      iet_mm[o] = t;
    } else if (ndoubl < 1) {
      const bool same = is_uv_row(i, ns) == is_uv_row(j, ns);
      ier_mp[o] = r;
      iet_pp[o] = t;
      ier_pm[o] = same ? r : -r;
      iet_mm[o] = same ? t : -t;
    } else {
      ier_mp[o] = is_uv_row(i, ns) ? -r : r;
      iet_pp[o] = t;
    }
  }
}

// get_elem_rt_SFI_RRS! (elemental_inelastic.jl:479-610)
template <typename T>
__global__ __launch_bounds__(256) void k_elemental_sfi_rrs(int N, int ns, int S, int m, int ndoubl, int i_mu0,
                                                           const int* __restrict__ shift, const T* __restrict__ varpi_ie,
                                                           const T* __restrict__ fscatt, const T* __restrict__ dtau,
                                                           const T* __restrict__ tau_sum, const T* __restrict__ F0,
                                                           const T* __restrict__ Zpp, const T* __restrict__ Zmp,
                                                           const T* __restrict__ mu, T* ieJ0_p, T* ieJ0_m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int n1 = blockIdx.y, dn = blockIdx.z;
  const int n0 = n1 + shift[dn];
  T jp = T(0), jm = T(0);
  if (n0 >= 0 && n0 < S) {
    const int i_start = ns * i_mu0;
    const T wct02 = (m == 0) ? T(0.5) : T(0.25);
    T zp = 0, zm = 0;
    for (int q = 0; q < ns; ++q) {
      const long long zo = i + (long long)N * (i_start + q);
      const T f0 = F0[q + (long long)ns * n0];
      zp += Zpp[zo] * f0;
      zm += Zmp[zo] * f0;
    }
    const T mi = mu[i], ms = mu[i_start];
    const T d1 = dtau[n1], d0 = dtau[n0];
    const T w = varpi_ie[dn], f = fscatt[n0];
    const T ratio = d1 / d0;
    if (i >= i_start && i < i_start + ns) {
      if (fabs(d0 - d1) > tol<T>::close())
        jp = w * f * zp * wct02 * expdiff_neg<T>(d1 / mi, d0 / mi) / (T(1) - ratio);
      else
        jp = (d0 / mi) * wct02 * w * f * zp * exp(-d0 / mi);
    } else {
      const T den = (mi / ms) - ratio;
      if (fabs(den) < tol<T>::close())
        jp = (d0 / mi) * wct02 * w * f * zp * exp(-d0 / ms);
      else
        jp = wct02 * w * f * zp * (T(1) / den) * expdiff_neg<T>(d1 / mi, d0 / ms);
    }
    jm = wct02 * w * f * zm * (T(1) / ((mi / ms) + ratio)) * (-expm1(-((d1 / mi) + (d0 / ms))));
    const T att = exp(-tau_sum[n0] / ms);
    jp *= att;
    jm *= att;
  }
  if (ndoubl >= 1 && is_uv_row(i, ns)) jm = -jm;
  const long long o = ((long long)n1 + (long long)S * dn) * N + i;
  ieJ0_p[o] = jp;
  ieJ0_m[o] = jm;
}

template <typename T>
struct added_rs {
  T *ier_mp, *iet_pp, *ier_pm, *iet_mm, *ieJ0_p, *ieJ0_m;
  int K;
};
template <typename T>
struct composite_rs {
  T *ieR_mp, *ieR_pm, *ieT_pp, *ieT_mm, *ieJ0_p, *ieJ0_m;
  int K;
};
template <typename T>
struct rrs_in {
  const int* shift;   // device int[K]
  const T* varpi_ie;  // device [K]
  const T* fscatt;    // device [S]
  const T* Zpp;       // device [N,N]
  const T* Zmp;
};

template <typename T>
static int elemental_inelastic_rrs(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* tau_sum, const T* F0,
                                   const rrs_in<T>& rs, const added_rs<T>& a, hipStream_t st) {
  if (S <= 0 || a.K <= 0) return VSM_OK;
  const int N = q.N;
  if (N > 128) {
    set_error("elemental_inelastic_rrs: N = %d exceeds 128", N);
    return VSM_ERR_UNSUPPORTED;
  }
  hipLaunchKernelGGL(k_elemental_rrs<T>, dim3(S, a.K), dim3(256), 0, st, N, q.n_stokes, S, m, ndoubl,
                     rs.shift, rs.varpi_ie, rs.fscatt, dtau, rs.Zpp, rs.Zmp, q.mu, q.wt, a.ier_mp, a.iet_pp, a.ier_pm,
                     a.iet_mm);
  VSM_LAUNCH_CHECK("k_elemental_rrs");
  hipLaunchKernelGGL(k_elemental_sfi_rrs<T>, dim3((N + 255) / 256, S, a.K), dim3(256), 0, st, N, q.n_stokes, S, m, ndoubl,
                     q.i_mu0, rs.shift, rs.varpi_ie, rs.fscatt, dtau, tau_sum, F0, rs.Zpp, rs.Zmp, q.mu, a.ieJ0_p, a.ieJ0_m);
  VSM_LAUNCH_CHECK("k_elemental_sfi_rrs");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// doubling_helper!(::RRS)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_scale_vec(int N, int S, const T* __restrict__ expk, const T* __restrict__ a, T* out) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)N * S) return;
  out[e] = a[e] * expk[e / N];
}
template <typename T>
__global__ void k_square_v(int S, T* x) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < S) x[s] = x[s] * x[s];
}
// apply_D_matrix! + apply_D_matrix_SFI! on a batch of B blocks (doubling.jl:178-252; for the 4-D inelastic
// arrays apply_D_IE_RRS! / apply_D_SFI_IE_RRS!, doubling_inelastic.jl:336-356,408-416, with B = S*K)
template <typename T>
__global__ void k_apply_D_batch(int N, int ns, T* r_mp, const T* __restrict__ t_pp, T* r_pm, T* t_mm, T* j0_m) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * N) return;
  const long long b = blockIdx.y;
  const int i = e % N, j = e / N;
  const long long o = b * N * N + e;
  const bool ui = is_uv_row(i, ns), uj = is_uv_row(j, ns);
  if (ns == 1) {
    r_pm[o] = r_mp[o];
    t_mm[o] = t_pp[o];
    return;
  }
  T r = r_mp[o];
  if (ui) r = -r;
  r_mp[o] = r;
  const T t = t_pp[o];
  r_pm[o] = (ui == uj) ? r : -r;
  t_mm[o] = (ui == uj) ? t : -t;
  if (j == 0 && ui) j0_m[b * N + i] = -j0_m[b * N + i];
}

// all lines of a recipient point in one workgroup (N <= 30): FP64 one wave per line, else one tile per wave
template <typename T>
static int raman_ia_lines(int N, int S, int K, const int* shift, const rs_ia_pass<T>& h, hipStream_t st) {
  if constexpr (std::is_same<T, double>::value) {
    int rc = raman_interaction_quad(N, S, K, shift, h, st);   // 3 <= N <= 22: four lines per wave on the 4 x 4 x 4 MFMA
    if (rc == VSM_ERR_UNSUPPORTED) rc = raman_interaction_wave(N, S, K, shift, h, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  return raman_interaction_lines<T>(N, S, K, shift, h, st);
}

template <typename T>
static size_t doubling_rs_work_elems(int N, int S, int K) {
  const size_t NN = (size_t)N * N;
  return 7 * NN * S + 6 * (size_t)N * S + 4 * NN * S * K + 5 * (size_t)N * S * K;
}

template <typename T>
static int doubling_inelastic_rrs(int N, int ns, int S, int ndoubl, T* expk, const int* shift, const added<T>& a,
                                  const added_rs<T>& ie, T* work, hipStream_t st) {
  if (ndoubl == 0 || S <= 0) return VSM_OK;  // doubling_inelastic.jl:35 (returns before apply_D)
  const int K = ie.K;
  const long long NN = (long long)N * N, per = NN * S, pv = (long long)N * S;
  T* p = work;
  auto take = [&](long long n) { T* r = p; p += n; return r; };
  T *gp = take(per), *ttg = take(per), *gt = take(per), *gr = take(per), *grt = take(per), *tM = take(per), *tM2 = take(per);
  T *j1p = take(pv), *j1m = take(pv), *tmp1 = take(pv), *tmp2 = take(pv), *u = take(pv), *u2 = take(pv);
  T *W1 = take(per * K), *W2 = take(per * K), *W3 = take(per * K), *W4 = take(per * K);
  T *ieJ1p = take(pv * K), *ieJ1m = take(pv * K), *V1 = take(pv * K), *V2 = take(pv * K), *V3 = take(pv * K);
  const T* nul = nullptr;
  const T one = T(1), zero = T(0);
  const rs_op<T> none{nullptr, 0, 0};
  int rc;
#define G3(...) if ((rc = gemm<T>(__VA_ARGS__, st))) return rc
#define G4(M_, Nc_, A_, B_, C_, D_) if ((rc = gemm_rs<T>(M_, Nc_, N, S, K, shift, A_, B_, C_, D_, st))) return rc
  const dim3 gv((unsigned)((pv + 255) / 256));
  static const bool sepD = ab_switch("VSM_RAMAN_SEPARATE_APPLY_D");
  bool ie_D_done = false;   // apply_D! of the inelastic operators done by the last step's line kernel
  // FP64, 20 <= N <= 22: the elastic chain of ALL steps first (every step's operands kept in library scratch: 9 x 6 N^2 S numbers on
  // the C5 shape), then ONE launch that walks the steps with the inelastic state of its lines on chip (k_raman_doubling_chain)
  if constexpr (std::is_same<T, double>::value) {
    if (raman_chain_supported(N, K) && !sepD && N <= fused_max_n<T>()) {
      T* stash = static_cast<T*>(scratch(raman_chain_stash_elems(N, S, ndoubl) * sizeof(T), 4, st));
      auto sp = [&](int step, int which) { return raman_chain_stash_ptr(stash, N, S, ndoubl, step, which); };
      bool chained = stash != nullptr;   // (no room for the steps' operands: the step kernels below need none)
      if (!chained) (void)hipGetLastError();
      for (int n = 0; chained && n < ndoubl; ++n) {
        rc = raman_elastic_pre<T>(N, S, a.r_mp, a.t_pp, a.j0_p, a.j0_m, expk, sp(n, 2), sp(n, 3), sp(n, 4), sp(n, 5), j1p, sp(n, 7), u,
                                  u2, sp(n, 8), sp(n, 9), st);
        if (rc == VSM_ERR_UNSUPPORTED && n == 0) {   // (an A/B build without the LDS-resident elastic kernels: nothing touched yet)
          chained = false;
          break;
        }
        if (rc) return rc;
        VSM_HIP(hipMemcpyAsync(sp(n, 0), a.r_mp, sizeof(T) * per, hipMemcpyDeviceToDevice, st));
        VSM_HIP(hipMemcpyAsync(sp(n, 1), a.t_pp, sizeof(T) * per, hipMemcpyDeviceToDevice, st));
        VSM_HIP(hipMemcpyAsync(sp(n, 6), a.j0_p, sizeof(T) * pv, hipMemcpyDeviceToDevice, st));
        VSM_HIP(hipMemcpyAsync(sp(n, 10), expk, sizeof(T) * S, hipMemcpyDeviceToDevice, st));
        if ((rc = raman_elastic_post<T>(N, S, a.r_mp, a.t_pp, sp(n, 2), u, u2, j1p, a.j0_p, a.j0_m, expk, st))) return rc;
      }
      if (chained) {
        if ((rc = raman_doubling_chain(N, S, K, ndoubl, shift, stash, ie.ier_mp, ie.iet_pp, ie.ieJ0_p, ie.ieJ0_m, ns, ie.ier_pm,
                                       ie.iet_mm, st)))
          return rc;
        const dim3 gm2((unsigned)((NN + 255) / 256), S);
        hipLaunchKernelGGL(k_apply_D_batch<T>, gm2, dim3(256), 0, st, N, ns, a.r_mp, a.t_pp, a.r_pm, a.t_mm, a.j0_m);
        VSM_LAUNCH_CHECK("k_apply_D_batch");
        return VSM_OK;
      }
    }
  }
  for (int n = 0; n < ndoubl; ++n) {
    // elastic operands of the step: one LDS-resident launch per point, else the operator chain
    rc = raman_elastic_pre<T>(N, S, a.r_mp, a.t_pp, a.j0_p, a.j0_m, expk, ttg, gt, gr, grt, j1p, j1m, u, u2, tmp1, tmp2, st);
    const bool fused_elastic = rc == VSM_OK;
    if (rc == VSM_ERR_UNSUPPORTED) {
      // gp = (I - r r)^-1 ; ttg = t gp
      if ((rc = inv_one_minus<T>(N, S, a.r_mp, NN, a.r_mp, NN, gp, tM, st))) return rc;
      G3(N, N, N, S, a.t_pp, NN, gp, NN, ttg, NN, one, nul, 0, zero, zero);
      // J1+- = J0+- expk
      hipLaunchKernelGGL(k_scale_vec<T>, gv, dim3(256), 0, st, N, S, expk, a.j0_p, j1p);
      hipLaunchKernelGGL(k_scale_vec<T>, gv, dim3(256), 0, st, N, S, expk, a.j0_m, j1m);
      VSM_LAUNCH_CHECK("k_scale_vec");
      // tmp1 = gp (J0+ + r J1-) ; tmp2 = gp (J1- + r J0+)
      G3(N, 1, N, S, a.r_mp, NN, j1m, N, u, N, one, a.j0_p, N, one, zero);
      G3(N, 1, N, S, gp, NN, u, N, tmp1, N, one, nul, 0, zero, zero);
      G3(N, 1, N, S, a.r_mp, NN, a.j0_p, N, u2, N, one, j1m, N, one, zero);
      G3(N, 1, N, S, gp, NN, u2, N, tmp2, N, one, nul, 0, zero, zero);
      // gt = gp t ; gr = gp r ; grt = gr t   (old r, t)
      G3(N, N, N, S, gp, NN, a.t_pp, NN, gt, NN, one, nul, 0, zero, zero);
      G3(N, N, N, S, gp, NN, a.r_mp, NN, gr, NN, one, nul, 0, zero, zero);
      G3(N, N, N, S, gr, NN, a.t_pp, NN, grt, NN, one, nul, 0, zero, zero);
    } else if (rc) {
      return rc;
    }
    // ---- inelastic recurrences of this step (they read the OLD elastic r, t, J0+, expk) -------------------------------
    rc = VSM_ERR_UNSUPPORTED;
    if constexpr (std::is_same<T, double>::value) {
      // FP64, 3 <= N <= 22: four lines per wave on the 4 x 4 x 4 MFMA; N <= 30: one wave per line on 16 x 16 x 4 tiles
      rc = raman_doubling_quad(N, S, K, shift, a.r_mp, a.t_pp, ttg, gt, gr, grt, a.j0_p, j1m, tmp1, tmp2, expk, ie.ier_mp,
                               ie.iet_pp, ie.ieJ0_p, ie.ieJ0_m, (n == ndoubl - 1 && !sepD) ? ns : 0, ie.ier_pm, ie.iet_mm, st);
      if (rc == VSM_ERR_UNSUPPORTED)
        rc = raman_doubling_wave(N, S, K, shift, a.r_mp, a.t_pp, ttg, gt, gr, grt, a.j0_p, j1m, tmp1, tmp2, expk, ie.ier_mp,
                                 ie.iet_pp, ie.ieJ0_p, ie.ieJ0_m, (n == ndoubl - 1 && !sepD) ? ns : 0, ie.ier_pm, ie.iet_mm, st);
    }
    ie_D_done = rc == VSM_OK && n == ndoubl - 1 && !sepD;
    if (rc == VSM_ERR_UNSUPPORTED)
      rc = raman_doubling_lines<T>(N, S, K, shift, a.r_mp, a.t_pp, ttg, gt, gr, grt, a.j0_p, j1m, tmp1, tmp2, expk,
                                   ie.ier_mp, ie.iet_pp, ie.ieJ0_p, ie.ieJ0_m, st);   // N <= 30: one workgroup per point
    if (rc == VSM_ERR_UNSUPPORTED) {
      // operator-level chain: one launch per batched operator over all (n1, dn) pairs
      // ieJ1+- = ieJ0+- expk[n0]
      if ((rc = copy_rs<T>(N, S, K, shift, ie.ieJ0_p, expk, ieJ1p, st))) return rc;
      if ((rc = copy_rs<T>(N, S, K, shift, ie.ieJ0_m, expk, ieJ1m, st))) return rc;
      // X = ier r[n0] + r[n1] ier -> W1
      G4(N, N, at_4d<T>(ie.ier_mp, NN), at_n0<T>(a.r_mp, NN), W1, none);
      G4(N, N, at_n1<T>(a.r_mp, NN), at_4d<T>(ie.ier_mp, NN), W1, at_4d<T>(W1, NN));
      // tmp3 = ieJ1+ + ttg[n1] (ieJ0+ + r[n1] ieJ1- + ier J1-[n0] + X tmp1[n0]) + iet tmp1[n0]   -> V2
      G4(N, 1, at_n1<T>(a.r_mp, NN), at_4d<T>(ieJ1m, N), V1, at_4d<T>(ie.ieJ0_p, N));
      G4(N, 1, at_4d<T>(ie.ier_mp, NN), at_n0<T>(j1m, N), V1, at_4d<T>(V1, N));
      G4(N, 1, at_4d<T>(W1, NN), at_n0<T>(tmp1, N), V1, at_4d<T>(V1, N));
      G4(N, 1, at_4d<T>(ie.iet_pp, NN), at_n0<T>(tmp1, N), V2, at_4d<T>(ieJ1p, N));
      G4(N, 1, at_n1<T>(ttg, NN), at_4d<T>(V1, N), V2, at_4d<T>(V2, N));
      // tmp4 = ieJ0- + ttg[n1] (ieJ1- + ier J0+[n0] + r[n1] ieJ0+ + X tmp2[n0]) + iet tmp2[n0]   -> V3
      G4(N, 1, at_4d<T>(ie.ier_mp, NN), at_n0<T>(a.j0_p, N), V1, at_4d<T>(ieJ1m, N));
      G4(N, 1, at_n1<T>(a.r_mp, NN), at_4d<T>(ie.ieJ0_p, N), V1, at_4d<T>(V1, N));
      G4(N, 1, at_4d<T>(W1, NN), at_n0<T>(tmp2, N), V1, at_4d<T>(V1, N));
      G4(N, 1, at_4d<T>(ie.iet_pp, NN), at_n0<T>(tmp2, N), V3, at_4d<T>(ie.ieJ0_m, N));
      G4(N, 1, at_n1<T>(ttg, NN), at_4d<T>(V1, N), V3, at_4d<T>(V3, N));
      if ((rc = copy_rs<T>(N, S, K, shift, V2, nul, ie.ieJ0_p, st))) return rc;
      if ((rc = copy_rs<T>(N, S, K, shift, V3, nul, ie.ieJ0_m, st))) return rc;
      // tmp5 = ttg[n1] (iet + X gt[n0]) + iet gt[n0]   -> W3
      G4(N, N, at_4d<T>(W1, NN), at_n0<T>(gt, NN), W2, at_4d<T>(ie.iet_pp, NN));
      G4(N, N, at_4d<T>(ie.iet_pp, NN), at_n0<T>(gt, NN), W3, none);
      G4(N, N, at_n1<T>(ttg, NN), at_4d<T>(W2, NN), W3, at_4d<T>(W3, NN));
      // tmp6 = ier + iet grt[n0] + ttg[n1] (r[n1] iet + (ier + X gr[n0]) t[n0])   -> W2
      G4(N, N, at_4d<T>(W1, NN), at_n0<T>(gr, NN), W2, at_4d<T>(ie.ier_mp, NN));
      G4(N, N, at_4d<T>(W2, NN), at_n0<T>(a.t_pp, NN), W4, none);
      G4(N, N, at_n1<T>(a.r_mp, NN), at_4d<T>(ie.iet_pp, NN), W4, at_4d<T>(W4, NN));
      G4(N, N, at_4d<T>(ie.iet_pp, NN), at_n0<T>(grt, NN), W2, at_4d<T>(ie.ier_mp, NN));
      G4(N, N, at_n1<T>(ttg, NN), at_4d<T>(W4, NN), W2, at_4d<T>(W2, NN));
      if ((rc = copy_rs<T>(NN, S, K, shift, W3, nul, ie.iet_pp, st))) return rc;
      if ((rc = copy_rs<T>(NN, S, K, shift, W2, nul, ie.ier_mp, st))) return rc;
    } else if (rc) {
      return rc;
    }
    rc = fused_elastic ? raman_elastic_post<T>(N, S, a.r_mp, a.t_pp, ttg, u, u2, j1p, a.j0_p, a.j0_m, expk, st) : VSM_ERR_UNSUPPORTED;
    if (rc == VSM_ERR_UNSUPPORTED) {
      // J0- += ttg (J1- + r J0+) ; J0+ = J1+ + ttg (J0+ + r J1-)      (u2, u from above: old J0+)
      G3(N, 1, N, S, ttg, NN, u2, N, a.j0_m, N, one, a.j0_m, N, one, zero);
      G3(N, 1, N, S, ttg, NN, u, N, a.j0_p, N, one, j1p, N, one, zero);
      hipLaunchKernelGGL(k_square_v<T>, dim3((S + 255) / 256), dim3(256), 0, st, S, expk);
      VSM_LAUNCH_CHECK("k_square_v");
      // r <- r + ttg r t ; t <- ttg t
      G3(N, N, N, S, ttg, NN, a.r_mp, NN, tM, NN, one, nul, 0, zero, zero);
      G3(N, N, N, S, ttg, NN, a.t_pp, NN, tM2, NN, one, nul, 0, zero, zero);
      G3(N, N, N, S, tM, NN, a.t_pp, NN, a.r_mp, NN, one, a.r_mp, NN, one, zero);
      if ((rc = copy_strided<T>(per, 1, tM2, 0, a.t_pp, st))) return rc;
    } else if (rc) {
      return rc;
    }
  }
  const dim3 gm((unsigned)((NN + 255) / 256), S);
  hipLaunchKernelGGL(k_apply_D_batch<T>, gm, dim3(256), 0, st, N, ns, a.r_mp, a.t_pp, a.r_pm, a.t_mm, a.j0_m);
  if (!ie_D_done) {
    const dim3 gk((unsigned)((NN + 255) / 256), (unsigned)((long long)S * K));
    hipLaunchKernelGGL(k_apply_D_batch<T>, gk, dim3(256), 0, st, N, ns, ie.ier_mp, ie.iet_pp, ie.ier_pm, ie.iet_mm, ie.ieJ0_m);
  }
  VSM_LAUNCH_CHECK("k_apply_D_batch");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// interaction_helper!(::RRS, ::ScatteringInterface_11)
// ---------------------------------------------------------------------------
template <typename T>
static size_t interaction_rs_work_elems(int N, int S, int K) {
  const size_t NN = (size_t)N * N;
  return (8 * NN * S + 4 * (size_t)N * S) + 4 * NN * S * K + 2 * (size_t)N * S * K + (3 * NN * S + 2 * (size_t)N * S);
}

template <typename T>
static int interaction_inelastic_rrs(int N, int S, const int* shift, const composite<T>& c, const composite_rs<T>& cie,
                                     const added<T>& a, const added_rs<T>& aie, T* work, size_t elastic_work_elems,
                                     hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int K = cie.K;
  const long long NN = (long long)N * N, per = NN * S, pv = (long long)N * S;
  const long long as = a.mat_stride;
  T* p = work;
  auto take = [&](long long n) { T* r = p; p += n; return r; };
  T *G = take(per), *Tinv = take(per), *tM = take(per), *gA = take(per), *gB = take(per), *tM2 = take(per);
  take(2 * per);  // (reserved)
  T *v = take(pv), *u = take(pv);
  take(2 * pv);
  T *W1 = take(per * K), *W2 = take(per * K), *W3 = take(per * K), *W4 = take(per * K);
  T *V1 = take(pv * K), *V2 = take(pv * K);
  T* ework = p;
  const T* nul = nullptr;
  const T one = T(1), zero = T(0);
  const rs_op<T> none{nullptr, 0, 0};
  int rc;
  // ---- pass 1: G1 = (I - r-+ R+-)^-1, T01 = T-- G1 ------------------------------------------------
  if ((rc = inv_one_minus<T>(N, S, a.r_mp, as, c.R_pm, NN, G, tM, st))) return rc;
  G3(N, N, N, S, c.T_mm, NN, G, NN, Tinv, NN, one, nul, 0, zero, zero);
  // v = G1 (j0- + r-+ J0+)
  G3(N, 1, N, S, a.r_mp, as, c.J0_p, N, u, N, one, a.j0_m, N, one, zero);
  G3(N, 1, N, S, G, NN, u, N, v, N, one, nul, 0, zero, zero);
  // gA = G1 r-+ T++ ; gB = G1 t--
  G3(N, N, N, S, G, NN, a.r_mp, as, tM, NN, one, nul, 0, zero, zero);
  G3(N, N, N, S, tM, NN, c.T_pp, NN, gA, NN, one, nul, 0, zero, zero);
  G3(N, N, N, S, G, NN, a.t_mm, as, gB, NN, one, nul, 0, zero, zero);
  bool fused1 = false;
  {
    rs_ia_pass<T> h{};
    h.L1 = aie.ier_mp; h.E0 = c.R_pm; h.sE0 = NN; h.L2 = a.r_mp; h.sL2 = as; h.I1 = cie.ieR_pm; h.TI = Tinv; h.YA = cie.ieT_mm;
    h.E3 = c.T_pp; h.sE3 = NN; h.I3 = cie.ieT_pp; h.ACCA = cie.ieR_mp; h.GX = gA; h.I4 = aie.iet_mm; h.GY = gB;
    h.OUTA = cie.ieR_mp; h.OUTB = cie.ieT_mm;
    h.VE0 = c.J0_p; h.VADD = aie.ieJ0_m; h.VI1 = cie.ieJ0_p; h.VACC = cie.ieJ0_m; h.VV = v; h.VOUT = cie.ieJ0_m;
    rc = raman_ia_lines<T>(N, S, K, shift, h, st);   // every line of a recipient point in one workgroup (N <= 30)
    if (rc == VSM_OK) fused1 = true;
    else if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  if (!fused1) {
  // Y = T01[n1] (ier R+-[n0] + r[n1] ieR+-) + ieT--   -> W2
  G4(N, N, at_4d<T>(aie.ier_mp, NN), at_n0<T>(c.R_pm, NN), W1, none);
  G4(N, N, at_n1<T>(a.r_mp, as), at_4d<T>(cie.ieR_pm, NN), W1, at_4d<T>(W1, NN));
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(W1, NN), W2, at_4d<T>(cie.ieT_mm, NN));
  // ieJ0- = ieJ0- + T01[n1] (ier J0+[n0] + r[n1] ieJ0+ + iej0-) + Y v[n0]
  G4(N, 1, at_4d<T>(aie.ier_mp, NN), at_n0<T>(c.J0_p, N), V1, at_4d<T>(aie.ieJ0_m, N));
  G4(N, 1, at_n1<T>(a.r_mp, as), at_4d<T>(cie.ieJ0_p, N), V1, at_4d<T>(V1, N));
  G4(N, 1, at_n1<T>(Tinv, NN), at_4d<T>(V1, N), V2, at_4d<T>(cie.ieJ0_m, N));
  if ((rc = gemm_rs<T>(N, 1, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(v, N), cie.ieJ0_m, at_4d<T>(V2, N), st, 1))) return rc;
  // ieR-+ = ieR-+ + T01[n1] (ier T++[n0] + r[n1] ieT++) + Y gA[n0]
  G4(N, N, at_4d<T>(aie.ier_mp, NN), at_n0<T>(c.T_pp, NN), W3, none);
  G4(N, N, at_n1<T>(a.r_mp, as), at_4d<T>(cie.ieT_pp, NN), W3, at_4d<T>(W3, NN));
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(W3, NN), W4, at_4d<T>(cie.ieR_mp, NN));
  if ((rc = gemm_rs<T>(N, N, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(gA, NN), cie.ieR_mp, at_4d<T>(W4, NN), st, 1))) return rc;
  // ieT-- = T01[n1] iet-- + Y gB[n0]
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(aie.iet_mm, NN), W3, none);
  if ((rc = gemm_rs<T>(N, N, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(gB, NN), cie.ieT_mm, at_4d<T>(W3, NN), st, 1))) return rc;
  }
  // ---- pass 2: G2 = (I - R+- r-+)^-1, T21 = t++ G2 ------------------------------------------------
  if ((rc = inv_one_minus<T>(N, S, c.R_pm, NN, a.r_mp, as, G, tM, st))) return rc;
  G3(N, N, N, S, a.t_pp, as, G, NN, Tinv, NN, one, nul, 0, zero, zero);
  // v = G2 (J0+ + R+- j0-)
  G3(N, 1, N, S, c.R_pm, NN, a.j0_m, N, u, N, one, c.J0_p, N, one, zero);
  G3(N, 1, N, S, G, NN, u, N, v, N, one, nul, 0, zero, zero);
  // gA = G2 T++ ; gB = G2 R+- t--
  G3(N, N, N, S, G, NN, c.T_pp, NN, gA, NN, one, nul, 0, zero, zero);
  G3(N, N, N, S, G, NN, c.R_pm, NN, tM2, NN, one, nul, 0, zero, zero);
  G3(N, N, N, S, tM2, NN, a.t_mm, as, gB, NN, one, nul, 0, zero, zero);
  bool fused2 = false;
  {
    rs_ia_pass<T> h{};
    h.L1 = cie.ieR_pm; h.E0 = a.r_mp; h.sE0 = as; h.L2 = c.R_pm; h.sL2 = NN; h.I1 = aie.ier_mp; h.TI = Tinv; h.YA = aie.iet_pp;
    h.E3 = a.t_mm; h.sE3 = as; h.I3 = aie.iet_mm; h.ACCA = aie.ier_pm; h.GX = gB; h.I4 = cie.ieT_pp; h.GY = gA;
    h.OUTA = cie.ieR_pm; h.OUTB = cie.ieT_pp;
    h.VE0 = a.j0_m; h.VADD = cie.ieJ0_p; h.VI1 = aie.ieJ0_m; h.VACC = aie.ieJ0_p; h.VV = v; h.VOUT = cie.ieJ0_p;
    rc = raman_ia_lines<T>(N, S, K, shift, h, st);   // every line of a recipient point in one workgroup (N <= 30)
    if (rc == VSM_OK) fused2 = true;
    else if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  if (!fused2) {
  // Y = T21[n1] (ieR+- r-+[n0] + R+-[n1] ier-+) + iet++   -> W2
  G4(N, N, at_4d<T>(cie.ieR_pm, NN), at_n0<T>(a.r_mp, as), W1, none);
  G4(N, N, at_n1<T>(c.R_pm, NN), at_4d<T>(aie.ier_mp, NN), W1, at_4d<T>(W1, NN));
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(W1, NN), W2, at_4d<T>(aie.iet_pp, NN));
  // ieJ0+ = iej0+ + T21[n1] (ieJ0+ + ieR+- j0-[n0] + R+-[n1] iej0-) + Y v[n0]
  G4(N, 1, at_4d<T>(cie.ieR_pm, NN), at_n0<T>(a.j0_m, N), V1, at_4d<T>(cie.ieJ0_p, N));
  G4(N, 1, at_n1<T>(c.R_pm, NN), at_4d<T>(aie.ieJ0_m, N), V1, at_4d<T>(V1, N));
  G4(N, 1, at_n1<T>(Tinv, NN), at_4d<T>(V1, N), V2, at_4d<T>(aie.ieJ0_p, N));
  if ((rc = gemm_rs<T>(N, 1, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(v, N), cie.ieJ0_p, at_4d<T>(V2, N), st, 1))) return rc;
  // ieR+- = ier+- + T21[n1] (ieR+- t--[n0] + R+-[n1] iet--) + Y gB[n0]     (before ieT++ is overwritten: independent)
  G4(N, N, at_4d<T>(cie.ieR_pm, NN), at_n0<T>(a.t_mm, as), W3, none);
  G4(N, N, at_n1<T>(c.R_pm, NN), at_4d<T>(aie.iet_mm, NN), W3, at_4d<T>(W3, NN));
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(W3, NN), W4, at_4d<T>(aie.ier_pm, NN));
  if ((rc = gemm_rs<T>(N, N, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(gB, NN), cie.ieR_pm, at_4d<T>(W4, NN), st, 1))) return rc;
  // ieT++ = T21[n1] ieT++ + Y gA[n0]
  G4(N, N, at_n1<T>(Tinv, NN), at_4d<T>(cie.ieT_pp, NN), W3, none);
  if ((rc = gemm_rs<T>(N, N, N, S, K, shift, at_4d<T>(W2, NN), at_n0<T>(gA, NN), cie.ieT_pp, at_4d<T>(W3, NN), st, 1))) return rc;
  }
  // ---- elastic part last (every inelastic right-hand side above used the pre-update composite) -----
  (void)elastic_work_elems;
  if (N <= fused_max_n<T>()) {   // LDS-resident k_interaction11 (one launch), else the operator chain
    static const bool off = ab_switch("VSM_NO_RAMAN_ELASTIC_FUSION");
    rc = off ? VSM_ERR_UNSUPPORTED : fused_interaction<T>(VSM_IFACE_11, N, S, c, a, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  return interaction_generic<T>(VSM_IFACE_11, N, S, c, a, ework, st);
}
#undef G3
#undef G4

// ---------------------------------------------------------------------------
// interaction_helper!(::RRS, ::ScatteringInterface_00 / _01 / _10)   (interaction_inelastic.jl:74-101, 103-154, 215-262)
// Operator level: one launch per batched operator over all (recipient, line) pairs; out-of-band blocks of the composite become
// (stay) zero, like the _11 pass.  The inelastic statements use the pre-update elastic composite; the elastic update comes last.
// (Upstream these three branches cannot run as committed: their loop headers name an undefined `ieJ1+`, :109,:221 -- the
// statements themselves are followed here.)
// ---------------------------------------------------------------------------
template <typename T>
static int interaction_inelastic_rrs_plain(int iface, int N, int S, const int* shift, const composite<T>& c,
                                           const composite_rs<T>& cie, const added<T>& a, const added_rs<T>& aie, T* work,
                                           hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int K = cie.K;
  const long long NN = (long long)N * N, per = NN * S, pv = (long long)N * S;
  const long long as = a.mat_stride;
  T* W1 = work;
  T* V1 = W1 + per * K;
  T* ework = V1 + pv * K;
  const rs_op<T> none{nullptr, 0, 0};
  int rc;
#define GZ(M_, Nc_, A_, B_, C_, D_) if ((rc = gemm_rs<T>(M_, Nc_, N, S, K, shift, A_, B_, C_, D_, st, 1))) return rc
  if (iface == VSM_IFACE_00) {
    for (T* x : {cie.ieR_mp, cie.ieR_pm, cie.ieT_pp, cie.ieT_mm})
      VSM_HIP(hipMemsetAsync(x, 0, sizeof(T) * (size_t)(per * K), st));
    for (T* x : {cie.ieJ0_p, cie.ieJ0_m}) VSM_HIP(hipMemsetAsync(x, 0, sizeof(T) * (size_t)(pv * K), st));
  } else if (iface == VSM_IFACE_01) {
    // ieJ0- = T--[n1] (ier-+ J0+[n0] + iej0-) ;  ieJ0+ = iej0+ + iet++ J0+[n0]
    GZ(N, 1, at_4d<T>(aie.ier_mp, NN), at_n0<T>(c.J0_p, N), V1, at_4d<T>(aie.ieJ0_m, N));
    GZ(N, 1, at_n1<T>(c.T_mm, NN), at_4d<T>(V1, N), cie.ieJ0_m, none);
    GZ(N, 1, at_4d<T>(aie.iet_pp, NN), at_n0<T>(c.J0_p, N), cie.ieJ0_p, at_4d<T>(aie.ieJ0_p, N));
    // ieR-+ = T--[n1] ier-+ T++[n0] ;  ieR+- = ier+- ;  ieT++ = iet++ T++[n0] ;  ieT-- = T--[n1] iet--
    GZ(N, N, at_4d<T>(aie.ier_mp, NN), at_n0<T>(c.T_pp, NN), W1, none);
    GZ(N, N, at_n1<T>(c.T_mm, NN), at_4d<T>(W1, NN), cie.ieR_mp, none);
    if ((rc = copy_strided<T>(per * K, 1, aie.ier_pm, 0, cie.ieR_pm, st))) return rc;
    GZ(N, N, at_4d<T>(aie.iet_pp, NN), at_n0<T>(c.T_pp, NN), cie.ieT_pp, none);
    GZ(N, N, at_n1<T>(c.T_mm, NN), at_4d<T>(aie.iet_mm, NN), cie.ieT_mm, none);
  } else {   // VSM_IFACE_10
    // ieJ0+ = t++[n1] (ieJ0+ + ieR+- j0-[n0]) ;  ieJ0- += ieT-- j0-[n0]
    GZ(N, 1, at_4d<T>(cie.ieR_pm, NN), at_n0<T>(a.j0_m, N), V1, at_4d<T>(cie.ieJ0_p, N));
    GZ(N, 1, at_n1<T>(a.t_pp, as), at_4d<T>(V1, N), cie.ieJ0_p, none);
    GZ(N, 1, at_4d<T>(cie.ieT_mm, NN), at_n0<T>(a.j0_m, N), V1, at_4d<T>(cie.ieJ0_m, N));
    if ((rc = copy_strided<T>(pv * K, 1, V1, 0, cie.ieJ0_m, st))) return rc;
    // ieT++ = t++[n1] ieT++ ;  ieT-- = ieT-- t--[n0] ;  ieR+- = t++[n1] ieR+- t--[n0]   (through W1: no operand is its own output)
    GZ(N, N, at_n1<T>(a.t_pp, as), at_4d<T>(cie.ieT_pp, NN), W1, none);
    if ((rc = copy_strided<T>(per * K, 1, W1, 0, cie.ieT_pp, st))) return rc;
    GZ(N, N, at_4d<T>(cie.ieT_mm, NN), at_n0<T>(a.t_mm, as), W1, none);
    if ((rc = copy_strided<T>(per * K, 1, W1, 0, cie.ieT_mm, st))) return rc;
    GZ(N, N, at_4d<T>(cie.ieR_pm, NN), at_n0<T>(a.t_mm, as), W1, none);
    GZ(N, N, at_n1<T>(a.t_pp, as), at_4d<T>(W1, NN), cie.ieR_pm, none);
  }
#undef GZ
  return interaction_generic<T>(iface, N, S, c, a, ework, st);
}

// postprocessing_vza!(::RRS) inelastic accumulation (postprocessing_vza.jl:139-142):
// ieR[v,k,s] += w[v,k] * sum_dn ieJ0-[row0[v]+k, s, dn]
struct pp_rs_args {
  int row0[16];
  double w[64];
};
template <typename T>
__global__ void k_postprocess_rs(int N, int ns, int S, int K, int nV, pp_rs_args pa, const T* __restrict__ ieJ_m,
                                 const T* __restrict__ ieJ_p, T* ieR, T* ieT) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)S * ns * nV) return;
  const int v = (int)(e % nV), k = (int)((e / nV) % ns);
  const long long s = e / ((long long)nV * ns);
  T sm = 0, sp = 0;
  for (int dn = 0; dn < K; ++dn) {
    const long long o = (s + (long long)S * dn) * N + pa.row0[v] + k;
    sm += ieJ_m[o];
    sp += ieJ_p[o];
  }
  const T w = (T)pa.w[v + nV * k];
  ieR[e] += w * sm;
  ieT[e] += w * sp;
}

}  // namespace vsm

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace vsm;

template <typename T, typename A>
static added_rs<T> cvt_ars(const A* a) {
  return added_rs<T>{a->ier_mp, a->iet_pp, a->ier_pm, a->iet_mm, a->ieJ0_p, a->ieJ0_m, a->K};
}
template <typename T, typename C>
static composite_rs<T> cvt_crs(const C* c) {
  return composite_rs<T>{c->ieR_mp, c->ieR_pm, c->ieT_pp, c->ieT_mm, c->ieJ0_p, c->ieJ0_m, c->K};
}
template <typename T, typename A>
static added<T> cvt_added_rs(const A* a) {
  added<T> r;
  r.r_mp = a->r_mp; r.t_pp = a->t_pp; r.r_pm = a->r_pm; r.t_mm = a->t_mm; r.j0_p = a->j0_p; r.j0_m = a->j0_m;
  r.mat_stride = a->mat_stride; r.d_symmetric = a->d_symmetric;
  return r;
}
template <typename T, typename C>
static composite<T> cvt_comp_rs(const C* c) {
  composite<T> r;
  r.R_mp = c->R_mp; r.R_pm = c->R_pm; r.T_pp = c->T_pp; r.T_mm = c->T_mm; r.J0_p = c->J0_p; r.J0_m = c->J0_m;
  return r;
}

extern "C" {

size_t vsm_doubling_inelastic_work_elems(int N, int S, int K) { return doubling_rs_work_elems<double>(N, S, K); }
size_t vsm_interaction_inelastic_work_elems(int N, int S, int K) { return interaction_rs_work_elems<double>(N, S, K); }

#define VSM_RAMAN_API(T, SFX)                                                                                          \
  int vsm_elemental_inelastic_rrs_##SFX(const vsm_quad_##SFX* q, int S, int m, int ndoubl, const T* dtau,              \
                                        const T* tau_sum, const T* F0, const vsm_rrs_##SFX* rs,                        \
                                        const vsm_added_rs_##SFX* added_rs_, void* stream) {                           \
    VSM_REQUIRE(q && rs && added_rs_ && dtau && tau_sum && F0, "elemental_inelastic_rrs: null argument");              \
    VSM_REQUIRE(q->mu && q->wt && q->N > 0 && q->n_stokes >= 1 && q->n_stokes <= 4 && q->N % q->n_stokes == 0,         \
                "elemental_inelastic_rrs: bad quad");                                                                  \
    VSM_REQUIRE(rs->shift && rs->varpi_ie && rs->fscatt && rs->Zpp && rs->Zmp, "elemental_inelastic_rrs: null RRS field"); \
    VSM_REQUIRE(added_rs_->K >= 0 && added_rs_->K <= 65535 && S >= 0 && S <= 65535,                                    \
                "elemental_inelastic_rrs: S and K must be <= 65535");                                                  \
    VSM_REQUIRE(added_rs_->ier_mp && added_rs_->iet_pp && added_rs_->ieJ0_p && added_rs_->ieJ0_m &&                    \
                    (ndoubl >= 1 || (added_rs_->ier_pm && added_rs_->iet_mm)),                                          \
                "elemental_inelastic_rrs: null AddedLayerRS field");                                                   \
    quad<T> qq;                                                                                                        \
    qq.mu = q->mu; qq.wt = q->wt; qq.N = q->N; qq.n_stokes = q->n_stokes; qq.i_mu0 = q->i_mu0; qq.mu0 = q->mu0;        \
    const rrs_in<T> r{rs->shift, rs->varpi_ie, rs->fscatt, rs->Zpp, rs->Zmp};                                          \
    return elemental_inelastic_rrs<T>(qq, S, m, ndoubl, dtau, tau_sum, F0, r, cvt_ars<T>(added_rs_), as_stream(stream)); \
  }                                                                                                                    \
  int vsm_doubling_inelastic_rrs_##SFX(int N, int n_stokes, int S, int ndoubl, T* expk, const int* shift,              \
                                       const vsm_added_##SFX* added_, const vsm_added_rs_##SFX* added_rs_, T* work,    \
                                       void* stream) {                                                                 \
    VSM_REQUIRE(added_ && added_rs_ && shift && expk && (work || ndoubl == 0), "doubling_inelastic_rrs: null argument"); \
    VSM_REQUIRE(added_->d_symmetric == 0 && added_->r_pm && added_->t_mm && added_->mat_stride == (long long)N * N,    \
                "doubling_inelastic_rrs: a full (non d_symmetric, per-point) AddedLayer is required");                 \
    VSM_REQUIRE(N > 0 && S >= 0 && S <= 65535 && added_rs_->K >= 0 && added_rs_->K <= 65535 &&                         \
                    (long long)S * added_rs_->K <= 2147483647LL,                                                       \
                "doubling_inelastic_rrs: bad N/S/K");                                                                  \
    VSM_REQUIRE(added_rs_->ier_mp && added_rs_->iet_pp && added_rs_->ier_pm && added_rs_->iet_mm &&                    \
                    added_rs_->ieJ0_p && added_rs_->ieJ0_m,                                                            \
                "doubling_inelastic_rrs: null AddedLayerRS field");                                                    \
    return doubling_inelastic_rrs<T>(N, n_stokes, S, ndoubl, expk, shift, cvt_added_rs<T>(added_), cvt_ars<T>(added_rs_), \
                                     work, as_stream(stream));                                                         \
  }                                                                                                                    \
  int vsm_interaction_inelastic_rrs_##SFX(int iface, int N, int S, const int* shift, const vsm_composite_##SFX* comp,  \
                                          const vsm_composite_rs_##SFX* comp_rs, const vsm_added_##SFX* added_,        \
                                          const vsm_added_rs_##SFX* added_rs_, T* work, void* stream) {                \
    VSM_REQUIRE(comp && comp_rs && added_ && added_rs_ && shift && work, "interaction_inelastic_rrs: null argument");  \
    VSM_REQUIRE(iface >= 0 && iface <= 3, "interaction_inelastic_rrs: unknown scattering interface %d", iface);         \
    VSM_REQUIRE(added_->d_symmetric == 0 && added_->r_pm && added_->t_mm,                                              \
                "interaction_inelastic_rrs: a full (non d_symmetric) AddedLayer is required");                         \
    VSM_REQUIRE(N > 0 && S >= 0 && S <= 65535 && comp_rs->K == added_rs_->K && comp_rs->K >= 0 && comp_rs->K <= 65535, \
                "interaction_inelastic_rrs: bad N/S/K");                                                               \
    if (iface != VSM_IFACE_11)                                                                                         \
      return interaction_inelastic_rrs_plain<T>(iface, N, S, shift, cvt_comp_rs<T>(comp), cvt_crs<T>(comp_rs),        \
                                                cvt_added_rs<T>(added_), cvt_ars<T>(added_rs_), work, as_stream(stream)); \
    return interaction_inelastic_rrs<T>(N, S, shift, cvt_comp_rs<T>(comp), cvt_crs<T>(comp_rs), cvt_added_rs<T>(added_), \
                                        cvt_ars<T>(added_rs_), work, 0, as_stream(stream));                            \
  }                                                                                                                    \
  int vsm_copy_added_to_composite_ie_##SFX(int N, int S, const vsm_added_rs_##SFX* a, const vsm_composite_rs_##SFX* c, \
                                           void* stream) {                                                             \
    VSM_REQUIRE(a && c && a->K == c->K && N > 0 && S >= 0, "copy_added_to_composite_ie: bad argument");                \
    const long long per = (long long)N * N * S * a->K, pv = (long long)N * S * a->K;                                   \
    hipStream_t st = as_stream(stream);                                                                                \
    int rc;                                                                                                            \
    if ((rc = copy_strided<T>(per, 1, a->iet_pp, 0, c->ieT_pp, st))) return rc;                                        \
    if ((rc = copy_strided<T>(per, 1, a->iet_mm, 0, c->ieT_mm, st))) return rc;                                        \
    if ((rc = copy_strided<T>(per, 1, a->ier_mp, 0, c->ieR_mp, st))) return rc;                                        \
    if ((rc = copy_strided<T>(per, 1, a->ier_pm, 0, c->ieR_pm, st))) return rc;                                        \
    if ((rc = copy_strided<T>(pv, 1, a->ieJ0_p, 0, c->ieJ0_p, st))) return rc;                                         \
    return copy_strided<T>(pv, 1, a->ieJ0_m, 0, c->ieJ0_m, st);                                                        \
  }                                                                                                                    \
  int vsm_postprocess_vza_ie_##SFX(int N, int n_stokes, int S, int K, int nV, const int* row0_h, const T* w_h,         \
                                   const T* ieJ0_m, const T* ieJ0_p, T* ieR, T* ieT, void* stream) {                   \
    VSM_REQUIRE(row0_h && w_h && ieJ0_m && ieJ0_p && ieR && ieT, "postprocess_vza_ie: null argument");                 \
    VSM_REQUIRE(nV >= 1 && nV <= 16 && n_stokes >= 1 && n_stokes <= 4, "postprocess_vza_ie: nV <= 16, nStokes <= 4");  \
    if (S <= 0 || K <= 0) return VSM_OK;                                                                               \
    pp_rs_args pa;                                                                                                     \
    for (int v = 0; v < nV; ++v) {                                                                                     \
      VSM_REQUIRE(row0_h[v] >= 0 && row0_h[v] + n_stokes <= N, "postprocess_vza_ie: row0 out of range");               \
      pa.row0[v] = row0_h[v];                                                                                          \
    }                                                                                                                  \
    for (int i = 0; i < nV * n_stokes; ++i) pa.w[i] = (double)w_h[i];                                                  \
    const long long tot = (long long)S * n_stokes * nV;                                                                \
    hipLaunchKernelGGL(k_postprocess_rs<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), N,   \
                       n_stokes, S, K, nV, pa, ieJ0_m, ieJ0_p, ieR, ieT);                                              \
    VSM_LAUNCH_CHECK("k_postprocess_rs");                                                                              \
    return VSM_OK;                                                                                                     \
  }
VSM_RAMAN_API(double, f64)
VSM_RAMAN_API(float, f32)

}  // extern "C"
