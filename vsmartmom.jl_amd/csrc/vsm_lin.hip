// Linearized (Jacobian) pass of the hot path, operator level: every statement of the reference's
// *_lin.jl kernels is one batched MFMA product over (spectral point, parameter) -- k_gemm with a second
// batch level -- or one elementwise kernel.  Inverses are NOT repeated per parameter (the reference reuses
// G as well: doubling_lin.jl:262-270, interaction_lin.jl:236-247).
//
//   elemental_lin           elemental_lin.jl:77-206, kernels :456-591 and :602-712
//   doubling_lin            doubling_lin.jl:216-339 (doubling_allparams_helper!), apply_D :374-421
//   interaction_lin         interaction_lin.jl:62-331
//   lambertian_surface_lin  Surfaces/lambertian_surface_lin.jl:48-162
//   postprocess_vza_lin     tools/postprocessing_vza_lin.jl:18-48
#include <type_traits>

#include "vsm_internal.h"

namespace vsm {

// ---------------------------------------------------------------------------
// elemental + chain rule (get_elem_rt_fused!, elemental_lin.jl:456-591)
// ---------------------------------------------------------------------------
// Component-mixed phase matrices (aerosol Jacobians): Z = sum_c fz[c, s] Zc[c] (or Zc[zsel]) and
// Z_dot[:, :, s, p] = sum_c zdcoef[c, p, s] Zc[c] over CT <= VSM_LIN_CT_MAX component blocks of one Fourier moment
// (vsm_layer_optics_lin_* produces fz / zdcoef) -- the [N, N, S, P] array Z_dot of the reference never exists.
#define VSM_LIN_CT_MAX 16
template <typename T>
struct zmix {
  const T *Zc_pp, *Zc_mp;   // [N, N, CT]
  const T* fz;              // [C, S] (zsel < 0)
  const T* zdcoef;          // [CT, P, S]
  int C, CT, zsel;
};

template <typename T, bool MIX>
__global__ __launch_bounds__(256) void k_elemental_lin(int N, int ns, int S, int m_rest, int n_m0, int ndoubl, int P,
                                                       const T* __restrict__ dtau, const T* __restrict__ varpi,
                                                       const T* __restrict__ Zpp, const T* __restrict__ Zmp,
                                                       long long zs, const T* __restrict__ dtau_dot,
                                                       const T* __restrict__ varpi_dot, const T* __restrict__ Zpp_dot,
                                                       const T* __restrict__ Zmp_dot, long long zds, long long zdp,
                                                       const zmix<T> zx, const T* __restrict__ mu,
                                                       const T* __restrict__ wt, T* r_mp, T* t_pp, T* r_pm, T* t_mm,
                                                       T* ap_r_mp, T* ap_t_pp, T* ap_r_pm, T* ap_t_mm) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const int m = (s < n_m0) ? 0 : m_rest;   // (a batch with Fourier moments folded into it: the first n_m0 points belong to m = 0)
  const int i = e % N, j = e / N;
  T zcp[MIX ? VSM_LIN_CT_MAX : 1], zcm[MIX ? VSM_LIN_CT_MAX : 1];
  T zmix_p = 0, zmix_m = 0;
  if constexpr (MIX) {
#pragma unroll
    for (int c = 0; c < VSM_LIN_CT_MAX; ++c) {
      zcp[c] = (c < zx.CT) ? zx.Zc_pp[(long long)c * N * N + e] : T(0);
      zcm[c] = (c < zx.CT) ? zx.Zc_mp[(long long)c * N * N + e] : T(0);
    }
#pragma unroll
    for (int c = 0; c < VSM_LIN_CT_MAX; ++c) {
      T f = T(0);
      if (c < zx.C) f = (zx.zsel < 0) ? zx.fz[c + (long long)zx.C * s] : (c == zx.zsel ? T(1) : T(0));
      zmix_p += f * zcp[c];
      zmix_m += f * zcm[c];
    }
  }
  const T wct = (m == 0) ? wt[j] / T(2) : wt[j] / T(4);
  const T mi = mu[i], mj = mu[j];
  const T d = dtau[s], w = varpi[s];
  const long long zo = (long long)s * zs + e;
  const bool ui = is_uv_row(i, ns), uj = is_uv_row(j, ns);
  const T dsign = (ui == uj) ? T(1) : T(-1);
  const T sign_r = (ndoubl >= 1 && ui) ? T(-1) : T(1);
  T r = 0, t = 0, r_tau = 0, r_w = 0, r_Z = 0, t_tau = 0, t_w = 0, t_Z = 0;
  if (wct > num<T>::eps()) {
    const T zm = MIX ? zmix_m : Zmp[zo], zp = MIX ? zmix_p : Zpp[zo];
    const T arg = d * ((T(1) / mi) + (T(1) / mj));
    const T geo = (mj / (mi + mj)) * wct * (-expm1(-arg));
    r = w * zm * geo;
    r_tau = w * zm * (T(1) / mi) * wct * exp(-arg);
    r_w = (w == T(0)) ? T(0) : r / w;
    r_Z = w * geo;
    if (mi == mj) {
      if (i == j) {
        const T ei = exp(-d / mi);
        t = ei * (T(1) + w * zp * (d / mi) * wct);
        t_tau = ei * (T(1) / mi) * (T(-1) + w * zp * wct * (T(1) - d / mi));
        t_w = ei * zp * (d / mi) * wct;
        t_Z = ei * w * (d / mi) * wct;
      } else {
        const T ej = exp(-d / mj);
        t = ej * (w * zp * (d / mi) * wct);
        t_tau = (ej * w * zp / mi) * (T(1) - d / mj) * wct;
        t_w = (w == T(0)) ? T(0) : t / w;
        t_Z = ej * w * (d / mi) * wct;
      }
    } else {
      const T g2 = (mj / (mi - mj)) * wct;
      const T ed = expdiff_neg<T>(d / mi, d / mj);
      t = w * zp * g2 * ed;
      t_tau = -w * zp * g2 * (exp(-d / mi) / mi - exp(-d / mj) / mj);
      t_w = (w == T(0)) ? T(0) : t / w;
      t_Z = w * g2 * ed;
    }
  } else if (i == j) {
    t = exp(-d / mi);
    t_tau = -t / mi;
  }
  const long long o = (long long)s * N * N + e;
  if (ndoubl < 1) {
    r_mp[o] = r;
    t_pp[o] = t;
    r_pm[o] = dsign * r;
    t_mm[o] = dsign * t;
  } else {
    r_mp[o] = ui ? -r : r;
    t_pp[o] = t;
  }
  const long long pstride = (long long)N * N * S;
  for (int p = 0; p < P; ++p) {
    const T td = dtau_dot[s + (long long)S * p], wd = varpi_dot[s + (long long)S * p];
    T zmd, zpd;
    if constexpr (MIX) {
      const T* cf = zx.zdcoef + (long long)zx.CT * (p + (long long)P * s);
      zmd = zpd = T(0);
#pragma unroll
      for (int c = 0; c < VSM_LIN_CT_MAX; ++c)
        if (c < zx.CT) {
          zpd += cf[c] * zcp[c];
          zmd += cf[c] * zcm[c];
        }
    } else {
      zmd = Zmp_dot ? Zmp_dot[(long long)s * zds + (long long)p * zdp + e] : T(0);
      zpd = Zpp_dot ? Zpp_dot[(long long)s * zds + (long long)p * zdp + e] : T(0);
    }
    const T vr = r_tau * td + r_w * wd + r_Z * zmd;
    const T vt = t_tau * td + t_w * wd + t_Z * zpd;
    ap_r_mp[o + p * pstride] = sign_r * vr;
    ap_t_pp[o + p * pstride] = vt;
    if (ndoubl < 1) {
      ap_r_pm[o + p * pstride] = dsign * (r_tau * td + r_w * wd) + r_Z * zmd;
      ap_t_mm[o + p * pstride] = dsign * (t_tau * td + t_w * wd) + t_Z * zpd;
    }
  }
}

// get_elem_rt_SFI_fused! (elemental_lin.jl:602-712)
template <typename T, bool MIX>
__global__ __launch_bounds__(256) void k_elemental_sfi_lin(int N, int ns, int S, int m_rest, int n_m0, int ndoubl, int i_mu0, int P,
                                                           const T* __restrict__ dtau, const T* __restrict__ varpi,
                                                           const T* __restrict__ tau_sum, const T* __restrict__ F0,
                                                           const T* __restrict__ Zpp, const T* __restrict__ Zmp,
                                                           long long zs, const T* __restrict__ dtau_dot,
                                                           const T* __restrict__ varpi_dot,
                                                           const T* __restrict__ tau_sum_dot,
                                                           const T* __restrict__ Zpp_dot, const T* __restrict__ Zmp_dot,
                                                           long long zds, long long zdp, const zmix<T> zx,
                                                           const T* __restrict__ mu, T* j0_p, T* j0_m, T* ap_J0_p,
                                                           T* ap_J0_m) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * S) return;
  const int i = e % N, s = e / N;
  const int m = (s < n_m0) ? 0 : m_rest;
  const int i_start = ns * i_mu0;
  const T wct02 = (m == 0) ? T(0.5) : T(0.25);
  T zp = 0, zm = 0;
  T zcp[MIX ? VSM_LIN_CT_MAX : 1], zcm[MIX ? VSM_LIN_CT_MAX : 1];   // (Zc[c] F0)_i per component
  if constexpr (MIX) {
#pragma unroll
    for (int c = 0; c < VSM_LIN_CT_MAX; ++c) {
      zcp[c] = zcm[c] = T(0);
      if (c < zx.CT)
        for (int q = 0; q < ns; ++q) {
          const long long zo = (long long)c * N * N + i + (long long)N * (i_start + q);
          const T f = F0[q + (long long)ns * s];
          zcp[c] += zx.Zc_pp[zo] * f;
          zcm[c] += zx.Zc_mp[zo] * f;
        }
    }
#pragma unroll
    for (int c = 0; c < VSM_LIN_CT_MAX; ++c) {
      T f = T(0);
      if (c < zx.C) f = (zx.zsel < 0) ? zx.fz[c + (long long)zx.C * s] : (c == zx.zsel ? T(1) : T(0));
      zp += f * zcp[c];
      zm += f * zcm[c];
    }
  } else {
    for (int q = 0; q < ns; ++q) {
      const long long zo = (long long)s * zs + i + (long long)N * (i_start + q);
      const T f = F0[q + (long long)ns * s];
      zp += Zpp[zo] * f;
      zm += Zmp[zo] * f;
    }
  }
  const T d = dtau[s], w = varpi[s];
  const T mi = mu[i], ms = mu[i_start];
  T jp, jp_tau;
  if (i >= i_start && i < i_start + ns) {
    jp = wct02 * w * zp * (d / mi) * exp(-d / mi);
    jp_tau = jp * (T(1) / d - T(1) / mi);
  } else {
    jp = wct02 * w * zp * (ms / (mi - ms)) * expdiff_neg<T>(d / mi, d / ms);
    jp_tau = -wct02 * w * zp * (ms / (mi - ms)) * (exp(-d / mi) / mi - exp(-d / ms) / ms);
  }
  T jp_w = (w == T(0)) ? T(0) : jp / w;
  T jp_Z = (zp == T(0)) ? T(0) : jp / zp;
  const T gm = ms / (mi + ms);
  const T arg = d * ((T(1) / mi) + (T(1) / ms));
  T jm = wct02 * w * zm * gm * (-expm1(-arg));
  T jm_tau = wct02 * w * zm * gm * exp(-arg) * ((T(1) / mi) + (T(1) / ms));
  T jm_w = (w == T(0)) ? T(0) : jm / w;
  T jm_Z = (zm == T(0)) ? T(0) : jm / zm;
  const T att = exp(-tau_sum[s] / ms);
  jp *= att; jp_tau *= att; jp_w *= att; jp_Z *= att;
  jm *= att; jm_tau *= att; jm_w *= att; jm_Z *= att;
  if (ndoubl >= 1 && is_uv_row(i, ns)) {
    jm = -jm; jm_tau = -jm_tau; jm_w = -jm_w; jm_Z = -jm_Z;
  }
  j0_p[e] = jp;
  j0_m[e] = jm;
  for (int p = 0; p < P; ++p) {
    T zpd = 0, zmd = 0;
    if constexpr (MIX) {
      const T* cf = zx.zdcoef + (long long)zx.CT * (p + (long long)P * s);
#pragma unroll
      for (int c = 0; c < VSM_LIN_CT_MAX; ++c)
        if (c < zx.CT) {
          zpd += cf[c] * zcp[c];
          zmd += cf[c] * zcm[c];
        }
    } else if (Zpp_dot) {
      for (int q = 0; q < ns; ++q) {
        const long long zo = (long long)s * zds + (long long)p * zdp + i + (long long)N * (i_start + q);
        const T f = F0[q + (long long)ns * s];
        zpd += Zpp_dot[zo] * f;
        zmd += Zmp_dot[zo] * f;
      }
    }
    const T td = dtau_dot[s + (long long)S * p], wd = varpi_dot[s + (long long)S * p];
    const T beam = -tau_sum_dot[s + (long long)S * p] / ms;
    ap_J0_p[e + (long long)N * S * p] = jp_tau * td + jp_w * wd + jp_Z * zpd + jp * beam;
    ap_J0_m[e + (long long)N * S * p] = jm_tau * td + jm_w * wd + jm_Z * zmd + jm * beam;
  }
}

template <typename T, bool MIX>
static int elemental_lin_impl(const quad<T>& q, int S, int m, int n_m0, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                              const T* F0, const T* Zpp, const T* Zmp, long long zs, int p_layer, const T* dtau_dot,
                              const T* varpi_dot, const T* tau_sum_dot, const T* Zpp_dot, const T* Zmp_dot, long long zds,
                              long long zdp, const zmix<T>& zx, const added<T>& a, const added_lin<T>& al, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int N = q.N;
  // zero every derivative slot first (elemental_lin.jl:128-141); slots >= p_layer stay zero
  const size_t mb = sizeof(T) * (size_t)N * N * S * al.P, vb = sizeof(T) * (size_t)N * S * al.P;
  VSM_HIP(hipMemsetAsync(al.ap_r_mp, 0, mb, st));
  VSM_HIP(hipMemsetAsync(al.ap_t_pp, 0, mb, st));
  VSM_HIP(hipMemsetAsync(al.ap_r_pm, 0, mb, st));
  VSM_HIP(hipMemsetAsync(al.ap_t_mm, 0, mb, st));
  VSM_HIP(hipMemsetAsync(al.ap_J0_p, 0, vb, st));
  VSM_HIP(hipMemsetAsync(al.ap_J0_m, 0, vb, st));
  hipLaunchKernelGGL((k_elemental_lin<T, MIX>), dim3(S, (N * N + 255) / 256), dim3(256), 0, st, N, q.n_stokes, S, m, n_m0, ndoubl,
                     p_layer, dtau, varpi, Zpp, Zmp, zs, dtau_dot, varpi_dot, Zpp_dot, Zmp_dot, zds, zdp, zx, q.mu, q.wt,
                     a.r_mp, a.t_pp, a.r_pm, a.t_mm, al.ap_r_mp, al.ap_t_pp, al.ap_r_pm, al.ap_t_mm);
  VSM_LAUNCH_CHECK("k_elemental_lin");
  hipLaunchKernelGGL((k_elemental_sfi_lin<T, MIX>), dim3((N * S + 255) / 256), dim3(256), 0, st, N, q.n_stokes, S, m, n_m0, ndoubl,
                     q.i_mu0, p_layer, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, dtau_dot, varpi_dot, tau_sum_dot,
                     Zpp_dot, Zmp_dot, zds, zdp, zx, q.mu, a.j0_p, a.j0_m, al.ap_J0_p, al.ap_J0_m);
  VSM_LAUNCH_CHECK("k_elemental_sfi_lin");
  return VSM_OK;
}

template <typename T>
int elemental_lin(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                  const T* F0, const T* Zpp, const T* Zmp, long long zs, int p_layer, const T* dtau_dot,
                  const T* varpi_dot, const T* tau_sum_dot, const T* Zpp_dot, const T* Zmp_dot, long long zds,
                  long long zdp, const added<T>& a, const added_lin<T>& al, hipStream_t st, int n_m0) {
  const zmix<T> none = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  return elemental_lin_impl<T, false>(q, S, m, n_m0, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, p_layer, dtau_dot, varpi_dot,
                                      tau_sum_dot, Zpp_dot, Zmp_dot, zds, zdp, none, a, al, st);
}

template <typename T>
int elemental_lin_mix(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                      const T* F0, int ncomp, int ncomp_total, const T* Zc_pp, const T* Zc_mp, int zsel, const T* fz,
                      int p_layer, const T* dtau_dot, const T* varpi_dot, const T* tau_sum_dot, const T* zdcoef,
                      const added<T>& a, const added_lin<T>& al, hipStream_t st) {
  if (ncomp_total > VSM_LIN_CT_MAX) {
    set_error("elemental_lin_mix: %d component blocks (limit %d: three aerosols with derivatives)", ncomp_total,
              VSM_LIN_CT_MAX);
    return VSM_ERR_UNSUPPORTED;
  }
  const zmix<T> zx = {Zc_pp, Zc_mp, fz, zdcoef, ncomp, ncomp_total, zsel};
  return elemental_lin_impl<T, true>(q, S, m, 0, ndoubl, dtau, varpi, tau_sum, F0, nullptr, nullptr, 0, p_layer, dtau_dot,
                                     varpi_dot, tau_sum_dot, nullptr, nullptr, 0, 0, zx, a, al, st);
}

// ---------------------------------------------------------------------------
// doubling, all parameters (doubling_lin.jl:216-339)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_dbl_src_prep(int N, int S, int P, const T* __restrict__ expk, T* ekl /*[S,P]*/,
                               const T* __restrict__ jp, const T* __restrict__ jm, const T* __restrict__ aJp,
                               const T* __restrict__ aJm, T* J1p, T* J1m, T* aJ1p, T* aJ1m) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)N * S) return;
  const int s = (int)(e / N);
  const T k = expk[s];
  const T vjp = jp[e], vjm = jm[e];
  J1p[e] = vjp * k;
  J1m[e] = vjm * k;
  for (int p = 0; p < P; ++p) {
    const T kl = ekl[s + (long long)S * p];
    const long long o = e + (long long)N * S * p;
    aJ1p[o] = aJp[o] * k + vjp * kl;
    aJ1m[o] = aJm[o] * k + vjm * kl;
  }
}
template <typename T>
__global__ void k_ekl_init(int S, int P, const T* __restrict__ expk, const T* __restrict__ dtau_dot, T mu0, T* ekl) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S * P) return;
  ekl[e] = -expk[e % S] / mu0 * dtau_dot[e];
}
template <typename T>
__global__ void k_ekl_step(int S, int P, const T* __restrict__ expk, T* ekl) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= S * P) return;
  ekl[e] = T(2) * expk[e % S] * ekl[e];
}
template <typename T>
__global__ void k_square_lin(int S, T* x) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < S) x[e] = x[e] * x[e];
}
// apply_D! / apply_D_SFI! with derivative slots (doubling_lin.jl:374-421)
template <typename T>
__global__ void k_apply_D_lin(int N, int ns, int S, int P, T* r_mp, const T* __restrict__ t_pp, T* r_pm, T* t_mm,
                              T* j0_m, T* ar, const T* __restrict__ at, T* arpm, T* atmm, T* aJm) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const int i = e % N, j = e / N;
  const long long o = (long long)s * N * N + e;
  const bool ui = is_uv_row(i, ns), uj = is_uv_row(j, ns);
  const T sg = (ui == uj) ? T(1) : T(-1);
  T r = r_mp[o];
  if (ui) r = -r;
  r_mp[o] = r;
  r_pm[o] = sg * r;
  t_mm[o] = sg * t_pp[o];
  const long long ps = (long long)N * N * S;
  for (int p = 0; p < P; ++p) {
    T x = ar[o + p * ps];
    if (ui) x = -x;
    ar[o + p * ps] = x;
    arpm[o + p * ps] = sg * x;
    atmm[o + p * ps] = sg * at[o + p * ps];
  }
  if (j == 0 && ui) {
    const long long v = (long long)s * N + i;
    j0_m[v] = -j0_m[v];
    for (int p = 0; p < P; ++p) aJm[v + (long long)N * S * p] = -aJm[v + (long long)N * S * p];
  }
}

template <typename T>
size_t doubling_lin_work_elems(int N, int S, int P) {
  return (size_t)N * N * S * (4 + 4 * (size_t)P) + (size_t)N * S * (6 + 5 * (size_t)P) + (size_t)S * P;
}
template size_t doubling_lin_work_elems<double>(int, int, int);
template size_t doubling_lin_work_elems<float>(int, int, int);

template <typename T>
int doubling_lin(int N, int ns, int S, int ndoubl, T* expk, const T* dtau_dot_all, T mu0, int n_active,
                 const added<T>& a, const added_lin<T>& al, T* work, hipStream_t st) {
  if (ndoubl == 0 || S <= 0) return VSM_OK;  // doubling_lin.jl:236
  const int Pall = al.P, P = n_active > 0 ? n_active : Pall;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  T* G = work;            // forward scratch
  T* tt = G + MS;
  T* W3 = tt + MS;
  T* rt = W3 + MS;
  T* X1 = rt + MS;        // per-parameter scratch [N,N,S,P]
  T* Gl = X1 + MS * P;
  T* ttl = Gl + MS * P;
  T* Q = ttl + MS * P;
  T* J1p = Q + MS * P;    // vectors
  T* J1m = J1p + VS;
  T* Av = J1m + VS;
  T* Bv = Av + VS;
  T* jmn = Bv + VS;
  T* jpn = jmn + VS;
  T* aJ1p = jpn + VS;
  T* aJ1m = aJ1p + VS * P;
  T* vv = aJ1m + VS * P;
  T* uv = vv + VS * P;
  T* tmpv = uv + VS * P;
  T* ekl = tmpv + VS * P;  // [S,P]
  const T one = T(1), zero = T(0);
  const T* nul = nullptr;
  int rc;
  T* ar = al.ap_r_mp;
  T* at = al.ap_t_pp;
  T* aJp = al.ap_J0_p;
  T* aJm = al.ap_J0_m;
#define MM(...) if ((rc = gemm2<T>(__VA_ARGS__, st))) return rc
  hipLaunchKernelGGL(k_ekl_init<T>, dim3((S * P + 255) / 256), dim3(256), 0, st, S, P, expk, dtau_dot_all, mu0, ekl);
  VSM_LAUNCH_CHECK("k_ekl_init");
  int n0 = 0;
  if constexpr (std::is_same<T, double>::value) {
    // one active parameter: all steps in one launch (state stays on the chip between the steps)
    // (apply_D! rides in its epilogue when every parameter slot is either active there or zero: P == n_active or the
    //  remaining slots' derivatives of r, t vanish in this layer, which the caller states with n_active > 0)
    static const bool sepD = ab_switch("VSM_LIN_SEPARATE_APPLY_D");
    rc = strip128_lin_dbl_supported(N) ? (int)VSM_ERR_UNSUPPORTED : strip_doubling_lin_multi(N, S, P, ndoubl, sepD ? 0 : ns, expk, ekl, a, al, st);
    // 60 < N <= 128, any number of active parameters: all steps in one persistent launch (vsm_strip128lin.hip)
    if (rc == VSM_ERR_UNSUPPORTED && !sepD) rc = strip128_doubling_lin(N, S, P, ndoubl, ns, expk, ekl, a, al, st);
    if (rc == VSM_OK && !sepD) return VSM_OK;
    if (rc == VSM_OK) n0 = ndoubl;
    else if (rc != VSM_ERR_UNSUPPORTED) return rc;
    // fused column-strip step (vsm_striplin.hip): one launch per doubling step, forward + all active parameters
    for (; n0 < ndoubl; ++n0) {
      rc = strip_doubling_lin_step(N, S, P, expk, ekl, a, al, st);
      if (rc == VSM_ERR_UNSUPPORTED && n0 == 0) break;
      if (rc) return rc;
    }
  } else {
    // Float32 (N <= 128): the FP64 kernels of vsm_strip128lin.hip on the FP32 arrays -- storage in single, arithmetic in double:
    // their FP64 MFMA rate is above what the FP32 operator chain below reaches (N = 96: 28.6 -> 41 TFLOP/s)
    rc = strip128_doubling_lin<T>(N, S, P, ndoubl, ns, expk, ekl, a, al, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  for (int n = n0; n < ndoubl; ++n) {
    // forward: G = (I - r r)^-1, tt = t G
    if ((rc = inv_one_minus<T>(N, S, a.r_mp, NN, a.r_mp, NN, G, G, st))) return rc;   // one fused launch when N fits on chip
    MM(N, N, N, S, 1, a.t_pp, NN, 0, G, NN, 0, tt, NN, 0, one, nul, 0, 0, zero, zero);
    // Gl_p = G (ar_p r + r ar_p) G ; ttl_p = at_p G + t Gl_p
    MM(N, N, N, S, P, ar, NN, MS, a.r_mp, NN, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, a.r_mp, NN, 0, ar, NN, MS, X1, NN, MS, one, X1, NN, MS, one, zero);
    MM(N, N, N, S, P, G, NN, 0, X1, NN, MS, Q, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, Q, NN, MS, G, NN, 0, Gl, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, at, NN, MS, G, NN, 0, ttl, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, a.t_pp, NN, 0, Gl, NN, MS, ttl, NN, MS, one, ttl, NN, MS, one, zero);
    // sources
    hipLaunchKernelGGL(k_dbl_src_prep<T>, dim3((unsigned)((VS + 255) / 256)), dim3(256), 0, st, N, S, P, expk, ekl,
                       a.j0_p, a.j0_m, aJp, aJm, J1p, J1m, aJ1p, aJ1m);
    VSM_LAUNCH_CHECK("k_dbl_src_prep");
    hipLaunchKernelGGL(k_ekl_step<T>, dim3((S * P + 255) / 256), dim3(256), 0, st, S, P, expk, ekl);
    VSM_LAUNCH_CHECK("k_ekl_step");
    MM(N, 1, N, S, 1, a.r_mp, NN, 0, a.j0_p, N, 0, Av, N, 0, one, J1m, N, 0, one, zero);   // A = J1- + r j0+
    MM(N, 1, N, S, 1, a.r_mp, NN, 0, J1m, N, 0, Bv, N, 0, one, a.j0_p, N, 0, one, zero);  // B = j0+ + r J1-
    // v_p = aJ1-_p + ar_p j0+ + r aJ+_p ; u_p = aJ+_p + ar_p J1- + r aJ1-_p    (old aJ+)
    MM(N, 1, N, S, P, ar, NN, MS, a.j0_p, N, 0, vv, N, VS, one, aJ1m, N, VS, one, zero);
    MM(N, 1, N, S, P, a.r_mp, NN, 0, aJp, N, VS, vv, N, VS, one, vv, N, VS, one, zero);
    MM(N, 1, N, S, P, ar, NN, MS, J1m, N, 0, uv, N, VS, one, aJp, N, VS, one, zero);
    MM(N, 1, N, S, P, a.r_mp, NN, 0, aJ1m, N, VS, uv, N, VS, one, uv, N, VS, one, zero);
    // aJ-_p += ttl_p A + tt v_p
    MM(N, 1, N, S, P, ttl, NN, MS, Av, N, 0, aJm, N, VS, one, aJm, N, VS, one, zero);
    MM(N, 1, N, S, P, tt, NN, 0, vv, N, VS, aJm, N, VS, one, aJm, N, VS, one, zero);
    // aJ+_p = aJ1+_p + ttl_p B + tt u_p
    MM(N, 1, N, S, P, ttl, NN, MS, Bv, N, 0, tmpv, N, VS, one, aJ1p, N, VS, one, zero);
    MM(N, 1, N, S, P, tt, NN, 0, uv, N, VS, aJp, N, VS, one, tmpv, N, VS, one, zero);
    // forward sources
    MM(N, 1, N, S, 1, tt, NN, 0, Av, N, 0, jmn, N, 0, one, a.j0_m, N, 0, one, zero);
    MM(N, 1, N, S, 1, tt, NN, 0, Bv, N, 0, jpn, N, 0, one, J1p, N, 0, one, zero);
    if ((rc = copy_strided<T>(VS, 1, jmn, 0, a.j0_m, st))) return rc;
    if ((rc = copy_strided<T>(VS, 1, jpn, 0, a.j0_p, st))) return rc;
    hipLaunchKernelGGL(k_square_lin<T>, dim3((S + 255) / 256), dim3(256), 0, st, S, expk);
    VSM_LAUNCH_CHECK("k_square_lin");
    // rt = r t ; Q_p = ar_p t + r at_p
    MM(N, N, N, S, 1, a.r_mp, NN, 0, a.t_pp, NN, 0, rt, NN, 0, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, ar, NN, MS, a.t_pp, NN, 0, Q, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, a.r_mp, NN, 0, at, NN, MS, Q, NN, MS, one, Q, NN, MS, one, zero);
    // ar_p += ttl_p rt + tt Q_p
    MM(N, N, N, S, P, ttl, NN, MS, rt, NN, 0, ar, NN, MS, one, ar, NN, MS, one, zero);
    MM(N, N, N, S, P, tt, NN, 0, Q, NN, MS, ar, NN, MS, one, ar, NN, MS, one, zero);
    // at_p = ttl_p t + tt at_p
    MM(N, N, N, S, P, tt, NN, 0, at, NN, MS, X1, NN, MS, one, nul, 0, 0, zero, zero);
    MM(N, N, N, S, P, ttl, NN, MS, a.t_pp, NN, 0, at, NN, MS, one, X1, NN, MS, one, zero);
    // forward r, t
    MM(N, N, N, S, 1, tt, NN, 0, rt, NN, 0, a.r_mp, NN, 0, one, a.r_mp, NN, 0, one, zero);
    MM(N, N, N, S, 1, tt, NN, 0, a.t_pp, NN, 0, W3, NN, 0, one, nul, 0, 0, zero, zero);
    if ((rc = copy_strided<T>(MS, 1, W3, 0, a.t_pp, st))) return rc;
  }
#undef MM
  hipLaunchKernelGGL(k_apply_D_lin<T>, dim3(S, (unsigned)((NN + 255) / 256)), dim3(256), 0, st, N, ns, S, Pall, a.r_mp,
                     a.t_pp, a.r_pm, a.t_mm, a.j0_m, al.ap_r_mp, al.ap_t_pp, al.ap_r_pm, al.ap_t_mm, al.ap_J0_m);
  VSM_LAUNCH_CHECK("k_apply_D_lin");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// interaction, ScatteringInterface_11 (interaction_lin.jl:217-331)
// ---------------------------------------------------------------------------
template <typename T>
size_t interaction_lin_work_elems(int N, int S, int P) {
  return (size_t)N * N * S * (8 + 6 * (size_t)P) + (size_t)N * S * (6 + 3 * (size_t)P);
}
template size_t interaction_lin_work_elems<double>(int, int, int);
template size_t interaction_lin_work_elems<float>(int, int, int);

template <typename T>
int interaction_lin(int iface, int N, int S, const composite<T>& c, const composite_lin<T>& cl, const added<T>& a,
                    const added_lin<T>& al, T* work, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (iface < VSM_IFACE_00 || iface > VSM_IFACE_11) {
    set_error("interaction_lin: unknown scattering interface %d", iface);
    return VSM_ERR_INVALID_ARG;
  }
  const int P = cl.P;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  const long long as = a.mat_stride;               // forward added matrices: 0 = shared surface block
  const long long als = al.mat_stride;             // derivative added matrices: stride over S (0 = shared)
  const long long alp = (als == 0) ? NN : MS;      // ... and over parameters
  T* G = work;
  T* T01 = G + MS;
  T* rT = T01 + MS;
  T* nRmp = rT + MS;
  T* nTmm = nRmp + MS;
  T* T21 = nTmm + MS;
  T* Rt = T21 + MS;
  T* W = Rt + MS;
  T* X1 = W + MS;          // [.,P]
  T* X2 = X1 + MS * P;
  T* TL = X2 + MS * P;     // T01_lin / T21_lin
  T* nA = TL + MS * P;     // new Rdot-+ then new Tdot++
  T* nB = nA + MS * P;     // new Tdot-- then new Rdot+-
  T* X3 = nB + MS * P;
  T* Av = X3 + MS * P;     // vectors
  T* Bv = Av + VS;
  T* nJm = Bv + VS;
  T* nJp = nJm + VS;
  T* v1 = nJp + VS;
  T* v2 = v1 + VS;
  T* nJml = v2 + VS;       // [.,P]
  T* nJpl = nJml + VS * P;
  T* wv = nJpl + VS * P;
  const T one = T(1), zero = T(0);
  const T* nul = nullptr;
  int rc;
#define MM(...) if ((rc = gemm2<T>(__VA_ARGS__, st))) return rc
#define CP(n, src, dst) if ((rc = copy_strided<T>(n, 1, src, 0, dst, st))) return rc
  if (iface != VSM_IFACE_11) {
    // ---- interfaces without multiple reflections between the two layers (interaction_lin.jl:62-215).  Every new
    // value goes to scratch first and is committed at the end, so all right-hand sides see pre-update values.
    if (iface == VSM_IFACE_00) {            // :62-96
      MM(N, 1, N, S, P, a.t_pp, as, 0, cl.J0_p, N, VS, nJpl, N, VS, one, al.ap_J0_p, N, VS, one, zero);
      MM(N, 1, N, S, P, al.ap_t_pp, als, alp, c.J0_p, N, 0, nJpl, N, VS, one, nJpl, N, VS, one, zero);
      MM(N, 1, N, S, P, c.T_mm, NN, 0, al.ap_J0_m, N, VS, nJml, N, VS, one, cl.J0_m, N, VS, one, zero);
      MM(N, 1, N, S, P, cl.T_mm, NN, MS, a.j0_m, N, 0, nJml, N, VS, one, nJml, N, VS, one, zero);
      MM(N, 1, N, S, 1, a.t_pp, as, 0, c.J0_p, N, 0, nJp, N, 0, one, a.j0_p, N, 0, one, zero);
      MM(N, 1, N, S, 1, c.T_mm, NN, 0, a.j0_m, N, 0, nJm, N, 0, one, c.J0_m, N, 0, one, zero);
      MM(N, N, N, S, P, al.ap_t_mm, als, alp, c.T_mm, NN, 0, nB, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.t_mm, as, 0, cl.T_mm, NN, MS, nB, NN, MS, one, nB, NN, MS, one, zero);      // new Tdot--
      MM(N, N, N, S, P, al.ap_t_pp, als, alp, c.T_pp, NN, 0, nA, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.t_pp, as, 0, cl.T_pp, NN, MS, nA, NN, MS, one, nA, NN, MS, one, zero);      // new Tdot++
      MM(N, N, N, S, 1, a.t_mm, as, 0, c.T_mm, NN, 0, nTmm, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, 1, a.t_pp, as, 0, c.T_pp, NN, 0, W, NN, 0, one, nul, 0, 0, zero, zero);
      CP(MS * P, nB, cl.T_mm);
      CP(MS * P, nA, cl.T_pp);
      CP(MS, nTmm, c.T_mm);
      CP(MS, W, c.T_pp);
    } else if (iface == VSM_IFACE_01) {     // :104-147
      MM(N, 1, N, S, 1, a.r_mp, as, 0, c.J0_p, N, 0, Av, N, 0, one, a.j0_m, N, 0, one, zero);
      MM(N, 1, N, S, P, al.ap_r_mp, als, alp, c.J0_p, N, 0, wv, N, VS, one, al.ap_J0_m, N, VS, one, zero);
      MM(N, 1, N, S, P, a.r_mp, as, 0, cl.J0_p, N, VS, wv, N, VS, one, wv, N, VS, one, zero);
      MM(N, 1, N, S, P, cl.T_mm, NN, MS, Av, N, 0, nJml, N, VS, one, cl.J0_m, N, VS, one, zero);
      MM(N, 1, N, S, P, c.T_mm, NN, 0, wv, N, VS, nJml, N, VS, one, nJml, N, VS, one, zero);
      MM(N, 1, N, S, P, al.ap_t_pp, als, alp, c.J0_p, N, 0, nJpl, N, VS, one, al.ap_J0_p, N, VS, one, zero);
      MM(N, 1, N, S, P, a.t_pp, as, 0, cl.J0_p, N, VS, nJpl, N, VS, one, nJpl, N, VS, one, zero);
      MM(N, 1, N, S, 1, c.T_mm, NN, 0, Av, N, 0, nJm, N, 0, one, c.J0_m, N, 0, one, zero);
      MM(N, 1, N, S, 1, a.t_pp, as, 0, c.J0_p, N, 0, nJp, N, 0, one, a.j0_p, N, 0, one, zero);
      MM(N, N, N, S, 1, a.r_mp, as, 0, c.T_pp, NN, 0, rT, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, al.ap_r_mp, als, alp, c.T_pp, NN, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.r_mp, as, 0, cl.T_pp, NN, MS, X1, NN, MS, one, X1, NN, MS, one, zero);
      MM(N, N, N, S, P, cl.T_mm, NN, MS, rT, NN, 0, nA, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, c.T_mm, NN, 0, X1, NN, MS, nA, NN, MS, one, nA, NN, MS, one, zero);           // new Rdot-+
      MM(N, N, N, S, P, al.ap_t_pp, als, alp, c.T_pp, NN, 0, nB, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.t_pp, as, 0, cl.T_pp, NN, MS, nB, NN, MS, one, nB, NN, MS, one, zero);      // new Tdot++
      MM(N, N, N, S, P, cl.T_mm, NN, MS, a.t_mm, as, 0, X2, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, c.T_mm, NN, 0, al.ap_t_mm, als, alp, X2, NN, MS, one, X2, NN, MS, one, zero); // new Tdot--
      MM(N, N, N, S, 1, c.T_mm, NN, 0, rT, NN, 0, nRmp, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, 1, a.t_pp, as, 0, c.T_pp, NN, 0, W, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, 1, c.T_mm, NN, 0, a.t_mm, as, 0, nTmm, NN, 0, one, nul, 0, 0, zero, zero);
      CP(MS * P, nA, cl.R_mp);
      for (int p = 0; p < P; ++p)
        if ((rc = copy_strided<T>(NN, S, al.ap_r_pm + (long long)p * alp, als, cl.R_pm + (long long)p * MS, st))) return rc;
      CP(MS * P, nB, cl.T_pp);
      CP(MS * P, X2, cl.T_mm);
      CP(MS, nRmp, c.R_mp);
      if ((rc = copy_strided<T>(NN, S, a.r_pm, as, c.R_pm, st))) return rc;
      CP(MS, W, c.T_pp);
      CP(MS, nTmm, c.T_mm);
    } else {                                // ScatteringInterface_10, :155-215
      MM(N, 1, N, S, 1, c.R_pm, NN, 0, a.j0_m, N, 0, Bv, N, 0, one, c.J0_p, N, 0, one, zero);
      MM(N, 1, N, S, P, cl.R_pm, NN, MS, a.j0_m, N, 0, wv, N, VS, one, cl.J0_p, N, VS, one, zero);
      MM(N, 1, N, S, P, c.R_pm, NN, 0, al.ap_J0_m, N, VS, wv, N, VS, one, wv, N, VS, one, zero);
      MM(N, 1, N, S, P, al.ap_t_pp, als, alp, Bv, N, 0, nJpl, N, VS, one, al.ap_J0_p, N, VS, one, zero);
      MM(N, 1, N, S, P, a.t_pp, as, 0, wv, N, VS, nJpl, N, VS, one, nJpl, N, VS, one, zero);
      MM(N, 1, N, S, P, cl.T_mm, NN, MS, a.j0_m, N, 0, nJml, N, VS, one, cl.J0_m, N, VS, one, zero);
      MM(N, 1, N, S, P, c.T_mm, NN, 0, al.ap_J0_m, N, VS, nJml, N, VS, one, nJml, N, VS, one, zero);
      MM(N, 1, N, S, 1, a.t_pp, as, 0, Bv, N, 0, nJp, N, 0, one, a.j0_p, N, 0, one, zero);
      MM(N, 1, N, S, 1, c.T_mm, NN, 0, a.j0_m, N, 0, nJm, N, 0, one, c.J0_m, N, 0, one, zero);
      MM(N, N, N, S, P, al.ap_t_pp, als, alp, c.T_pp, NN, 0, nA, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.t_pp, as, 0, cl.T_pp, NN, MS, nA, NN, MS, one, nA, NN, MS, one, zero);      // new Tdot++
      MM(N, N, N, S, P, cl.T_mm, NN, MS, a.t_mm, as, 0, X2, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, c.T_mm, NN, 0, al.ap_t_mm, als, alp, X2, NN, MS, one, X2, NN, MS, one, zero); // new Tdot--
      MM(N, N, N, S, 1, c.R_pm, NN, 0, a.t_mm, as, 0, Rt, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, cl.R_pm, NN, MS, a.t_mm, as, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, c.R_pm, NN, 0, al.ap_t_mm, als, alp, X1, NN, MS, one, X1, NN, MS, one, zero);
      MM(N, N, N, S, P, al.ap_t_pp, als, alp, Rt, NN, 0, nB, NN, MS, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, P, a.t_pp, as, 0, X1, NN, MS, nB, NN, MS, one, nB, NN, MS, one, zero);           // new Rdot+-
      MM(N, N, N, S, 1, a.t_pp, as, 0, c.T_pp, NN, 0, W, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, 1, c.T_mm, NN, 0, a.t_mm, as, 0, nTmm, NN, 0, one, nul, 0, 0, zero, zero);
      MM(N, N, N, S, 1, a.t_pp, as, 0, Rt, NN, 0, G, NN, 0, one, nul, 0, 0, zero, zero);
      CP(MS * P, nA, cl.T_pp);
      CP(MS * P, X2, cl.T_mm);
      CP(MS * P, nB, cl.R_pm);
      CP(MS, W, c.T_pp);
      CP(MS, nTmm, c.T_mm);
      CP(MS, G, c.R_pm);
    }
    CP(VS, nJp, c.J0_p);
    CP(VS, nJm, c.J0_m);
    CP(VS * P, nJpl, cl.J0_p);
    CP(VS * P, nJml, cl.J0_m);
    (void)v1;
    (void)v2;
    return VSM_OK;
  }
  if constexpr (std::is_same<T, double>::value) {
    // fused: two launches (first half, second half) instead of ~60, every N <= 128 (vsm_strip128lin.hip)
    rc = strip128_interaction11_lin(N, S, c, cl, a, al, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  } else {
    rc = strip128_interaction11_lin<T>(N, S, c, cl, a, al, st);     // Float32 arrays, FP64 arithmetic (see doubling_lin)
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  // ---- first half: G1, T01_inv and everything that hangs off them --------------------------------
  if ((rc = inv_one_minus<T>(N, S, a.r_mp, as, c.R_pm, NN, G, G, st))) return rc;
  MM(N, N, N, S, 1, c.T_mm, NN, 0, G, NN, 0, T01, NN, 0, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, 1, a.r_mp, as, 0, c.T_pp, NN, 0, rT, NN, 0, one, nul, 0, 0, zero, zero);
  // G1l_p = G (ar_p R+- + r Rdot+-_p) G
  MM(N, N, N, S, P, al.ap_r_mp, als, alp, c.R_pm, NN, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, a.r_mp, as, 0, cl.R_pm, NN, MS, X1, NN, MS, one, X1, NN, MS, one, zero);
  MM(N, N, N, S, P, G, NN, 0, X1, NN, MS, X2, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, X2, NN, MS, G, NN, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);          // X1 = G1l
  // T01l_p = Tdot--_p G + T-- G1l_p
  MM(N, N, N, S, P, cl.T_mm, NN, MS, G, NN, 0, TL, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, c.T_mm, NN, 0, X1, NN, MS, TL, NN, MS, one, TL, NN, MS, one, zero);
  // new Rdot-+_p = Rdot-+_p + T01l_p rT + T01 (ar_p T++ + r Tdot++_p)
  MM(N, N, N, S, P, al.ap_r_mp, als, alp, c.T_pp, NN, 0, X2, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, a.r_mp, as, 0, cl.T_pp, NN, MS, X2, NN, MS, one, X2, NN, MS, one, zero);
  MM(N, N, N, S, P, TL, NN, MS, rT, NN, 0, nA, NN, MS, one, cl.R_mp, NN, MS, one, zero);
  MM(N, N, N, S, P, T01, NN, 0, X2, NN, MS, nA, NN, MS, one, nA, NN, MS, one, zero);
  // new Tdot--_p = T01l_p t-- + T01 at--_p
  MM(N, N, N, S, P, TL, NN, MS, a.t_mm, as, 0, nB, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, T01, NN, 0, al.ap_t_mm, als, alp, nB, NN, MS, one, nB, NN, MS, one, zero);
  // sources (-): A = r J0+ + j0- ; new J0- ; new Jdot-_p
  MM(N, 1, N, S, 1, a.r_mp, as, 0, c.J0_p, N, 0, Av, N, 0, one, a.j0_m, N, 0, one, zero);
  MM(N, 1, N, S, 1, T01, NN, 0, Av, N, 0, nJm, N, 0, one, c.J0_m, N, 0, one, zero);
  MM(N, 1, N, S, P, al.ap_r_mp, als, alp, c.J0_p, N, 0, wv, N, VS, one, al.ap_J0_m, N, VS, one, zero);
  MM(N, 1, N, S, P, a.r_mp, as, 0, cl.J0_p, N, VS, wv, N, VS, one, wv, N, VS, one, zero);
  MM(N, 1, N, S, P, TL, NN, MS, Av, N, 0, nJml, N, VS, one, cl.J0_m, N, VS, one, zero);
  MM(N, 1, N, S, P, T01, NN, 0, wv, N, VS, nJml, N, VS, one, nJml, N, VS, one, zero);
  // new R-+, T--
  MM(N, N, N, S, 1, T01, NN, 0, rT, NN, 0, nRmp, NN, 0, one, c.R_mp, NN, 0, one, zero);
  MM(N, N, N, S, 1, T01, NN, 0, a.t_mm, as, 0, nTmm, NN, 0, one, nul, 0, 0, zero, zero);
  // commit the (-) derivative results that the second half does not read: Rdot-+ and Tdot-- are only inputs of
  // the first half; Tdot++ (input of the second half) is still the old one.
  if ((rc = copy_strided<T>(MS * P, 1, nA, 0, cl.R_mp, st))) return rc;
  if ((rc = copy_strided<T>(MS * P, 1, nB, 0, cl.T_mm, st))) return rc;
  // ---- second half: G2, T21_inv ---------------------------------------------------------------------
  if ((rc = inv_one_minus<T>(N, S, c.R_pm, NN, a.r_mp, as, G, G, st))) return rc;
  MM(N, N, N, S, 1, a.t_pp, as, 0, G, NN, 0, T21, NN, 0, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, 1, c.R_pm, NN, 0, a.t_mm, as, 0, Rt, NN, 0, one, nul, 0, 0, zero, zero);
  // G2l_p = G (R+- ar_p + Rdot+-_p r) G
  MM(N, N, N, S, P, c.R_pm, NN, 0, al.ap_r_mp, als, alp, X1, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, cl.R_pm, NN, MS, a.r_mp, as, 0, X1, NN, MS, one, X1, NN, MS, one, zero);
  MM(N, N, N, S, P, G, NN, 0, X1, NN, MS, X2, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, X2, NN, MS, G, NN, 0, X1, NN, MS, one, nul, 0, 0, zero, zero);          // X1 = G2l
  // T21l_p = at++_p G + t++ G2l_p
  MM(N, N, N, S, P, al.ap_t_pp, als, alp, G, NN, 0, TL, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, a.t_pp, as, 0, X1, NN, MS, TL, NN, MS, one, TL, NN, MS, one, zero);
  // new Tdot++_p = T21l_p T++ + T21 Tdot++_p
  MM(N, N, N, S, P, TL, NN, MS, c.T_pp, NN, 0, nA, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, T21, NN, 0, cl.T_pp, NN, MS, nA, NN, MS, one, nA, NN, MS, one, zero);
  // new Rdot+-_p = ar+-_p + T21l_p Rt + T21 (Rdot+-_p t-- + R+- at--_p)
  MM(N, N, N, S, P, cl.R_pm, NN, MS, a.t_mm, as, 0, X3, NN, MS, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, P, c.R_pm, NN, 0, al.ap_t_mm, als, alp, X3, NN, MS, one, X3, NN, MS, one, zero);
  MM(N, N, N, S, P, TL, NN, MS, Rt, NN, 0, nB, NN, MS, one, al.ap_r_pm, als, alp, one, zero);
  MM(N, N, N, S, P, T21, NN, 0, X3, NN, MS, nB, NN, MS, one, nB, NN, MS, one, zero);
  // sources (+): B = J0+ + R+- j0- ; new Jdot+_p ; new J0+
  MM(N, 1, N, S, 1, c.R_pm, NN, 0, a.j0_m, N, 0, Bv, N, 0, one, c.J0_p, N, 0, one, zero);
  MM(N, 1, N, S, P, cl.R_pm, NN, MS, a.j0_m, N, 0, wv, N, VS, one, cl.J0_p, N, VS, one, zero);
  MM(N, 1, N, S, P, c.R_pm, NN, 0, al.ap_J0_m, N, VS, wv, N, VS, one, wv, N, VS, one, zero);
  MM(N, 1, N, S, P, TL, NN, MS, Bv, N, 0, nJpl, N, VS, one, al.ap_J0_p, N, VS, one, zero);
  MM(N, 1, N, S, P, T21, NN, 0, wv, N, VS, nJpl, N, VS, one, nJpl, N, VS, one, zero);
  MM(N, 1, N, S, 1, T21, NN, 0, Bv, N, 0, nJp, N, 0, one, a.j0_p, N, 0, one, zero);
  // new T++ (-> W), new R+- (-> G is free now)
  MM(N, N, N, S, 1, T21, NN, 0, c.T_pp, NN, 0, W, NN, 0, one, nul, 0, 0, zero, zero);
  MM(N, N, N, S, 1, T21, NN, 0, Rt, NN, 0, G, NN, 0, one, a.r_pm, as, 0, one, zero);
  // ---- write back (interaction_lin.jl:306-330) ------------------------------------------------------------
  if ((rc = copy_strided<T>(VS, 1, nJp, 0, c.J0_p, st))) return rc;
  if ((rc = copy_strided<T>(VS, 1, nJm, 0, c.J0_m, st))) return rc;
  if ((rc = copy_strided<T>(VS * P, 1, nJpl, 0, cl.J0_p, st))) return rc;
  if ((rc = copy_strided<T>(VS * P, 1, nJml, 0, cl.J0_m, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, G, 0, c.R_pm, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, nTmm, 0, c.T_mm, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, nRmp, 0, c.R_mp, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, W, 0, c.T_pp, st))) return rc;
  if ((rc = copy_strided<T>(MS * P, 1, nB, 0, cl.R_pm, st))) return rc;
  if ((rc = copy_strided<T>(MS * P, 1, nA, 0, cl.T_pp, st))) return rc;
#undef MM
#undef CP
  (void)v1;
  (void)v2;
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// Lambertian surface with albedo derivative (lambertian_surface_lin.jl:48-162)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_lambertian_lin_mats(int N, int ns, int m, T albedo, int iparam, int P, const T* __restrict__ mu,
                                      const T* __restrict__ wt, T* r_mp, T* r_pm, T* t_pp, T* t_mm, T* ar, T* arpm,
                                      T* at, T* atmm) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int i = e % N, j = e / N;
  const bool ii = (m == 0) && (i % ns == 0) && (j % ns == 0);
  r_mp[e] = ii ? T(2) * albedo * (mu[j] * wt[j]) : T(0);
  r_pm[e] = T(0);
  t_pp[e] = (i == j) ? T(1) : T(0);
  t_mm[e] = T(0);                                  // (reference quirk: the linearized builder zeroes t--)
  for (int p = 0; p < P; ++p) {
    const long long o = e + (long long)N * N * p;
    ar[o] = (ii && p == iparam) ? T(2) * (mu[j] * wt[j]) : T(0);
    arpm[o] = T(0);
    at[o] = T(0);
    atmm[o] = T(0);
  }
}
template <typename T>
__global__ void k_lambertian_lin_src(int N, int ns, int S, int m, T albedo, int iparam, int P, int p_layer, int i_mu0,
                                     T mu0, const T* __restrict__ tau_sum, const T* __restrict__ tau_sum_dot,
                                     const T* __restrict__ F0, T* j0_p, T* j0_m, T* aJp, T* aJm) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * S) return;
  const int i = e % N, s = e / N;
  T jm = T(0), base = T(0);
  if (m == 0 && (i % ns) == 0) {
    base = mu0 * F0[(long long)ns * s] * exp(-tau_sum[s] / mu0);  // mu0 * (R_surf F0_N) without the albedo factor 2a
    jm = T(2) * albedo * base;
  }
  j0_p[e] = T(0);
  j0_m[e] = jm;
  (void)i_mu0;
  for (int p = 0; p < P; ++p) {
    const long long o = e + (long long)N * S * p;
    aJp[o] = T(0);
    T v = T(0);
    if (p < p_layer) v = -jm * tau_sum_dot[s + (long long)S * p] / mu0;
    if (p == iparam) v = T(2) * base;
    aJm[o] = v;
  }
}
template <typename T>
int lambertian_surface_lin(const quad<T>& q, int S, int m, T albedo, int iparam, const T* tau_sum,
                           const T* tau_sum_dot, int p_layer, const T* F0, const added<T>& a, const added_lin<T>& al,
                           hipStream_t st) {
  if (a.mat_stride != 0 || al.mat_stride != 0) {
    set_error("lambertian_surface_lin: surface layers use ONE shared block (mat_stride must be 0)");
    return VSM_ERR_INVALID_ARG;
  }
  const int N = q.N;
  hipLaunchKernelGGL(k_lambertian_lin_mats<T>, dim3((N * N + 255) / 256), dim3(256), 0, st, N, q.n_stokes, m, albedo,
                     iparam, al.P, q.mu, q.wt, a.r_mp, a.r_pm, a.t_pp, a.t_mm, al.ap_r_mp, al.ap_r_pm, al.ap_t_pp,
                     al.ap_t_mm);
  VSM_LAUNCH_CHECK("k_lambertian_lin_mats");
  if (S > 0) {
    hipLaunchKernelGGL(k_lambertian_lin_src<T>, dim3((N * S + 255) / 256), dim3(256), 0, st, N, q.n_stokes, S, m,
                       albedo, iparam, al.P, p_layer, q.i_mu0, q.mu0, tau_sum, tau_sum_dot, F0, a.j0_p, a.j0_m,
                       al.ap_J0_p, al.ap_J0_m);
    VSM_LAUNCH_CHECK("k_lambertian_lin_src");
  }
  return VSM_OK;
}

template <typename T>
int copy_added_to_composite_lin(int N, int S, const added_lin<T>& al, const composite_lin<T>& cl, hipStream_t st) {
  const long long MS = (long long)N * N * S * al.P, VS = (long long)N * S * al.P;
  int rc;
  if (al.mat_stride == 0) {
    set_error("copy_added_to_composite_lin: shared (surface) derivative blocks cannot be the TOA layer");
    return VSM_ERR_INVALID_ARG;
  }
  if ((rc = copy_strided<T>(MS, 1, al.ap_t_pp, 0, cl.T_pp, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, al.ap_t_mm, 0, cl.T_mm, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, al.ap_r_mp, 0, cl.R_mp, st))) return rc;
  if ((rc = copy_strided<T>(MS, 1, al.ap_r_pm, 0, cl.R_pm, st))) return rc;
  if ((rc = copy_strided<T>(VS, 1, al.ap_J0_p, 0, cl.J0_p, st))) return rc;
  if ((rc = copy_strided<T>(VS, 1, al.ap_J0_m, 0, cl.J0_m, st))) return rc;
  return VSM_OK;
}

struct pp_args_lin {
  int row0[64];
  double w[256];
};
template <typename T>
__global__ void k_postprocess_lin(int N, int ns, int S, int nV, int nVtot, int v0, int P, pp_args_lin pa, const T* __restrict__ Jm,
                                  const T* __restrict__ Jp, T* Rd, T* Td) {
  // one chunk of <= 64 viewing angles [v0, v0 + nV) of the nVtot of the output arrays [P][S][ns][nVtot]
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)nV * ns * S;
  if (e >= per * P) return;
  const int p = (int)(e / per);
  const long long x = e % per;
  const int v = (int)(x % nV);
  const int k = (int)((x / nV) % ns);
  const long long s = x / ((long long)nV * ns);
  const T w = (T)pa.w[v + nV * k];
  const long long src = (long long)N * S * p + s * N + pa.row0[v] + k;
  const long long dst = (((long long)p * S + s) * ns + k) * nVtot + v0 + v;
  Rd[dst] += w * Jm[src];
  Td[dst] += w * Jp[src];
}
template <typename T>
int postprocess_vza_lin(int N, int ns, int S, int nV, int P, const int* row0_h, const T* w_h, const T* Jd_m,
                        const T* Jd_p, T* Rd, T* Td, hipStream_t st) {
  if (ns > 4) {
    set_error("postprocess_vza_lin: n_stokes <= 4 (got %d)", ns);
    return VSM_ERR_UNSUPPORTED;
  }
  if (S <= 0 || nV <= 0 || P <= 0) return VSM_OK;
  for (int v0 = 0; v0 < nV; v0 += 64) {
    const int nc = nV - v0 < 64 ? nV - v0 : 64;
    pp_args_lin pa;
    for (int v = 0; v < nc; ++v) pa.row0[v] = row0_h[v0 + v];
    for (int k = 0; k < ns; ++k)
      for (int v = 0; v < nc; ++v) pa.w[v + nc * k] = (double)w_h[v0 + v + nV * k];
    const long long tot = (long long)nc * ns * S * P;
    hipLaunchKernelGGL(k_postprocess_lin<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, ns, S, nc, nV, v0, P, pa,
                       Jd_m, Jd_p, Rd, Td);
    VSM_LAUNCH_CHECK("k_postprocess_lin");
  }
  return VSM_OK;
}

#define VSM_INST_L(T)                                                                                                 \
  template int elemental_lin<T>(const quad<T>&, int, int, int, const T*, const T*, const T*, const T*, const T*,     \
                                const T*, long long, int, const T*, const T*, const T*, const T*, const T*,          \
                                long long, long long, const added<T>&, const added_lin<T>&, hipStream_t, int);       \
  template int elemental_lin_mix<T>(const quad<T>&, int, int, int, const T*, const T*, const T*, const T*, int, int,   \
                                    const T*, const T*, int, const T*, int, const T*, const T*, const T*, const T*,     \
                                    const added<T>&, const added_lin<T>&, hipStream_t);                                 \
  template int doubling_lin<T>(int, int, int, int, T*, const T*, T, int, const added<T>&, const added_lin<T>&, T*,   \
                               hipStream_t);                                                                          \
  template int interaction_lin<T>(int, int, int, const composite<T>&, const composite_lin<T>&, const added<T>&,      \
                                  const added_lin<T>&, T*, hipStream_t);                                              \
  template int lambertian_surface_lin<T>(const quad<T>&, int, int, T, int, const T*, const T*, int, const T*,        \
                                         const added<T>&, const added_lin<T>&, hipStream_t);                          \
  template int copy_added_to_composite_lin<T>(int, int, const added_lin<T>&, const composite_lin<T>&, hipStream_t);  \
  template int postprocess_vza_lin<T>(int, int, int, int, int, const int*, const T*, const T*, const T*, T*, T*,     \
                                      hipStream_t);
VSM_INST_L(double)
VSM_INST_L(float)

}  // namespace vsm
