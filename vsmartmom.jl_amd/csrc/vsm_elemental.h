// Element formulas of the elemental layer shared by the FP64 / FP32 strip kernels and their pre-pass kernels.
#pragma once
#include "vsm_common.h"

namespace vsm {

// exp(-xi) - exp(-xj)  (expdiff_neg, src/CoreRT/CoreKernel/rt_helpers.jl:32-40) from tabulated e = exp(-x), a = expm1(-x):
//  * thin layers (every x < 1/2): a_i - a_j where the arguments are well separated (relative error 16 eps at most: the
//    subtraction is exact, each a carries one rounding), otherwise e_j expm1(-(x_i - x_j)) with expm1 by its Taylor polynomial
//    (|x_i - x_j| <= x_max / 8 < 1/16: degree 11, truncation < 2^-60 relative) -- no transcendental per matrix element;
//  * otherwise the reference's form: exp(-min) (-expm1(-|x_i - x_j|)) with the sign of x_j - x_i; one expm1 per element.
template <typename T>
__device__ __forceinline__ T expm1_small(T y) {
  T q = T(1.0 / 39916800.0);
  q = fma(q, y, T(1.0 / 3628800.0));
  q = fma(q, y, T(1.0 / 362880.0));
  q = fma(q, y, T(1.0 / 40320.0));
  q = fma(q, y, T(1.0 / 5040.0));
  q = fma(q, y, T(1.0 / 720.0));
  q = fma(q, y, T(1.0 / 120.0));
  q = fma(q, y, T(1.0 / 24.0));
  q = fma(q, y, T(1.0 / 6.0));
  q = fma(q, y, T(0.5));
  q = fma(q, y, T(1.0));
  return q * y;
}
template <typename T>
__device__ __forceinline__ T expdiff_tab_thin(T xi, T xj, T ai, T aj, T ej) {
  const T dlt = xi - xj;
  return (fabs(dlt) > T(0.125) * fmax(xi, xj)) ? (ai - aj) : ej * expm1_small<T>(-dlt);
}
template <typename T>
__device__ __forceinline__ T expdiff_tab_thick(T xi, T xj, T ei, T ej) {
  const T dlt = xi - xj;
  const T v = ((dlt < T(0)) ? ei : ej) * (-expm1(-fabs(dlt)));
  return (dlt < T(0)) ? v : -v;   // (dlt == 0: v = 0)
}

// quotient of the geometric factors: IEEE division in FP64; in FP32 reciprocal * numerator (v_rcp_f32, 1 ulp: two instructions
// instead of ten -- the FP32 pre-pass is VALU bound)
__device__ __forceinline__ double geo_div(double a, double b) { return a / b; }
__device__ __forceinline__ float geo_div(float a, float b) { return __fdividef(a, b); }

// One element of the elemental layer (elemental.jl:289-334) from the per-row / per-column tables (x = dtau / mu, e = exp(-x),
// a = expm1(-x)); straight-line selects instead of the reference's branches:
//   r-+_ij = varpi Z-+_ij  mu_j / (mu_i + mu_j) w_j (1 - e^{-x_i} e^{-x_j}),   1 - e^{-x_i} e^{-x_j} = -(a_i + a_j + a_i a_j)
//   t++_ij = varpi Z++_ij  mu_j / (mu_i - mu_j) w_j (e^{-x_i} - e^{-x_j})      (mu_i != mu_j)
//          = delta_ij e^{-x_i} + e^{-x_j} varpi Z++_ij x_i w_j                 (mu_i == mu_j)
// The SFI source (elemental.jl:348-392) has the same form with the solar column in place of column j (mu_j -> mu_0,
// x_j -> dtau / mu_0, w_j -> (1 + delta_m0) / 4, Z_ij -> sum_q Z_{i, i0 + q} F0_q): j0+ is the "t" formula, j0- the "r" one.
template <typename T>
__device__ __forceinline__ void elemental_pair(T w, T zp, T zm, T mi, T xi, T ai, T ei, T mj, T xj, T aj, T ej, T wct, bool diag,
                                               bool thick, T& rr, T& tt) {
  rr = w * zm * geo_div(mj, mi + mj) * wct * (-(ai + aj + ai * aj));
  T ediff;
  if (thick) ediff = expdiff_tab_thick<T>(xi, xj, ei, ej); else ediff = expdiff_tab_thin<T>(xi, xj, ai, aj, ej);
  const T t_off = w * zp * geo_div(mj, mi - mj) * wct * ediff;
  const T t_1 = w * zp * xi * wct;
  const T t_same = diag ? ei * (T(1) + t_1) : ej * t_1;
  tt = (mi == mj) ? t_same : t_off;
}

}  // namespace vsm
