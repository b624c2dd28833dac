// Per-scene layer optics on the device: the reference assembles these on the host once per Fourier
// moment and ships [N,N,nSpec] / [nSpec] arrays per layer (H2D inside the layer loop).  Here the raw optical depths are
// uploaded once and everything the layer kernels consume is produced in HBM.
//
//   compute_Z_moments   Z++(m), Z-+(m) of one scatterer from its Greek coefficients
//                       src/Scattering/compute_Z_matrices.jl:26-110 with the generalized spherical functions of
//                       src/Scattering/legendre_functions.jl:24-183 and Pi / B of mie_helper_functions.jl:532-615
//   layer_optics        constructCoreOpticalProperties + extractEffectiveProps
//                       src/CoreRT/LayerOpticalProperties/compEffectiveLayerProperties.jl:11-93 (createAero delta-M :67-72),
//                       mixing `+` of CoreScatteringOpticalProperties src/CoreRT/types.jl:1262-1308
//                       -> tau, varpi, tau_sum per (point, layer), per-point component weights, max(tau*varpi) per layer
//   layer_dtau          dtau = tau / 2^ndoubl (rt_kernel.jl:266-287) once the host has turned the maxima into ndoubl
//
// All arithmetic is FP64 (the reference's host code works in Float64 model arrays); results are stored in T.
#include "vsm_internal.h"

// keep the FP64 arithmetic identical to the host mirror (numpy does not contract a*b+c into an fma)
#pragma clang fp contract(off)

namespace vsm {

// one step l of the (P, R, T) recurrences at fixed Fourier order m for x = c (s = sqrt(1 - c^2)); `p1`/`p2` are the values
// at l-1 / l-2 (zero below l = m).  T is the internal T (the reference publishes -T).
struct prt {
  double P, R, T;
};
__device__ __forceinline__ prt prt_step(int l, int m, double c, double s, prt p1, prt p2) {
  prt o = {0.0, 0.0, 0.0};
  if (m == 0) {
    if (l == 0) {
      o.P = 1.0;
    } else if (l == 1) {
      o.P = c;
    } else if (l == 2) {
      o.P = 0.5 * (3 * c * c - 1);
      o.R = 0.5 * sqrt(1.5) * s * s;
    } else {
      o.P = (p1.P * (2 * l - 1) * c - p2.P * (l - 1)) / l;
      o.R = (p1.R * (2 * l - 1) * c - p2.R * sqrt((double)(l + 1) * (l - 3))) / sqrt((double)l * l - 4);
    }
    return o;
  }
  if (l == m && m == 1) {
    o.P = sqrt(0.5) * s;
    return o;
  }
  if (m == 1 && l == 2) {
    const double m1 = sqrt(1.0 / 6.0);
    o.P = m1 * 3 * c * s;
    o.R = -m1 * c * sqrt(1.5) * s;
    o.T = m1 * sqrt(1.5) * s;
    return o;
  }
  if (l == m) {  // m >= 2
    double f1 = 1.0, f2 = 1.0;
    for (int i = 1; i <= m; ++i) {
      f1 = f1 * ((2 * i - 1) * s) / sqrt((double)i * (i + m));
      f2 = f2 * (s / 2) * (i > 2 ? sqrt((double)(m + i) / (i - 2)) : 1.0);
    }
    const bool ok = s > 1e-8;
    const double lim = (m == 2) ? 0.5 : 0.0;
    o.P = f1;
    o.R = ok ? f2 * (1 + c * c) / (s * s) : lim;
    o.T = -(ok ? f2 * (2 * c) / (s * s) : lim);
    return o;
  }
  const double zc = (2.0 * m * (2 * l - 1)) / ((double)l * (l - 1));
  const double xr = ((double)(l - m) / l) * sqrt((double)l * l - 4);
  if (l == m + 1 && m >= 2) {
    const double m1 = sqrt(1.0 / (l + m));
    o.P = (m1 * p1.P * (2 * l - 1) * c) / (l - m);
    o.R = (m1 * p1.R * (2 * l - 1) * c + m1 * p1.T * zc) / xr;
    o.T = (m1 * p1.T * (2 * l - 1) * c + m1 * p1.R * zc) / xr;
    return o;
  }
  double m1, m2;
  if (m == 1) {
    m1 = sqrt((double)(l - 1) / (l + 1));
    m2 = m1 * sqrt((double)(l - 2) / l);
  } else {
    m1 = sqrt((double)(l - m) / (l + m));
    m2 = m1 * sqrt((double)(l - m - 1) / (l + m - 1));
  }
  const double yr = ((double)(l + m - 1) / (l - 1)) * sqrt((double)(l - 3) * (l + 1));
  o.P = (m1 * p1.P * (2 * l - 1) * c - m2 * p2.P * (l - 1 + m)) / (l - m);
  o.R = (m1 * p1.R * (2 * l - 1) * c - m2 * p2.R * yr + m1 * p1.T * zc) / xr;
  o.T = (m1 * p1.T * (2 * l - 1) * c - m2 * p2.T * yr + m1 * p1.R * zc) / xr;
  return o;
}

// Pi_l^m(x) restricted to n Stokes components (mie_helper_functions.jl:532-582): diag(P, R, R, P) with -(-T) = T on (Q,U),(U,Q)
__device__ __forceinline__ void fill_pi(const prt& v, int n, double (&Pi)[4][4]) {
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) Pi[a][b] = 0.0;
  Pi[0][0] = v.P;
  if (n >= 2) Pi[1][1] = v.R;
  if (n >= 3) {
    Pi[1][2] = v.T;
    Pi[2][1] = v.T;
    Pi[2][2] = v.R;
  }
  if (n == 4) Pi[3][3] = v.P;
}

// one thread per stream pair (i, j): the n x n blocks of Z++ and Z-+ (compute_Z_matrices.jl:26-110)
template <typename T>
__global__ void k_z_moments(int Nq, int n, int m, int lmax, const T* __restrict__ muN, const double* __restrict__ greek, T* Zpp,
                            T* Zmp) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Nq * Nq) return;
  const int i = e % Nq, j = e / Nq;
  const double ci = (double)muN[i * n], cj = (double)muN[j * n];
  const double si = sqrt(1.0 - ci * ci), sj = sqrt(1.0 - cj * cj);
  const double* al = greek;
  const double* be = greek + lmax;
  const double* ga = greek + 2 * lmax;
  const double* de = greek + 3 * lmax;
  const double* ep = greek + 4 * lmax;
  const double* ze = greek + 5 * lmax;
  prt a1 = {0, 0, 0}, a2 = {0, 0, 0}, b1 = {0, 0, 0}, b2 = {0, 0, 0}, c1 = {0, 0, 0}, c2 = {0, 0, 0};
  double App[4][4], Amp[4][4];
  for (int a = 0; a < 4; ++a)
    for (int b = 0; b < 4; ++b) App[a][b] = Amp[a][b] = 0.0;
  for (int l = m; l < lmax; ++l) {
    const prt vi = prt_step(l, m, ci, si, a1, a2);    // Pi(+mu_i)
    const prt vp = prt_step(l, m, cj, sj, b1, b2);    // Pi(+mu_j)
    const prt vm = prt_step(l, m, -cj, sj, c1, c2);   // Pi(-mu_j)
    a2 = a1; a1 = vi; b2 = b1; b1 = vp; c2 = c1; c1 = vm;
    double Pi[4][4], Pp[4][4], Pm[4][4], B[4][4], L[4][4];
    fill_pi(vi, n, Pi);
    fill_pi(vp, n, Pp);
    fill_pi(vm, n, Pm);
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) B[a][b] = 0.0;
    B[0][0] = be[l];
    if (n >= 2) {
      B[0][1] = B[1][0] = ga[l];
      B[1][1] = al[l];
    }
    if (n >= 3) B[2][2] = ze[l];
    if (n == 4) {
      B[2][3] = ep[l];
      B[3][2] = -ep[l];
      B[3][3] = de[l];
    }
    for (int a = 0; a < n; ++a)
      for (int c = 0; c < n; ++c) {
        double acc = 0.0;
        for (int b = 0; b < n; ++b) acc += Pi[a][b] * B[b][c];
        L[a][c] = acc;
      }
    for (int a = 0; a < n; ++a)
      for (int d = 0; d < n; ++d) {
        double sp = 0.0, sm = 0.0;
        for (int c = 0; c < n; ++c) {
          sp += L[a][c] * Pp[c][d];
          sm += L[a][c] * Pm[c][d];
        }
        App[a][d] += sp;
        Amp[a][d] += sm;
      }
  }
  const double fact = (m == 0) ? 0.5 : 1.0;
  const int N = Nq * n;
  for (int a = 0; a < n; ++a)
    for (int d = 0; d < n; ++d) {
      const double sg = ((a < 2) != (d < 2)) ? -1.0 : 1.0;   // sign flip on the (I,Q) x (U,V) cross blocks of Z-+
      const long long o = (long long)(i * n + a) + (long long)N * (j * n + d);
      Zpp[o] = (T)(2 * fact * App[a][d]);
      Zmp[o] = (T)(2 * fact * Amp[a][d] * sg);
    }
}

template <typename T>
int compute_Z_moments(int Nq, int n_stokes, const T* muN, int m, int lmax, const double* greek, T* Zpp, T* Zmp, hipStream_t st) {
  const int tot = Nq * Nq;
  hipLaunchKernelGGL(k_z_moments<T>, dim3((tot + 63) / 64), dim3(64), 0, st, Nq, n_stokes, m, lmax, muN, greek, Zpp, Zmp);
  VSM_LAUNCH_CHECK("k_z_moments");
  return VSM_OK;
}

// ---- layer optics ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_max_nonneg(double* p, double v) {
  atomicMax(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v));
}
__device__ __forceinline__ void atomic_max_nonneg(float* p, float v) {
  atomicMax(reinterpret_cast<unsigned int*>(p), __float_as_uint(v));
}

// one thread per spectral point, walking the layers from TOA down (the tau_sum prefix is sequential in l).
// mode[ia + nAer l]: how `x + createAero(...)` resolves for aerosol ia in layer l (batch-global branches of types.jl:1262-1292):
//   0 = per-point mix, 1 = every point has tau_x varpi_x == 0 -> Z of the aerosol, 2 = aerosol does not scatter -> Z of x.
template <typename T>
__global__ void k_layer_optics(int S, int L, int nAer, const double* __restrict__ tau_rayl, const double* __restrict__ tau_abs,
                               double varpi_cab, const double* __restrict__ tau_aer, const double* __restrict__ ssa,
                               const double* __restrict__ ftrunc, const int* __restrict__ mode, T* tau, T* varpi, T* tau_sum,
                               T* fcomp, T* max_tw) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = s < S;
  const int C = nAer + 1;
  double acc = 0.0;
  if (live) tau_sum[s] = T(0);
  for (int l = 0; l < L; ++l) {
    T tw = T(0);
    if (live) {
      const long long o = (long long)s + (long long)S * l;
      double t = tau_rayl[o], w = varpi_cab;
      double coef[8];
      for (int k = 0; k < C; ++k) coef[k] = (k == 0) ? 1.0 : 0.0;
      for (int ia = 0; ia < nAer; ++ia) {
        const double f = ftrunc[ia], om = ssa[ia], ta = tau_aer[ia + nAer * l];
        const double ty = (1 - f * om) * ta, wy_ = (1 - f) * om / (1 - f * om);   // createAero (delta-M)
        const double t2 = t + ty;
        const double wx = t * w, wy = ty * wy_;
        const double ws = wx + wy;
        const double w2 = ws / (t2 > 0 ? t2 : 1.0);
        const int md = mode[ia + nAer * l];
        if (md == 1) {
          for (int k = 0; k < C; ++k) coef[k] = (k == ia + 1) ? 1.0 : 0.0;
        } else if (md == 0) {
          const double fx = wx / ws, fy = wy / ws;
          for (int k = 0; k < C; ++k) coef[k] = fx * coef[k] + ((k == ia + 1) ? fy : 0.0);
        }
        t = t2;
        w = w2;
      }
      const double t3 = t + tau_abs[o];
      const double w3 = (t * w) / (t3 > 0 ? t3 : 1.0);
      const T tT = (T)t3, wT = (T)w3;
      tau[o] = tT;
      varpi[o] = wT;
      tw = tT * wT;
      acc += t3;
      tau_sum[o + S] = (T)acc;
      if (fcomp)
        for (int k = 0; k < C; ++k) fcomp[k + (long long)C * o] = (T)coef[k];
    }
    // max over the workgroup's points, then one atomic per wave
    T mx = tw;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (T)__shfl_down(mx, off, 64));
    if ((threadIdx.x & 63) == 0) atomic_max_nonneg(&max_tw[l], mx);
  }
}

template <typename T>
int layer_optics(int S, int L, int nAer, const double* tau_rayl, const double* tau_abs, double varpi_cab, const double* tau_aer,
                 const double* ssa, const double* ftrunc, const int* mode, T* tau, T* varpi, T* tau_sum, T* fcomp, T* max_tw,
                 hipStream_t st) {
  if (nAer > 7) {
    set_error("layer_optics: at most 7 aerosol components (got %d)", nAer);
    return VSM_ERR_UNSUPPORTED;
  }
  VSM_HIP(hipMemsetAsync(max_tw, 0, sizeof(T) * (size_t)L, st));
  if (S <= 0 || L <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_layer_optics<T>, dim3((S + 255) / 256), dim3(256), 0, st, S, L, nAer, tau_rayl, tau_abs, varpi_cab,
                     tau_aer, ssa, ftrunc, mode, tau, varpi, tau_sum, fcomp, max_tw);
  VSM_LAUNCH_CHECK("k_layer_optics");
  return VSM_OK;
}

template <typename T>
__global__ void k_layer_dtau(long long S, int L, const int* __restrict__ nd, const T* __restrict__ tau, T* dtau) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= S * L) return;
  const int l = (int)(e / S);
  dtau[e] = tau[e] / (T)ldexp(1.0, nd[l]);
}
template <typename T>
int layer_dtau(int S, int L, const int* nd, const T* tau, T* dtau, hipStream_t st) {
  if (S <= 0 || L <= 0) return VSM_OK;
  const long long tot = (long long)S * L;
  hipLaunchKernelGGL(k_layer_dtau<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (long long)S, L, nd, tau, dtau);
  VSM_LAUNCH_CHECK("k_layer_dtau");
  return VSM_OK;
}


// ---- linearized layer optics ---------------------------------------------------------------------------------------------
// constructCoreOpticalProperties with lin_model (compEffectiveLayerProperties_lin.jl:43-197; createAero with derivatives
// :330-395; the quotient rule of the pairwise `+`, types_lin.jl:196-380) in closed form.  A layer's optics are
//   tau = sum_c tau_c + tau_abs,  W = sum_c w_c  (w_c = tau_c varpi_c),  varpi = W / tau,  Z = sum_c (w_c / W) Z_c
// over the scatterers c = Rayleigh, aerosol 1, ...  For slot k of aerosol a (w_dot = tau_dot_a varpi_a + tau_a varpi_dot_a):
//   tau_dot = tau_dot_a,  varpi_dot = (w_dot - varpi tau_dot) / tau,  Z_dot = (w_dot / W) (Z_a - Z) + (w_a / W) Zdot_{a,k}
// and for a gas: tau_dot = tau_abs_dot, varpi_dot = -varpi tau_dot / tau, Z_dot = 0.  Z_dot is NOT materialised: the kernel
// writes its coefficients over the component matrices [Z_0 .. Z_{C-1}, Zdot_{0,0..3}, Zdot_{1,0..3}, ...] (CT = C + 4 nAer
// blocks of N x N per Fourier moment) and elemental! (lin) forms Z_dot where it consumes it (vsm_elemental_lin_mix).
// One thread per spectral point of the rank's block [lo, lo + S) of the full axis (inputs are indexed on the full axis,
// outputs are block-local); the tau_sum_dot prefix walks the layers.  Slot order: 7 per aerosol, then the gases
// (parameter_layout.jl:28-56); dtau_dot_all has P >= pl columns (surface slots stay zero).
template <typename T>
__global__ void k_layer_optics_lin(int S_full, int lo, int S, int L, int nAer, int nGas, int P,
                                   const double* __restrict__ tau_rayl, const double* __restrict__ tau_abs, double varpi_cab,
                                   const double* __restrict__ tau_aer, const double* __restrict__ ssa,
                                   const double* __restrict__ ftrunc, const double* __restrict__ tau_abs_dot,
                                   const double* __restrict__ tau_aer_dot, const double* __restrict__ ssa_dot,
                                   const double* __restrict__ ftrunc_dot, const int* __restrict__ nd, T* dtau_dot_all,
                                   T* varpi_dot, T* tau_sum_dot, T* fz, T* zdcoef) {
  const int sl = blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= S) return;
  const int s = lo + sl;
  const int C = nAer + 1, CT = C + 4 * nAer, pl = 7 * nAer + nGas;
  const long long SP = (long long)S * pl;
  // tau_sum_dot[:, :, l] = sum of tau_dot over the layers above l (rt_run_lin.jl:214-222), accumulated in FP64 per slot
  for (int p = 0; p < pl; ++p) {
    double acc = 0.0;
    tau_sum_dot[sl + (long long)S * p] = T(0);
    for (int l = 0; l < L; ++l) {
      double td;
      if (p < 7 * nAer) {
        const int ia = p / 7, k = p % 7;
        const bool mie = k >= 1 && k <= 4;
        const double f = ftrunc[ia], om = ssa[ia], ta = tau_aer[ia + nAer * l];
        const double wd = mie ? ssa_dot[(k - 1) + 4 * ia] : 0.0, fd = mie ? ftrunc_dot[(k - 1) + 4 * ia] : 0.0;
        td = (1.0 - f * om) * tau_aer_dot[k + 7 * (ia + nAer * l)] - (f * wd + om * fd) * ta;
      } else {
        td = tau_abs_dot[(long long)s + (long long)S_full * l + (long long)S_full * L * (p - 7 * nAer)];
      }
      acc += td;
      tau_sum_dot[SP * (l + 1) + sl + (long long)S * p] = (T)acc;
    }
  }
  for (int l = 0; l < L; ++l) {
    const long long o = (long long)s + (long long)S_full * l;
    const double scale = ldexp(1.0, -nd[l]);
    T* dd = dtau_dot_all + (long long)S * P * l;
    T* vd = varpi_dot + SP * l;
    for (int p = pl; p < P; ++p) dd[sl + (long long)S * p] = T(0);
    // scatterers: tau_c, varpi_c (createAero: delta-M), their sums
    double tc[8], vc[8];
    tc[0] = tau_rayl[o];
    vc[0] = varpi_cab;
    double tsum = tc[0], W = tc[0] * vc[0];
    for (int ia = 0; ia < nAer; ++ia) {
      const double f = ftrunc[ia], om = ssa[ia], ta = tau_aer[ia + nAer * l];
      const double g = 1.0 - f * om;
      tc[ia + 1] = g * ta;
      vc[ia + 1] = (1.0 - f) * om / g;
      tsum += tc[ia + 1];
      W += tc[ia + 1] * vc[ia + 1];
    }
    const double tau = tsum + tau_abs[o];
    const double tsafe = tau > 0 ? tau : 1.0;
    const double varpi = W / tsafe;
    const double Winv = W > 0 ? 1.0 / W : 0.0;
    if (fz)
      for (int c = 0; c < C; ++c) fz[c + (long long)C * (sl + (long long)S * l)] = (T)(tc[c] * vc[c] * Winv);
    T* zc = zdcoef ? zdcoef + (long long)CT * pl * (sl + (long long)S * l) : nullptr;
    for (int ia = 0; ia < nAer; ++ia) {
      const double f = ftrunc[ia], om = ssa[ia], ta = tau_aer[ia + nAer * l];
      const double g = 1.0 - f * om;
      const double fa = tc[ia + 1] * vc[ia + 1] * Winv;
      for (int k = 0; k < 7; ++k) {
        const int p = 7 * ia + k;
        const bool mie = k >= 1 && k <= 4;   // n_r, n_i, r_m, sigma_r move ssa and f_trunc (and the Greek coefficients)
        const double wd = mie ? ssa_dot[(k - 1) + 4 * ia] : 0.0, fd = mie ? ftrunc_dot[(k - 1) + 4 * ia] : 0.0;
        const double td = g * tau_aer_dot[k + 7 * (ia + nAer * l)] - (f * wd + om * fd) * ta;
        const double vdk = (wd * (1.0 - f) - fd * (om * (1.0 - om))) / (g * g);
        const double wdot = td * vc[ia + 1] + tc[ia + 1] * vdk;
        dd[sl + (long long)S * p] = (T)(td * scale);
        vd[sl + (long long)S * p] = (T)((wdot - varpi * td) / tsafe);
        if (zc) {
          const double r = wdot * Winv;
          for (int c = 0; c < C; ++c) zc[c + CT * p] = (T)(r * ((c == ia + 1 ? 1.0 : 0.0) - tc[c] * vc[c] * Winv));
          for (int c = C; c < CT; ++c) zc[c + CT * p] = (T)((mie && c == C + 4 * ia + (k - 1)) ? fa : 0.0);
        }
      }
    }
    for (int gi = 0; gi < nGas; ++gi) {
      const int p = 7 * nAer + gi;
      const double td = tau_abs_dot[o + (long long)S_full * L * gi];
      dd[sl + (long long)S * p] = (T)(td * scale);
      vd[sl + (long long)S * p] = (T)(-(varpi / tsafe) * td);
      if (zc)
        for (int c = 0; c < CT; ++c) zc[c + CT * p] = T(0);
    }
  }
}

template <typename T>
int layer_optics_lin(int S_full, int lo, int S, int L, int nAer, int nGas, int P, const double* tau_rayl, const double* tau_abs,
                     double varpi_cab, const double* tau_aer, const double* ssa, const double* ftrunc, const double* tau_abs_dot,
                     const double* tau_aer_dot, const double* ssa_dot, const double* ftrunc_dot, const int* nd, T* dtau_dot_all,
                     T* varpi_dot, T* tau_sum_dot, T* fz, T* zdcoef, hipStream_t st) {
  if (nAer > 7) {
    set_error("layer_optics_lin: at most 7 aerosol components (got %d)", nAer);
    return VSM_ERR_UNSUPPORTED;
  }
  if (S <= 0 || L <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_layer_optics_lin<T>, dim3((S + 127) / 128), dim3(128), 0, st, S_full, lo, S, L, nAer, nGas, P, tau_rayl,
                     tau_abs, varpi_cab, tau_aer, ssa, ftrunc, tau_abs_dot, tau_aer_dot, ssa_dot, ftrunc_dot, nd, dtau_dot_all,
                     varpi_dot, tau_sum_dot, fz, zdcoef);
  VSM_LAUNCH_CHECK("k_layer_optics_lin");
  return VSM_OK;
}

// expk = exp(-dtau / mu0) (rt_kernel.jl:339-349 init_layer): doubling! squares it in place, so a run refreshes it per layer
template <typename T>
__global__ void k_layer_expk(int S, const T* __restrict__ dtau, T mu0, T* expk) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s < S) expk[s] = exp(-dtau[s] / mu0);
}
template <typename T>
int layer_expk(int S, const T* dtau, T mu0, T* expk, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_layer_expk<T>, dim3((S + 255) / 256), dim3(256), 0, st, S, dtau, mu0, expk);
  VSM_LAUNCH_CHECK("k_layer_expk");
  return VSM_OK;
}

#define VSM_INST_OPT(T)                                                                                                      \
  template int compute_Z_moments<T>(int, int, const T*, int, int, const double*, T*, T*, hipStream_t);                        \
  template int layer_optics<T>(int, int, int, const double*, const double*, double, const double*, const double*,             \
                               const double*, const int*, T*, T*, T*, T*, T*, hipStream_t);                                   \
  template int layer_dtau<T>(int, int, const int*, const T*, T*, hipStream_t);                                                \
  template int layer_optics_lin<T>(int, int, int, int, int, int, int, const double*, const double*, double, const double*,    \
                                   const double*, const double*, const double*, const double*, const double*, const double*, \
                                   const int*, T*, T*, T*, T*, T*, hipStream_t);                                              \
  template int layer_expk<T>(int, const T*, T, T*, hipStream_t);
VSM_INST_OPT(double)
VSM_INST_OPT(float)

}  // namespace vsm
