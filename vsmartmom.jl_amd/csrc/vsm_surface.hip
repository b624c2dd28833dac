// BRDF surfaces on the device: Cox-Munk ocean (forward + wind-speed derivative) and the generic BRDF surface layer.
//
//   coxmunk_reflectance   reflectance / reflectance_and_deriv     src/CoreRT/Surfaces/coxmunk_surface.jl:381-460
//     (facet geometry :146-267, Fresnel Mueller fresnel.jl:25-124, slope PDF / Smith shadowing / whitecaps :23-128,
//      Mueller matrix + d/dU :277-370), 100-point Gauss-Legendre over azimuth handed in by the host
//   brdf_surface          create_surface_layer!(::AbstractSurfaceType)           Surfaces/rpv_surface.jl:51-97
//   brdf_surface_lin      create_surface_layer!(::noRS, ::CoxMunkSurface, lin)   Surfaces/coxmunk_surface_lin.jl:27-102
//   coxmunk_ss_correction apply_ss_correction! (TMS)                             Surfaces/coxmunk_surface.jl:481-569
//
// The reflectance matrix does not depend on the spectral point (the reference evaluates the water index at 550 nm for
// every call site), so it is ONE N x N block per Fourier moment: one workgroup per stream pair (i, j), one lane per
// azimuth node, the n x n Mueller block reduced across the workgroup.  Everything here is VALU / transcendental work
// of O(Nquad^2 * 100) per moment -- negligible next to the layer kernels; it lives on the device so that a scene never
// waits for the host between the last layer and the surface interaction.
#include "vsm_internal.h"

namespace vsm {

template <typename T>
struct cplx {
  T re, im;
};
template <typename T>
__device__ __forceinline__ cplx<T> cmul(cplx<T> a, cplx<T> b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
template <typename T>
__device__ __forceinline__ cplx<T> cdiv(cplx<T> a, cplx<T> b) {
  const T d = b.re * b.re + b.im * b.im;
  return {(a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d};
}
// principal square root, stable for |im| << |re| (the water index has k ~ 1e-9)
template <typename T>
__device__ __forceinline__ cplx<T> csqrt(cplx<T> z) {
  const T r = sqrt(z.re * z.re + z.im * z.im);
  if (r == T(0)) return {T(0), T(0)};
  if (z.re >= T(0)) {
    const T a = sqrt((r + z.re) / 2);
    return {a, z.im / (2 * a)};
  }
  const T b = sqrt((r - z.re) / 2);
  const T bs = z.im >= T(0) ? b : -b;
  return {z.im / (2 * bs), bs};
}

template <typename T>
__device__ __forceinline__ T pi_v() {
  return T(3.141592653589793238462643383279502884);
}

// smith_Lambda and its sigma^2 derivative (coxmunk_surface.jl:44-55, 76-99)
template <typename T>
__device__ __forceinline__ void smith_lambda(T mu, T s2, T& L, T& dL) {
  if (mu <= T(0)) {
    L = T(1e10);
    dL = T(0);
    return;
  }
  const T sig = sqrt(s2);
  const T cot = mu / sqrt(max(T(1e-30), T(1) - mu * mu));
  const T nu = cot / (sqrt(T(2)) * sig);
  const T e = exp(-nu * nu);
  const T raw = (e / (sqrt(2 * pi_v<T>()) * nu) - erfc(nu)) / 2;
  L = max(T(0), raw);
  if (raw <= T(0)) {
    dL = T(0);
    return;
  }
  const T dLdnu = (e * (T(-2) * nu * nu - T(1)) / (sqrt(2 * pi_v<T>()) * nu * nu) + T(2) / sqrt(pi_v<T>()) * e) / 2;
  dL = dLdnu * (-nu / (2 * s2));
}

// Mueller matrix M[NS][NS] of the Cox-Munk BRDF at (first argument mu_i, second mu_r, relative azimuth dphi) and
// its derivative with respect to wind speed (coxmunk_surface.jl:277-370; geometry :146-267).
template <typename T, int NS>
__device__ void cm_brdf(const cm_surf<T>& sf, T mu_i, T mu_r, T dphi, T (&M)[NS][NS], T (&dM)[NS][NS]) {
  const T U = sf.wind_speed;
  const T s2 = T(0.003) + T(0.00512) * U;
  // ---- geometry
  const T si = sqrt(max(T(0), T(1) - mu_i * mu_i));
  const T sr = sqrt(max(T(0), T(1) - mu_r * mu_r));
  const T cd = cos(dphi), sd = sin(dphi);
  T nx = -si + sr * cd, ny = sr * sd, nz = mu_i + mu_r;
  const T nrm = sqrt(nx * nx + ny * ny + nz * nz);
  T cos_b = T(1), cos_loc = T(1), zx = T(0), zy = T(0), a1 = T(0), a2 = T(0);
  if (!(nrm < T(1e-15))) {
    nx /= nrm;
    ny /= nrm;
    nz /= nrm;
    cos_b = max(T(1e-10), nz);
    cos_loc = min(max((mu_i + mu_r) / (2 * cos_b), T(0)), T(1));
    zx = -nx / cos_b;
    zy = -ny / cos_b;
    const T cosT = -mu_i * mu_r + si * sr * cd;
    const T sinT = sqrt(max(T(0), T(1) - cosT * cosT));
    if (!(sinT < T(1e-12))) {
      const T spx = -mu_i * sr * sd, spy = mu_i * sr * cd - si * mu_r, spz = -si * sr * sd;
      const T msp = sqrt(spx * spx + spy * spy + spz * spz);
      {  // alpha1: scattering plane vs incidence plane (k_i x n), sign from k_i . (sp x ip)
        const T ipx = mu_i * ny, ipy = -mu_i * nx - si * nz, ipz = si * ny;
        const T mip = sqrt(ipx * ipx + ipy * ipy + ipz * ipz);
        if (!(msp < T(1e-15) || mip < T(1e-15))) {
          const T c = min(max((spx * ipx + spy * ipy + spz * ipz) / (msp * mip), T(-1)), T(1));
          const T cx = spy * ipz - spz * ipy, cz = spx * ipy - spy * ipx;
          const T sgn = si * cx + (-mu_i) * cz;
          a1 = sgn >= T(0) ? acos(c) : -acos(c);
        }
      }
      {  // alpha2: scattering plane vs reflection plane (k_r x n), sign from k_r . (sp x rp)
        const T rpx = (-sr * sd) * nz - mu_r * ny, rpy = mu_r * nx - (-sr * cd) * nz, rpz = (-sr * cd) * ny - (-sr * sd) * nx;
        const T mrp = sqrt(rpx * rpx + rpy * rpy + rpz * rpz);
        if (!(msp < T(1e-15) || mrp < T(1e-15))) {
          const T c = min(max((spx * rpx + spy * rpy + spz * rpz) / (msp * mrp), T(-1)), T(1));
          const T cx = spy * rpz - spz * rpy, cy = spz * rpx - spx * rpz, cz = spx * rpy - spy * rpx;
          const T sgn = (-sr * cd) * cx + (-sr * sd) * cy + mu_r * cz;
          a2 = sgn >= T(0) ? acos(c) : -acos(c);
        }
      }
    }
  }
  // ---- Fresnel Mueller matrix (fresnel.jl:25-90)
  const cplx<T> nw = {sf.n_re, sf.n_im};
  const T sin2 = max(T(0), T(1) - cos_loc * cos_loc);
  const cplx<T> q = cdiv<T>({sin2, T(0)}, cmul(nw, nw));
  const cplx<T> ct = csqrt<T>({T(1) - q.re, -q.im});
  const cplx<T> nct = cmul(nw, ct);
  const cplx<T> rs = cdiv<T>({cos_loc - nct.re, -nct.im}, {cos_loc + nct.re, nct.im});
  const cplx<T> rp = cdiv<T>({nw.re * cos_loc - ct.re, nw.im * cos_loc - ct.im}, {nw.re * cos_loc + ct.re, nw.im * cos_loc + ct.im});
  const T rs2 = rs.re * rs.re + rs.im * rs.im, rp2 = rp.re * rp.re + rp.im * rp.im;
  const T re_rsp = rs.re * rp.re + rs.im * rp.im, im_rsp = rs.im * rp.re - rs.re * rp.im;  // rs * conj(rp)
  T MF[NS][NS], L1[NS][NS], L2[NS][NS];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) MF[a][b] = L1[a][b] = L2[a][b] = T(0);
  MF[0][0] = (rs2 + rp2) / 2;
  L1[0][0] = L2[0][0] = T(1);
  if (NS >= 2) {
    MF[1 % NS][1 % NS] = (rs2 + rp2) / 2;
    MF[0][1 % NS] = MF[1 % NS][0] = (rs2 - rp2) / 2;
    const T c1 = cos(2 * (-a1)), s1 = sin(2 * (-a1)), c2 = cos(2 * a2), s2r = sin(2 * a2);
    L1[1 % NS][1 % NS] = c1;
    L2[1 % NS][1 % NS] = c2;
    if (NS >= 3) {
      MF[2 % NS][2 % NS] = re_rsp;
      L1[2 % NS][2 % NS] = c1;
      L1[1 % NS][2 % NS] = s1;
      L1[2 % NS][1 % NS] = -s1;
      L2[2 % NS][2 % NS] = c2;
      L2[1 % NS][2 % NS] = s2r;
      L2[2 % NS][1 % NS] = -s2r;
    }
    if (NS == 4) {
      MF[3 % NS][3 % NS] = re_rsp;
      MF[2 % NS][3 % NS] = im_rsp;
      MF[3 % NS][2 % NS] = -im_rsp;
      L1[3 % NS][3 % NS] = L2[3 % NS][3 % NS] = T(1);
    }
  }
  T tmp[NS][NS], Mf[NS][NS];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < NS; ++k) acc += L2[a][k] * MF[k][b];
      tmp[a][b] = acc;
    }
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < NS; ++k) acc += tmp[a][k] * L1[k][b];
      Mf[a][b] = acc;
    }
  // ---- weights
  const T cb2 = cos_b * cos_b;
  const T gw = T(1) / (T(4) * mu_i * mu_r * (cb2 * cb2));
  const T Z2 = zx * zx + zy * zy;
  const T P = exp(-Z2 / (2 * s2)) / (2 * pi_v<T>() * s2);
  const T dP = P * (Z2 - 2 * s2) / (2 * s2 * s2);
  T pre, dpre;
  if (sf.shadowing) {
    T Li, dLi, Lr, dLr;
    smith_lambda(mu_i, s2, Li, dLi);
    smith_lambda(mu_r, s2, Lr, dLr);
    const T Sh = T(1) / (T(1) + Li + Lr);
    const T dSh = -Sh * Sh * (dLi + dLr);
    pre = P * Sh * gw;
    dpre = (dP * Sh + P * dSh) * gw;
  } else {
    pre = P * gw;
    dpre = dP * gw;
  }
  const T dscale = dpre * T(0.00512);
  T f = T(0), df = T(0);
  if (sf.include_whitecaps && U > T(0)) {
    f = T(2.95e-6) * pow(U, T(3.52));
    df = T(2.95e-6) * T(3.52) * pow(U, T(2.52));
  }
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) {
      const T g = pre * Mf[a][b], dg = dscale * Mf[a][b];
      if (sf.include_whitecaps) {
        const T wc = (a == 0 && b == 0) ? sf.whitecap_albedo / pi_v<T>() : T(0);
        M[a][b] = (T(1) - f) * g + f * wc;
        dM[a][b] = (T(1) - f) * dg + df * (wc - g);
      } else {
        M[a][b] = g;
        dM[a][b] = dg;
      }
    }
}

// sum over the workgroup (128 lanes = 2 waves); result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum128(T v, T* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1];
}

// rho[(i NS + a) + N (j NS + b)] = ff/pi sum_phi w M_ab(mu_i, mu_j, phi) az_ab(m phi)   (coxmunk_surface.jl:381-460)
template <typename T, int NS>
__global__ void __launch_bounds__(128) k_cm_reflectance(cm_surf<T> sf, int Nmu, const T* __restrict__ muN, int m, int nphi,
                                                        const T* __restrict__ phi, const T* __restrict__ wphi, T* rho, T* drho) {
  __shared__ T red[2];
  const int i = blockIdx.x, j = blockIdx.y, f = threadIdx.x;
  const int N = Nmu * NS;
  T M[NS][NS], dM[NS][NS];
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) M[a][b] = dM[a][b] = T(0);
  T cm = T(0), sm = T(0), w = T(0);
  if (f < nphi) {
    const T ph = phi[f];
    cm_brdf<T, NS>(sf, muN[i * NS], muN[j * NS], ph, M, dM);
    cm = cos(T(m) * ph);
    sm = sin(T(m) * ph);
    w = wphi[f];
  }
  const T ff = (m == 0 ? T(1) : T(2));
#pragma unroll
  for (int a = 0; a < NS; ++a)
#pragma unroll
    for (int b = 0; b < NS; ++b) {
      const T az = ((a <= 1) == (b <= 1)) ? cm : sm;
      const T v = block_sum128<T>(w * M[a][b] * az, red);
      const T dv = drho ? block_sum128<T>(w * dM[a][b] * az, red) : T(0);
      if (f == 0) {
        const long long o = (long long)(i * NS + a) + (long long)N * (j * NS + b);
        rho[o] = ff * v / pi_v<T>();
        if (drho) drho[o] = ff * dv / pi_v<T>();
      }
    }
}

template <typename T>
int coxmunk_reflectance(const cm_surf<T>& sf, int n_stokes, int Nmu, const T* muN, int m, int nphi, const T* phi, const T* wphi,
                        T* rho, T* drho, hipStream_t st) {
  if (nphi > 128 || nphi < 1) {
    set_error("coxmunk_reflectance: 1 <= nphi <= 128 azimuth nodes (got %d)", nphi);
    return VSM_ERR_UNSUPPORTED;
  }
  const dim3 g(Nmu, Nmu), b(128);
  switch (n_stokes) {
    case 1: hipLaunchKernelGGL((k_cm_reflectance<T, 1>), g, b, 0, st, sf, Nmu, muN, m, nphi, phi, wphi, rho, drho); break;
    case 2: hipLaunchKernelGGL((k_cm_reflectance<T, 2>), g, b, 0, st, sf, Nmu, muN, m, nphi, phi, wphi, rho, drho); break;
    case 3: hipLaunchKernelGGL((k_cm_reflectance<T, 3>), g, b, 0, st, sf, Nmu, muN, m, nphi, phi, wphi, rho, drho); break;
    case 4: hipLaunchKernelGGL((k_cm_reflectance<T, 4>), g, b, 0, st, sf, Nmu, muN, m, nphi, phi, wphi, rho, drho); break;
    default: set_error("coxmunk_reflectance: n_stokes must be 1..4"); return VSM_ERR_INVALID_ARG;
  }
  VSM_LAUNCH_CHECK("k_cm_reflectance");
  return VSM_OK;
}

// ---- generic BRDF surface layer ------------------------------------------------------------------------------------
// matrices (one shared block): r-+ = f rho diag(mu w), r+- = 0, t++ = I, t-- = I (forward) | 0 (linearized builder)
template <typename T>
__global__ void k_brdf_mats(int N, int m, const T* __restrict__ rho, const T* __restrict__ drho, const T* __restrict__ mu,
                            const T* __restrict__ wt, T* r_mp, T* r_pm, T* t_pp, T* t_mm, int lin, int iparam, int P, T* ar,
                            T* arpm, T* at, T* atmm) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int i = e % N, j = e / N;
  const T f = (m == 0) ? T(2) : T(1);
  const T sc = mu[j] * wt[j];
  r_mp[e] = (f * rho[e]) * sc;
  r_pm[e] = T(0);
  const T d = (i == j) ? T(1) : T(0);
  t_pp[e] = d;
  t_mm[e] = lin ? T(0) : d;
  if (lin) {
    for (int p = 0; p < P; ++p) {
      const long long o = e + (long long)N * N * p;
      ar[o] = (p == iparam) ? (f * drho[e]) * sc : T(0);
      arpm[o] = T(0);
      at[o] = T(0);
      atmm[o] = T(0);
    }
  }
}
// forward sources (rpv_surface.jl:80-87): j0+ = I0_N att, j0- = mu0 (R_surf I0_N) att with I0 = e1 on the SZA stream
template <typename T>
__global__ void k_brdf_src(int N, int ns, int S, int m, const T* __restrict__ rho, int i_mu0, T mu0,
                           const T* __restrict__ tau_sum, T* j0_p, T* j0_m) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)N * S) return;
  const int i = (int)(e % N);
  const long long s = e / N;
  const T att = exp(-tau_sum[s] / mu0);
  const int i0 = ns * i_mu0;
  const T f = (m == 0) ? T(2) : T(1);
  j0_p[e] = (i == i0) ? att : T(0);
  j0_m[e] = (mu0 * (f * rho[i + (long long)N * i0])) * att;
}
// linearized sources (coxmunk_surface_lin.jl:62-83): F0 (all Stokes components) instead of I0, j0+ = 0
template <typename T>
__global__ void k_brdf_src_lin(int N, int ns, int S, int m, const T* __restrict__ rho, const T* __restrict__ drho, int iparam,
                               int P, int p_layer, int i_mu0, T mu0, const T* __restrict__ tau_sum,
                               const T* __restrict__ tau_sum_dot, const T* __restrict__ F0, T* j0_p, T* j0_m, T* aJp, T* aJm) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)N * S) return;
  const int i = (int)(e % N);
  const long long s = e / N;
  const T att = exp(-tau_sum[s] / mu0);
  const int i0 = ns * i_mu0;
  const T f = (m == 0) ? T(2) : T(1);
  T a = T(0), da = T(0);
  for (int k = 0; k < ns; ++k) {
    const T Fk = F0[k + (long long)ns * s] * att;
    a += (f * rho[i + (long long)N * (i0 + k)]) * Fk;
    da += (f * drho[i + (long long)N * (i0 + k)]) * Fk;
  }
  const T jm = mu0 * a;
  j0_p[e] = T(0);
  j0_m[e] = jm;
  for (int p = 0; p < P; ++p) {
    const long long o = e + (long long)N * S * p;
    aJp[o] = T(0);
    T v = T(0);
    if (p < p_layer) v = -jm * tau_sum_dot[s + (long long)S * p] / mu0;
    if (p == iparam) v = mu0 * da;
    aJm[o] = v;
  }
}

template <typename T>
int brdf_surface(const quad<T>& q, int S, int m, const T* rho, const T* tau_sum, const added<T>& a, hipStream_t st) {
  if (a.mat_stride != 0) {
    set_error("brdf_surface: added.mat_stride must be 0 (one shared surface block)");
    return VSM_ERR_INVALID_ARG;
  }
  const int N = q.N;
  hipLaunchKernelGGL(k_brdf_mats<T>, dim3((N * N + 255) / 256), dim3(256), 0, st, N, m, rho, (const T*)nullptr, q.mu, q.wt,
                     a.r_mp, a.r_pm, a.t_pp, a.t_mm, 0, -1, 0, (T*)nullptr, (T*)nullptr, (T*)nullptr, (T*)nullptr);
  VSM_LAUNCH_CHECK("k_brdf_mats");
  if (S > 0) {
    hipLaunchKernelGGL(k_brdf_src<T>, dim3((unsigned)(((long long)N * S + 255) / 256)), dim3(256), 0, st, N, q.n_stokes, S, m,
                       rho, q.i_mu0, q.mu0, tau_sum, a.j0_p, a.j0_m);
    VSM_LAUNCH_CHECK("k_brdf_src");
  }
  return VSM_OK;
}

template <typename T>
int brdf_surface_lin(const quad<T>& q, int S, int m, const T* rho, const T* drho, int iparam, const T* tau_sum,
                     const T* tau_sum_dot, int p_layer, const T* F0, const added<T>& a, const added_lin<T>& al, hipStream_t st) {
  if (a.mat_stride != 0 || al.mat_stride != 0) {
    set_error("brdf_surface_lin: surface layers use ONE shared block (mat_stride must be 0)");
    return VSM_ERR_INVALID_ARG;
  }
  const int N = q.N;
  hipLaunchKernelGGL(k_brdf_mats<T>, dim3((N * N + 255) / 256), dim3(256), 0, st, N, m, rho, drho, q.mu, q.wt, a.r_mp, a.r_pm,
                     a.t_pp, a.t_mm, 1, iparam, al.P, al.ap_r_mp, al.ap_r_pm, al.ap_t_pp, al.ap_t_mm);
  VSM_LAUNCH_CHECK("k_brdf_mats(lin)");
  if (S > 0) {
    hipLaunchKernelGGL(k_brdf_src_lin<T>, dim3((unsigned)(((long long)N * S + 255) / 256)), dim3(256), 0, st, N, q.n_stokes, S,
                       m, rho, drho, iparam, al.P, p_layer, q.i_mu0, q.mu0, tau_sum, tau_sum_dot, F0, a.j0_p, a.j0_m,
                       al.ap_J0_p, al.ap_J0_m);
    VSM_LAUNCH_CHECK("k_brdf_src_lin");
  }
  return VSM_OK;
}

// ---- Lambertian surface with a spectrally varying albedo (LambertianSurfaceLegendre / LambertianSurfaceSpline,
// lambertian_surface.jl:97-213): per-point blocks r-+[:,:,s] = 2 a_s E11 (x) (mu w); the builder's quirks kept: j0+ = 0, and for
// m > 0 the transmission blocks are ZERO (the scalar builder writes the identity there).
template <typename T>
__global__ void k_lambertian_spectral(int N, int ns, long long S, int m, const T* __restrict__ albedo, const T* __restrict__ mu,
                                      const T* __restrict__ wt, int i_mu0, T mu0, const T* __restrict__ tau_sum, T* r_mp, T* r_pm,
                                      T* t_pp, T* t_mm, T* j0_p, T* j0_m) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long NN = (long long)N * N;
  if (e >= NN * S) return;
  const long long s = e / NN;
  const int x = (int)(e % NN), i = x % N, j = x / N;
  const T rho = T(2) * albedo[s];
  const bool ii = (m == 0) && (i % ns == 0) && (j % ns == 0);
  r_mp[e] = ii ? rho * (mu[j] * wt[j]) : T(0);
  r_pm[e] = T(0);
  const T d = (m == 0 && i == j) ? T(1) : T(0);
  t_pp[e] = d;
  t_mm[e] = d;
  if (j == 0) {
    const long long o = s * N + i;
    j0_p[o] = T(0);
    j0_m[o] = (m == 0 && (i % ns) == 0) ? mu0 * (rho * exp(-tau_sum[s] / mu0)) : T(0);
  }
  (void)i_mu0;
}
template <typename T>
int lambertian_surface_spectral(const quad<T>& q, int S, int m, const T* albedo, const T* tau_sum, const added<T>& a, hipStream_t st) {
  const int N = q.N;
  if (a.mat_stride != (long long)N * N) {
    set_error("lambertian_surface_spectral: the surface layer needs one block per spectral point (mat_stride = N*N)");
    return VSM_ERR_INVALID_ARG;
  }
  if (S <= 0) return VSM_OK;
  const long long tot = (long long)N * N * S;
  hipLaunchKernelGGL(k_lambertian_spectral<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, q.n_stokes, (long long)S, m,
                     albedo, q.mu, q.wt, q.i_mu0, q.mu0, tau_sum, a.r_mp, a.r_pm, a.t_pp, a.t_mm, a.j0_p, a.j0_m);
  VSM_LAUNCH_CHECK("k_lambertian_spectral");
  return VSM_OK;
}

// ---- TMS single-scattering correction (coxmunk_surface.jl:481-569) ---------------------------------------------------
struct pp_args_s {
  int row0[64];
  double w[256];
};
struct ss_geom {
  double mu_v[64];
  double dphi[64];
};
// coef[v + nV k] = M_exact[k,1](mu_v, mu0, dphi_v) - sum_m w_m az_k1(m dphi_v) c_m[k],
// c_m[k] = ff_m/pi sum_phi w M[k,1](mu_v, mu0, phi) az_k1(m phi): one workgroup per viewing geometry; the azimuth sums run
// in the reference's order (one thread per (m, k) walks the nodes).
template <typename T, int NS>
__global__ void __launch_bounds__(128) k_cm_ss_coef(cm_surf<T> sf, int nV, ss_geom g, T mu0, int m_max, int nphi,
                                                    const T* __restrict__ phi, const T* __restrict__ wphi, T* coef) {
  __shared__ T col[NS][128];
  __shared__ T terms[1024];
  const int v = blockIdx.x, f = threadIdx.x;
  const T mu_v = (T)g.mu_v[v], dphi_v = (T)g.dphi[v];
  T M[NS][NS], dM[NS][NS];
  if (f < nphi) {
    cm_brdf<T, NS>(sf, mu_v, mu0, phi[f], M, dM);
#pragma unroll
    for (int k = 0; k < NS; ++k) col[k][f] = wphi[f] * M[k][0];
  }
  __syncthreads();
  for (int wi = f; wi < (m_max + 1) * NS; wi += 128) {
    const int mm = wi / NS, k = wi % NS;
    T acc = T(0);
    for (int x = 0; x < nphi; ++x) acc += col[k][x] * ((k <= 1) ? cos(T(mm) * phi[x]) : sin(T(mm) * phi[x]));
    const T ff = (mm == 0) ? T(1) : T(2), wm = (mm == 0) ? T(0.5) : T(1);
    const T az = (k <= 1) ? cos(T(mm) * dphi_v) : sin(T(mm) * dphi_v);
    terms[wi] = wm * az * (ff * acc / pi_v<T>());
  }
  __syncthreads();
  if (f == 0) {
    cm_brdf<T, NS>(sf, mu_v, mu0, dphi_v, M, dM);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      T mf = T(0);
      for (int mm = 0; mm <= m_max; ++mm) mf += terms[mm * NS + k];
      coef[v + nV * k] = M[k][0] - mf;
    }
  }
}
template <typename T>
__global__ void k_ss_apply(int nV, int ns, int S, T mu0, const T* __restrict__ tau, const T* __restrict__ coef, T* R) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)nV * ns * S) return;
  const int vk = (int)(e % (nV * ns));
  const long long s = e / (nV * ns);
  R[e] += (mu0 * exp(-tau[s] / mu0)) * coef[vk];
}

template <typename T>
int coxmunk_ss_correction(const cm_surf<T>& sf, int n_stokes, int S, int nV, const T* mu_v_h, const T* dphi_h, T mu0, int m_max,
                          int nphi, const T* phi, const T* wphi, const T* tau_total, T* coef, T* R_SFI, hipStream_t st) {
  if (nV > 64 || nV < 1 || nphi > 128 || nphi < 1 || (m_max + 1) * n_stokes > 1024 || m_max < 0) {
    set_error("coxmunk_ss_correction: 1 <= nV <= 64 viewing geometries, nphi <= 128, (m_max+1) n_stokes <= 1024 (got %d, %d, %d)", nV,
              nphi, m_max);
    return VSM_ERR_UNSUPPORTED;
  }
  ss_geom g;
  for (int v = 0; v < nV; ++v) {
    g.mu_v[v] = (double)mu_v_h[v];
    g.dphi[v] = (double)dphi_h[v];
  }
  const dim3 gr(nV), b(128);
  switch (n_stokes) {
    case 1: hipLaunchKernelGGL((k_cm_ss_coef<T, 1>), gr, b, 0, st, sf, nV, g, mu0, m_max, nphi, phi, wphi, coef); break;
    case 2: hipLaunchKernelGGL((k_cm_ss_coef<T, 2>), gr, b, 0, st, sf, nV, g, mu0, m_max, nphi, phi, wphi, coef); break;
    case 3: hipLaunchKernelGGL((k_cm_ss_coef<T, 3>), gr, b, 0, st, sf, nV, g, mu0, m_max, nphi, phi, wphi, coef); break;
    case 4: hipLaunchKernelGGL((k_cm_ss_coef<T, 4>), gr, b, 0, st, sf, nV, g, mu0, m_max, nphi, phi, wphi, coef); break;
    default: set_error("coxmunk_ss_correction: n_stokes must be 1..4"); return VSM_ERR_INVALID_ARG;
  }
  VSM_LAUNCH_CHECK("k_cm_ss_coef");
  if (S > 0 && R_SFI) {
    const long long tot = (long long)nV * n_stokes * S;
    hipLaunchKernelGGL(k_ss_apply<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, nV, n_stokes, S, mu0, tau_total,
                       coef, R_SFI);
    VSM_LAUNCH_CHECK("k_ss_apply");
  }
  return VSM_OK;
}

// ---- HDRF / BHR diagnostics (CoreKernel/interaction_hdrf.jl:4-42; postprocessing_vza_hdrf!, postprocessing_vza.jl:103-115) ----
// after the surface interaction: hdr_J0-[:, s] = r-+_surf J0+ + j0-_surf (the upwelling field just above the surface) and, for
// m = 0, the hemispheric fluxes per Stokes component  bhr_uw[c, s] = sum_{j = c mod n} hdr_J0-[j] w_j mu_j,
// bhr_dw[c, s] = sum_{j = c mod n} J0+[j] w_j mu_j + j0+_surf[i_mu0 n] mu[i_mu0 n]  (the reference adds the I entry of the direct
// beam to every component).  One workgroup per spectral point.
template <typename T>
__global__ void __launch_bounds__(512) k_interaction_hdrf(int N, int ns, int m, int i_mu0, const T* __restrict__ mu,
                                                          const T* __restrict__ wt, const T* __restrict__ r_mp, long long rstride,
                                                          const T* __restrict__ j0_p, const T* __restrict__ j0_m,
                                                          const T* __restrict__ J0_p, T* hdr_J, T* bhr_uw, T* bhr_dw) {
  __shared__ T Jp[512];
  __shared__ T up[512];
  const long long s = blockIdx.x;
  const int i = threadIdx.x;
  Jp[i] = (i < N) ? J0_p[s * N + i] : T(0);
  __syncthreads();
  T h = T(0);
  if (i < N) {
    const T* r = r_mp + s * rstride;
    for (int k = 0; k < N; ++k) h += r[i + (long long)N * k] * Jp[k];
    h += j0_m[s * N + i];
    hdr_J[s * N + i] = h;
  }
  up[i] = h;
  __syncthreads();
  if (m == 0 && i < ns) {
    T u = T(0), d = T(0);
    for (int j = i; j < N; j += ns) {
      u += up[j] * wt[j] * mu[j];
      d += Jp[j] * wt[j] * mu[j];
    }
    const int i0 = ns * i_mu0;
    bhr_uw[i + (long long)ns * s] = u;
    bhr_dw[i + (long long)ns * s] = d + j0_p[s * N + i0] * mu[i0];
  }
}
template <typename T>
__global__ void k_postprocess_hdrf(int N, int ns, long long S, int nV, int nVtot, int v0, pp_args_s pa, const T* __restrict__ hdr_J,
                                   T* hdr) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)nV * ns * S) return;
  const int v = (int)(e % nV), k = (int)((e / nV) % ns);
  const long long s = e / ((long long)nV * ns);
  hdr[(s * ns + k) * nVtot + v0 + v] += (T)pa.w[v + nV * k] * hdr_J[s * N + pa.row0[v] + k];
}
template <typename T>
int interaction_hdrf(const quad<T>& q, int S, int m, const composite<T>& c, const added<T>& a, T* hdr_J, T* bhr_uw, T* bhr_dw,
                     hipStream_t st) {
  if (q.N > 512) {
    set_error("interaction_hdrf: N <= 512 (got %d)", q.N);
    return VSM_ERR_UNSUPPORTED;
  }
  if (S <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_interaction_hdrf<T>, dim3(S), dim3(q.N <= 128 ? 128 : 512), 0, st, q.N, q.n_stokes, m, q.i_mu0, q.mu, q.wt, a.r_mp,
                     a.mat_stride, a.j0_p, a.j0_m, c.J0_p, hdr_J, bhr_uw, bhr_dw);
  VSM_LAUNCH_CHECK("k_interaction_hdrf");
  return VSM_OK;
}
template <typename T>
int postprocess_vza_hdrf(int N, int ns, int S, int nV, const int* row0_h, const T* w_h, const T* hdr_J, T* hdr, hipStream_t st) {
  if (ns > 4) {
    set_error("postprocess_vza_hdrf: n_stokes <= 4 (got %d)", ns);
    return VSM_ERR_UNSUPPORTED;
  }
  if (S <= 0 || nV <= 0) return VSM_OK;
  for (int v0 = 0; v0 < nV; v0 += 64) {
    const int nc = nV - v0 < 64 ? nV - v0 : 64;
    pp_args_s pa;
    for (int v = 0; v < nc; ++v) pa.row0[v] = row0_h[v0 + v];
    for (int k = 0; k < ns; ++k)
      for (int v = 0; v < nc; ++v) pa.w[v + nc * k] = (double)w_h[v0 + v + nV * k];
    const long long tot = (long long)nc * ns * S;
    hipLaunchKernelGGL(k_postprocess_hdrf<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, ns, (long long)S, nc, nV, v0,
                       pa, hdr_J, hdr);
    VSM_LAUNCH_CHECK("k_postprocess_hdrf");
  }
  return VSM_OK;
}

#define VSM_INST_SURF(T)                                                                                                     \
  template int coxmunk_reflectance<T>(const cm_surf<T>&, int, int, const T*, int, int, const T*, const T*, T*, T*, hipStream_t); \
  template int brdf_surface<T>(const quad<T>&, int, int, const T*, const T*, const added<T>&, hipStream_t);                   \
  template int lambertian_surface_spectral<T>(const quad<T>&, int, int, const T*, const T*, const added<T>&, hipStream_t);    \
  template int brdf_surface_lin<T>(const quad<T>&, int, int, const T*, const T*, int, const T*, const T*, int, const T*,      \
                                   const added<T>&, const added_lin<T>&, hipStream_t);                                        \
  template int interaction_hdrf<T>(const quad<T>&, int, int, const composite<T>&, const added<T>&, T*, T*, T*, hipStream_t);  \
  template int postprocess_vza_hdrf<T>(int, int, int, int, const int*, const T*, const T*, T*, hipStream_t);                  \
  template int coxmunk_ss_correction<T>(const cm_surf<T>&, int, int, int, const T*, const T*, T, int, int, const T*, const T*, \
                                        const T*, T*, T*, hipStream_t);
VSM_INST_SURF(double)
VSM_INST_SURF(float)

}  // namespace vsm
