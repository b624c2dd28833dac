// Typed (templated) mirrors of the C-ABI structs + declarations of the launchers
// implemented in vsm_generic.hip / vsm_fused.hip and called from vsm_api.hip.
#pragma once
#include "vsm_common.h"

namespace vsm {

template <typename T>
struct quad {
  const T* mu;
  const T* wt;
  int N;
  int n_stokes;
  int i_mu0;
  T mu0;
};
template <typename T>
struct added {
  T *r_mp, *t_pp, *r_pm, *t_mm, *j0_p, *j0_m;
  long long mat_stride;
  int d_symmetric;  // 0, or nStokes: r+-/t-- derived by D-symmetry, never touched
};
template <typename T>
struct composite {
  T *R_mp, *R_pm, *T_pp, *T_mm, *J0_p, *J0_m;
};

// ---- generic (operator-level) path: vsm_generic.hip --------------------------
template <typename T>
int gemm(int M, int Nc, int K, int S, const T* A, long long sa, const T* B, long long sb, T* C, long long sc,
         T alpha, const T* D, long long sd, T beta, T gamma, hipStream_t st);
template <typename T>
int gemm2(int M, int Nc, int K, int S, int P, const T* A, long long sa, long long pa, const T* B, long long sb,
          long long pb, T* C, long long sc, long long pc, T alpha, const T* D, long long sd, long long pd, T beta,
          T gamma, hipStream_t st);
template <typename T>
int batch_inv(int N, int S, const T* A, T* X, int* info, hipStream_t st);
template <typename T>
int elemental(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
              const T* F0, const T* Zpp, const T* Zmp, long long zs, const added<T>& a, hipStream_t st);
template <typename T>
int doubling(int N, int n_stokes, int S, int ndoubl, T* expk, const added<T>& a, T* work, hipStream_t st);
template <typename T>
int noscat_layer(const quad<T>& q, int S, const T* tau, const added<T>& a, hipStream_t st);
template <typename T>
int thermal_source(const quad<T>& q, int S, const T* dtau, const T* varpi, const T* B, const added<T>& a, hipStream_t st);
template <typename T>
int copy_added_to_composite(int N, int S, const added<T>& a, const composite<T>& c, hipStream_t st);
template <typename T>
int interaction_generic(int iface, int N, int S, const composite<T>& c, const added<T>& a, T* work, hipStream_t st);
template <typename T>
int lambertian_surface(const quad<T>& q, int S, int m, T albedo, const T* tau_sum, const added<T>& a, hipStream_t st);
template <typename T>
int postprocess_vza(int N, int n_stokes, int S, int nV, const int* row0_h, const T* w_h, const T* J0_m, const T* J0_p,
                    T* R, T* Tt, hipStream_t st);
template <typename T>
int copy_strided(long long per, int S, const T* src, long long ss, T* dst, hipStream_t st);

// ---- linearized (Jacobian) pass, operator level: vsm_lin.hip ---------------------------------------
template <typename T>
struct added_lin {
  T *ap_r_mp, *ap_t_pp, *ap_r_pm, *ap_t_mm, *ap_J0_p, *ap_J0_m;  // [N,N,S,P] / [N,1,S,P]
  int P;
  long long mat_stride;  // N*N, or 0: ONE block per parameter shared by all S (surface)
};
template <typename T>
struct composite_lin {
  T *R_mp, *R_pm, *T_pp, *T_mm, *J0_p, *J0_m;
  int P;
};
template <typename T>
int elemental_lin(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                  const T* F0, const T* Zpp, const T* Zmp, long long zs, int p_layer, const T* dtau_dot,
                  const T* varpi_dot, const T* tau_sum_dot, const T* Zpp_dot, const T* Zmp_dot, long long zds,
                  long long zdp, const added<T>& a, const added_lin<T>& al, hipStream_t st, int n_m0 = 0);
template <typename T>
int elemental_lin_mix(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                      const T* F0, int ncomp, int ncomp_total, const T* Zc_pp, const T* Zc_mp, int zsel, const T* fz,
                      int p_layer, const T* dtau_dot, const T* varpi_dot, const T* tau_sum_dot, const T* zdcoef,
                      const added<T>& a, const added_lin<T>& al, hipStream_t st);
template <typename T>
size_t doubling_lin_work_elems(int N, int S, int P);
template <typename T>
int doubling_lin(int N, int n_stokes, int S, int ndoubl, T* expk, const T* dtau_dot_all, T mu0, int n_active,
                 const added<T>& a, const added_lin<T>& al, T* work, hipStream_t st);
template <typename T>
size_t interaction_lin_work_elems(int N, int S, int P);
template <typename T>
int interaction_lin(int iface, int N, int S, const composite<T>& c, const composite_lin<T>& cl, const added<T>& a,
                    const added_lin<T>& al, T* work, hipStream_t st);
template <typename T>
int lambertian_surface_lin(const quad<T>& q, int S, int m, T albedo, int iparam, const T* tau_sum,
                           const T* tau_sum_dot, int p_layer, const T* F0, const added<T>& a, const added_lin<T>& al,
                           hipStream_t st);
template <typename T>
int copy_added_to_composite_lin(int N, int S, const added_lin<T>& al, const composite_lin<T>& cl, hipStream_t st);
template <typename T>
int postprocess_vza_lin(int N, int n_stokes, int S, int nV, int P, const int* row0_h, const T* w_h, const T* Jd_m,
                        const T* Jd_p, T* Rd, T* Td, hipStream_t st);

// ---- BRDF surfaces (Cox-Munk reflectance, generic BRDF surface layer, TMS correction): vsm_surface.hip ----
template <typename T>
struct cm_surf {
  T wind_speed, n_re, n_im, whitecap_albedo;
  int include_whitecaps, shadowing;
};
template <typename T>
int coxmunk_reflectance(const cm_surf<T>& sf, int n_stokes, int Nmu, const T* muN, int m, int nphi, const T* phi, const T* wphi,
                        T* rho, T* drho, hipStream_t st);
template <typename T>
int brdf_surface(const quad<T>& q, int S, int m, const T* rho, const T* tau_sum, const added<T>& a, hipStream_t st);
template <typename T>
int lambertian_surface_spectral(const quad<T>& q, int S, int m, const T* albedo, const T* tau_sum, const added<T>& a, hipStream_t st);
template <typename T>
int brdf_surface_lin(const quad<T>& q, int S, int m, const T* rho, const T* drho, int iparam, const T* tau_sum,
                     const T* tau_sum_dot, int p_layer, const T* F0, const added<T>& a, const added_lin<T>& al, hipStream_t st);
template <typename T>
int interaction_hdrf(const quad<T>& q, int S, int m, const composite<T>& c, const added<T>& a, T* hdr_J, T* bhr_uw, T* bhr_dw,
                     hipStream_t st);
template <typename T>
int postprocess_vza_hdrf(int N, int ns, int S, int nV, const int* row0_h, const T* w_h, const T* hdr_J, T* hdr, hipStream_t st);
template <typename T>
int coxmunk_ss_correction(const cm_surf<T>& sf, int n_stokes, int S, int nV, const T* mu_v_h, const T* dphi_h, T mu0, int m_max,
                          int nphi, const T* phi, const T* wphi, const T* tau_total, T* coef, T* R_SFI, hipStream_t st);

// ---- per-scene layer optics on the device: vsm_optics.hip ----
template <typename T>
int compute_Z_moments(int Nq, int n_stokes, const T* muN, int m, int lmax, const double* greek, T* Zpp, T* Zmp, hipStream_t st);
template <typename T>
int layer_optics(int S, int L, int nAer, const double* tau_rayl, const double* tau_abs, double varpi_cab, const double* tau_aer,
                 const double* ssa, const double* ftrunc, const int* mode, T* tau, T* varpi, T* tau_sum, T* fcomp, T* max_tw,
                 hipStream_t st);
template <typename T>
int layer_dtau(int S, int L, const int* nd, const T* tau, T* dtau, hipStream_t st);
template <typename T>
int layer_optics_lin(int S_full, int lo, int S, int L, int nAer, int nGas, int P, const double* tau_rayl, const double* tau_abs,
                     double varpi_cab, const double* tau_aer, const double* ssa, const double* ftrunc, const double* tau_abs_dot,
                     const double* tau_aer_dot, const double* ssa_dot, const double* ftrunc_dot, const int* nd, T* dtau_dot_all,
                     T* varpi_dot, T* tau_sum_dot, T* fz, T* zdcoef, hipStream_t st);
template <typename T>
int layer_expk(int S, const T* dtau, T mu0, T* expk, hipStream_t st);

// ---- fused (LDS-resident) path: vsm_fused.hip ---------------------------------
template <typename T>
int fused_max_n();
// elemental + ndoubl doublings + apply_D in one launch, one workgroup per spectral point
template <typename T>
int fused_elemental_doubling(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi,
                             const T* tau_sum, const T* F0, const T* Zpp, const T* Zmp, long long zs,
                             const added<T>& a, hipStream_t st);
template <typename T>
int fused_interaction(int iface, int N, int S, const composite<T>& c, const added<T>& a, hipStream_t st);
// X = (I - A B)^-1, product + series/Gauss-Jordan inverse in one LDS-resident launch (N <= fused_max_n, else
// VSM_ERR_UNSUPPORTED).  sa / sb: element strides between spectral slices (0 = shared block).
template <typename T>
int inv_one_minus_product(int N, int S, const T* A, long long sa, const T* B, long long sb, T* X, hipStream_t st);
bool strip128_supported(int N);
int strip128_inv_one_minus(int N, int S, const double* A, long long sa, const double* B, long long sb, double* X, hipStream_t st);
// (I - A B)^-1 by the fused kernel when N fits, else gemm + batch_inv through `tmp` ([N,N,S] scratch)
template <typename T>
inline int inv_one_minus(int N, int S, const T* A, long long sa, const T* B, long long sb, T* X, T* tmp, hipStream_t st) {
  int rc = inv_one_minus_product<T>(N, S, A, sa, B, sb, X, st);
  if (rc != VSM_ERR_UNSUPPORTED) return rc;
  if constexpr (sizeof(T) == 8) {   // FP64, 64 < N <= 128: product + series / squaring levels in one strip kernel
    if (strip128_supported(N)) return strip128_inv_one_minus(N, S, A, sa, B, sb, X, st);
  }
  const long long NN = (long long)N * N;
  if ((rc = gemm<T>(N, N, N, S, A, sa, B, sb, tmp, NN, T(-1), (const T*)nullptr, 0, T(0), T(1), st))) return rc;
  return batch_inv<T>(N, S, tmp, X, nullptr, st);
}
// Raman: inelastic part of one doubling step for all lines, LDS-resident (N <= 30; else VSM_ERR_UNSUPPORTED)
// One pass (G1/T01 or G2/T21) of the inelastic ScatteringInterface_11 interaction, all lines of a recipient point in one
// workgroup (vsm_fused.hip: k_raman_interaction_lines).  "4d": inelastic arrays [N,N,S,K] / [N,S,K] at (n1, line);
// "n0" / "n1": elastic arrays at the donor / recipient point.
template <typename T>
struct rs_ia_pass {
  const T *L1 /*4d, left*/, *E0 /*n0*/, *L2 /*n1, left*/, *I1 /*4d*/, *TI /*n1, left*/, *YA /*4d addend*/;
  const T *E3 /*n0*/, *I3 /*4d*/, *ACCA /*4d addend*/, *GX /*n0*/, *I4 /*4d*/, *GY /*n0*/;
  long long sE0, sL2, sE3;          // strides over the spectral axis of elastic operands that may be shared (0)
  T *OUTA, *OUTB;                   // 4d
  const T *VE0 /*n0*/, *VADD /*4d*/, *VI1 /*4d*/, *VACC /*4d*/, *VV /*n0*/;
  T* VOUT;                          // 4d
};
template <typename T>
int raman_interaction_lines(int N, int S, int K, const int* shift, const rs_ia_pass<T>& h, hipStream_t st);
// elastic part of one Raman doubling step, LDS-resident per spectral point (vsm_fused.hip); VSM_ERR_UNSUPPORTED past the
// on-chip limit
template <typename T>
int raman_elastic_pre(int N, int S, const T* r, const T* t, const T* j0p, const T* j0m, const T* expk, T* ttg, T* gt, T* gr,
                      T* grt, T* j1p, T* j1m, T* u, T* u2, T* tmp1, T* tmp2, hipStream_t st);
template <typename T>
int raman_elastic_post(int N, int S, T* r, T* t, const T* ttg, const T* u, const T* u2, const T* j1p, T* j0p, T* j0m,
                       T* expk, hipStream_t st);
// one wave per Raman line (vsm_raman_wave.hip): FP64, N <= 30
int raman_interaction_wave(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st);
int raman_doubling_wave(int N, int S, int K, const int* shift, const double* r, const double* t, const double* ttg,
                        const double* gt, const double* gr, const double* grt, const double* jp, const double* j1m,
                        const double* tmp1, const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp,
                        double* ieJm, int ns_last, double* ier_pm, double* iet_mm, hipStream_t st);
// four lines per wave on the 4 x 4 x 4 MFMA (vsm_raman_quad.hip): FP64, 3 <= N <= 22
int raman_doubling_quad(int N, int S, int K, const int* shift, const double* r, const double* t, const double* ttg,
                        const double* gt, const double* gr, const double* grt, const double* jp, const double* j1m,
                        const double* tmp1, const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp,
                        double* ieJm, int ns_last, double* ier_pm, double* iet_mm, hipStream_t st);
int raman_interaction_quad(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st);
// all doubling steps of the inelastic recurrences in one launch, the state of a line on chip (vsm_raman_chain.hip: FP64, 20 <= N <= 22); `stash`: the
// elastic operands of every step (which: 0..5 = r, t, ttg, gt, gr, grt; 6..9 = jp, j1m, tmp1, tmp2; 10 = expk)
bool raman_chain_supported(int N, int K);
size_t raman_chain_stash_elems(int N, int S, int nd);
double* raman_chain_stash_ptr(double* stash, int N, int S, int nd, int step, int which);
int raman_doubling_chain(int N, int S, int K, int nd, const int* shift, double* stash, double* ier, double* iet, double* ieJp,
                         double* ieJm, int ns, double* ier_pm, double* iet_mm, hipStream_t st);
template <typename T>
int raman_doubling_lines(int N, int S, int K, const int* shift, const T* r, const T* t, const T* ttg, const T* gt, const T* gr,
                         const T* grt, const T* jp, const T* j1m, const T* tmp1, const T* tmp2, const T* expk, T* ier, T* iet,
                         T* ieJp, T* ieJm, hipStream_t st);
template <typename T>
int test_lds_mm(int N, int S, const T* A, const T* B, T* C, hipStream_t st);
template <typename T>
int test_lds_inv(int N, int S, const T* A, T* X, int mode, int* path_out, hipStream_t st);

// Where a layer's Z++ / Z-+ come from: one N x N block per point (stride zs; 0 = shared by all points), or -- ncomp > 0
// -- the per-point mix  sum_k fcomp[s, k] Z_k  of ncomp <= 4 component blocks stacked in Zpp / Zmp ([N,N,ncomp]).
template <typename T>
struct zsrc {
  const T* Zpp;
  const T* Zmp;
  long long zs;
  int ncomp;
  const T* fcomp;  // [ncomp, S] column-major: fcomp[s * ncomp + k]
};
template <typename T>
int mix_Z(int N, int S, int ncomp, const T* Zpp_comp, const T* Zmp_comp, const T* fcomp, T* Zpp, T* Zmp, hipStream_t st);
template <typename T>
int mix_Z_moments(int N, int S, int ncomp, int nm, const T* const* Zpp_comp, const T* const* Zmp_comp, int single, const T* fcomp,
                  T* Zpp, T* Zmp, hipStream_t st);

// ---- column-strip kernels (FP64, 32 < N <= 60; two workgroups per CU): vsm_strip.hip ----------------
bool strip_supported(int N);
bool strip_layer_supported(int N);   // the layer step and interaction!(_11): 32 < N <= 64
int strip_elemental_doubling(const quad<double>& q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                             const double* tau_sum, const double* F0, const zsrc<double>& z, const added<double>& a,
                             hipStream_t st);

int strip_gemm(int M, int Nc, int K, int S, int P, const double* A, long long sa, long long pa, const double* B, long long sb,
               long long pb, double* C, long long sc, long long pc, double alpha, const double* D, long long sd, long long pd,
               double beta, double gamma, hipStream_t st);
int strip_interaction11(int N, int S, const composite<double>& c, const added<double>& a, hipStream_t st);
int strip_doubling_lin_step(int N, int S, int P, double* expk, double* ekl, const added<double>& a,
                            const added_lin<double>& al, hipStream_t st);
int strip_doubling_lin_multi(int N, int S, int P, int nd, int ns, double* expk, double* ekl, const added<double>& a,
                             const added_lin<double>& al, hipStream_t st);
// several Fourier moments of one layer in ONE launch (gridDim.y = nm <= VSM_MM_MAX): same dtau / varpi / tau_sum / F0, per
// moment its m, Z source and composite -- three times the workgroups per launch, i.e. a third of the launch tails
constexpr int VSM_MM_MAX = 24;   // (the argument block stays under the 4 KB kernel-argument limit: 24 x (m, Z source, composite) = 2.2 KB)
template <typename T>
struct layer_mm_args {
  int m[VSM_MM_MAX];
  zsrc<T> z[VSM_MM_MAX];
  composite<T> c[VSM_MM_MAX];
};
int strip_layer_forward_mm(const quad<double>& q, int S, int nm, int ndoubl, const double* dtau, const double* varpi,
                           const double* tau_sum, const double* F0, const layer_mm_args<double>& a, int toa, hipStream_t st);
// thermal != 0: the layer's `:thermal` source slot (F0 = B[S], expk = 1) instead of the solar beam
int strip_layer_forward(const quad<double>& q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                        const double* tau_sum, const double* F0, const zsrc<double>& z, int toa, const composite<double>& c,
                        hipStream_t st, int thermal = 0);

// ---- native-layout family (vsm_native.hip): interaction!(_11) on the reference's arrays, FP64, N <= 64 ----
int native_interaction11(int N, int S, const composite<double>& c, const added<double>& a, hipStream_t st);

// ---- column-strip kernels, FP32 (64 < N <= 96; two workgroups of 6 waves per CU): vsm_strip32.hip ----
bool strip32_supported(int N);
int strip32_layer_forward(const quad<float>& q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const zsrc<float>& z, int toa, const composite<float>& c,
                          hipStream_t st, int thermal = 0);
int strip32_layer_forward_mm(const quad<float>& q, int S, int nm, int ndoubl, const float* dtau, const float* varpi,
                             const float* tau_sum, const float* F0, const layer_mm_args<float>& a, int toa, hipStream_t st);
int strip32_interaction11(int N, int S, const composite<float>& c, const added<float>& a, hipStream_t st);

// ---- column-strip kernels, FP64, 64 < N <= 126 (one persistent workgroup of <= 8 waves per CU): vsm_strip128.hip ----
bool strip128_supported(int N);
template <typename ST>   // ST = double, or float (Float32 runs of 96 < N <= 128: storage in single, arithmetic in double)
int strip128_doubling(int N, int n_stokes, int S, int ndoubl, ST* expk, const added<ST>& a, hipStream_t st);
template <typename ST>
int strip128_interaction11(int N, int S, const composite<ST>& c, const added<ST>& a, hipStream_t st);
inline bool strip128_f32_supported(int N) { return N > 96 && N <= 128; }   // (64 < N <= 96 is the FP32 strip kernels')

// ---- linearized column-strip kernels, FP64, 60 < N <= 128 (one A-form, parked strips): vsm_strip128lin.hip ----
bool strip128_lin_dbl_supported(int N);   // which FP64 shapes take k_dbl128_lin (k_ia128_lin: every N <= 128)
template <typename ST>   // ST = double, or float (storage in single, arithmetic in double)
int strip128_doubling_lin(int N, int S, int P, int nd, int ns, ST* expk, ST* ekl, const added<ST>& a, const added_lin<ST>& al,
                          hipStream_t st);
template <typename ST>
int strip128_interaction11_lin(int N, int S, const composite<ST>& c, const composite_lin<ST>& cl, const added<ST>& a,
                               const added_lin<ST>& al, hipStream_t st);

// Library-owned device scratch, keyed by (current device, stream, slot): two streams -- or two devices driven from one
// process -- never share a buffer, so the entry points that use it keep the contract "calls on one stream are ordered, calls on
// different streams are independent".  Grow-only per key; a buffer that is outgrown is retired behind an event recorded on
// its stream and freed only once that event has completed (never under running work).  nullptr + error set on failure.
void* scratch(size_t bytes, int slot, hipStream_t st);
int* device_status();   // 4 zero-initialised device words of the current device (vsm_device_status); nullptr + error set on failure
int release_scratch();   // current device: synchronise, free every scratch buffer
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: raise it once per (device, kernel).
int ensure_dyn_lds(const void* kern, size_t bytes, const char* what);
// compute units of the CURRENT device (cached per device)
int cu_count();

}  // namespace vsm
