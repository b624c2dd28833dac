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
int batch_inv(int N, int S, const T* A, T* X, int* info, hipStream_t st);
template <typename T>
int elemental(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
              const T* F0, const T* Zpp, const T* Zmp, long long zs, const added<T>& a, hipStream_t st);
template <typename T>
int doubling(int N, int n_stokes, int S, int ndoubl, T* expk, const added<T>& a, T* work, hipStream_t st);
template <typename T>
int noscat_layer(const quad<T>& q, int S, const T* tau, const added<T>& a, hipStream_t st);
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
template <typename T>
int test_lds_mm(int N, int S, const T* A, const T* B, T* C, hipStream_t st);
template <typename T>
int test_lds_inv(int N, int S, const T* A, T* X, int mode, int* path_out, hipStream_t st);

// grow-only device scratch (one per element type); not for concurrent streams.
void* scratch(size_t bytes, int slot);

}  // namespace vsm
