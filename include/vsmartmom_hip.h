/*
 * vsmartmom_hip.h -- C ABI of libvsmartmom_hip.so, the MI355X (gfx950) engine for
 * vSmartMOM.jl's rt_run CoreRT hot path.
 *
 * This is the drop-in boundary: every entry point replaces one
 * method that the reference's CUDA extension defines for CuArray, or one
 * host-level CoreKernel function whose KernelAbstractions kernels the AMD
 * backend overrides.  The reference-side binding (Julia `ccall`) is shown in
 * INTEGRATION.md and julia/vSmartMOMROCmExt.jl.
 *
 * Conventions
 *  - All array pointers are DEVICE pointers (HBM) unless the name ends in _h.
 *  - Layout is the reference's: Julia column-major. A matrix batch `A[N,N,S]`
 *    has element (i,j,s) at  i + N*j + N*N*s  (0-based); a vector batch
 *    `v[N,1,S]` has (i,s) at i + N*s. Per-spectral-point scalars are `x[S]`.
 *    Each spectral point's N*N block is contiguous; the spectral (batch) axis is
 *    the slowest, so one workgroup streams one point with fully coalesced reads.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls
 *    are asynchronous w.r.t. the host, like the CUBLAS calls they replace.
 *  - STREAM CONTRACT (every entry point that takes `stream`): all device work of
 *    a call is enqueued on `stream` and nowhere else; calls on one stream execute
 *    in call order; calls on DIFFERENT streams are independent and may be issued
 *    concurrently from one or several host threads as long as the arrays they
 *    write do not overlap.  Scratch the caller passes (`work`, `z_scratch`, ...)
 *    belongs to the call's stream until the call's work has completed.  Scratch
 *    the LIBRARY owns (the entry points marked "library scratch" below) is keyed
 *    by (current device, stream): two streams never share it, an outgrown buffer
 *    is freed only after the work queued on its stream has completed, and no
 *    call synchronises the device.  vsm_release_scratch() returns it.
 *  - DEVICE CONTRACT: a call runs on the device that is current in the calling
 *    host thread (hipSetDevice) and `stream` must belong to that device.  One
 *    process may drive several devices (one host thread per device or
 *    hipSetDevice between calls): per-kernel launch attributes and the library
 *    scratch are kept per device.  The library never calls hipSetDevice.
 *  - Return value: 0 = VSM_OK, otherwise a vsm_status; vsm_last_error() returns
 *    a thread-local message.  No C++ exception crosses this boundary.
 *  - Suffix _f64 / _f32 = the reference's `FT` (Float64 / Float32).
 *  - N = Nquad*nStokes (the reference's NquadN), S = nSpec.
 */
#ifndef VSMARTMOM_HIP_H
#define VSMARTMOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vsm_status {
  VSM_OK = 0,
  VSM_ERR_INVALID_ARG = 1,   /* bad size / null pointer */
  VSM_ERR_UNSUPPORTED = 2,   /* shape outside what the kernels are built for */
  VSM_ERR_HIP = 3,           /* a HIP runtime call failed; see vsm_last_error() */
  VSM_ERR_NO_DEVICE = 4
} vsm_status;

/* ScatteringInterface tags (src/CoreRT/types.jl ScatteringInterface_00/01/10/11;
 * selection rule src/CoreRT/tools/rt_helper_functions.jl:15-33). */
typedef enum vsm_iface {
  VSM_IFACE_00 = 0,
  VSM_IFACE_01 = 1,
  VSM_IFACE_10 = 2,
  VSM_IFACE_11 = 3
} vsm_iface;

/* ---- library / device ---------------------------------------------------- */
int vsm_version(void);                 /* 10000*major + 100*minor + patch */
const char* vsm_last_error(void);      /* thread-local, never NULL */
const char* vsm_build_id(void);        /* 16 hex digits: SHA-256 over the library's sources (csrc/Makefile); profiles name it */
int vsm_device_count(int* count);      /* Architectures.jl:68-96 `_has_cuda`-style probe */
int vsm_device_name(int device, char* buf, size_t buflen);
int vsm_sync(void* stream);            /* Architectures.synchronize_if_gpu (Architectures.jl:96) */
/* Device-side status of kernels that invert inside an asynchronous launch and therefore cannot return `info` the way
 * vsm_batch_inv does (the reference's LU raises SingularException on the host, cpu_batched.jl:32-47).  flags_h[0] = OR of the
 * vsm_devstat bits raised on the CURRENT device since the last reset; flags_h[1] = number of pivoted (Gauss-Jordan) inverses the
 * 64 < N <= 128 kernels ran (k_dbl128 / k_ia128 / k_inv1m128: every inverse whose norm bound gives no series order);
 * flags_h[2] / flags_h[3] = spectral points k_dbl128 / k_ia128 processed (which kernel family a run landed on).  SYNCHRONOUS: waits for `stream` before it reads.  reset != 0 clears the words afterwards. */
typedef enum vsm_devstat {
  VSM_DEVSTAT_SINGULAR = 1,    /* an exactly zero pivot: (I - R r) or (I - r r) is singular; the results of that point are not finite */
  VSM_DEVSTAT_NONFINITE = 2,   /* an operand of an in-kernel inverse was NaN / Inf */
  VSM_DEVSTAT_MASK = 4         /* vsm_run_layer: a phase matrix has a non-zero element outside the declared Stokes coupling (the run's
                                  `coupling`, or a block that `layer_coupling_h` called zero): that coupling was dropped, the results of
                                  the run are not those of the dense walk */
} vsm_devstat;
int vsm_device_status(int* flags_h, int reset, void* stream);
/* Frees the library-owned scratch of the CURRENT device (all streams) after a
 * device synchronisation.  Optional: the scratch is grow-only and reused. */
int vsm_release_scratch(void);
/* Largest N the fused (LDS-resident) layer kernels accept for the element size
 * (8 = f64, 4 = f32); larger N use the operator-level kernels below. */
int vsm_fused_max_n(int elem_size);

/* ---- L1 operator API: batched_mul / batch_inv! ---------------------------
 * Replace ext/gpu_batched_cuda.jl:208-233 (CUBLAS.gemm_strided_batched) and
 * :97-182 (getrf/getri batched); CPU semantics src/CoreRT/tools/cpu_batched.jl:25-82.
 * C[M,Nc,S] = A[M,K,S] * B[K,Nc,S].  A batch stride of 0 broadcasts one matrix
 * over all S (sa/sb are element strides between consecutive slices; pass M*K and
 * K*Nc for dense batches). S==1 is handled natively (no singleton-batch guard needed). */
int vsm_batched_mul_f64(int M, int Nc, int K, int S, const double* A, long long sa,
                        const double* B, long long sb, double* C, void* stream);
int vsm_batched_mul_f32(int M, int Nc, int K, int S, const float* A, long long sa,
                        const float* B, long long sb, float* C, void* stream);
/* X[:,:,s] = inv(A[:,:,s]), partial pivoting (same pivot rule as getrf). A is NOT
 * clobbered (the reference allows clobbering; callers may alias X == A).
 * info (nullable, int[S]): 0 ok, k>0 = exact zero pivot at step k (LAPACK convention). */
int vsm_batch_inv_f64(int N, int S, const double* A, double* X, int* info, void* stream);
int vsm_batch_inv_f32(int N, int S, const float* A, float* X, int* info, void* stream);

/* batch_solve!(X, A, B) (ext/gpu_batched_cuda.jl:72-94 getrf+getrs; CPU: cpu_batched.jl:25-29): X[:,:,s] = A[:,:,s] \ B[:,:,s],
 * A [N,N,S], B and X [N,Nrhs,S].  Executed as the pivoted Gauss-Jordan inverse of vsm_batch_inv into `work` (N*N*S elements)
 * followed by one batched product; A and B are not clobbered; X must not alias B.  info as in vsm_batch_inv.  (Unused on
 * the reference's forward hot path; exported because it is part of the operator API a backend extension provides.) */
int vsm_batch_solve_f64(int N, int Nrhs, int S, const double* A, const double* B, double* X, double* work, int* info,
                        void* stream);
int vsm_batch_solve_f32(int N, int Nrhs, int S, const float* A, const float* B, float* X, float* work, int* info,
                        void* stream);

/* ---- L2 CoreKernel API ----------------------------------------------------
 * Layer state containers (src/CoreRT/types.jl:155-230 AddedLayer / CompositeLayer). */
typedef struct vsm_added_f64 {
  double *r_mp, *t_pp, *r_pm, *t_mm;  /* r⁻⁺ t⁺⁺ r⁺⁻ t⁻⁻  [N,N,S] */
  double *j0_p, *j0_m;                /* j₀⁺ j₀⁻          [N,1,S] */
  long long mat_stride;               /* element stride between spectral slices of the
                                         matrices: N*N, or 0 if one matrix is shared by all S
                                         (surface layers) */
  int d_symmetric;                    /* 0: all four matrices are materialised (the reference's AddedLayer).
                                         n>0 (= nStokes): r⁺⁻/t⁻⁻ are NOT read or written; consumers derive
                                         them by the D-symmetry r⁺⁻ = D r⁻⁺ D, t⁻⁻ = D t⁺⁺ D
                                         (doubling.jl:178-201), halving the layer's HBM traffic.  Only the
                                         fused kernels accept n>0. */
  int reserved;
} vsm_added_f64;
typedef struct vsm_added_f32 {
  float *r_mp, *t_pp, *r_pm, *t_mm;
  float *j0_p, *j0_m;
  long long mat_stride;
  int d_symmetric;
  int reserved;
} vsm_added_f32;
typedef struct vsm_composite_f64 {
  double *R_mp, *R_pm, *T_pp, *T_mm;  /* R⁻⁺ R⁺⁻ T⁺⁺ T⁻⁻  [N,N,S] */
  double *J0_p, *J0_m;                /* J₀⁺ J₀⁻          [N,1,S] */
} vsm_composite_f64;
typedef struct vsm_composite_f32 {
  float *R_mp, *R_pm, *T_pp, *T_mm;
  float *J0_p, *J0_m;
} vsm_composite_f32;

/* Quadrature + polarization, as rt_kernel! sees them (QuadPoints, types.jl; pol_type.n). */
typedef struct vsm_quad_f64 {
  const double* mu;   /* qp_μN[N]  (device) */
  const double* wt;   /* wt_μN[N]  (device) */
  int N;              /* Nquad*nStokes */
  int n_stokes;       /* pol_type.n  (1..4) */
  int i_mu0;          /* 0-based node index of the SZA stream (iμ₀-1) */
  double mu0;         /* quad_points.μ₀ */
} vsm_quad_f64;
typedef struct vsm_quad_f32 {
  const float* mu;
  const float* wt;
  int N;
  int n_stokes;
  int i_mu0;
  float mu0;
} vsm_quad_f32;

/* elemental! + doubling! for one homogeneous layer and one Fourier moment m
 * (src/CoreRT/CoreKernel/elemental.jl:174-230 with kernels :289-334,:348-392,:403-422;
 *  doubling.jl:38-99 with rt_helpers.jl:102-166 and apply_D doubling.jl:178-252).
 * Inputs per spectral point: dtau[S] (= τ/2^ndoubl, rt_kernel.jl:266-287 -- ndoubl is
 * decided by the host from the batch-global max(τϖ), exactly as the reference does),
 * varpi[S], tau_sum[S] (optical depth above the layer), F0[n_stokes,S].
 * Zpp/Zmp are Z⁺⁺/Z⁻⁺ [N,N,S] with slice stride z_stride (0 = one Z for all S).
 * Fills all six fields of `added`.  ndoubl == 0 reproduces the un-doubled branch.
 * Stream: see Conventions; library scratch (doubling work of the operator-level path, N > vsm_fused_max_n). */
int vsm_elemental_doubling_f64(const vsm_quad_f64* q, int S, int m, int ndoubl,
                               const double* dtau, const double* varpi, const double* tau_sum,
                               const double* F0, const double* Zpp, const double* Zmp,
                               long long z_stride, const vsm_added_f64* added, void* stream);
int vsm_elemental_doubling_f32(const vsm_quad_f32* q, int S, int m, int ndoubl,
                               const float* dtau, const float* varpi, const float* tau_sum,
                               const float* F0, const float* Zpp, const float* Zmp,
                               long long z_stride, const vsm_added_f32* added, void* stream);

/* The two halves separately (operator-for-operator with the reference; used for
 * N above vsm_fused_max_n and by the per-kernel parity tests).  vsm_doubling_f64 with 64 < N <= 128 is ONE launch for all
 * ndoubl steps (k_dbl128); vsm_interaction_f64 (interface 11) likewise one launch (k_ia128).
 * Stream: see Conventions; vsm_doubling_f64 / vsm_interaction_f64 with 64 < N <= 128 park strips in library scratch (keyed by
 * device and stream). */
int vsm_elemental_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau,
                      const double* varpi, const double* tau_sum, const double* F0,
                      const double* Zpp, const double* Zmp, long long z_stride,
                      const vsm_added_f64* added, void* stream);
int vsm_elemental_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau,
                      const float* varpi, const float* tau_sum, const float* F0,
                      const float* Zpp, const float* Zmp, long long z_stride,
                      const vsm_added_f32* added, void* stream);
/* doubling!: expk[S] = exp(-dtau/μ₀) is updated in place (squared ndoubl times) like the
 * reference does.  work = scratch of vsm_doubling_work_elems(N,S) elements. */
size_t vsm_doubling_work_elems(int N, int S);
int vsm_doubling_f64(int N, int n_stokes, int S, int ndoubl, double* expk,
                     const vsm_added_f64* added, double* work, void* stream);
int vsm_doubling_f32(int N, int n_stokes, int S, int ndoubl, float* expk,
                     const vsm_added_f32* added, float* work, void* stream);

/* Non-scattering layer (rt_helpers.jl:174-180 zero_added_noscat! + rt_kernel.jl:36-45):
 * r⁻⁺ = r⁺⁻ = 0, j₀⁻ = 0, t±± = diag exp(-τ/μ).  j₀⁺ is left untouched, as in the reference. */
int vsm_noscat_layer_f64(const vsm_quad_f64* q, int S, const double* tau,
                         const vsm_added_f64* added, void* stream);
int vsm_noscat_layer_f32(const vsm_quad_f32* q, int S, const float* tau,
                         const vsm_added_f32* added, void* stream);

/* Thermal-emission source slot (`:thermal` of j₀_by_src): contribute!(::PreparedThermalEmission, ...)
 * (src/CoreRT/Sources/thermal_emission.jl:241-301) -- j₀⁺ = j₀⁻ = 2π (1-ϖ) B (1 - exp(-dτ/μᵢ)) on the Stokes-I rows of the
 * elemental layer, zero on Q/U/V; B[S] = Planck radiance of the layer per spectral point.  Written into added->j0_p / j0_m
 * AFTER vsm_elemental_* and BEFORE vsm_doubling_*, which then runs with the slot's own expk = 1 (doubling.jl:62-81). */
int vsm_thermal_source_f64(const vsm_quad_f64* q, int S, const double* dtau, const double* varpi, const double* B,
                           const vsm_added_f64* added, void* stream);
int vsm_thermal_source_f32(const vsm_quad_f32* q, int S, const float* dtau, const float* varpi, const float* B,
                           const vsm_added_f32* added, void* stream);

/* copy_added_to_composite! (rt_helpers.jl:188-200), TOA layer. */
int vsm_copy_added_to_composite_f64(int N, int S, const vsm_added_f64* added,
                                    const vsm_composite_f64* comp, void* stream);
int vsm_copy_added_to_composite_f32(int N, int S, const vsm_added_f32* added,
                                    const vsm_composite_f32* comp, void* stream);

/* interaction! (src/CoreRT/CoreKernel/interaction.jl:52-285): composite (above) ⊕ added
 * (below) -> composite, in place, statement order as in the reference.
 * work = scratch of vsm_interaction_work_elems(N,S) elements (only used when
 * N > vsm_fused_max_n; may be NULL otherwise -- and NULL beyond the fused limit = library scratch).
 * Stream: see Conventions. */
size_t vsm_interaction_work_elems(int N, int S);
int vsm_interaction_f64(int iface, int N, int S, const vsm_composite_f64* comp,
                        const vsm_added_f64* added, double* work, void* stream);
int vsm_interaction_f32(int iface, int N, int S, const vsm_composite_f32* comp,
                        const vsm_added_f32* added, float* work, void* stream);
/* Same contract, but always executed operator-for-operator (batched products + batch_inv!,
 * the way the reference issues it); work must be non-NULL.  Used for N above the fused limit
 * and as an independent cross-check of the fused kernel. */
int vsm_interaction_oplevel_f64(int iface, int N, int S, const vsm_composite_f64* comp,
                                const vsm_added_f64* added, double* work, void* stream);
int vsm_interaction_oplevel_f32(int iface, int N, int S, const vsm_composite_f32* comp,
                                const vsm_added_f32* added, float* work, void* stream);

/* create_surface_layer!(::LambertianSurfaceScalar) (src/CoreRT/Surfaces/lambertian_surface.jl:41-95).
 * Writes ONE shared N×N block into each matrix of `added` (added->mat_stride must be 0)
 * and the per-point source vectors j₀±[N,1,S] from tau_sum[S] (total column). */
int vsm_lambertian_surface_f64(const vsm_quad_f64* q, int S, int m, double albedo,
                               const double* tau_sum, const vsm_added_f64* added, void* stream);
int vsm_lambertian_surface_f32(const vsm_quad_f32* q, int S, int m, float albedo,
                               const float* tau_sum, const vsm_added_f32* added, void* stream);

/* postprocessing_vza! noRS/SFI (src/CoreRT/tools/postprocessing_vza.jl:23-94):
 *   R[v,k,s] += w[v,k] * J₀⁻[row0[v]+k, s],  T[v,k,s] += w[v,k] * J₀⁺[row0[v]+k, s]
 * R,T are [nV,n_stokes,S] column-major (v fastest); row0_h[nV] (host) = n_stokes*(iμ_v-1),
 * w_h[nV*n_stokes] (host, (v,k) at v + nV*k) = weight*{cos mφ,cos mφ,sin mφ,sin mφ}. */
int vsm_postprocess_vza_f64(int N, int n_stokes, int S, int nV, const int* row0_h,
                            const double* w_h, const double* J0_m, const double* J0_p,
                            double* R, double* T, void* stream);
int vsm_postprocess_vza_f32(int N, int n_stokes, int S, int nV, const int* row0_h,
                            const float* w_h, const float* J0_m, const float* J0_p,
                            float* R, float* T, void* stream);

/* ---- linearized (Jacobian) pass: rt_run(model, lin_model, NAer, NGas, NSurf) ----------------------
 * Derivative stacks follow the reference's layout [N,N,S,P] / [N,1,S,P] (parameter axis slowest;
 * src/CoreRT/types_lin.jl:20-97 AddedLayerLin `ap_*` fields / CompositeLayerLin). */
typedef struct vsm_added_lin_f64 {
  double *ap_r_mp, *ap_t_pp, *ap_r_pm, *ap_t_mm;  /* ap_ṙ⁻⁺ ap_ṫ⁺⁺ ap_ṙ⁺⁻ ap_ṫ⁻⁻ */
  double *ap_J0_p, *ap_J0_m;                      /* ap_J̇₀⁺ ap_J̇₀⁻ */
  int P;                                          /* Nparams */
  int reserved;
  long long mat_stride;                           /* N*N, or 0: one block per parameter shared by all S */
} vsm_added_lin_f64;
typedef struct vsm_added_lin_f32 {
  float *ap_r_mp, *ap_t_pp, *ap_r_pm, *ap_t_mm;
  float *ap_J0_p, *ap_J0_m;
  int P;
  int reserved;
  long long mat_stride;
} vsm_added_lin_f32;
typedef struct vsm_composite_lin_f64 {
  double *R_mp, *R_pm, *T_pp, *T_mm, *J0_p, *J0_m;  /* Ṙ⁻⁺ Ṙ⁺⁻ Ṫ⁺⁺ Ṫ⁻⁻ J̇₀⁺ J̇₀⁻ */
  int P;
  int reserved;
} vsm_composite_lin_f64;
typedef struct vsm_composite_lin_f32 {
  float *R_mp, *R_pm, *T_pp, *T_mm, *J0_p, *J0_m;
  int P;
  int reserved;
} vsm_composite_lin_f32;

/* elemental! (lin) = get_elem_rt_fused! + get_elem_rt_SFI_fused! (elemental_lin.jl:77-206,456-712): forward
 * r,t,j AND the chain rule to the first p_layer parameter slots.  dtau_dot[S,p_layer] = τ̇/2^ndoubl,
 * varpi_dot[S,p_layer], tau_sum_dot[S,p_layer]; Zpp_dot/Zmp_dot (nullable = 0): element (i,j,s,p) at
 * i + N*j + s*zd_stride_s + p*zd_stride_p.  All P slots of added_lin are zeroed first. */
int vsm_elemental_lin_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                          const double* tau_sum, const double* F0, const double* Zpp, const double* Zmp,
                          long long z_stride, int p_layer, const double* dtau_dot, const double* varpi_dot,
                          const double* tau_sum_dot, const double* Zpp_dot, const double* Zmp_dot,
                          long long zd_stride_s, long long zd_stride_p, const vsm_added_f64* added,
                          const vsm_added_lin_f64* added_lin, void* stream);
int vsm_elemental_lin_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp,
                          long long z_stride, int p_layer, const float* dtau_dot, const float* varpi_dot,
                          const float* tau_sum_dot, const float* Zpp_dot, const float* Zmp_dot,
                          long long zd_stride_s, long long zd_stride_p, const vsm_added_f32* added,
                          const vsm_added_lin_f32* added_lin, void* stream);
/* vsm_elemental_lin_fold_*: vsm_elemental_lin_* (no Z_dot) for a batch with Fourier moments folded into it -- a small batch walks
 * the layers with its (moment, point) pairs as ONE batch (per-point Z: z_stride = N*N), and the moment enters elemental! only
 * through m == 0 (the weight 1/2 against 1/4, elemental_lin.jl:77-206): the first n_m0 points of the batch are pairs of m = 0, the
 * others of moments m > 0 (`m`: any of them). */
int vsm_elemental_lin_fold_f64(const vsm_quad_f64* q, int S, int m, int n_m0, int ndoubl, const double* dtau, const double* varpi,
                               const double* tau_sum, const double* F0, const double* Zpp, const double* Zmp, long long z_stride,
                               int p_layer, const double* dtau_dot, const double* varpi_dot, const double* tau_sum_dot,
                               const vsm_added_f64* added, const vsm_added_lin_f64* al, void* stream);
int vsm_elemental_lin_fold_f32(const vsm_quad_f32* q, int S, int m, int n_m0, int ndoubl, const float* dtau, const float* varpi,
                               const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp, long long z_stride,
                               int p_layer, const float* dtau_dot, const float* varpi_dot, const float* tau_sum_dot,
                               const vsm_added_f32* added, const vsm_added_lin_f32* al, void* stream);
/* elemental! (lin) for a layer whose phase matrix is a per-point mix of component matrices (aerosol Jacobians): as
 * vsm_elemental_lin_* with Z = sum_{c < ncomp} fz[c, s] Zc[c] (zsel < 0) or Z = Zc[zsel] (zsel >= 0, fz unused) and
 * Z_dot[:, :, s, p] = sum_{c < ncomp_total} zdcoef[c, p, s] Zc[c]; Zc_pp / Zc_mp: [N, N, ncomp_total] blocks of one Fourier
 * moment, fz [ncomp, S], zdcoef [ncomp_total, p_layer, S] (the layer's slices of vsm_layer_optics_lin_*'s outputs).
 * ncomp_total <= 16. */
int vsm_elemental_lin_mix_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                              const double* tau_sum, const double* F0, int ncomp, int ncomp_total, const double* Zc_pp,
                              const double* Zc_mp, int zsel, const double* fz, int p_layer, const double* dtau_dot,
                              const double* varpi_dot, const double* tau_sum_dot, const double* zdcoef,
                              const vsm_added_f64* added, const vsm_added_lin_f64* added_lin, void* stream);
int vsm_elemental_lin_mix_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                              const float* tau_sum, const float* F0, int ncomp, int ncomp_total, const float* Zc_pp,
                              const float* Zc_mp, int zsel, const float* fz, int p_layer, const float* dtau_dot,
                              const float* varpi_dot, const float* tau_sum_dot, const float* zdcoef,
                              const vsm_added_f32* added, const vsm_added_lin_f32* added_lin, void* stream);
/* doubling_allparams! (doubling_lin.jl:216-339): forward + the first n_active parameter slots; the D-symmetry at
 * the end covers all P slots.  dtau_dot_all[S,P] (zero beyond the layer parameters); expk[S] updated in place. */
size_t vsm_doubling_lin_work_elems(int N, int S, int P);
int vsm_doubling_lin_f64(int N, int n_stokes, int S, int ndoubl, double* expk, const double* dtau_dot_all, double mu0,
                         int n_active, const vsm_added_f64* added, const vsm_added_lin_f64* added_lin, double* work,
                         void* stream);
int vsm_doubling_lin_f32(int N, int n_stokes, int S, int ndoubl, float* expk, const float* dtau_dot_all, float mu0,
                         int n_active, const vsm_added_f32* added, const vsm_added_lin_f32* added_lin, float* work,
                         void* stream);
/* interaction! (lin), ScatteringInterface_11 (interaction_lin.jl:217-331); forward composite updated as well. */
size_t vsm_interaction_lin_work_elems(int N, int S, int P);
int vsm_interaction_lin_f64(int iface, int N, int S, const vsm_composite_f64* comp, const vsm_composite_lin_f64* comp_lin,
                            const vsm_added_f64* added, const vsm_added_lin_f64* added_lin, double* work, void* stream);
int vsm_interaction_lin_f32(int iface, int N, int S, const vsm_composite_f32* comp, const vsm_composite_lin_f32* comp_lin,
                            const vsm_added_f32* added, const vsm_added_lin_f32* added_lin, float* work, void* stream);
/* The same for the parameter slots [p_lo, p_hi) only (0 <= p_lo < p_hi <= P); the other slots of comp_lin are left as they are.
 * For callers that KNOW the other slots to be zero in both operands -- the reference's loop `for iparam = 1:Nparams`
 * (interaction_lin.jl:242,291) then computes exact zeros for them: the composite above the surface does not depend on a surface
 * parameter, so rt_run's atmospheric layers need the layer slots only (n_layer_params of parameter_layout.jl) and only the
 * surface interaction needs all.  Identical results; the forward composite is updated as in vsm_interaction_lin. */
int vsm_interaction_lin_range_f64(int iface, int N, int S, const vsm_composite_f64* comp, const vsm_composite_lin_f64* comp_lin,
                                  const vsm_added_f64* added, const vsm_added_lin_f64* added_lin, int p_lo, int p_hi, double* work,
                                  void* stream);
int vsm_interaction_lin_range_f32(int iface, int N, int S, const vsm_composite_f32* comp, const vsm_composite_lin_f32* comp_lin,
                                  const vsm_added_f32* added, const vsm_added_lin_f32* added_lin, int p_lo, int p_hi, float* work,
                                  void* stream);
/* TOA copy of the derivative stacks (rt_kernel_lin.jl:148-166). */
int vsm_copy_added_to_composite_lin_f64(int N, int S, const vsm_added_lin_f64* added_lin,
                                        const vsm_composite_lin_f64* comp_lin, void* stream);
int vsm_copy_added_to_composite_lin_f32(int N, int S, const vsm_added_lin_f32* added_lin,
                                        const vsm_composite_lin_f32* comp_lin, void* stream);
/* create_surface_layer! (lin, LambertianSurfaceScalar; lambertian_surface_lin.jl:48-162).  iparam = slot of the
 * albedo (0-based); shared blocks (mat_stride 0) for the matrices of `added` and `added_lin`. */
int vsm_lambertian_surface_lin_f64(const vsm_quad_f64* q, int S, int m, double albedo, int iparam, const double* tau_sum,
                                   const double* tau_sum_dot, int p_layer, const double* F0, const vsm_added_f64* added,
                                   const vsm_added_lin_f64* added_lin, void* stream);
int vsm_lambertian_surface_lin_f32(const vsm_quad_f32* q, int S, int m, float albedo, int iparam, const float* tau_sum,
                                   const float* tau_sum_dot, int p_layer, const float* F0, const vsm_added_f32* added,
                                   const vsm_added_lin_f32* added_lin, void* stream);
/* postprocessing_vza! (lin) (postprocessing_vza_lin.jl:18-48): Rdot/Tdot [nV,n_stokes,S,P] += w * Jdot0∓. */
int vsm_postprocess_vza_lin_f64(int N, int n_stokes, int S, int nV, int P, const int* row0_h, const double* w_h,
                                const double* Jdot0_m, const double* Jdot0_p, double* Rdot, double* Tdot, void* stream);
int vsm_postprocess_vza_lin_f32(int N, int n_stokes, int S, int nV, int P, const int* row0_h, const float* w_h,
                                const float* Jdot0_m, const float* Jdot0_p, float* Rdot, float* Tdot, void* stream);

/* create_surface_layer!(::LambertianSurfaceLegendre / ::LambertianSurfaceSpline) (src/CoreRT/Surfaces/lambertian_surface.jl:97-213):
 * Lambertian surface whose albedo varies over the band, albedo[S] (device; the host evaluates the Legendre series / the spline
 * on the band grid).  One block per spectral point (added->mat_stride = N*N): r-+[:,:,s] = 2 albedo[s] E11 (x) (mu w), r+- = 0,
 * t++ = t-- = I, j0- = mu0 2 albedo[s] exp(-tau_sum/mu0) on the I rows; as in the reference j0+ = 0 and for m > 0 ALL blocks
 * (including t++ / t--) are zero. */
int vsm_lambertian_surface_spectral_f64(const vsm_quad_f64* q, int S, int m, const double* albedo, const double* tau_sum,
                                        const vsm_added_f64* added, void* stream);
int vsm_lambertian_surface_spectral_f32(const vsm_quad_f32* q, int S, int m, const float* albedo, const float* tau_sum,
                                        const vsm_added_f32* added, void* stream);

/* ---- BRDF surfaces: Cox-Munk ocean + the generic BRDF surface layer ------------------------------------------
 * CoxMunkSurface{FT} (src/CoreRT/types.jl:525-536); n_water is the complex index the reference's call sites use
 * (`_get_n_water(surf, 550)`, coxmunk_surface.jl:434-444: the Segelstein table at 550 nm unless the surface carries one). */
typedef struct vsm_coxmunk_f64 {
  double wind_speed, n_water_re, n_water_im, whitecap_albedo;
  int include_whitecaps, shadowing;
} vsm_coxmunk_f64;
typedef struct vsm_coxmunk_f32 {
  float wind_speed, n_water_re, n_water_im, whitecap_albedo;
  int include_whitecaps, shadowing;
} vsm_coxmunk_f32;
/* reflectance(surf, pol, qp_mu, m) and reflectance_and_deriv (coxmunk_surface.jl:381-460): the Fourier moment m of the
 * BRDF Mueller matrix over the streams of `q` (node k = q->mu[k*n_stokes]) and, if drho_dU != NULL, of its derivative
 * with respect to wind speed: rho[N,N] column-major, rho[(i n+a) + N (j n+b)] = ff/pi sum_phi w M_ab(mu_i, mu_j, phi) az_ab(m phi),
 * ff = 1 (m = 0) | 2.  phi/wphi (device, nphi <= 128) are the azimuth quadrature on [0, pi] the host hands over (the
 * reference: 100-point Gauss-Legendre, CanopyOptics.gauleg). */
int vsm_coxmunk_reflectance_f64(const vsm_coxmunk_f64* surf, const vsm_quad_f64* q, int m, int nphi, const double* phi,
                                const double* wphi, double* rho, double* drho_dU, void* stream);
int vsm_coxmunk_reflectance_f32(const vsm_coxmunk_f32* surf, const vsm_quad_f32* q, int m, int nphi, const float* phi,
                                const float* wphi, float* rho, float* drho_dU, void* stream);
/* create_surface_layer!(brdf::AbstractSurfaceType, ...) (src/CoreRT/Surfaces/rpv_surface.jl:51-97) from a Fourier
 * reflectance block rho[N,N] = reflectance(brdf, pol, qp_mu, m): r-+ = f rho diag(mu w) (f = 2 for m = 0, else 1), r+- = 0,
 * t++ = t-- = I, j0+ = I0_N exp(-tau_sum/mu0), j0- = mu0 (f rho I0_N) exp(-tau_sum/mu0).  Shared block (mat_stride 0). */
int vsm_brdf_surface_f64(const vsm_quad_f64* q, int S, int m, const double* rho, const double* tau_sum,
                         const vsm_added_f64* added, void* stream);
int vsm_brdf_surface_f32(const vsm_quad_f32* q, int S, int m, const float* rho, const float* tau_sum,
                         const vsm_added_f32* added, void* stream);
/* create_surface_layer!(::noRS, ::CoxMunkSurface, added, added_lin, iparam, ...) (coxmunk_surface_lin.jl:27-102): the same
 * with the surface-parameter derivative block drho[N,N] in slot iparam, the beam-attenuation derivatives in the first
 * p_layer slots, F0 [n_stokes,S] instead of I0, and the linearized builder's quirks (j0+ = 0, t-- = 0). */
int vsm_brdf_surface_lin_f64(const vsm_quad_f64* q, int S, int m, const double* rho, const double* drho, int iparam,
                             const double* tau_sum, const double* tau_sum_dot, int p_layer, const double* F0,
                             const vsm_added_f64* added, const vsm_added_lin_f64* added_lin, void* stream);
int vsm_brdf_surface_lin_f32(const vsm_quad_f32* q, int S, int m, const float* rho, const float* drho, int iparam,
                             const float* tau_sum, const float* tau_sum_dot, int p_layer, const float* F0,
                             const vsm_added_f32* added, const vsm_added_lin_f32* added_lin, void* stream);
/* interaction_hdrf! (src/CoreRT/CoreKernel/interaction_hdrf.jl:4-42), called after the surface interaction of each Fourier
 * moment (rt_run.jl:467-476): hdr_J[N,S] = r-+_surf J0+ + j0-_surf; for m == 0 also the hemispheric fluxes bhr_uw / bhr_dw
 * [n_stokes, S] (for m > 0 they are not touched).  The surface layer may be shared (mat_stride 0) or per point. */
int vsm_interaction_hdrf_f64(const vsm_quad_f64* q, int S, int m, const vsm_composite_f64* comp,
                             const vsm_added_f64* added_surface, double* hdr_J, double* bhr_uw, double* bhr_dw, void* stream);
int vsm_interaction_hdrf_f32(const vsm_quad_f32* q, int S, int m, const vsm_composite_f32* comp,
                             const vsm_added_f32* added_surface, float* hdr_J, float* bhr_uw, float* bhr_dw, void* stream);
/* postprocessing_vza_hdrf! (tools/postprocessing_vza.jl:103-115): hdr[v,k,s] += w[v,k] hdr_J[row0[v]+k, s]; row0_h / w_h as in
 * vsm_postprocess_vza_*. */
int vsm_postprocess_vza_hdrf_f64(int N, int n_stokes, int S, int nV, const int* row0_h, const double* w_h, const double* hdr_J,
                                 double* hdr, void* stream);
int vsm_postprocess_vza_hdrf_f32(int N, int n_stokes, int S, int nV, const int* row0_h, const float* w_h, const float* hdr_J,
                                 float* hdr, void* stream);
/* apply_ss_correction! (TMS; coxmunk_surface.jl:481-569, called at rt_run.jl:520-524):
 *   coef[v + nV k] = M_exact[k,1](mu_v, mu0, dphi_v) - sum_{m<=m_max} w_m az_k1(m dphi_v) c_m[k,1](mu_v, mu0)
 *   R_SFI[v,k,s]  += mu0 exp(-tau_total[s]/mu0) coef[v + nV k]
 * mu_v_h / dphi_h: host arrays [nV] (cos of the viewing zenith, relative azimuth in radians); coef: device [nV*n_stokes]
 * (output); R_SFI [nV,n_stokes,S] may be NULL (coefficients only). */
int vsm_coxmunk_ss_correction_f64(const vsm_coxmunk_f64* surf, int n_stokes, int S, int nV, const double* mu_v_h,
                                  const double* dphi_h, double mu0, int m_max, int nphi, const double* phi,
                                  const double* wphi, const double* tau_total, double* coef, double* R_SFI, void* stream);
int vsm_coxmunk_ss_correction_f32(const vsm_coxmunk_f32* surf, int n_stokes, int S, int nV, const float* mu_v_h,
                                  const float* dphi_h, float mu0, int m_max, int nphi, const float* phi, const float* wphi,
                                  const float* tau_total, float* coef, float* R_SFI, void* stream);

/* ---- per-scene layer optics on the device ---------------------------------------------------
 * The reference builds these on the host once per Fourier moment (rt_run.jl:390-394) and ships per-layer arrays inside
 * the layer loop (expandOpticalProperties, compEffectiveLayerProperties.jl:106-117).  Here the raw optical depths are
 * uploaded once per scene and the inputs of the layer kernels are produced in HBM.
 *
 * compute_Z_moments(pol, mu, greek_coefs, m) (src/Scattering/compute_Z_matrices.jl:26-110): Z++ / Z-+ [N,N] over the
 * streams of `q` (node k = q->mu[k*n_stokes]) from the Greek coefficients greek[6*lmax] = alpha | beta | gamma | delta |
 * epsilon | zeta (device, FP64), each of length lmax. */
int vsm_compute_Z_moments_f64(const vsm_quad_f64* q, int m, int lmax, const double* greek, double* Zpp, double* Zmp,
                              void* stream);
int vsm_compute_Z_moments_f32(const vsm_quad_f32* q, int m, int lmax, const double* greek, float* Zpp, float* Zmp,
                              void* stream);
/* constructCoreOpticalProperties + extractEffectiveProps (compEffectiveLayerProperties.jl:11-93) for one band, noRS:
 * Rayleigh (tau_rayl[S,L], varpi_cabannes) + nAer aerosols (tau_aer[nAer,L], ssa[nAer], ftrunc[nAer]; createAero delta-M
 * scaling :67-72) + absorption (tau_abs[S,L]); arrays are the reference's column-major [nSpec, Nz] ((s,l) at s + S*l), all
 * device pointers, FP64 inputs.  mode[nAer,L] (int, (ia,l) at ia + nAer*l) resolves the batch-global branches of the
 * mixing `+` (types.jl:1262-1292): 0 = per-point mix, 1 = no point scatters before this aerosol (take its Z),
 * 2 = the aerosol does not scatter (keep Z).  Outputs: tau, varpi [S,L]; tau_sum [S,L+1] (column l = optical depth above
 * layer l, column L = total); fcomp (nullable) [nAer+1, S, L] per-point weights of the component phase matrices
 * (Rayleigh, aerosol 1, ...) as consumed by vsm_layer_forward_mix_*; max_tau_varpi[L] = maximum(tau .* varpi) per layer
 * (rt_kernel.jl:197,282 -- the host turns it into ndoubl and the scattering-interface tags). */
int vsm_layer_optics_f64(int S, int L, int nAer, const double* tau_rayl, const double* tau_abs, double varpi_cabannes,
                         const double* tau_aer, const double* ssa, const double* ftrunc, const int* mode, double* tau,
                         double* varpi, double* tau_sum, double* fcomp, double* max_tau_varpi, void* stream);
int vsm_layer_optics_f32(int S, int L, int nAer, const double* tau_rayl, const double* tau_abs, double varpi_cabannes,
                         const double* tau_aer, const double* ssa, const double* ftrunc, const int* mode, float* tau,
                         float* varpi, float* tau_sum, float* fcomp, float* max_tau_varpi, void* stream);
/* dtau[S,L] = tau ./ 2^ndoubl[l] (get_dtau_ndoubl, rt_kernel.jl:266-287); ndoubl: device int[L]. */
int vsm_layer_dtau_f64(int S, int L, const int* ndoubl, const double* tau, double* dtau, void* stream);
int vsm_layer_dtau_f32(int S, int L, const int* ndoubl, const float* tau, float* dtau, void* stream);
/* constructCoreOpticalProperties with lin_model (compEffectiveLayerProperties_lin.jl:43-197; createAero with derivatives
 * :330-395; quotient rule of the pairwise `+`, types_lin.jl:196-380) for the rank's block [lo, lo + S) of a band of S_full
 * points: the derivatives of the layer optics with respect to the pl = 7 nAer + nGas layer parameters (slot order of
 * parameter_layout.jl:28-56: per aerosol tau_ref, n_r, n_i, r_m, sigma_r, p0, sigma_p; then the gases).  Inputs (device, FP64,
 * indexed on the FULL axis): tau_rayl, tau_abs [S_full, L]; tau_aer [nAer, L]; ssa, ftrunc [nAer]; tau_abs_dot [S_full, L, nGas]
 * (lin_model.tau_abs_dot); tau_aer_dot [7, nAer, L] (lin_model.tau_aer_dot); ssa_dot, ftrunc_dot [4, nAer]
 * (lin_aerosol_optics: d/d(n_r, n_i, r_m, sigma_r)); ndoubl: device int[L].  Outputs (block-local, column-major):
 * dtau_dot_all [S, P, L] = tau_dot / 2^ndoubl (columns >= pl zero: the surface slots of doubling_allparams!),
 * varpi_dot [S, pl, L], tau_sum_dot [S, pl, L+1] (layer l: derivative of the optical depth above it; L: total), and for
 * nAer > 0 the per-point weights fz [nAer+1, S, L] of the component phase matrices in Z and the coefficients
 * zdcoef [CT, pl, S, L] of Z_dot over the CT = (nAer+1) + 4 nAer component blocks
 * [Z_Rayleigh, Z_aer1.., dZ_aer1/d(n_r, n_i, r_m, sigma_r), dZ_aer2/d.., ..] -- Z_dot itself is never materialised
 * (vsm_elemental_lin_mix_* forms it where it is consumed). */
int vsm_layer_optics_lin_f64(int S_full, int lo, int S, int L, int nAer, int nGas, int P, const double* tau_rayl,
                             const double* tau_abs, double varpi_cabannes, const double* tau_aer, const double* ssa,
                             const double* ftrunc, const double* tau_abs_dot, const double* tau_aer_dot,
                             const double* ssa_dot, const double* ftrunc_dot, const int* ndoubl, double* dtau_dot_all,
                             double* varpi_dot, double* tau_sum_dot, double* fz, double* zdcoef, void* stream);
int vsm_layer_optics_lin_f32(int S_full, int lo, int S, int L, int nAer, int nGas, int P, const double* tau_rayl,
                             const double* tau_abs, double varpi_cabannes, const double* tau_aer, const double* ssa,
                             const double* ftrunc, const double* tau_abs_dot, const double* tau_aer_dot,
                             const double* ssa_dot, const double* ftrunc_dot, const int* ndoubl, float* dtau_dot_all,
                             float* varpi_dot, float* tau_sum_dot, float* fz, float* zdcoef, void* stream);
/* expk[S] = exp(-dtau / mu0) (init_layer, rt_kernel.jl:339-349); doubling! squares it in place. */
int vsm_layer_expk_f64(int S, const double* dtau, double mu0, double* expk, void* stream);
int vsm_layer_expk_f32(int S, const float* dtau, float mu0, float* expk, void* stream);

/* rt_kernel!(::noRS) for ONE scattering layer (src/CoreRT/CoreKernel/rt_kernel.jl:175-250): elemental! + doubling!
 * followed by copy_added_to_composite! (toa != 0, i.e. iz == 1; rt_helpers.jl:188-200) or
 * interaction!(::ScatteringInterface_11) (interaction.jl:207-266).  Arguments as vsm_elemental_doubling_*.
 * FP64 with 32 < N <= 64 (and FP32 with 64 < N <= 96) runs as a pre-pass + ONE launch whose added layer never leaves the chip
 * (`added_scratch` is then not touched and may be NULL); other shapes run elemental / doubling / interaction as separate
 * launches through `added_scratch` -- FP64 with 64 < N <= 128: the whole doubling loop and the interaction of a point each in
 * one persistent workgroup (vsm_strip128.hip).
 * Stream: see Conventions; library scratch (the pre-pass images of the layer; parked strips for 64 < N <= 128). */
int vsm_layer_forward_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                          const double* tau_sum, const double* F0, const double* Zpp, const double* Zmp, long long z_stride,
                          int toa, const vsm_composite_f64* comp, const vsm_added_f64* added_scratch, void* stream);
int vsm_layer_forward_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp, long long z_stride,
                          int toa, const vsm_composite_f32* comp, const vsm_added_f32* added_scratch, void* stream);

/* vsm_layer_forward(_mix)_* for nm Fourier moments of ONE layer (the moments of rt_run's outer loop, rt_run.jl:383, are
 * independent until post-processing): m[nm] (host), Zpp[nm] / Zmp[nm] (host arrays of device pointers: the moment's Z block or
 * component stack), comps[nm] (host array: one CompositeLayer per moment); dtau, varpi, tau_sum, F0, fcomp, ndoubl are the
 * layer's and shared.  FP64 with 32 < N <= 60 runs the moments in ONE launch (three times the workgroups per launch: a third of
 * the launch tails); other shapes run vsm_layer_forward(_mix) moment by moment through `added_scratch` / `z_scratch`.
 * Stream: see Conventions; library scratch as vsm_layer_forward_*. */
int vsm_layer_forward_multi_f64(const vsm_quad_f64* q, int S, int nm, const int* m, int ndoubl, const double* dtau,
                                const double* varpi, const double* tau_sum, const double* F0, int ncomp,
                                const double* const* Zpp, const double* const* Zmp, long long z_stride, const double* fcomp,
                                double* z_scratch, int toa, const vsm_composite_f64* comps, const vsm_added_f64* added_scratch,
                                void* stream);
int vsm_layer_forward_multi_f32(const vsm_quad_f32* q, int S, int nm, const int* m, int ndoubl, const float* dtau,
                                const float* varpi, const float* tau_sum, const float* F0, int ncomp, const float* const* Zpp,
                                const float* const* Zmp, long long z_stride, const float* fcomp, float* z_scratch, int toa,
                                const vsm_composite_f32* comps, const vsm_added_f32* added_scratch, void* stream);

/* The `:thermal` per-source slot of one scattering layer (rt_kernel.jl:205-232: contribute!(::PreparedThermalEmission)
 * between elemental! and doubling!, Sources/thermal_emission.jl:241-301; the slot's own expk = 1, doubling.jl:62-81; the
 * per-source recurrences of interaction.jl) in the launch of vsm_layer_forward(_mix)_*: m = 0, the solar source replaced by
 * j0+- = 2 pi (1 - varpi) B (1 - exp(-dtau/mu_i)) on the I rows, thermal_B[S] = Planck radiance of the layer per point.
 * ncomp = 0: Zpp / Zmp [N,N,S] with z_stride (0 = shared block); ncomp >= 1: component stacks [N,N,ncomp] + fcomp[ncomp,S].
 * `comp` is the composite of the thermal slot.  Fused for FP64 with 32 < N <= 60 and FP32 with 64 < N <= 96, ncomp <= 4
 * (vsm_layer_thermal_fused(N, is_f64) != 0); any other shape returns VSM_ERR_UNSUPPORTED (callers then run
 * vsm_elemental + vsm_thermal_source + vsm_doubling + vsm_interaction). */
int vsm_layer_thermal_fused(int N, int is_f64);
int vsm_layer_forward_thermal_f64(const vsm_quad_f64* q, int S, int ndoubl, const double* dtau, const double* varpi,
                                  const double* thermal_B, int ncomp, const double* Zpp, const double* Zmp, long long z_stride,
                                  const double* fcomp, int toa, const vsm_composite_f64* comp, void* stream);
int vsm_layer_forward_thermal_f32(const vsm_quad_f32* q, int S, int ndoubl, const float* dtau, const float* varpi,
                                  const float* thermal_B, int ncomp, const float* Zpp, const float* Zmp, long long z_stride,
                                  const float* fcomp, int toa, const vsm_composite_f32* comp, void* stream);

/* Layer optics with several scatterers: the reference mixes Z per spectral point on the host,
 * Z = sum_k (tau_k varpi_k Z_k) / sum_k (tau_k varpi_k)  (`+` of CoreScatteringOpticalProperties, src/CoreRT/types.jl:1262-1292;
 * compEffectiveLayerProperties.jl:43-54) and ships [N,N,nSpec] arrays to the device.  Here the ncomp component
 * matrices Zpp_comp / Zmp_comp [N,N,ncomp] (one stack per Fourier moment) and the per-point weights fcomp [ncomp,S]
 * (fcomp[k + ncomp*s]) are handed over instead.
 * vsm_layer_forward_mix_*: as vsm_layer_forward_*, mixing Z where the elemental step consumes it (fused strip kernel,
 *   ncomp <= 4); other shapes materialise Z into z_scratch (2*N*N*S elements; NULL = library scratch) first.
 * vsm_mix_Z_*: the materialising kernel on its own: Zpp/Zmp [N,N,S]. */
int vsm_layer_forward_mix_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                              const double* tau_sum, const double* F0, int ncomp, const double* Zpp_comp,
                              const double* Zmp_comp, const double* fcomp, double* z_scratch, int toa,
                              const vsm_composite_f64* comp, const vsm_added_f64* added_scratch, void* stream);
int vsm_layer_forward_mix_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                              const float* tau_sum, const float* F0, int ncomp, const float* Zpp_comp, const float* Zmp_comp,
                              const float* fcomp, float* z_scratch, int toa, const vsm_composite_f32* comp,
                              const vsm_added_f32* added_scratch, void* stream);
int vsm_mix_Z_f64(int N, int S, int ncomp, const double* Zpp_comp, const double* Zmp_comp, const double* fcomp, double* Zpp,
                  double* Zmp, void* stream);
int vsm_mix_Z_f32(int N, int S, int ncomp, const float* Zpp_comp, const float* Zmp_comp, const float* fcomp, float* Zpp,
                  float* Zmp, void* stream);
/* vsm_mix_Z_moments_*: the same for nm <= 24 Fourier moments of one layer in one launch, the moments folded into the batch axis
 * (the layer walk of a small batch takes its moments as ONE batch: rt_kernel!'s arguments of layer iz for (moment, point) pairs):
 * Zpp_comp_h / Zmp_comp_h[nm] = HOST arrays of device pointers to the moments' component stacks [N,N,ncomp]; block im * S + s of
 * Zpp / Zmp [N,N,nm*S] = sum_k fcomp[k + ncomp*s] Z_k(moment im).  ncomp = 0: a layer with one scatterer -- block `single` of
 * every stack, copied (fcomp unused). */
int vsm_mix_Z_moments_f64(int N, int S, int ncomp, int nm, const double* const* Zpp_comp_h, const double* const* Zmp_comp_h, int single,
                          const double* fcomp, double* Zpp, double* Zmp, void* stream);
int vsm_mix_Z_moments_f32(int N, int S, int ncomp, int nm, const float* const* Zpp_comp_h, const float* const* Zmp_comp_h, int single,
                          const float* fcomp, float* Zpp, float* Zmp, void* stream);

/* ---- a run with the CompositeLayer in kernel-native layout ---------------------
 * The layer loop of rt_run (src/CoreRT/rt_run.jl:383-453: `for iz = 1:Nz ... rt_kernel!(RS_type, pol_type, SFI, added_layer,
 * composite_layer, ...)`) reads and writes the CompositeLayer (make_composite_layer, tools/rt_helper_functions.jl:259-270;
 * allocated once per run, rt_run.jl:326-335) in every layer step, and nothing else touches it until the surface interaction
 * (rt_run.jl:455-470).  A vsm_run is that CompositeLayer for nm Fourier moments of S points, kept in the layer kernels' own
 * strip layout for the whole loop:
 *   vsm_run_create   = make_composite_layer (the storage is the caller's `workspace`: vsm_run_workspace_bytes bytes, 16-byte
 *                      aligned device memory that must outlive the run; q->mu / q->wt must outlive it too)
 *   vsm_run_layer    = rt_kernel!(::noRS) for ONE scattering layer and all nm moments (rt_kernel.jl:175-250: elemental! +
 *                      doubling! + interaction!(::ScatteringInterface_11), or copy_added_to_composite! when toa != 0); arguments as
 *                      vsm_layer_forward_multi_f64 (Zpp[nm] / Zmp[nm]: host arrays of device pointers, ncomp <= 4)
 *   vsm_run_export   = the composite as the reference's arrays (comps[nm]: one CompositeLayer per moment), for the surface
 *                      step and postprocessing_vza!; vsm_run_import is the inverse (a run that continues from existing arrays)
 *   vsm_run_destroy  frees the host-side handle only.
 * Stokes blocks.  `coupling[nm]` (NULL or -1 per moment = dense) tells per moment which Stokes components the phase matrices
 * couple: bit 4 a + b set = some element Z[i, j] with i % n_stokes == a, j % n_stokes == b is non-zero.  Components that do not
 * couple run as independent sub-problems (for m = 0 every phase matrix has exactly zero (I,Q) x (U,V) blocks,
 * src/Scattering/compute_Z_matrices.jl:26-110: N = 60, n_stokes = 3 runs as 40 x 40 + 20 x 20) -- products with exact zeros are
 * not formed, results are those of the dense run.  The mask MUST cover every Z later handed to vsm_run_layer (every call checks
 * the layer's matrices against the masks on the device and raises VSM_DEVSTAT_MASK, vsm_device_status, on a violation):
 * vsm_stokes_coupling_f64 computes one mask per matrix of a stack of `nblocks` matrices on the device (mask_d[nblocks]: DEVICE
 * ints, written asynchronously on `stream`); OR the masks of all scatterers of a moment.
 * `layer_coupling_h[nm]` of vsm_run_layer (NULL = the run's) is the same mask for the phase matrices of THIS layer (the OR over the
 * scatterers present in it): a block the layer's phase matrices leave exactly zero (U at m = 0 in a Rayleigh layer; any block of
 * a Rayleigh-only layer at m >= 3, where the Rayleigh phase matrix vanishes) has r = 0, j = 0, t = diag(exp(-tau / mu)) and its
 * interaction is a scaling of the composite's rows and columns -- one elementwise pass instead of the products.
 * Every block of coupled components must fit the native
 * kernels: (N / n_stokes) * (components in the block) <= 64 (vsm_run_supported != 0), else VSM_ERR_UNSUPPORTED.
 * _f32: the same for a Float32 model in the model's own float type, as the reference runs it (doubling.jl:38-131 and
 * interaction.jl:207-266 compute in FT): the caller's arrays, the native records and the arithmetic (v_mfma_f32_16x16x4_f32) are
 * single precision, blocks of up to 128 rows (vsm_run_supported_f32; workspace: vsm_run_workspace_bytes_f32 -- half of
 * vsm_run_workspace_bytes, which stays sufficient).  A run is used with the entry points of its own type.
 * Stream: see Conventions; library scratch (the pre-pass images of the layer). */
typedef struct vsm_run vsm_run;
int vsm_run_supported(int N, int n_stokes, int coupling);
size_t vsm_run_workspace_bytes(int N, int n_stokes, int S, int nm, const int* coupling_h);
int vsm_run_supported_f32(int N, int n_stokes, int coupling);
size_t vsm_run_workspace_bytes_f32(int N, int n_stokes, int S, int nm, const int* coupling_h);
int vsm_run_create_f64(const vsm_quad_f64* q, int S, int nm, const int* m_h, const int* coupling_h, void* workspace,
                       size_t workspace_bytes, vsm_run** run);
int vsm_run_create_f32(const vsm_quad_f32* q, int S, int nm, const int* m_h, const int* coupling_h, void* workspace,
                       size_t workspace_bytes, vsm_run** run);
int vsm_run_layer_f64(vsm_run* run, int ndoubl, const double* dtau, const double* varpi, const double* tau_sum,
                      const double* F0, int ncomp, const double* const* Zpp, const double* const* Zmp, long long z_stride,
                      const double* fcomp, int toa, const int* layer_coupling_h, void* stream);
int vsm_run_layer_f32(vsm_run* run, int ndoubl, const float* dtau, const float* varpi, const float* tau_sum,
                      const float* F0, int ncomp, const float* const* Zpp, const float* const* Zmp, long long z_stride,
                      const float* fcomp, int toa, const int* layer_coupling_h, void* stream);
int vsm_run_export_f64(vsm_run* run, const vsm_composite_f64* comps, void* stream);
int vsm_run_export_f32(vsm_run* run, const vsm_composite_f32* comps, void* stream);
int vsm_run_import_f64(vsm_run* run, const vsm_composite_f64* comps, void* stream);
int vsm_run_import_f32(vsm_run* run, const vsm_composite_f32* comps, void* stream);
int vsm_run_destroy(vsm_run* run);
int vsm_stokes_coupling_f64(int N, int n_stokes, int nblocks, const double* Zpp, const double* Zmp, int* mask_d, void* stream);
int vsm_stokes_coupling_f32(int N, int n_stokes, int nblocks, const float* Zpp, const float* Zmp, int* mask_d, void* stream);

/* ---- rotational Raman scattering (RRS), operator level ---------------------
 * Inelastic layer state (src/CoreRT/types.jl:278-335 AddedLayerRS / CompositeLayerRS): 4-D arrays
 * [N,N,S,K] / [N,1,S,K], K = number of Raman offsets (length of RS_type.i_lambda1lambda0); element
 * (i,j,n1,dn) at i + N*j + N*N*n1 + N*N*S*dn.  Block (n1,dn) couples recipient point n1 with donor point
 * n0 = n1 + shift[dn] (src/Inelastic/inelastic_helper.jl:19-26); pairs with n0 outside [0,S) stay zero. */
typedef struct vsm_added_rs_f64 {
  double *ier_mp, *iet_pp, *ier_pm, *iet_mm;  /* ier-+ iet++ ier+- iet--  [N,N,S,K] */
  double *ieJ0_p, *ieJ0_m;                    /* ieJ0+ ieJ0-              [N,1,S,K] */
  int K;
  int reserved;
} vsm_added_rs_f64;
typedef struct vsm_added_rs_f32 {
  float *ier_mp, *iet_pp, *ier_pm, *iet_mm;
  float *ieJ0_p, *ieJ0_m;
  int K;
  int reserved;
} vsm_added_rs_f32;
typedef struct vsm_composite_rs_f64 {
  double *ieR_mp, *ieR_pm, *ieT_pp, *ieT_mm;
  double *ieJ0_p, *ieJ0_m;
  int K;
  int reserved;
} vsm_composite_rs_f64;
typedef struct vsm_composite_rs_f32 {
  float *ieR_mp, *ieR_pm, *ieT_pp, *ieT_mm;
  float *ieJ0_p, *ieJ0_m;
  int K;
  int reserved;
} vsm_composite_rs_f32;
/* The fields of RRS{FT} (src/Inelastic/types.jl) read by the CoreRT kernels; all DEVICE pointers. */
typedef struct vsm_rrs_f64 {
  const int* shift;        /* i_lambda1lambda0[K] */
  const double* varpi_ie;  /* varpi_lambda1lambda0[K] */
  const double* fscatt;    /* fscattRayl[S] of the current layer (rt_run.jl:221-223) */
  const double* Zpp;       /* Z++_lambda1lambda0 [N,N] of the current Fourier moment (inelastic_helper.jl:917-924) */
  const double* Zmp;       /* Z-+_lambda1lambda0 [N,N] */
} vsm_rrs_f64;
typedef struct vsm_rrs_f32 {
  const int* shift;
  const float* varpi_ie;
  const float* fscatt;
  const float* Zpp;
  const float* Zmp;
} vsm_rrs_f32;
/* elemental_inelastic!(::RRS) (CoreKernel/elemental_inelastic.jl:23-105; kernels get_elem_rt_RRS! :117-206,
 * get_elem_rt_SFI_RRS! :479-610, apply_D_elemental_RRS! :619-637).  Fills the six fields of added_rs
 * (ier+-/iet-- only when ndoubl < 1, like the reference). */
int vsm_elemental_inelastic_rrs_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau,
                                    const double* tau_sum, const double* F0, const vsm_rrs_f64* rs,
                                    const vsm_added_rs_f64* added_rs, void* stream);
int vsm_elemental_inelastic_rrs_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau,
                                    const float* tau_sum, const float* F0, const vsm_rrs_f32* rs,
                                    const vsm_added_rs_f32* added_rs, void* stream);
/* doubling_inelastic! = doubling_helper!(::RRS) (CoreKernel/doubling_inelastic.jl:13-164): doubles the elastic AND
 * the inelastic fields together (the elastic recurrences are interleaved with the inelastic ones), then applies
 * the D-matrix kernels (:336-356, :408-416).  expk[S] is squared in place ndoubl times.  `added` must be a full
 * per-point AddedLayer.  work: vsm_doubling_inelastic_work_elems(N,S,K) elements. */
size_t vsm_doubling_inelastic_work_elems(int N, int S, int K);
int vsm_doubling_inelastic_rrs_f64(int N, int n_stokes, int S, int ndoubl, double* expk, const int* shift,
                                   const vsm_added_f64* added, const vsm_added_rs_f64* added_rs, double* work, void* stream);
int vsm_doubling_inelastic_rrs_f32(int N, int n_stokes, int S, int ndoubl, float* expk, const int* shift,
                                   const vsm_added_f32* added, const vsm_added_rs_f32* added_rs, float* work, void* stream);
/* interaction!(RS_type::RRS, scattering_interface, ...) (CoreKernel/interaction_inelastic.jl:523-539): updates the
 * inelastic composite from pre-update elastic/inelastic values, then the elastic composite.  iface = VSM_IFACE_11:
 * interaction_helper!(::RRS, ::ScatteringInterface_11) (:319-521, fused line kernels for N <= 30); VSM_IFACE_00 / _01 / _10:
 * the statements of :74-101, :103-154, :215-262 over the in-band (recipient, line) pairs, operator level (the RRS rt_kernel!
 * hard-wires scatter = true, rt_kernel.jl:365, but dispatches the interaction on the tag, :385).  `added` may be a
 * shared-block surface layer (mat_stride 0) with an all-zero added_rs.
 * work: vsm_interaction_inelastic_work_elems(N,S,K) elements. */
size_t vsm_interaction_inelastic_work_elems(int N, int S, int K);
int vsm_interaction_inelastic_rrs_f64(int iface, int N, int S, const int* shift, const vsm_composite_f64* comp,
                                      const vsm_composite_rs_f64* comp_rs, const vsm_added_f64* added,
                                      const vsm_added_rs_f64* added_rs, double* work, void* stream);
int vsm_interaction_inelastic_rrs_f32(int iface, int N, int S, const int* shift, const vsm_composite_f32* comp,
                                      const vsm_composite_rs_f32* comp_rs, const vsm_added_f32* added,
                                      const vsm_added_rs_f32* added_rs, float* work, void* stream);
/* copy_added_to_composite_ie! (rt_helpers.jl:222-228), inelastic fields only (pair with vsm_copy_added_to_composite). */
int vsm_copy_added_to_composite_ie_f64(int N, int S, const vsm_added_rs_f64* added_rs, const vsm_composite_rs_f64* comp_rs,
                                       void* stream);
int vsm_copy_added_to_composite_ie_f32(int N, int S, const vsm_added_rs_f32* added_rs, const vsm_composite_rs_f32* comp_rs,
                                       void* stream);
/* postprocessing_vza!(::RRS) inelastic accumulation (tools/postprocessing_vza.jl:139-142):
 * ieR/ieT [nV,n_stokes,S] += w * sum_dn ieJ0-/+[rows(vza), 1, s, dn].  row0_h/w_h as in vsm_postprocess_vza. */
int vsm_postprocess_vza_ie_f64(int N, int n_stokes, int S, int K, int nV, const int* row0_h, const double* w_h,
                               const double* ieJ0_m, const double* ieJ0_p, double* ieR, double* ieT, void* stream);
int vsm_postprocess_vza_ie_f32(int N, int n_stokes, int S, int K, int nV, const int* row0_h, const float* w_h,
                               const float* ieJ0_m, const float* ieJ0_p, float* ieR, float* ieT, void* stream);

/* ---- diagnostics used by the parity tests -------------------------------- */
/* Runs the LDS-resident MFMA tile product used inside the fused kernels on plain
 * [N,N,S] inputs: C = A*B.  (Checks fragment layouts / swizzle independently.) */
int vsm_test_lds_mm_f64(int N, int S, const double* A, const double* B, double* C, void* stream);
int vsm_test_lds_mm_f32(int N, int S, const float* A, const float* B, float* C, void* stream);
/* Inverse via the fused kernels' LDS path (series fast path + pivoted Gauss-Jordan fallback).
 * mode: 0 = automatic, 1 = force Gauss-Jordan, 2 = force Neumann series. path_out (nullable,
 * int[S]) receives the path taken per point (1 = GJ, 2.. = series order + 1). */
int vsm_test_lds_inv_f64(int N, int S, const double* A, double* X, int mode, int* path_out, void* stream);
int vsm_test_lds_inv_f32(int N, int S, const float* A, float* X, int mode, int* path_out, void* stream);

/* Fills the whole LDS of every CU with NaN bit patterns (the GPU tests call it before every test: no kernel may depend
 * on what a previous workgroup left in LDS). */
int vsm_test_poison_lds(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VSMARTMOM_HIP_H */
