// Internal helpers shared by the HIP translation units of libvsmartmom_hip.so.
// gfx950 (MI355X / CDNA4) only: wave64, v_mfma_{f64,f32}_16x16x4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "vsmartmom_hip.h"

namespace vsm {

// ---- error plumbing ---------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);           // failed HIP API call: records the message, returns VSM_ERR_HIP
int hip_launch_fail(hipError_t e, const char* kernel);  // failed launch: VSM_ERR_UNSUPPORTED for an invalid configuration only
#define VSM_HIP(call)                                            \
  do {                                                           \
    hipError_t _e = (call);                                      \
    if (_e != hipSuccess) return ::vsm::hip_fail(_e, #call);     \
  } while (0)
#define VSM_LAUNCH_CHECK(name)                                   \
  do {                                                           \
    hipError_t _e = hipGetLastError();                           \
    if (_e != hipSuccess) return ::vsm::hip_launch_fail(_e, name);      \
  } while (0)
#define VSM_REQUIRE(cond, ...)                                   \
  do {                                                           \
    if (!(cond)) {                                               \
      ::vsm::set_error(__VA_ARGS__);                             \
      return VSM_ERR_INVALID_ARG;                                \
    }                                                            \
  } while (0)

// ---- A/B switches ---------------------------------------------------------------
// Environment switches that force a shape off its fused kernel family onto the next one (VSM_NO_STRIP, VSM_NO_RAMAN_WAVE, ...)
// exist only in a diagnostic build (make EXTRA=-DVSM_AB_SWITCHES): the shipped library has ONE kernel per shape, the one the
// parity tests exercise.  Every family is still covered by the tests through the shapes it owns.
#ifdef VSM_AB_SWITCHES
inline bool ab_switch(const char* name) { return getenv(name) != nullptr; }
#else
constexpr bool ab_switch(const char*) { return false; }
#endif

// ---- MFMA traits --------------------------------------------------------------
// One MFMA = a 16x16 output tile, K = 4.  Operands: lane l holds A[i = l&15][k = l>>4]
// and B[k = l>>4][j = l&15] (one element each).  The accumulator holds 4 elements per
// lane at column (l&15); the row map differs between f64 and f32 (CDNA4 ISA, and
// /opt/skills/guides/cdna_hip_programming.md "Fragment layout").
typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct mfma;
template <>
struct mfma<double> {
  using acc_t = d4_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct mfma<float> {
  using acc_t = f4_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <typename T>
__device__ __forceinline__ typename mfma<T>::acc_t acc_zero() {
  typename mfma<T>::acc_t z = {0, 0, 0, 0};
  return z;
}

// D_i of the reference: +1 for Stokes I,Q rows, -1 for U,V rows (mod1(i, n) > 2).
__device__ __forceinline__ bool is_uv_row(int i, int n_stokes) { return (i % n_stokes) >= 2; }

template <typename T>
struct num;
template <>
struct num<double> {
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
};
template <>
struct num<float> {
  static __device__ __forceinline__ float eps() { return 1.1920929e-07f; }
};

// exp(-a) - exp(-b) without cancellation (src/CoreRT/CoreKernel/rt_helpers.jl:32-40).
template <typename T>
__device__ __forceinline__ T expdiff_neg(T a, T b) {
  if (a == b) return T(0);
  if (a < b) return exp(-a) * (-expm1(-(b - a)));
  return -exp(-b) * (-expm1(-(a - b)));
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace vsm
