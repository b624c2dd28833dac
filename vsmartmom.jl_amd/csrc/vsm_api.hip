// extern "C" entry points of libvsmartmom_hip.so (see include/vsmartmom_hip.h).
#include <map>
#include <mutex>
#include <set>
#include <utility>
#include <vector>
#include <stdlib.h>
#include <string.h>

#include "build_id.h"
#include "vsm_internal.h"

namespace vsm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
  // a failed HIP API call (hipMalloc, hipMemsetAsync, hipFuncSetAttribute, hipMemcpy ...) is always VSM_ERR_HIP: callers that
  // fall through to another kernel family on VSM_ERR_UNSUPPORTED must never swallow it
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return VSM_ERR_HIP;
}
int hip_launch_fail(hipError_t e, const char* kernel) {
  // only a launch that the hardware limits refuse (grid dimension y/z > 65535, block size, dynamic LDS) is "unsupported shape";
  // the entry points whose kernels put the spectral axis or the Raman lines on gridDim.y/z check S, K <= 65535 themselves
  if (e == hipErrorInvalidConfiguration) {
    set_error("HIP error %d (%s) launching %s -- invalid launch configuration (a grid dimension y/z above 65535?)", (int)e,
              hipGetErrorString(e), kernel);
    return VSM_ERR_UNSUPPORTED;
  }
  return hip_fail(e, kernel);
}

// Library-owned scratch (vsm_internal.h).  One buffer per (device, stream, slot): a host that drives several streams (the
// Fourier-moment lanes of the linearized run) or several devices from one process never shares a buffer between them.
// A buffer is outgrown -> it goes on a retire list behind an event recorded on ITS stream and is freed by a later call once
// that event has completed: nothing is freed under running work and no call synchronises the device.
namespace {
struct scratch_key {
  int dev;
  hipStream_t st;
  int slot;
  bool operator<(const scratch_key& o) const {
    if (dev != o.dev) return dev < o.dev;
    if (st != o.st) return st < o.st;
    return slot < o.slot;
  }
};
struct scratch_buf {
  void* ptr = nullptr;
  size_t sz = 0;
};
struct retired_buf {
  int dev;
  void* ptr;
  hipEvent_t ev;
};
std::mutex g_scratch_mu;
std::map<scratch_key, scratch_buf> g_scratch;
std::vector<retired_buf> g_retired;

// frees the retired buffers of device `dev` whose event has completed (caller holds the mutex, `dev` is current)
void reap_retired(int dev, bool wait) {
  for (size_t i = 0; i < g_retired.size();) {
    retired_buf& r = g_retired[i];
    if (r.dev == dev && (wait ? hipEventSynchronize(r.ev) == hipSuccess : hipEventQuery(r.ev) == hipSuccess)) {
      (void)hipEventDestroy(r.ev);
      (void)hipFree(r.ptr);
      g_retired[i] = g_retired.back();
      g_retired.pop_back();
    } else {
      ++i;
    }
  }
}
}  // namespace

void* scratch(size_t bytes, int slot, hipStream_t st) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    hip_fail(e, "hipGetDevice(scratch)");
    return nullptr;
  }
  if (st == hipStreamPerThread) {   // one handle value, a different stream in every host thread: it cannot key a buffer
    set_error("hipStreamPerThread is not accepted by the entry points that use library scratch: pass the thread's own stream handle");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  scratch_buf& b = g_scratch[scratch_key{dev, st, slot}];
  if (bytes <= b.sz) return b.ptr;
  if (!g_retired.empty()) reap_retired(dev, false);
  // geometric growth bounds the retired bytes by the live ones
  size_t want = bytes;
  if (b.sz && want < b.sz + b.sz / 2) want = b.sz + b.sz / 2;
  void* p = nullptr;
  e = hipMalloc(&p, want);
  if (e != hipSuccess && want > bytes) {
    (void)hipGetLastError();
    want = bytes;
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    hip_fail(e, "hipMalloc(scratch)");
    return nullptr;
  }
  if (b.ptr) {   // work already queued on `st` may still use the old buffer: retire it behind an event on that stream
    hipEvent_t ev = nullptr;
    e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, st);
    if (e != hipSuccess) {
      if (ev) (void)hipEventDestroy(ev);
      (void)hipFree(p);
      hip_fail(e, "hipEventRecord(scratch retire)");
      return nullptr;
    }
    g_retired.push_back(retired_buf{dev, b.ptr, ev});
  }
  b.ptr = p;
  b.sz = want;
  return p;
}

int release_scratch() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice(release_scratch)");
  if ((e = hipDeviceSynchronize()) != hipSuccess) return hip_fail(e, "hipDeviceSynchronize(release_scratch)");
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  reap_retired(dev, true);
  for (auto it = g_scratch.begin(); it != g_scratch.end();) {
    if (it->first.dev == dev) {
      if (it->second.ptr) (void)hipFree(it->second.ptr);
      it = g_scratch.erase(it);
    } else {
      ++it;
    }
  }
  return VSM_OK;
}

namespace {
std::mutex g_attr_mu;
std::set<std::pair<int, const void*>> g_attr_done;
std::mutex g_cu_mu;
std::map<int, int> g_cu;
}  // namespace

namespace {
std::mutex g_status_mu;
std::map<int, int*> g_status;
}  // namespace
int* device_status() {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    hip_fail(e, "hipGetDevice(device_status)");
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_status_mu);
  auto it = g_status.find(dev);
  if (it != g_status.end()) return it->second;
  int* p = nullptr;
  e = hipMalloc(reinterpret_cast<void**>(&p), 4 * sizeof(int));
  // zeroed once, then the device is synchronised: the null-stream memset is not ordered against non-blocking streams (torch's,
  // AMDGPU.jl's), so the first kernel on such a stream could otherwise have its flag wiped
  if (e == hipSuccess) e = hipMemset(p, 0, 4 * sizeof(int));
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    hip_fail(e, "hipMalloc(device_status)");
    return nullptr;
  }
  g_status[dev] = p;
  return p;
}

int ensure_dyn_lds(const void* kern, size_t bytes, const char* what) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice(ensure_dyn_lds)");
  std::lock_guard<std::mutex> lk(g_attr_mu);
  const std::pair<int, const void*> key(dev, kern);
  if (g_attr_done.count(key)) return VSM_OK;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return hip_fail(e, what);
  g_attr_done.insert(key);
  return VSM_OK;
}

int cu_count() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(g_cu_mu);
  auto it = g_cu.find(dev);
  if (it != g_cu.end()) return it->second;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  g_cu[dev] = v;
  return v;
}

template <typename T, typename Q>
static quad<T> cvt_quad(const Q* q) {
  quad<T> r;
  r.mu = q->mu;
  r.wt = q->wt;
  r.N = q->N;
  r.n_stokes = q->n_stokes;
  r.i_mu0 = q->i_mu0;
  r.mu0 = q->mu0;
  return r;
}
template <typename T, typename A>
static added<T> cvt_added(const A* a) {
  added<T> r;
  r.r_mp = a->r_mp;
  r.t_pp = a->t_pp;
  r.r_pm = a->r_pm;
  r.t_mm = a->t_mm;
  r.j0_p = a->j0_p;
  r.j0_m = a->j0_m;
  r.mat_stride = a->mat_stride;
  r.d_symmetric = a->d_symmetric;
  return r;
}
template <typename T, typename C>
static composite<T> cvt_comp(const C* c) {
  composite<T> r;
  r.R_mp = c->R_mp;
  r.R_pm = c->R_pm;
  r.T_pp = c->T_pp;
  r.T_mm = c->T_mm;
  r.J0_p = c->J0_p;
  r.J0_m = c->J0_m;
  return r;
}

template <typename Q>
static int check_quad(const Q* q) {
  VSM_REQUIRE(q != nullptr, "quad: null");
  VSM_REQUIRE(q->mu && q->wt, "quad: null mu/wt");
  VSM_REQUIRE(q->N > 0 && q->n_stokes >= 1 && q->n_stokes <= 4 && q->N % q->n_stokes == 0,
              "quad: bad N=%d / n_stokes=%d", q->N, q->n_stokes);
  VSM_REQUIRE(q->i_mu0 >= 0 && (q->i_mu0 + 1) * q->n_stokes <= q->N, "quad: i_mu0=%d out of range", q->i_mu0);
  return VSM_OK;
}
template <typename A>
static int check_added(const A* a) {
  VSM_REQUIRE(a != nullptr, "added: null");
  VSM_REQUIRE(a->r_mp && a->t_pp && a->j0_p && a->j0_m, "added: null field");
  VSM_REQUIRE(a->d_symmetric > 0 || (a->r_pm && a->t_mm), "added: null r_pm/t_mm without d_symmetric");
  VSM_REQUIRE(a->d_symmetric >= 0 && a->d_symmetric <= 4, "added: bad d_symmetric");
  return VSM_OK;
}
template <typename C>
static int check_comp(const C* c) {
  VSM_REQUIRE(c != nullptr, "composite: null");
  VSM_REQUIRE(c->R_mp && c->R_pm && c->T_pp && c->T_mm && c->J0_p && c->J0_m, "composite: null field");
  return VSM_OK;
}

template <typename T, typename Q, typename A>
static int elemental_doubling_impl(const Q* q, int S, int m, int ndoubl, const T* dtau, const T* varpi,
                                   const T* tau_sum, const T* F0, const T* Zpp, const T* Zmp, long long zs,
                                   const A* ad, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_added(ad))) return rc;
  VSM_REQUIRE(S >= 0 && m >= 0 && ndoubl >= 0, "elemental_doubling: bad S/m/ndoubl");
  VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp, "elemental_doubling: null input");
  const quad<T> qq = cvt_quad<T>(q);
  const added<T> aa = cvt_added<T>(ad);
  hipStream_t st = as_stream(stream);
  if (q->N <= fused_max_n<T>())
    return fused_elemental_doubling<T>(qq, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, aa, st);
  // operator-level path for N that does not fit on-chip
  VSM_REQUIRE(ad->d_symmetric == 0, "elemental_doubling: d_symmetric layers need the fused kernels (N=%d too large)", q->N);
  if ((rc = elemental<T>(qq, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, aa, st))) return rc;
  if (ndoubl == 0) return VSM_OK;
  const size_t we = vsm_doubling_work_elems(q->N, S);
  T* work = static_cast<T*>(scratch(we * sizeof(T) + (size_t)S * sizeof(T), 0, st));
  if (!work) return VSM_ERR_HIP;
  T* expk = work + we;
  // expk = exp(-dtau/mu0)  (rt_kernel.jl:339-349 init_layer)
  if ((rc = layer_expk<T>(S, dtau, (T)q->mu0, expk, st))) return rc;
  return doubling<T>(q->N, q->n_stokes, S, ndoubl, expk, aa, work, st);
}

template <typename T, typename C, typename A>
static int interaction_impl(int iface, int N, int S, const C* c, const A* a, T* work, void* stream, bool oplevel) {
  int rc;
  if ((rc = check_comp(c)) || (rc = check_added(a))) return rc;
  VSM_REQUIRE(N > 0 && S >= 0, "interaction: bad N/S");
  VSM_REQUIRE(iface >= 0 && iface <= 3, "interaction: unknown scattering interface %d", iface);
  hipStream_t st = as_stream(stream);
  if (!oplevel && iface == VSM_IFACE_11 && N <= fused_max_n<T>())
    return fused_interaction<T>(iface, N, S, cvt_comp<T>(c), cvt_added<T>(a), st);
  VSM_REQUIRE(a->d_symmetric == 0, "interaction: d_symmetric layers are only accepted by the fused 11 kernel");
  if constexpr (sizeof(T) == 8) {
    static const bool no_strip = ab_switch("VSM_NO_STRIP128");
    if (!oplevel && !no_strip && iface == VSM_IFACE_11 && strip128_supported(N))
      return strip128_interaction11(N, S, cvt_comp<T>(c), cvt_added<T>(a), st);
  } else {
    if (!oplevel && iface == VSM_IFACE_11 && strip128_f32_supported(N))
      return strip128_interaction11<T>(N, S, cvt_comp<T>(c), cvt_added<T>(a), st);
  }
  if (!work) {
    work = static_cast<T*>(scratch(vsm_interaction_work_elems(N, S) * sizeof(T), 1, st));
    if (!work) return VSM_ERR_HIP;
  }
  return interaction_generic<T>(iface, N, S, cvt_comp<T>(c), cvt_added<T>(a), work, st);
}

template <typename T, typename Q, typename C, typename A>
static int layer_forward_impl(const Q* q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
                              const T* F0, const T* Zpp, const T* Zmp, long long zs, int ncomp, const T* fcomp, T* z_scratch,
                              int toa, const C* c, const A* ad, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_comp(c))) return rc;
  VSM_REQUIRE(S >= 0 && m >= 0 && ndoubl >= 0, "layer_forward: bad S/m/ndoubl");
  VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp, "layer_forward: null input");
  VSM_REQUIRE(ncomp >= 0 && (ncomp == 0 || fcomp), "layer_forward: bad component mix");
  hipStream_t st = as_stream(stream);
  if constexpr (sizeof(T) == 8) {
    static const bool no_fuse = ab_switch("VSM_NO_STRIP") || ab_switch("VSM_NO_LAYER_FUSION");
    if (!no_fuse && strip_layer_supported(q->N) && ncomp <= 4)
      return strip_layer_forward(cvt_quad<T>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, zsrc<T>{Zpp, Zmp, zs, ncomp, fcomp},
                                 toa, cvt_comp<T>(c), st);
  }
  if constexpr (sizeof(T) == 4) {
    static const bool no_fuse32 = ab_switch("VSM_NO_LAYER_FUSION");
    if (!no_fuse32 && strip32_supported(q->N) && ncomp <= 4)
      return strip32_layer_forward(cvt_quad<T>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, zsrc<T>{Zpp, Zmp, zs, ncomp, fcomp},
                                   toa, cvt_comp<T>(c), st);
  }
  if (ncomp > 0) {  // materialise Z[N,N,S] for the kernels that do not mix on the fly
    const long long per = (long long)q->N * q->N * S;
    if (!z_scratch) {
      z_scratch = static_cast<T*>(scratch((size_t)(2 * per) * sizeof(T), 2, st));
      if (!z_scratch) return VSM_ERR_HIP;
    }
    if ((rc = mix_Z<T>(q->N, S, ncomp, Zpp, Zmp, fcomp, z_scratch, z_scratch + per, st))) return rc;
    Zpp = z_scratch;
    Zmp = z_scratch + per;
    zs = (long long)q->N * q->N;
  }
  // two launches through the caller's AddedLayer
  VSM_REQUIRE(ad != nullptr, "layer_forward: this shape needs an AddedLayer as scratch (N=%d)", q->N);
  if ((rc = elemental_doubling_impl<T>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, ad, stream))) return rc;
  if (toa) return copy_added_to_composite<T>(q->N, S, cvt_added<T>(ad), cvt_comp<T>(c), st);
  return interaction_impl<T>(VSM_IFACE_11, q->N, S, c, ad, (T*)nullptr, stream, false);
}

// vsm_layer_forward(_mix) for nm Fourier moments of one layer: ONE launch with gridDim.y = nm where the strip kernel takes the
// shape (FP64, 32 < N <= 60, ncomp <= 4, groups of VSM_MM_MAX moments), else the single-moment path moment by moment
template <typename T, typename Q, typename C, typename A>
static int layer_forward_multi_impl(const Q* q, int S, int nm, const int* m, int ndoubl, const T* dtau, const T* varpi,
                                    const T* tau_sum, const T* F0, int ncomp, const T* const* Zpp, const T* const* Zmp,
                                    long long zs, const T* fcomp, T* z_scratch, int toa, const C* comps, const A* ad,
                                    void* stream) {
  int rc;
  if ((rc = check_quad(q))) return rc;
  VSM_REQUIRE(nm >= 0 && (nm == 0 || (m && Zpp && Zmp && comps)), "layer_forward_multi: bad moment list");
  VSM_REQUIRE(S >= 0 && ndoubl >= 0, "layer_forward_multi: bad S/ndoubl");
  VSM_REQUIRE(ncomp >= 0 && (ncomp == 0 || fcomp), "layer_forward_multi: bad component mix");
  for (int i = 0; i < nm; ++i) {
    if ((rc = check_comp(&comps[i]))) return rc;
    VSM_REQUIRE(m[i] >= 0 && Zpp[i] && Zmp[i], "layer_forward_multi: bad moment %d", i);
  }
  if constexpr (sizeof(T) == 8) {
    static const bool no_fuse = ab_switch("VSM_NO_STRIP") || ab_switch("VSM_NO_LAYER_FUSION") ||
                                ab_switch("VSM_NO_MOMENT_BATCH");
    if (!no_fuse && strip_layer_supported(q->N) && ncomp <= 4) {
      VSM_REQUIRE(dtau && varpi && tau_sum && F0, "layer_forward_multi: null input");
      for (int i0 = 0; i0 < nm; i0 += VSM_MM_MAX) {
        const int n = nm - i0 < VSM_MM_MAX ? nm - i0 : VSM_MM_MAX;
        layer_mm_args<double> a;
        for (int i = 0; i < VSM_MM_MAX; ++i) {
          const int j = i0 + (i < n ? i : 0);
          a.m[i] = m[j];
          a.z[i] = zsrc<double>{Zpp[j], Zmp[j], ncomp ? 0 : zs, ncomp, fcomp};
          a.c[i] = cvt_comp<double>(&comps[j]);
        }
        if ((rc = strip_layer_forward_mm(cvt_quad<double>(q), S, n, ndoubl, dtau, varpi, tau_sum, F0, a, toa, as_stream(stream))))
          return rc;
      }
      return VSM_OK;
    }
  }
  if constexpr (sizeof(T) == 4) {
    static const bool no_fuse32 = ab_switch("VSM_NO_LAYER_FUSION") || ab_switch("VSM_NO_MOMENT_BATCH");
    if (!no_fuse32 && strip32_supported(q->N) && ncomp <= 4) {
      VSM_REQUIRE(dtau && varpi && tau_sum && F0, "layer_forward_multi: null input");
      for (int i0 = 0; i0 < nm; i0 += VSM_MM_MAX) {
        const int n = nm - i0 < VSM_MM_MAX ? nm - i0 : VSM_MM_MAX;
        layer_mm_args<float> a;
        for (int i = 0; i < VSM_MM_MAX; ++i) {
          const int j = i0 + (i < n ? i : 0);
          a.m[i] = m[j];
          a.z[i] = zsrc<float>{Zpp[j], Zmp[j], ncomp ? 0 : zs, ncomp, fcomp};
          a.c[i] = cvt_comp<float>(&comps[j]);
        }
        if ((rc = strip32_layer_forward_mm(cvt_quad<float>(q), S, n, ndoubl, dtau, varpi, tau_sum, F0, a, toa, as_stream(stream))))
          return rc;
      }
      return VSM_OK;
    }
  }
  for (int i = 0; i < nm; ++i)
    if ((rc = layer_forward_impl<T>(q, S, m[i], ndoubl, dtau, varpi, tau_sum, F0, Zpp[i], Zmp[i], ncomp ? 0 : zs, ncomp, fcomp,
                                    z_scratch, toa, &comps[i], ad, stream)))
      return rc;
  return VSM_OK;
}

}  // namespace vsm

using namespace vsm;

extern "C" {

int vsm_version(void) { return 100; /* 0.1.0 */ }
int vsm_release_scratch(void) { return release_scratch(); }
const char* vsm_build_id(void) { return VSM_BUILD_ID; }
int vsm_device_status(int* flags_h, int reset, void* stream) {
  VSM_REQUIRE(flags_h != nullptr, "device_status: null");
  int* d = device_status();
  if (!d) return VSM_ERR_HIP;
  // read and reset ON the caller's stream (ordered behind its kernels, and the reset ahead of whatever it launches next); kernels
  // still running on OTHER streams keep raising flags that a reset here may or may not clear -- synchronise those first
  hipStream_t st = as_stream(stream);
  VSM_HIP(hipMemcpyAsync(flags_h, d, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  if (reset) VSM_HIP(hipMemsetAsync(d, 0, 4 * sizeof(int), st));
  VSM_HIP(hipStreamSynchronize(st));
  return VSM_OK;
}
const char* vsm_last_error(void) { return g_err; }
int vsm_device_count(int* count) {
  VSM_REQUIRE(count != nullptr, "device_count: null");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    hip_fail(e, "hipGetDeviceCount");
    return VSM_ERR_NO_DEVICE;
  }
  *count = n;
  return VSM_OK;
}
int vsm_device_name(int device, char* buf, size_t buflen) {
  VSM_REQUIRE(buf && buflen > 0, "device_name: bad buffer");
  hipDeviceProp_t p;
  VSM_HIP(hipGetDeviceProperties(&p, device));
  snprintf(buf, buflen, "%s (%s)", p.name, p.gcnArchName);
  return VSM_OK;
}
int vsm_sync(void* stream) {
  VSM_HIP(hipStreamSynchronize(as_stream(stream)));
  return VSM_OK;
}
int vsm_fused_max_n(int elem_size) { return elem_size == 8 ? fused_max_n<double>() : fused_max_n<float>(); }

// ---- batched_mul / batch_inv! -------------------------------------------------
int vsm_batched_mul_f64(int M, int Nc, int K, int S, const double* A, long long sa, const double* B, long long sb,
                        double* C, void* stream) {
  if (S == 0 && M > 0 && Nc > 0 && K > 0) return VSM_OK;   // empty batch: nothing to do (pointers may be null)
  VSM_REQUIRE(M > 0 && Nc > 0 && K > 0 && S >= 0 && A && B && C, "batched_mul: bad argument");
  return gemm<double>(M, Nc, K, S, A, sa, B, sb, C, (long long)M * Nc, 1.0, nullptr, 0, 0.0, 0.0, as_stream(stream));
}
int vsm_batched_mul_f32(int M, int Nc, int K, int S, const float* A, long long sa, const float* B, long long sb,
                        float* C, void* stream) {
  if (S == 0 && M > 0 && Nc > 0 && K > 0) return VSM_OK;   // empty batch: nothing to do (pointers may be null)
  VSM_REQUIRE(M > 0 && Nc > 0 && K > 0 && S >= 0 && A && B && C, "batched_mul: bad argument");
  return gemm<float>(M, Nc, K, S, A, sa, B, sb, C, (long long)M * Nc, 1.f, nullptr, 0, 0.f, 0.f, as_stream(stream));
}
int vsm_batch_inv_f64(int N, int S, const double* A, double* X, int* info, void* stream) {
  if (S == 0 && N > 0) return VSM_OK;
  VSM_REQUIRE(N > 0 && S >= 0 && A && X, "batch_inv: bad argument");
  return batch_inv<double>(N, S, A, X, info, as_stream(stream));
}
int vsm_batch_inv_f32(int N, int S, const float* A, float* X, int* info, void* stream) {
  if (S == 0 && N > 0) return VSM_OK;
  VSM_REQUIRE(N > 0 && S >= 0 && A && X, "batch_inv: bad argument");
  return batch_inv<float>(N, S, A, X, info, as_stream(stream));
}

// ---- CoreKernel -----------------------------------------------------------------
int vsm_elemental_doubling_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau,
                               const double* varpi, const double* tau_sum, const double* F0, const double* Zpp,
                               const double* Zmp, long long z_stride, const vsm_added_f64* added, void* stream) {
  return elemental_doubling_impl<double>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, added, stream);
}
int vsm_elemental_doubling_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                               const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp,
                               long long z_stride, const vsm_added_f32* added, void* stream) {
  return elemental_doubling_impl<float>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, added, stream);
}

int vsm_elemental_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                      const double* tau_sum, const double* F0, const double* Zpp, const double* Zmp,
                      long long z_stride, const vsm_added_f64* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_elemental_f64: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp, "elemental: null input");
  return elemental<double>(cvt_quad<double>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride,
                           cvt_added<double>(added), as_stream(stream));
}
int vsm_elemental_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                      const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp, long long z_stride,
                      const vsm_added_f32* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_elemental_f32: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp, "elemental: null input");
  return elemental<float>(cvt_quad<float>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride,
                          cvt_added<float>(added), as_stream(stream));
}

size_t vsm_doubling_work_elems(int N, int S) {
  return (size_t)3 * N * N * (size_t)S + (size_t)4 * N * (size_t)S;
}
int vsm_doubling_f64(int N, int n_stokes, int S, int ndoubl, double* expk, const vsm_added_f64* added, double* work,
                     void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_doubling_f64: d_symmetric layers are not accepted here");
  if ((rc = check_added(added))) return rc;
  VSM_REQUIRE(N > 0 && S >= 0 && ndoubl >= 0 && expk && (work || ndoubl == 0), "doubling: bad argument");
  return doubling<double>(N, n_stokes, S, ndoubl, expk, cvt_added<double>(added), work, as_stream(stream));
}
int vsm_doubling_f32(int N, int n_stokes, int S, int ndoubl, float* expk, const vsm_added_f32* added, float* work,
                     void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_doubling_f32: d_symmetric layers are not accepted here");
  if ((rc = check_added(added))) return rc;
  VSM_REQUIRE(N > 0 && S >= 0 && ndoubl >= 0 && expk && (work || ndoubl == 0), "doubling: bad argument");
  return doubling<float>(N, n_stokes, S, ndoubl, expk, cvt_added<float>(added), work, as_stream(stream));
}

int vsm_thermal_source_f64(const vsm_quad_f64* q, int S, const double* dtau, const double* varpi, const double* B,
                           const vsm_added_f64* added, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(S >= 0 && (S == 0 || (dtau && varpi && B)), "thermal_source: null input");
  return thermal_source<double>(cvt_quad<double>(q), S, dtau, varpi, B, cvt_added<double>(added), as_stream(stream));
}
int vsm_thermal_source_f32(const vsm_quad_f32* q, int S, const float* dtau, const float* varpi, const float* B,
                           const vsm_added_f32* added, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(S >= 0 && (S == 0 || (dtau && varpi && B)), "thermal_source: null input");
  return thermal_source<float>(cvt_quad<float>(q), S, dtau, varpi, B, cvt_added<float>(added), as_stream(stream));
}
int vsm_noscat_layer_f64(const vsm_quad_f64* q, int S, const double* tau, const vsm_added_f64* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_noscat_layer_f64: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(tau != nullptr, "noscat_layer: null tau");
  return noscat_layer<double>(cvt_quad<double>(q), S, tau, cvt_added<double>(added), as_stream(stream));
}
int vsm_noscat_layer_f32(const vsm_quad_f32* q, int S, const float* tau, const vsm_added_f32* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_noscat_layer_f32: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(tau != nullptr, "noscat_layer: null tau");
  return noscat_layer<float>(cvt_quad<float>(q), S, tau, cvt_added<float>(added), as_stream(stream));
}

int vsm_copy_added_to_composite_f64(int N, int S, const vsm_added_f64* added, const vsm_composite_f64* comp,
                                    void* stream) {
  int rc;
  if ((rc = check_added(added)) || (rc = check_comp(comp))) return rc;
  return copy_added_to_composite<double>(N, S, cvt_added<double>(added), cvt_comp<double>(comp), as_stream(stream));
}
int vsm_copy_added_to_composite_f32(int N, int S, const vsm_added_f32* added, const vsm_composite_f32* comp,
                                    void* stream) {
  int rc;
  if ((rc = check_added(added)) || (rc = check_comp(comp))) return rc;
  return copy_added_to_composite<float>(N, S, cvt_added<float>(added), cvt_comp<float>(comp), as_stream(stream));
}

size_t vsm_interaction_work_elems(int N, int S) {
  return (size_t)3 * N * N * (size_t)S + (size_t)2 * N * (size_t)S;
}
int vsm_interaction_f64(int iface, int N, int S, const vsm_composite_f64* comp, const vsm_added_f64* added,
                        double* work, void* stream) {
  return interaction_impl<double>(iface, N, S, comp, added, work, stream, false);
}
int vsm_interaction_oplevel_f64(int iface, int N, int S, const vsm_composite_f64* comp, const vsm_added_f64* added,
                                double* work, void* stream) {
  return interaction_impl<double>(iface, N, S, comp, added, work, stream, true);
}
int vsm_interaction_f32(int iface, int N, int S, const vsm_composite_f32* comp, const vsm_added_f32* added,
                        float* work, void* stream) {
  return interaction_impl<float>(iface, N, S, comp, added, work, stream, false);
}
int vsm_interaction_oplevel_f32(int iface, int N, int S, const vsm_composite_f32* comp, const vsm_added_f32* added,
                                float* work, void* stream) {
  return interaction_impl<float>(iface, N, S, comp, added, work, stream, true);
}
int vsm_layer_forward_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                          const double* tau_sum, const double* F0, const double* Zpp, const double* Zmp, long long z_stride,
                          int toa, const vsm_composite_f64* comp, const vsm_added_f64* added_scratch, void* stream) {
  return layer_forward_impl<double>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, 0, nullptr, nullptr, toa,
                                    comp, added_scratch, stream);
}
int vsm_layer_forward_mix_f64(const vsm_quad_f64* q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                              const double* tau_sum, const double* F0, int ncomp, const double* Zpp_comp,
                              const double* Zmp_comp, const double* fcomp, double* z_scratch, int toa,
                              const vsm_composite_f64* comp, const vsm_added_f64* added_scratch, void* stream) {
  VSM_REQUIRE(ncomp >= 1, "layer_forward_mix: ncomp >= 1 required");
  return layer_forward_impl<double>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp_comp, Zmp_comp, 0, ncomp, fcomp, z_scratch,
                                    toa, comp, added_scratch, stream);
}
int vsm_layer_forward_mix_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                              const float* tau_sum, const float* F0, int ncomp, const float* Zpp_comp, const float* Zmp_comp,
                              const float* fcomp, float* z_scratch, int toa, const vsm_composite_f32* comp,
                              const vsm_added_f32* added_scratch, void* stream) {
  VSM_REQUIRE(ncomp >= 1, "layer_forward_mix: ncomp >= 1 required");
  return layer_forward_impl<float>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp_comp, Zmp_comp, 0, ncomp, fcomp, z_scratch,
                                   toa, comp, added_scratch, stream);
}
int vsm_mix_Z_f64(int N, int S, int ncomp, const double* Zpp_comp, const double* Zmp_comp, const double* fcomp, double* Zpp,
                  double* Zmp, void* stream) {
  VSM_REQUIRE(N > 0 && S >= 0 && ncomp >= 1 && Zpp_comp && Zmp_comp && fcomp && Zpp && Zmp, "mix_Z: bad argument");
  return mix_Z<double>(N, S, ncomp, Zpp_comp, Zmp_comp, fcomp, Zpp, Zmp, as_stream(stream));
}
int vsm_mix_Z_f32(int N, int S, int ncomp, const float* Zpp_comp, const float* Zmp_comp, const float* fcomp, float* Zpp,
                  float* Zmp, void* stream) {
  VSM_REQUIRE(N > 0 && S >= 0 && ncomp >= 1 && Zpp_comp && Zmp_comp && fcomp && Zpp && Zmp, "mix_Z: bad argument");
  return mix_Z<float>(N, S, ncomp, Zpp_comp, Zmp_comp, fcomp, Zpp, Zmp, as_stream(stream));
}
int vsm_mix_Z_moments_f64(int N, int S, int ncomp, int nm, const double* const* Zpp_comp, const double* const* Zmp_comp, int single,
                          const double* fcomp, double* Zpp, double* Zmp, void* stream) {
  VSM_REQUIRE(N > 0 && S >= 0 && ncomp >= 0 && nm >= 0 && Zpp_comp && Zmp_comp && (ncomp == 0 || fcomp) && Zpp && Zmp && single >= 0,
              "mix_Z_moments: bad argument");
  return mix_Z_moments<double>(N, S, ncomp, nm, Zpp_comp, Zmp_comp, single, fcomp, Zpp, Zmp, as_stream(stream));
}
int vsm_mix_Z_moments_f32(int N, int S, int ncomp, int nm, const float* const* Zpp_comp, const float* const* Zmp_comp, int single,
                          const float* fcomp, float* Zpp, float* Zmp, void* stream) {
  VSM_REQUIRE(N > 0 && S >= 0 && ncomp >= 0 && nm >= 0 && Zpp_comp && Zmp_comp && (ncomp == 0 || fcomp) && Zpp && Zmp && single >= 0,
              "mix_Z_moments: bad argument");
  return mix_Z_moments<float>(N, S, ncomp, nm, Zpp_comp, Zmp_comp, single, fcomp, Zpp, Zmp, as_stream(stream));
}
int vsm_layer_forward_multi_f64(const vsm_quad_f64* q, int S, int nm, const int* m, int ndoubl, const double* dtau,
                                const double* varpi, const double* tau_sum, const double* F0, int ncomp,
                                const double* const* Zpp, const double* const* Zmp, long long z_stride, const double* fcomp,
                                double* z_scratch, int toa, const vsm_composite_f64* comps, const vsm_added_f64* added_scratch,
                                void* stream) {
  return layer_forward_multi_impl<double>(q, S, nm, m, ndoubl, dtau, varpi, tau_sum, F0, ncomp, Zpp, Zmp, z_stride, fcomp,
                                          z_scratch, toa, comps, added_scratch, stream);
}
int vsm_layer_forward_multi_f32(const vsm_quad_f32* q, int S, int nm, const int* m, int ndoubl, const float* dtau,
                                const float* varpi, const float* tau_sum, const float* F0, int ncomp, const float* const* Zpp,
                                const float* const* Zmp, long long z_stride, const float* fcomp, float* z_scratch, int toa,
                                const vsm_composite_f32* comps, const vsm_added_f32* added_scratch, void* stream) {
  return layer_forward_multi_impl<float>(q, S, nm, m, ndoubl, dtau, varpi, tau_sum, F0, ncomp, Zpp, Zmp, z_stride, fcomp,
                                         z_scratch, toa, comps, added_scratch, stream);
}
// does vsm_layer_forward_thermal_* fuse this shape?  (FP64: the strip kernels, 32 < N <= 60; FP32: 64 < N <= 96)
int vsm_layer_thermal_fused(int N, int is_f64) {
  static const bool off = ab_switch("VSM_NO_LAYER_FUSION") || ab_switch("VSM_NO_STRIP");
  if (off) return 0;
  return is_f64 ? (strip_layer_supported(N) ? 1 : 0) : (strip32_supported(N) ? 1 : 0);
}
// the `:thermal` per-source slot of a scattering layer through the fused layer kernel (m = 0; FP64, 32 < N <= 60, ncomp <= 4):
// same launch as vsm_layer_forward(_mix) with the solar source replaced by the thermal one and expk = 1
int vsm_layer_forward_thermal_f64(const vsm_quad_f64* q, int S, int ndoubl, const double* dtau, const double* varpi,
                                  const double* thermal_B, int ncomp, const double* Zpp, const double* Zmp, long long z_stride,
                                  const double* fcomp, int toa, const vsm_composite_f64* comp, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_comp(comp))) return rc;
  VSM_REQUIRE(S >= 0 && ndoubl >= 0 && dtau && varpi && thermal_B && Zpp && Zmp, "layer_forward_thermal: bad argument");
  VSM_REQUIRE(ncomp >= 0 && (ncomp == 0 || fcomp), "layer_forward_thermal: bad component mix");
  if (!vsm_layer_thermal_fused(q->N, 1) || ncomp > 4) {
    set_error("layer_forward_thermal: only the FP64 strip shapes (32 < N <= 60, ncomp <= 4) are fused; N=%d ncomp=%d -- use "
              "vsm_elemental / vsm_thermal_source / vsm_doubling / vsm_interaction", q->N, ncomp);
    return VSM_ERR_UNSUPPORTED;
  }
  return strip_layer_forward(cvt_quad<double>(q), S, 0, ndoubl, dtau, varpi, dtau /* tau_sum: not read */, thermal_B,
                             zsrc<double>{Zpp, Zmp, ncomp ? 0 : z_stride, ncomp, fcomp}, toa, cvt_comp<double>(comp),
                             as_stream(stream), 1);
}
int vsm_layer_forward_thermal_f32(const vsm_quad_f32* q, int S, int ndoubl, const float* dtau, const float* varpi,
                                  const float* thermal_B, int ncomp, const float* Zpp, const float* Zmp, long long z_stride,
                                  const float* fcomp, int toa, const vsm_composite_f32* comp, void* stream) {
  int rc;
  if ((rc = check_quad(q)) || (rc = check_comp(comp))) return rc;
  VSM_REQUIRE(S >= 0 && ndoubl >= 0 && dtau && varpi && thermal_B && Zpp && Zmp, "layer_forward_thermal: bad argument");
  VSM_REQUIRE(ncomp >= 0 && (ncomp == 0 || fcomp), "layer_forward_thermal: bad component mix");
  if (!vsm_layer_thermal_fused(q->N, 0) || ncomp > 4) {
    set_error("layer_forward_thermal: only the FP32 strip shapes (64 < N <= 96, ncomp <= 4) are fused; N=%d ncomp=%d -- use "
              "vsm_elemental / vsm_thermal_source / vsm_doubling / vsm_interaction", q->N, ncomp);
    return VSM_ERR_UNSUPPORTED;
  }
  return strip32_layer_forward(cvt_quad<float>(q), S, 0, ndoubl, dtau, varpi, dtau /* tau_sum: not read */, thermal_B,
                               zsrc<float>{Zpp, Zmp, ncomp ? 0 : z_stride, ncomp, fcomp}, toa, cvt_comp<float>(comp),
                               as_stream(stream), 1);
}
int vsm_layer_forward_f32(const vsm_quad_f32* q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const float* Zpp, const float* Zmp, long long z_stride,
                          int toa, const vsm_composite_f32* comp, const vsm_added_f32* added_scratch, void* stream) {
  return layer_forward_impl<float>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, 0, nullptr, nullptr, toa,
                                   comp, added_scratch, stream);
}

int vsm_lambertian_surface_f64(const vsm_quad_f64* q, int S, int m, double albedo, const double* tau_sum,
                               const vsm_added_f64* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_lambertian_surface_f64: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(tau_sum != nullptr, "lambertian_surface: null tau_sum");
  return lambertian_surface<double>(cvt_quad<double>(q), S, m, albedo, tau_sum, cvt_added<double>(added),
                                    as_stream(stream));
}
int vsm_lambertian_surface_f32(const vsm_quad_f32* q, int S, int m, float albedo, const float* tau_sum,
                               const vsm_added_f32* added, void* stream) {
  int rc;
  VSM_REQUIRE(added && added->d_symmetric == 0, "vsm_lambertian_surface_f32: d_symmetric layers are not accepted here");
  if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;
  VSM_REQUIRE(tau_sum != nullptr, "lambertian_surface: null tau_sum");
  return lambertian_surface<float>(cvt_quad<float>(q), S, m, albedo, tau_sum, cvt_added<float>(added),
                                   as_stream(stream));
}

int vsm_postprocess_vza_f64(int N, int n_stokes, int S, int nV, const int* row0_h, const double* w_h,
                            const double* J0_m, const double* J0_p, double* R, double* T, void* stream) {
  VSM_REQUIRE(row0_h && w_h && J0_m && J0_p && R && T, "postprocess_vza: null argument");
  return postprocess_vza<double>(N, n_stokes, S, nV, row0_h, w_h, J0_m, J0_p, R, T, as_stream(stream));
}
int vsm_postprocess_vza_f32(int N, int n_stokes, int S, int nV, const int* row0_h, const float* w_h,
                            const float* J0_m, const float* J0_p, float* R, float* T, void* stream) {
  VSM_REQUIRE(row0_h && w_h && J0_m && J0_p && R && T, "postprocess_vza: null argument");
  return postprocess_vza<float>(N, n_stokes, S, nV, row0_h, w_h, J0_m, J0_p, R, T, as_stream(stream));
}

// ---- diagnostics ----------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_poison_lds(int n64) {
  extern __shared__ unsigned long long poison_smem[];
  for (int i = threadIdx.x; i < n64; i += 256) poison_smem[i] = 0x7ff8dead7fc0beefULL;   // NaN as f64 and as 2 x f32
  __syncthreads();
  if (poison_smem[(threadIdx.x * 97) % n64] == 0) asm volatile("s_nop 0");   // keep the stores
}
}  // namespace
int vsm_test_poison_lds(void* stream) {
  const int bytes = 160 * 1024;
  const int prepared = vsm::ensure_dyn_lds(reinterpret_cast<const void*>(k_poison_lds), 160 * 1024, "hipFuncSetAttribute(k_poison_lds)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(k_poison_lds, dim3(1024), dim3(256), bytes, as_stream(stream), bytes / 8);
  VSM_LAUNCH_CHECK("k_poison_lds");
  return VSM_OK;
}
int vsm_test_lds_mm_f64(int N, int S, const double* A, const double* B, double* C, void* stream) {
  VSM_REQUIRE(N > 0 && A && B && C, "test_lds_mm: bad argument");
  return test_lds_mm<double>(N, S, A, B, C, as_stream(stream));
}
int vsm_test_lds_mm_f32(int N, int S, const float* A, const float* B, float* C, void* stream) {
  VSM_REQUIRE(N > 0 && A && B && C, "test_lds_mm: bad argument");
  return test_lds_mm<float>(N, S, A, B, C, as_stream(stream));
}
int vsm_test_lds_inv_f64(int N, int S, const double* A, double* X, int mode, int* path_out, void* stream) {
  VSM_REQUIRE(N > 0 && A && X, "test_lds_inv: bad argument");
  return test_lds_inv<double>(N, S, A, X, mode, path_out, as_stream(stream));
}
int vsm_test_lds_inv_f32(int N, int S, const float* A, float* X, int mode, int* path_out, void* stream) {
  VSM_REQUIRE(N > 0 && A && X, "test_lds_inv: bad argument");
  return test_lds_inv<float>(N, S, A, X, mode, path_out, as_stream(stream));
}

}  // extern "C"

// ---- linearized pass ----------------------------------------------------------------
namespace vsm {
template <typename T, typename A>
static added_lin<T> cvt_al(const A* a) {
  added_lin<T> r;
  r.ap_r_mp = a->ap_r_mp; r.ap_t_pp = a->ap_t_pp; r.ap_r_pm = a->ap_r_pm; r.ap_t_mm = a->ap_t_mm;
  r.ap_J0_p = a->ap_J0_p; r.ap_J0_m = a->ap_J0_m; r.P = a->P; r.mat_stride = a->mat_stride;
  return r;
}
template <typename T, typename C>
static composite_lin<T> cvt_cl(const C* c) {
  composite_lin<T> r;
  r.R_mp = c->R_mp; r.R_pm = c->R_pm; r.T_pp = c->T_pp; r.T_mm = c->T_mm; r.J0_p = c->J0_p; r.J0_m = c->J0_m; r.P = c->P;
  return r;
}
template <typename A>
static int check_al(const A* a) {
  VSM_REQUIRE(a != nullptr, "added_lin: null");
  VSM_REQUIRE(a->ap_r_mp && a->ap_t_pp && a->ap_r_pm && a->ap_t_mm && a->ap_J0_p && a->ap_J0_m, "added_lin: null field");
  VSM_REQUIRE(a->P >= 1 && a->P <= 64, "added_lin: bad P=%d", a->P);
  return VSM_OK;
}
template <typename C>
static int check_cl(const C* c) {
  VSM_REQUIRE(c != nullptr, "composite_lin: null");
  VSM_REQUIRE(c->R_mp && c->R_pm && c->T_pp && c->T_mm && c->J0_p && c->J0_m, "composite_lin: null field");
  VSM_REQUIRE(c->P >= 1 && c->P <= 64, "composite_lin: bad P=%d", c->P);
  return VSM_OK;
}
}  // namespace vsm

#define VSM_LIN_API(T, SFX)                                                                                            \
  int vsm_elemental_lin_##SFX(const vsm_quad_##SFX* q, int S, int m, int ndoubl, const T* dtau, const T* varpi,        \
                              const T* tau_sum, const T* F0, const T* Zpp, const T* Zmp, long long z_stride,           \
                              int p_layer, const T* dtau_dot, const T* varpi_dot, const T* tau_sum_dot,                \
                              const T* Zpp_dot, const T* Zmp_dot, long long zds, long long zdp,                        \
                              const vsm_added_##SFX* added, const vsm_added_lin_##SFX* al, void* stream) {             \
    int rc;                                                                                                            \
    if ((rc = check_quad(q)) || (rc = check_added(added)) || (rc = check_al(al))) return rc;                           \
    VSM_REQUIRE(added->d_symmetric == 0, "elemental_lin: d_symmetric layers are not accepted here");                   \
    VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp && dtau_dot && varpi_dot && tau_sum_dot,                  \
                "elemental_lin: null input");                                                                          \
    VSM_REQUIRE(p_layer >= 0 && p_layer <= al->P && (!Zpp_dot == !Zmp_dot), "elemental_lin: bad p_layer / Zdot");      \
    return elemental_lin<T>(cvt_quad<T>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, p_layer,       \
                            dtau_dot, varpi_dot, tau_sum_dot, Zpp_dot, Zmp_dot, zds, zdp, cvt_added<T>(added),         \
                            cvt_al<T>(al), as_stream(stream));                                                         \
  }                                                                                                                    \
  int vsm_elemental_lin_fold_##SFX(const vsm_quad_##SFX* q, int S, int m, int n_m0, int ndoubl, const T* dtau,         \
                                   const T* varpi, const T* tau_sum, const T* F0, const T* Zpp, const T* Zmp,          \
                                   long long z_stride, int p_layer, const T* dtau_dot, const T* varpi_dot,             \
                                   const T* tau_sum_dot, const vsm_added_##SFX* added, const vsm_added_lin_##SFX* al,  \
                                   void* stream) {                                                                     \
    int rc;                                                                                                            \
    if ((rc = check_quad(q)) || (rc = check_added(added)) || (rc = check_al(al))) return rc;                           \
    VSM_REQUIRE(added->d_symmetric == 0, "elemental_lin_fold: d_symmetric layers are not accepted here");              \
    VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zpp && Zmp && dtau_dot && varpi_dot && tau_sum_dot,                  \
                "elemental_lin_fold: null input");                                                                     \
    VSM_REQUIRE(p_layer >= 0 && p_layer <= al->P && n_m0 >= 0 && n_m0 <= S && (m > 0 || n_m0 == 0 || n_m0 == S),      \
                "elemental_lin_fold: bad p_layer / n_m0");                                                             \
    return elemental_lin<T>(cvt_quad<T>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, z_stride, p_layer,       \
                            dtau_dot, varpi_dot, tau_sum_dot, (const T*)nullptr, (const T*)nullptr, 0, 0,              \
                            cvt_added<T>(added), cvt_al<T>(al), as_stream(stream), n_m0);                              \
  }                                                                                                                    \
  int vsm_elemental_lin_mix_##SFX(const vsm_quad_##SFX* q, int S, int m, int ndoubl, const T* dtau, const T* varpi,    \
                                  const T* tau_sum, const T* F0, int ncomp, int ncomp_total, const T* Zc_pp,           \
                                  const T* Zc_mp, int zsel, const T* fz, int p_layer, const T* dtau_dot,               \
                                  const T* varpi_dot, const T* tau_sum_dot, const T* zdcoef,                           \
                                  const vsm_added_##SFX* added, const vsm_added_lin_##SFX* al, void* stream) {         \
    int rc;                                                                                                            \
    if ((rc = check_quad(q)) || (rc = check_added(added)) || (rc = check_al(al))) return rc;                           \
    VSM_REQUIRE(added->d_symmetric == 0, "elemental_lin_mix: d_symmetric layers are not accepted here");               \
    VSM_REQUIRE(dtau && varpi && tau_sum && F0 && Zc_pp && Zc_mp && dtau_dot && varpi_dot && tau_sum_dot && zdcoef,    \
                "elemental_lin_mix: null input");                                                                      \
    VSM_REQUIRE(ncomp >= 1 && ncomp_total >= ncomp && zsel < ncomp && (zsel >= 0 || fz),                               \
                "elemental_lin_mix: bad component selection");                                                         \
    VSM_REQUIRE(p_layer >= 0 && p_layer <= al->P, "elemental_lin_mix: bad p_layer");                                   \
    return elemental_lin_mix<T>(cvt_quad<T>(q), S, m, ndoubl, dtau, varpi, tau_sum, F0, ncomp, ncomp_total, Zc_pp,     \
                                Zc_mp, zsel, fz, p_layer, dtau_dot, varpi_dot, tau_sum_dot, zdcoef,                    \
                                cvt_added<T>(added), cvt_al<T>(al), as_stream(stream));                                \
  }                                                                                                                    \
  int vsm_doubling_lin_##SFX(int N, int n_stokes, int S, int ndoubl, T* expk, const T* dtau_dot_all, T mu0,            \
                             int n_active, const vsm_added_##SFX* added, const vsm_added_lin_##SFX* al, T* work,       \
                             void* stream) {                                                                           \
    int rc;                                                                                                            \
    if ((rc = check_added(added)) || (rc = check_al(al))) return rc;                                                   \
    VSM_REQUIRE(added->d_symmetric == 0, "doubling_lin: d_symmetric layers are not accepted here");                    \
    VSM_REQUIRE(N > 0 && S >= 0 && ndoubl >= 0 && expk && dtau_dot_all && (work || ndoubl == 0) &&         \
                    n_active >= 0 && n_active <= al->P, "doubling_lin: bad argument");                                 \
    return doubling_lin<T>(N, n_stokes, S, ndoubl, expk, dtau_dot_all, mu0, n_active, cvt_added<T>(added),             \
                           cvt_al<T>(al), work, as_stream(stream));                                                    \
  }                                                                                                                    \
  int vsm_interaction_lin_##SFX(int iface, int N, int S, const vsm_composite_##SFX* comp,                              \
                                const vsm_composite_lin_##SFX* cl, const vsm_added_##SFX* added,                       \
                                const vsm_added_lin_##SFX* al, T* work, void* stream) {                                \
    int rc;                                                                                                            \
    if ((rc = check_comp(comp)) || (rc = check_cl(cl)) || (rc = check_added(added)) || (rc = check_al(al))) return rc; \
    VSM_REQUIRE(added->d_symmetric == 0, "interaction_lin: d_symmetric layers are not accepted here");                 \
    VSM_REQUIRE(N > 0 && S >= 0 && work && cl->P == al->P, "interaction_lin: bad argument");               \
    return interaction_lin<T>(iface, N, S, cvt_comp<T>(comp), cvt_cl<T>(cl), cvt_added<T>(added), cvt_al<T>(al), work, \
                              as_stream(stream));                                                                      \
  }                                                                                                                    \
  int vsm_interaction_lin_range_##SFX(int iface, int N, int S, const vsm_composite_##SFX* comp,                        \
                                      const vsm_composite_lin_##SFX* cl, const vsm_added_##SFX* added,                 \
                                      const vsm_added_lin_##SFX* al, int p_lo, int p_hi, T* work, void* stream) {      \
    int rc;                                                                                                            \
    if ((rc = check_comp(comp)) || (rc = check_cl(cl)) || (rc = check_added(added)) || (rc = check_al(al))) return rc; \
    VSM_REQUIRE(added->d_symmetric == 0, "interaction_lin: d_symmetric layers are not accepted here");                 \
    VSM_REQUIRE(N > 0 && S >= 0 && work && cl->P == al->P && p_lo >= 0 && p_lo < p_hi && p_hi <= cl->P,                \
                "interaction_lin_range: bad argument");                                                                \
    /* the slots [p_lo, p_hi) as a view of the parameter axis (the slowest one of [N,N,S,P] / [N,S,P]) */              \
    composite_lin<T> c2 = cvt_cl<T>(cl);                                                                               \
    added_lin<T> a2 = cvt_al<T>(al);                                                                                   \
    const long long MS = (long long)N * N * S, VS = (long long)N * S;                                                  \
    const long long alp = a2.mat_stride == 0 ? (long long)N * N : MS;                                                  \
    c2.R_mp += p_lo * MS; c2.R_pm += p_lo * MS; c2.T_pp += p_lo * MS; c2.T_mm += p_lo * MS;                            \
    c2.J0_p += p_lo * VS; c2.J0_m += p_lo * VS;                                                                        \
    a2.ap_r_mp += p_lo * alp; a2.ap_r_pm += p_lo * alp; a2.ap_t_pp += p_lo * alp; a2.ap_t_mm += p_lo * alp;            \
    a2.ap_J0_p += p_lo * VS; a2.ap_J0_m += p_lo * VS;                                                                  \
    c2.P = a2.P = p_hi - p_lo;                                                                                         \
    return interaction_lin<T>(iface, N, S, cvt_comp<T>(comp), c2, cvt_added<T>(added), a2, work, as_stream(stream));   \
  }                                                                                                                    \
  int vsm_copy_added_to_composite_lin_##SFX(int N, int S, const vsm_added_lin_##SFX* al,                               \
                                            const vsm_composite_lin_##SFX* cl, void* stream) {                         \
    int rc;                                                                                                            \
    if ((rc = check_al(al)) || (rc = check_cl(cl))) return rc;                                                         \
    VSM_REQUIRE(al->P == cl->P, "copy_added_to_composite_lin: P mismatch");                                            \
    return copy_added_to_composite_lin<T>(N, S, cvt_al<T>(al), cvt_cl<T>(cl), as_stream(stream));                      \
  }                                                                                                                    \
  int vsm_lambertian_surface_lin_##SFX(const vsm_quad_##SFX* q, int S, int m, T albedo, int iparam, const T* tau_sum,  \
                                       const T* tau_sum_dot, int p_layer, const T* F0, const vsm_added_##SFX* added,   \
                                       const vsm_added_lin_##SFX* al, void* stream) {                                  \
    int rc;                                                                                                            \
    if ((rc = check_quad(q)) || (rc = check_added(added)) || (rc = check_al(al))) return rc;                           \
    VSM_REQUIRE(tau_sum && F0 && (tau_sum_dot || p_layer == 0) && iparam >= 0 && iparam < al->P && p_layer <= al->P,   \
                "lambertian_surface_lin: bad argument");                                                               \
    return lambertian_surface_lin<T>(cvt_quad<T>(q), S, m, albedo, iparam, tau_sum, tau_sum_dot, p_layer, F0,          \
                                     cvt_added<T>(added), cvt_al<T>(al), as_stream(stream));                           \
  }                                                                                                                    \
  int vsm_postprocess_vza_lin_##SFX(int N, int n_stokes, int S, int nV, int P, const int* row0_h, const T* w_h,        \
                                    const T* Jm, const T* Jp, T* Rd, T* Td, void* stream) {                            \
    VSM_REQUIRE(row0_h && w_h && Jm && Jp && Rd && Td, "postprocess_vza_lin: null argument");                          \
    return postprocess_vza_lin<T>(N, n_stokes, S, nV, P, row0_h, w_h, Jm, Jp, Rd, Td, as_stream(stream));              \
  }

extern "C" {
size_t vsm_doubling_lin_work_elems(int N, int S, int P) { return doubling_lin_work_elems<double>(N, S, P); }
size_t vsm_interaction_lin_work_elems(int N, int S, int P) { return interaction_lin_work_elems<double>(N, S, P); }
VSM_LIN_API(double, f64)
VSM_LIN_API(float, f32)
}

// ---- BRDF surfaces ---------------------------------------------------------------------------------------------------
#define VSM_SURF_API(T, SFX)                                                                                           \
  static cm_surf<T> cvt_cm(const vsm_coxmunk_##SFX* c) {                                                               \
    cm_surf<T> r;                                                                                                      \
    r.wind_speed = c->wind_speed;                                                                                      \
    r.n_re = c->n_water_re;                                                                                            \
    r.n_im = c->n_water_im;                                                                                            \
    r.whitecap_albedo = c->whitecap_albedo;                                                                            \
    r.include_whitecaps = c->include_whitecaps;                                                                        \
    r.shadowing = c->shadowing;                                                                                        \
    return r;                                                                                                          \
  }                                                                                                                    \
  extern "C" int vsm_coxmunk_reflectance_##SFX(const vsm_coxmunk_##SFX* surf, const vsm_quad_##SFX* q, int m, int nphi, \
                                               const T* phi, const T* wphi, T* rho, T* drho_dU, void* stream) {        \
    int rc;                                                                                                            \
    if ((rc = check_quad(q))) return rc;                                                                               \
    VSM_REQUIRE(surf && m >= 0 && phi && wphi && rho && q->N % q->n_stokes == 0, "coxmunk_reflectance: bad argument"); \
    return coxmunk_reflectance<T>(cvt_cm(surf), q->n_stokes, q->N / q->n_stokes, q->mu, m, nphi, phi, wphi, rho,       \
                                  drho_dU, as_stream(stream));                                                         \
  }                                                                                                                    \
  extern "C" int vsm_brdf_surface_##SFX(const vsm_quad_##SFX* q, int S, int m, const T* rho, const T* tau_sum,         \
                                        const vsm_added_##SFX* added, void* stream) {                                  \
    int rc;                                                                                                            \
    VSM_REQUIRE(added && added->d_symmetric == 0, "brdf_surface: d_symmetric layers are not accepted here");           \
    if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;                                                  \
    VSM_REQUIRE(S >= 0 && m >= 0 && rho && (tau_sum || S == 0), "brdf_surface: bad argument");                         \
    return brdf_surface<T>(cvt_quad<T>(q), S, m, rho, tau_sum, cvt_added<T>(added), as_stream(stream));                \
  }                                                                                                                    \
  extern "C" int vsm_lambertian_surface_spectral_##SFX(const vsm_quad_##SFX* q, int S, int m, const T* albedo,         \
                                                       const T* tau_sum, const vsm_added_##SFX* added, void* stream) { \
    int rc;                                                                                                            \
    VSM_REQUIRE(added && added->d_symmetric == 0, "lambertian_surface_spectral: d_symmetric layers are not accepted"); \
    if ((rc = check_quad(q)) || (rc = check_added(added))) return rc;                                                  \
    VSM_REQUIRE(S >= 0 && m >= 0 && (S == 0 || (albedo && tau_sum)), "lambertian_surface_spectral: bad argument");     \
    return lambertian_surface_spectral<T>(cvt_quad<T>(q), S, m, albedo, tau_sum, cvt_added<T>(added),                  \
                                          as_stream(stream));                                                          \
  }                                                                                                                    \
  extern "C" int vsm_brdf_surface_lin_##SFX(const vsm_quad_##SFX* q, int S, int m, const T* rho, const T* drho,        \
                                            int iparam, const T* tau_sum, const T* tau_sum_dot, int p_layer,           \
                                            const T* F0, const vsm_added_##SFX* added, const vsm_added_lin_##SFX* al,  \
                                            void* stream) {                                                            \
    int rc;                                                                                                            \
    VSM_REQUIRE(added && added->d_symmetric == 0, "brdf_surface_lin: d_symmetric layers are not accepted here");       \
    if ((rc = check_quad(q)) || (rc = check_added(added)) || (rc = check_al(al))) return rc;                           \
    VSM_REQUIRE(S >= 0 && m >= 0 && rho && drho && tau_sum && F0 && (tau_sum_dot || p_layer == 0) && iparam >= 0 &&    \
                    iparam < al->P && p_layer >= 0 && p_layer <= al->P,                                                \
                "brdf_surface_lin: bad argument");                                                                     \
    return brdf_surface_lin<T>(cvt_quad<T>(q), S, m, rho, drho, iparam, tau_sum, tau_sum_dot, p_layer, F0,             \
                               cvt_added<T>(added), cvt_al<T>(al), as_stream(stream));                                 \
  }                                                                                                                    \
  extern "C" int vsm_interaction_hdrf_##SFX(const vsm_quad_##SFX* q, int S, int m, const vsm_composite_##SFX* comp,    \
                                            const vsm_added_##SFX* added_surface, T* hdr_J, T* bhr_uw, T* bhr_dw,      \
                                            void* stream) {                                                            \
    int rc;                                                                                                            \
    if ((rc = check_quad(q)) || (rc = check_comp(comp)) || (rc = check_added(added_surface))) return rc;               \
    VSM_REQUIRE(S >= 0 && m >= 0 && (S == 0 || (hdr_J && (m > 0 || (bhr_uw && bhr_dw)))), "interaction_hdrf: bad argument"); \
    return interaction_hdrf<T>(cvt_quad<T>(q), S, m, cvt_comp<T>(comp), cvt_added<T>(added_surface), hdr_J, bhr_uw,    \
                               bhr_dw, as_stream(stream));                                                             \
  }                                                                                                                    \
  extern "C" int vsm_postprocess_vza_hdrf_##SFX(int N, int n_stokes, int S, int nV, const int* row0_h, const T* w_h,   \
                                                const T* hdr_J, T* hdr, void* stream) {                                \
    VSM_REQUIRE(row0_h && w_h && hdr_J && hdr, "postprocess_vza_hdrf: null argument");                                 \
    return postprocess_vza_hdrf<T>(N, n_stokes, S, nV, row0_h, w_h, hdr_J, hdr, as_stream(stream));                    \
  }                                                                                                                    \
  extern "C" int vsm_coxmunk_ss_correction_##SFX(const vsm_coxmunk_##SFX* surf, int n_stokes, int S, int nV,           \
                                                 const T* mu_v_h, const T* dphi_h, T mu0, int m_max, int nphi,         \
                                                 const T* phi, const T* wphi, const T* tau_total, T* coef, T* R_SFI,   \
                                                 void* stream) {                                                       \
    VSM_REQUIRE(surf && mu_v_h && dphi_h && phi && wphi && coef && S >= 0 && (R_SFI == nullptr || tau_total || S == 0), \
                "coxmunk_ss_correction: bad argument");                                                                \
    return coxmunk_ss_correction<T>(cvt_cm(surf), n_stokes, S, nV, mu_v_h, dphi_h, mu0, m_max, nphi, phi, wphi,        \
                                    tau_total, coef, R_SFI, as_stream(stream));                                        \
  }
VSM_SURF_API(double, f64)
VSM_SURF_API(float, f32)

// ---- per-scene layer optics --------------------------------------------------------------------------------------------
#define VSM_OPT_API(T, SFX)                                                                                            \
  extern "C" int vsm_compute_Z_moments_##SFX(const vsm_quad_##SFX* q, int m, int lmax, const double* greek, T* Zpp,    \
                                             T* Zmp, void* stream) {                                                   \
    int rc;                                                                                                            \
    if ((rc = check_quad(q))) return rc;                                                                               \
    VSM_REQUIRE(m >= 0 && lmax >= 0 && (greek || lmax == 0) && Zpp && Zmp && q->N % q->n_stokes == 0,                  \
                "compute_Z_moments: bad argument");                                                                    \
    return compute_Z_moments<T>(q->N / q->n_stokes, q->n_stokes, q->mu, m, lmax, greek, Zpp, Zmp, as_stream(stream));  \
  }                                                                                                                    \
  extern "C" int vsm_layer_optics_##SFX(int S, int L, int nAer, const double* tau_rayl, const double* tau_abs,         \
                                        double varpi_cabannes, const double* tau_aer, const double* ssa,               \
                                        const double* ftrunc, const int* mode, T* tau, T* varpi, T* tau_sum, T* fcomp, \
                                        T* max_tau_varpi, void* stream) {                                              \
    VSM_REQUIRE(S >= 0 && L >= 0 && nAer >= 0 && max_tau_varpi &&                                                      \
                    (S == 0 || (tau_rayl && tau_abs && tau && varpi && tau_sum)) &&                                    \
                    (nAer == 0 || (tau_aer && ssa && ftrunc && mode)),                                                 \
                "layer_optics: bad argument");                                                                         \
    return layer_optics<T>(S, L, nAer, tau_rayl, tau_abs, varpi_cabannes, tau_aer, ssa, ftrunc, mode, tau, varpi,      \
                           tau_sum, fcomp, max_tau_varpi, as_stream(stream));                                          \
  }                                                                                                                    \
  extern "C" int vsm_layer_dtau_##SFX(int S, int L, const int* ndoubl, const T* tau, T* dtau, void* stream) {          \
    VSM_REQUIRE(S >= 0 && L >= 0 && (S == 0 || L == 0 || (ndoubl && tau && dtau)), "layer_dtau: bad argument");       \
    return layer_dtau<T>(S, L, ndoubl, tau, dtau, as_stream(stream));                                                  \
  }                                                                                                                    \
  extern "C" int vsm_layer_optics_lin_##SFX(int S_full, int lo, int S, int L, int nAer, int nGas, int P,               \
                                            const double* tau_rayl, const double* tau_abs, double varpi_cabannes,      \
                                            const double* tau_aer, const double* ssa, const double* ftrunc,            \
                                            const double* tau_abs_dot, const double* tau_aer_dot,                      \
                                            const double* ssa_dot, const double* ftrunc_dot, const int* ndoubl,        \
                                            T* dtau_dot_all, T* varpi_dot, T* tau_sum_dot, T* fz, T* zdcoef,           \
                                            void* stream) {                                                            \
    VSM_REQUIRE(S_full >= 0 && lo >= 0 && S >= 0 && lo + S <= S_full && L >= 0 && nAer >= 0 && nGas >= 0 &&           \
                    P >= 7 * nAer + nGas && P <= 64,                                                                   \
                "layer_optics_lin: bad size (0 <= lo, lo + S <= S_full, 7 nAer + nGas <= P <= 64)");                   \
    VSM_REQUIRE(S == 0 || L == 0 || (tau_rayl && tau_abs && ndoubl && dtau_dot_all && varpi_dot && tau_sum_dot),       \
                "layer_optics_lin: null argument");                                                                    \
    VSM_REQUIRE(nGas == 0 || S == 0 || L == 0 || tau_abs_dot, "layer_optics_lin: tau_abs_dot missing");                \
    VSM_REQUIRE(nAer == 0 || S == 0 || L == 0 ||                                                                       \
                    (tau_aer && ssa && ftrunc && tau_aer_dot && ssa_dot && ftrunc_dot && fz && zdcoef),                \
                "layer_optics_lin: aerosol tables / coefficient outputs missing");                                     \
    return layer_optics_lin<T>(S_full, lo, S, L, nAer, nGas, P, tau_rayl, tau_abs, varpi_cabannes, tau_aer, ssa,       \
                               ftrunc, tau_abs_dot, tau_aer_dot, ssa_dot, ftrunc_dot, ndoubl, dtau_dot_all, varpi_dot, \
                               tau_sum_dot, nAer ? fz : nullptr, nAer ? zdcoef : nullptr, as_stream(stream));          \
  }                                                                                                                    \
  extern "C" int vsm_layer_expk_##SFX(int S, const T* dtau, T mu0, T* expk, void* stream) {                            \
    VSM_REQUIRE(S >= 0 && (S == 0 || (dtau && expk)) && mu0 > T(0), "layer_expk: bad argument");                       \
    return layer_expk<T>(S, dtau, mu0, expk, as_stream(stream));                                                       \
  }
VSM_OPT_API(double, f64)
VSM_OPT_API(float, f32)

// ---- batch_solve! ------------------------------------------------------------------------------------------------------
#define VSM_SOLVE_API(T, SFX)                                                                                          \
  extern "C" int vsm_batch_solve_##SFX(int N, int Nrhs, int S, const T* A, const T* B, T* X, T* work, int* info,       \
                                       void* stream) {                                                                 \
    VSM_REQUIRE(N > 0 && Nrhs > 0 && S >= 0, "batch_solve: bad size");                                                 \
    if (S == 0) return VSM_OK;                                                                                         \
    VSM_REQUIRE(A && B && X && work && X != B, "batch_solve: null argument / X must not alias B");                     \
    int rc = batch_inv<T>(N, S, A, work, info, as_stream(stream));                                                     \
    if (rc) return rc;                                                                                                 \
    return gemm<T>(N, Nrhs, N, S, work, (long long)N * N, B, (long long)N * Nrhs, X, (long long)N * Nrhs, T(1),        \
                   (const T*)nullptr, 0, T(0), T(0), as_stream(stream));                                               \
  }
VSM_SOLVE_API(double, f64)
VSM_SOLVE_API(float, f32)
