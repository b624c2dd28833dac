// The run object behind vsm_run_* (a CompositeLayer kept in the layer kernels' native strip layout, include/vsmartmom_hip.h) as the
// two kernel families see it: vsm_native.hip (FP64 records and arithmetic: FP64 models) and vsm_native32.hip (FP32 records and
// arithmetic: Float32 models).
#pragma once
#include <vector>

#include "vsm_internal.h"

namespace vsm {

constexpr int NSUB_MAX = VSM_MM_MAX;   // sub-problems per launch (gridDim.y)

struct nat_sub {
  int im;          // index of the Fourier moment in the run's list
  int m;
  int gsz, g[4];   // the Stokes components of the block
  unsigned uvmask;
  int n, rt, ks;
  size_t comp_off;   // elements (of the run's record type) from the workspace base
};

}  // namespace vsm

struct vsm_run {
  const void *mu, *wt;     // the quadrature arrays of the caller (element type: elem_size)
  int N, ns, i_mu0;
  double mu0;
  int elem_size;           // 8: FP64 model (FP64 records and arithmetic), 4: Float32 model (FP32 records and arithmetic)
  int S, nm;
  std::vector<int> m;
  std::vector<vsm::nat_sub> subs;
  std::vector<std::vector<int>> classes;   // indices into subs with equal (n, gsz, uvmask)
  void* ws;
  size_t ws_elems;
  std::vector<char> pure_diag;   // per sub-problem: its composite is still (R = 0, T = diag, J = 0): only diagonal steps so far
};

namespace vsm {

// ---- FP32 family (vsm_native32.hip) ------------------------------------------------------------------------------------------
constexpr int NATIVE_MAX_ROWS = 96;     // FP64 family (vsm_native.hip): blocks of up to six row tiles
constexpr int NATIVE32_MAX_ROWS = 128;   // FP32 family (vsm_native32.hip): up to eight row tiles
int native32_rt_of(int n);                 // row tiles of a block of n rows
size_t native32_comp_stride(int rt);       // floats per point of a native composite
int native32_run_layer(vsm_run* run, int ndoubl, const float* dtau, const float* varpi, const float* tau_sum, const float* F0,
                       int ncomp, const float* const* Zpp, const float* const* Zmp, long long z_stride, const float* fcomp, int toa,
                       const int* layer_coupling, int* status, hipStream_t st);
int native32_convert(vsm_run* run, int im, const composite<float>& c, bool import, hipStream_t st);

}  // namespace vsm
