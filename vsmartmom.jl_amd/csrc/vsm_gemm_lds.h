// Device body of the LDS-staged operator-level product (see k_gemm_lds in vsm_generic.hip): one workgroup of 256 threads
// forms C = alpha A*B + beta D + gamma I for ONE matrix triple, 16 < M <= 16 MT, Nc <= 64 CT, any K.
#pragma once
#include "vsm_common.h"

namespace vsm {

template <int MT>
struct gemm_lds_cfg {
  static constexpr int KC = 16, MP = 16 * MT, LDA = MP + 4;
};

// As_lds: KC * LDA elements of LDS.  All 256 threads must call (contains barriers).
template <typename T, int MT, int CT>
__device__ __forceinline__ void gemm_lds_body(int M, int Nc, int K, const T* __restrict__ Ag, const T* __restrict__ Bg, T* Cg,
                                              const T* Dg, T alpha, T beta, T gamma, T* As_lds) {
  constexpr int KC = gemm_lds_cfg<MT>::KC, MP = gemm_lds_cfg<MT>::MP, LDA = gemm_lds_cfg<MT>::LDA;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  typename mfma<T>::acc_t acc[CT][MT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ta = 0; ta < MT; ++ta) acc[ct][ta] = acc_zero<T>();
  // staging map of the A chunk: element e of this thread is (row, k) = ((tid + 256 e) % MP, (tid + 256 e) / MP)
  T areg[MT], breg[CT][4];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int e = 0; e < MT; ++e) {
      const int idx = tid + 256 * e, row = idx % MP, k = kc + idx / MP;
      areg[e] = (row < M && k < K) ? Ag[row + (long long)M * k] : T(0);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int col = 16 * (wave + 4 * ct) + li;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = kc + 4 * kq + t;
        breg[ct][t] = (col < Nc && k < K) ? Bg[k + (long long)K * col] : T(0);
      }
    }
  };
  load_chunk(0);
  for (int kc = 0; kc < K; kc += KC) {
    __syncthreads();   // the previous chunk's fragment reads are done
#pragma unroll
    for (int e = 0; e < MT; ++e) {
      const int idx = tid + 256 * e;
      As_lds[(idx / MP) * LDA + idx % MP] = areg[e];
    }
    T bcur[CT][4];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int t = 0; t < 4; ++t) bcur[ct][t] = breg[ct][t];
    __syncthreads();
    if (kc + KC < K) load_chunk(kc + KC);   // in flight behind this chunk's MFMAs
    if (16 * wave < Nc) {                   // wave-uniform: waves without a column tile only help staging
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        T a[MT];
#pragma unroll
        for (int ta = 0; ta < MT; ++ta) a[ta] = As_lds[(4 * kq + t) * LDA + 16 * ta + li];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int ta = 0; ta < MT; ++ta) acc[ct][ta] = mfma<T>::mma(a[ta], bcur[ct][t], acc[ct][ta]);
      }
    }
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = 16 * (wave + 4 * ct) + li;
    if (col < Nc) {
#pragma unroll
      for (int ta = 0; ta < MT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * ta + mfma<T>::crow(lane, r);
          if (row < M) {
            T v = alpha * acc[ct][ta][r];
            if (Dg) v += beta * Dg[row + (long long)M * col];
            if (row == col) v += gamma;
            Cg[row + (long long)M * col] = v;
          }
        }
    }
  }
}

}  // namespace vsm
