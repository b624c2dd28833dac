// Device body of the LDS-staged operator-level product (see k_gemm_lds in vsm_generic.hip): one workgroup of NW waves
// forms C = alpha A*B + beta D + gamma I for ONE matrix triple, 16 < M <= 16 MT, Nc <= 16 NW, any K.
#pragma once
#include "vsm_common.h"

namespace vsm {

template <int MT, int KC>
struct gemm_lds_cfg {
  static_assert(KC == 16 || KC == 32, "chunk of 16 or 32 contraction indices");
  // leading dimension MP + 64 / KC: the four kq groups of an A-fragment read (rows KC/4 apart) fall on disjoint bank sets in
  // both precisions (FP64: KC/4 * LDA * 8 B = 128 mod 256; FP32: KC/4 * LDA * 4 B = 64 mod 256)
  static constexpr int MP = 16 * MT, LDA = MP + 64 / KC, LDS_ELEMS = KC * LDA;
};

// As_lds: KC * LDA elements of LDS.  All 64 NW threads must call (contains barriers).  Wave w owns the 16-column tile w of C
// and loads, per chunk, KC/4 CONSECUTIVE k of its column of B straight into the MFMA B-operand registers (lane (j, kq) takes
// k = kc + (KC/4) kq + t; the A fragments use the same permutation of the contraction index).
template <typename T, int MT, int NW, int KC>
__device__ __forceinline__ void gemm_lds_body(int M, int Nc, int K, const T* __restrict__ Ag, const T* __restrict__ Bg, T* Cg,
                                              const T* Dg, T alpha, T beta, T gamma, T* As_lds) {
  constexpr int MP = gemm_lds_cfg<MT, KC>::MP, LDA = gemm_lds_cfg<MT, KC>::LDA, NT = 64 * NW, KQ = KC / 4;
  constexpr int AE = (MP * KC + NT - 1) / NT;   // A-chunk elements per thread
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
  const int col = 16 * wave + li;
  typename mfma<T>::acc_t acc[MT];
#pragma unroll
  for (int ta = 0; ta < MT; ++ta) acc[ta] = acc_zero<T>();
  // staging map of the A chunk: element e of this thread is (row, k) = ((tid + NT e) % MP, (tid + NT e) / MP)
  T areg[AE], breg[KQ];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int idx = tid + NT * e, row = idx % MP, k = kc + idx / MP;
      areg[e] = (idx < MP * KC && row < M && k < K) ? Ag[row + (long long)M * k] : T(0);
    }
#pragma unroll
    for (int t = 0; t < KQ; ++t) {
      const int k = kc + KQ * kq + t;
      breg[t] = (col < Nc && k < K) ? Bg[k + (long long)K * col] : T(0);
    }
  };
  load_chunk(0);
  for (int kc = 0; kc < K; kc += KC) {
    __syncthreads();   // the previous chunk's fragment reads are done
#pragma unroll
    for (int e = 0; e < AE; ++e) {
      const int idx = tid + NT * e;
      if (idx < MP * KC) As_lds[(idx / MP) * LDA + idx % MP] = areg[e];
    }
    T bcur[KQ];
#pragma unroll
    for (int t = 0; t < KQ; ++t) bcur[t] = breg[t];
    __syncthreads();
    if (kc + KC < K) load_chunk(kc + KC);   // in flight behind this chunk's MFMAs
    if (16 * wave < Nc) {                   // wave-uniform: waves without a column tile only help staging
#pragma unroll
      for (int t = 0; t < KQ; ++t) {
        T a[MT];
#pragma unroll
        for (int ta = 0; ta < MT; ++ta) a[ta] = As_lds[(KQ * kq + t) * LDA + 16 * ta + li];
#pragma unroll
        for (int ta = 0; ta < MT; ++ta) acc[ta] = mfma<T>::mma(a[ta], bcur[t], acc[ta]);
      }
    }
  }
  if (col < Nc) {
#pragma unroll
    for (int ta = 0; ta < MT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * ta + mfma<T>::crow(lane, r);
        if (row < M) {
          T v = alpha * acc[ta][r];
          if (Dg) v += beta * Dg[row + (long long)M * col];
          if (row == col) v += gamma;
          Cg[row + (long long)M * col] = v;
        }
      }
  }
}

// launch shape for (M, Nc): MT row tiles in {2, 4, 6, 8}, NW waves in {4, 6, 8}
#define VSM_GEMM_LDS_DISPATCH(M_, Nc_, CALL) \
  do {                                       \
    const int mt_ = ((M_) + 15) / 16, nt_ = ((Nc_) + 15) / 16; \
    if (mt_ <= 2) { if (nt_ <= 4) CALL(2, 4); else if (nt_ <= 6) CALL(2, 6); else CALL(2, 8); } \
    else if (mt_ <= 4) { if (nt_ <= 4) CALL(4, 4); else if (nt_ <= 6) CALL(4, 6); else CALL(4, 8); } \
    else if (mt_ <= 6) { if (nt_ <= 4) CALL(6, 4); else if (nt_ <= 6) CALL(6, 6); else CALL(6, 8); } \
    else { if (nt_ <= 4) CALL(8, 4); else if (nt_ <= 6) CALL(8, 6); else CALL(8, 8); } \
  } while (0)

}  // namespace vsm
