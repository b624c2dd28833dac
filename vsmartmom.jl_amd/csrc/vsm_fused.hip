// Fused, LDS-resident kernels: ONE workgroup owns ONE spectral point and keeps its N x N
// operators on-chip for a whole layer step.
//
//   k_elemental_doubling : elemental! + ndoubl x doubling step + apply_D!   (1 launch / layer)
//   k_interaction11      : interaction_helper!(::ScatteringInterface_11)      (1 launch / layer)
//
// LDS layout: column-major NP x NP (NP = N rounded up to 32), row index XOR-swizzled by column so
// that both MFMA operand fetch patterns (A: 16 rows x 2 k-columns per half-wave, B: 2 k-rows x 16
// columns per half-wave) are bank-conflict free for single ds_read_b64/b32.
// MFMA: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32.  Waves form a WR x WC grid, each wave owns
// a TMR x TMC block of 16x16 accumulator tiles:
//      NP = 32 : 4 waves (2x2), 1x1 tiles        NP = 64 : 8 waves (4x2), 1x2 tiles
//      NP = 96 : 4 waves (2x2), 3x3 tiles (f32 only; 4 N x N buffers must fit 160 KB of LDS)
// Two waves per SIMD (NP = 64) let one wave's LDS latency / epilogue hide behind the other's MFMAs;
// measured on MI355X: 70.3 TF/s (1 wave/SIMD) vs 77.0 TF/s (2 waves/SIMD) for back-to-back f64 MFMA.
//
// The inverse (I - E)^-1 that the reference takes from batched getrf/getri is produced on-chip either
// by a Neumann series built with repeated squaring (orders up to 31; truncation error bounded below
// rounding by a norm bound on E), or by the pivoted Gauss-Jordan of vsm_inverse.h (general case).
#include "vsm_internal.h"
#include "vsm_inverse.h"
#include <stdlib.h>

#include "vsm_lds.h"

namespace vsm {

// Optional phase timing (build with -DVSM_PHASE_TIMING; tools/phase_timing.py): workgroup 0 / thread 0
// accumulates s_memtime deltas per phase into a device symbol.  Compiled out of the product build.
#ifdef VSM_PHASE_TIMING
__device__ unsigned long long vsm_phase_cycles[32];
#define VSM_STAMP_DECL unsigned long long _t_prev = __builtin_readcyclecounter()
#define VSM_STAMP(i)                                                     \
  do {                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                           \
      const unsigned long long _t = __builtin_readcyclecounter();        \
      vsm_phase_cycles[i] += _t - _t_prev;                               \
      _t_prev = _t;                                                      \
    }                                                                    \
  } while (0)
#else
#define VSM_STAMP_DECL
#define VSM_STAMP(i)
#endif

template <int NP, int NW_>
struct fcfg {
  static_assert(NP == 32 || NP == 64 || NP == 96, "NP must be 32, 64 or 96");
  static_assert(NW_ == 4 || (NW_ == 8 && NP == 64) || (NW_ == 6 && NP == 96), "4 waves, 8 for NP = 64, or 6 for NP = 96");
  static constexpr int NW = NW_;                      // waves per workgroup
  static constexpr int NT = 64 * NW;                  // threads
  static constexpr int WC = 2;                        // wave grid columns
  static constexpr int WR = NW / WC;                  // wave grid rows
  static constexpr int TMR = NP / 16 / WR;            // tiles per wave, rows
  static constexpr int TMC = NP / 16 / WC;            // tiles per wave, cols
  static constexpr int KS = NP / 4;                   // max k-steps
  static constexpr int TPR = (NT >= 512) ? 4 : 2;     // threads per row in the mat-vec (power of two; NT / TPR rows >= NP)
  static_assert(TMR * WR * 16 == NP && TMC * WC * 16 == NP, "tile grid must cover NP");
};

template <typename T, int NP, int NW>
struct fsmem {
  T L[4][NP * NP];
  T vec[10][NP];
  float red[2][8];
  int flag[2];
  gj_scratch<T, NP> gj;
};

template <typename T, int NP, int NW>
struct acc_block {
  using C = fcfg<NP, NW>;
  typename mfma<T>::acc_t v[C::TMR][C::TMC];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < C::TMR; ++a)
#pragma unroll
      for (int b = 0; b < C::TMC; ++b) v[a][b] = acc_zero<T>();
  }
};

template <int NP, int NW>
struct wave_pos {
  int lane, l15, kq, rowA, colB;  // rowA/colB: first row / column this lane touches in tile 0 of its wave
  __device__ __forceinline__ wave_pos() {
    using C = fcfg<NP, NW>;
    lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    rowA = 16 * ((wave / C::WC) * C::TMR) + l15;
    colB = 16 * ((wave % C::WC) * C::TMC) + l15;
  }
};

// ---- MFMA k-loops, software-pipelined by hand: the fragments of k-step (k0+4) are requested before
// the MFMAs of k-step k0 issue (hipcc does not pipeline a runtime-trip-count loop). -------------------
template <typename T, int NP, int NW>
struct frag_set {
  using C = fcfg<NP, NW>;
  T a[C::TMR], b[C::TMC];
  __device__ __forceinline__ void load(const T* A, const T* B, int k, const wave_pos<NP, NW>& w) {
#pragma unroll
    for (int t = 0; t < C::TMR; ++t) a[t] = A[lidx<NP>(w.rowA + 16 * t, k)];
#pragma unroll
    for (int t = 0; t < C::TMC; ++t) b[t] = B[lidx<NP>(k, w.colB + 16 * t)];
  }
  __device__ __forceinline__ void mma(acc_block<T, NP, NW>& acc) const {
#pragma unroll
    for (int x = 0; x < C::TMR; ++x)
#pragma unroll
      for (int y = 0; y < C::TMC; ++y) acc.v[x][y] = mfma<T>::mma(a[x], b[y], acc.v[x][y]);
  }
};

// acc += A * B with A, B in (swizzled) LDS.  Kend: multiple of 4 covering N.
template <typename T, int NP, int NW>
__device__ __forceinline__ void mm_ll(acc_block<T, NP, NW>& acc, const T* A, const T* B, int Kend) {
  const wave_pos<NP, NW> w;
  frag_set<T, NP, NW> f0, f1;
  f0.load(A, B, w.kq, w);
  int k0 = 0;
  for (; k0 + 8 <= Kend; k0 += 8) {
    f1.load(A, B, k0 + 4 + w.kq, w);
    f0.mma(acc);
    if (k0 + 8 < Kend) f0.load(A, B, k0 + 8 + w.kq, w);
    f1.mma(acc);
  }
  if (k0 < Kend) f0.mma(acc);  // odd number of k-steps
}

// two products sharing the B operand: acc1 += A1*B, acc2 += A2*B
template <typename T, int NP, int NW>
__device__ __forceinline__ void mm_ll2(acc_block<T, NP, NW>& acc1, acc_block<T, NP, NW>& acc2, const T* A1, const T* A2,
                                       const T* B, int Kend) {
  using C = fcfg<NP, NW>;
  const wave_pos<NP, NW> w;
  T a1[2][C::TMR], a2[2][C::TMR], bf[2][C::TMC];
  auto load = [&](int buf, int k) {
#pragma unroll
    for (int t = 0; t < C::TMR; ++t) {
      const int ia = lidx<NP>(w.rowA + 16 * t, k);
      a1[buf][t] = A1[ia];
      a2[buf][t] = A2[ia];
    }
#pragma unroll
    for (int t = 0; t < C::TMC; ++t) bf[buf][t] = B[lidx<NP>(k, w.colB + 16 * t)];
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int x = 0; x < C::TMR; ++x)
#pragma unroll
      for (int y = 0; y < C::TMC; ++y) {
        acc1.v[x][y] = mfma<T>::mma(a1[buf][x], bf[buf][y], acc1.v[x][y]);
        acc2.v[x][y] = mfma<T>::mma(a2[buf][x], bf[buf][y], acc2.v[x][y]);
      }
  };
  load(0, w.kq);
  int k0 = 0;
  for (; k0 + 8 <= Kend; k0 += 8) {
    load(1, k0 + 4 + w.kq);
    mma(0);
    if (k0 + 8 < Kend) load(0, k0 + 8 + w.kq);
    mma(1);
  }
  if (k0 < Kend) mma(0);
}

// A read straight from global memory (column-major N x N), B in LDS.  The A fragments of the whole
// k-range are requested up front so the L2/HBM latency is paid once and overlaps whatever the caller
// does between prefetch() and run().
template <typename T, int NP, int NW>
struct gl_operand_a {
  using C = fcfg<NP, NW>;
  T a[C::KS][C::TMR];
  __device__ __forceinline__ void prefetch(const T* __restrict__ Ag, int N) {
    const wave_pos<NP, NW> w;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int t = 0; t < C::TMR; ++t) {
        const int row = w.rowA + 16 * t, k = 4 * ks + w.kq;
        a[ks][t] = (row < N && k < N) ? Ag[row + (long long)N * k] : T(0);
      }
  }
  __device__ __forceinline__ void run(acc_block<T, NP, NW>& acc, const T* B) const {  // acc += A * B
    const wave_pos<NP, NW> w;
    T bf[2][C::TMC];
#pragma unroll
    for (int t = 0; t < C::TMC; ++t) bf[0][t] = B[lidx<NP>(w.kq, w.colB + 16 * t)];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      if (ks + 1 < C::KS) {
#pragma unroll
        for (int t = 0; t < C::TMC; ++t) bf[(ks + 1) & 1][t] = B[lidx<NP>(4 * (ks + 1) + w.kq, w.colB + 16 * t)];
      }
#pragma unroll
      for (int x = 0; x < C::TMR; ++x)
#pragma unroll
        for (int y = 0; y < C::TMC; ++y) acc.v[x][y] = mfma<T>::mma(a[ks][x], bf[ks & 1][y], acc.v[x][y]);
    }
  }
};
// B read straight from global memory, A in LDS.
template <typename T, int NP, int NW>
struct gl_operand_b {
  using C = fcfg<NP, NW>;
  T b[C::KS][C::TMC];
  __device__ __forceinline__ void prefetch(const T* __restrict__ Bg, int N) {
    const wave_pos<NP, NW> w;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int t = 0; t < C::TMC; ++t) {
        const int col = w.colB + 16 * t, k = 4 * ks + w.kq;
        b[ks][t] = (col < N && k < N) ? Bg[k + (long long)N * col] : T(0);
      }
  }
  // B = D Bg D (t-- from t++): sign (+) when row k and column have the same U/V parity
  __device__ __forceinline__ void prefetch_dsym(const T* __restrict__ Bg, int N, int ns) {
    const wave_pos<NP, NW> w;
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks)
#pragma unroll
      for (int t = 0; t < C::TMC; ++t) {
        const int col = w.colB + 16 * t, k = 4 * ks + w.kq;
        const T v = (col < N && k < N) ? Bg[k + (long long)N * col] : T(0);
        b[ks][t] = (is_uv_row(k, ns) == is_uv_row(col, ns)) ? v : -v;
      }
  }
  __device__ __forceinline__ void run(acc_block<T, NP, NW>& acc, const T* A) const {  // acc += A * B
    const wave_pos<NP, NW> w;
    T af[2][C::TMR];
#pragma unroll
    for (int t = 0; t < C::TMR; ++t) af[0][t] = A[lidx<NP>(w.rowA + 16 * t, w.kq)];
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
      if (ks + 1 < C::KS) {
#pragma unroll
        for (int t = 0; t < C::TMR; ++t) af[(ks + 1) & 1][t] = A[lidx<NP>(w.rowA + 16 * t, 4 * (ks + 1) + w.kq)];
      }
#pragma unroll
      for (int x = 0; x < C::TMR; ++x)
#pragma unroll
        for (int y = 0; y < C::TMC; ++y) acc.v[x][y] = mfma<T>::mma(af[ks & 1][x], b[ks][y], acc.v[x][y]);
    }
  }
};

// dst(row,col) = f(acc(row,col), row, col, old) for every accumulator element of this wave.
// The swizzled element offsets are loop invariant: built once per kernel (acc_map) and reused by every
// store of the layer step.
template <int NP, int NW>
struct acc_map {
  using C = fcfg<NP, NW>;
  int ix[C::TMR][C::TMC][4];
  int r0, c0, lane;
  __device__ __forceinline__ acc_map() {
    lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    r0 = 16 * ((wave / C::WC) * C::TMR);
    c0 = 16 * ((wave % C::WC) * C::TMC) + (lane & 15);
  }
  template <typename T>
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int a = 0; a < C::TMR; ++a)
#pragma unroll
      for (int b = 0; b < C::TMC; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) ix[a][b][r] = lidx<NP>(r0 + 16 * a + mfma<T>::crow(lane, r), c0 + 16 * b);
  }
};
template <typename T, int NP, int NW, typename F>
__device__ __forceinline__ void acc_store(T* dst, const acc_block<T, NP, NW>& acc, F f) {
  using C = fcfg<NP, NW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = 16 * ((wave / C::WC) * C::TMR), c0 = 16 * ((wave % C::WC) * C::TMC) + (lane & 15);
#pragma unroll
  for (int a = 0; a < C::TMR; ++a)
#pragma unroll
    for (int b = 0; b < C::TMC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * a + mfma<T>::crow(lane, r);
        const int col = c0 + 16 * b;
        const int ix = lidx<NP>(row, col);
        dst[ix] = f(acc.v[a][b][r], row, col, dst[ix]);
      }
}
template <typename T, int NP, int NW, typename F>
__device__ __forceinline__ void acc_store(T* dst, const acc_block<T, NP, NW>& acc, const acc_map<NP, NW>& m, F f) {
  using C = fcfg<NP, NW>;
#pragma unroll
  for (int a = 0; a < C::TMR; ++a)
#pragma unroll
    for (int b = 0; b < C::TMC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m.r0 + 16 * a + mfma<T>::crow(m.lane, r);
        const int col = m.c0 + 16 * b;
        const int ix = m.ix[a][b][r];
        dst[ix] = f(acc.v[a][b][r], row, col, dst[ix]);
      }
}

// global (column-major N x N) -> swizzled LDS, split in two halves so the global latency overlaps
// other work: load() issues the reads into registers, store() writes the LDS image (zero padded).
template <typename T, int NP, int NW>
struct stage_regs {
  using C = fcfg<NP, NW>;
  static constexpr int CNT = NP * NP / C::NT;
  T v[CNT];
  __device__ __forceinline__ void load(const T* __restrict__ src, int N) {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + C::NT * c;
      const int i = e % NP, j = e / NP;
      v[c] = (i < N && j < N) ? src[i + (long long)N * j] : T(0);
    }
  }
  __device__ __forceinline__ void store(T* dst) const {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + C::NT * c;
      dst[lidx<NP>(e % NP, e / NP)] = v[c];
    }
  }
};
// N x N block of global memory held in registers in flat (coalesced) order: element e = tid + NT*c.
template <typename T, int NP, int NW>
struct flat_regs {
  using C = fcfg<NP, NW>;
  static constexpr int CNT = NP * NP / C::NT;
  T v[CNT];
  __device__ __forceinline__ void load(const T* __restrict__ src, int N) {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + C::NT * c;
      v[c] = (e < N * N) ? src[e] : T(0);
    }
  }
  __device__ __forceinline__ void load_dsym(const T* __restrict__ src, int N, int ns) {  // D src D
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + C::NT * c;
      const T x = (e < N * N) ? src[e] : T(0);
      v[c] = (is_uv_row(e % N, ns) == is_uv_row(e / N, ns)) ? x : -x;
    }
  }
  // dst[e] = v[e] + L(e)   (L swizzled LDS)
  __device__ __forceinline__ void add_lds_store(T* __restrict__ dst, const T* L, int N) const {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + C::NT * c;
      if (e < N * N) dst[e] = v[c] + L[lidx<NP>(e % N, e / N)];
    }
  }
};
template <typename T, int NP, int NW>
__device__ __forceinline__ void stage(T* dst, const T* __restrict__ src, int N) {
  stage_regs<T, NP, NW> s;
  s.load(src, N);
  s.store(dst);
}
template <typename T, int NP, int NW>
__device__ __forceinline__ void lds_to_global(T* __restrict__ dst, const T* L, int N) {
  for (int e = threadIdx.x; e < N * N; e += fcfg<NP, NW>::NT) dst[e] = L[lidx<NP>(e % N, e / N)];
}

// ---- norm of the N x N block held in the accumulators -----------------------------------------------
// ||E||_2 <= ||E||_F.  Deterministic reduction: per-lane sum of squares (in T), rounded UP to float,
// DPP row-rotation adds inside a wave, one LDS slot per wave, one barrier, fixed-order final sum; the
// result is inflated by 1e-3 to cover the float rounding of the <= 80 additions.  `slot` alternates
// between calls so no extra barrier is needed before the slots are reused.
template <typename T, int NP, int NW>
__device__ __forceinline__ T acc_norm_bound(const acc_block<T, NP, NW>& acc, int N, fsmem<T, NP, NW>& sm, int& slot) {
  using C = fcfg<NP, NW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = 16 * ((wave / C::WC) * C::TMR), c0 = 16 * ((wave % C::WC) * C::TMC) + (lane & 15);
  T ss = 0;
#pragma unroll
  for (int a = 0; a < C::TMR; ++a)
#pragma unroll
    for (int b = 0; b < C::TMC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * a + mfma<T>::crow(lane, r), col = c0 + 16 * b;
        const T v = acc.v[a][b][r];
        if (row < N && col < N) ss += v * v;
      }
  const float ws = wave_sum(to_float_up(ss));
  if (lane == 0) sm.red[slot][wave] = ws;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tot += sm.red[slot][w];
  slot ^= 1;
  // NaN/Inf propagate: every comparison in the caller then fails and the general path is taken
  return T(sqrtf(tot) * 1.001f);
}

// In-place pivoted Gauss-Jordan of the swizzled LDS matrix V (only the N x N block is used).
template <typename T, int NP, int NW>
__device__ __noinline__ void gj_lds(T* V, int N, gj_scratch<T, NP>* sc) {
  using G = gj_cfg<NP, fcfg<NP, NW>::NT>;
  const int tr = threadIdx.x % G::TR, tc = threadIdx.x / G::TR;
  T g[G::RB][G::CB];
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      g[rb][cb] = (i < N && j < N) ? V[lidx<NP>(i, j)] : ((i == j) ? T(1) : T(0));
    }
  gj_invert<T, NP, fcfg<NP, NW>::NT>(g, N, *sc);
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      if (i < N && j < N) V[lidx<NP>(i, sc->dst[j])] = g[rb][cb];
    }
  __syncthreads();
}

// G = (I - E)^-1 with E given in the accumulators.  Result -> V.  W is scratch.
//
// Series path: G = sum_{k<=K} E^k, built by repeated squaring
//     (I+E)(I+E^2)(I+E^4)... = sum_{k < 2^p} E^k          (2 MFMA products per doubling of the order)
// optionally closed with "+ E^(2^p)" (1 product).  Orders K = 1,2,3,4,7,8,15,16,31 cost
// 0,1,2,3,4,5,6,7,8 products.  K is the smallest order whose truncation error bound
// b^(K+1)/(1-b), b >= ||E||_2, stays below eps/4: the result is the inverse to working precision,
// the same contract as the LU-based inverse of the reference.  b >= 0.3: pivoted Gauss-Jordan.
// mode 0 = automatic, 1 = force Gauss-Jordan, 2 = force series (order 31 if the bound fails).
// Returns 1 for Gauss-Jordan, 1+K for a series of order K.  Ends with a barrier; on exit every
// thread may read V.  Precondition: nobody is still reading V or W.
template <typename T, int NP, int NW>
__device__ __forceinline__ int invert_one_minus(acc_block<T, NP, NW>& acc, T* V, T* W, int N, int Kend,
                                                fsmem<T, NP, NW>& sm, int& slot, int mode,
                                                const acc_map<NP, NW>* mp = nullptr) {
  acc_map<NP, NW> local_map;
  if (!mp) {
    local_map.template init<T>();
    mp = &local_map;
  }
  const acc_map<NP, NW>& amap = *mp;
  const T nrm = acc_norm_bound<T, NP, NW>(acc, N, sm, slot);
  const T tol = num<T>::eps() * T(0.25);
  int K = 0;
  if (nrm < T(0.3)) {
    const T lim = tol * (T(1) - nrm);
    const T n2 = nrm * nrm, n4 = n2 * n2, n8 = n4 * n4, n16 = n8 * n8;
    if (n2 <= lim) K = 1;
    else if (n2 * nrm <= lim) K = 2;
    else if (n4 <= lim) K = 3;
    else if (n4 * nrm <= lim) K = 4;
    else if (n8 <= lim) K = 7;
    else if (n8 * nrm <= lim) K = 8;
    else if (n16 <= lim) K = 15;
    else if (n16 * nrm <= lim) K = 16;
    else if (n16 * n16 <= lim) K = 31;
  }
  if (mode == 1) K = 0;
  if (mode == 2 && K == 0) K = 31;
  auto keep = [N](T a, int r, int c) { return (r < N && c < N) ? a : T(0); };
  if (K > 0) {
    acc_store<T, NP, NW>(V, acc, amap, [=](T a, int r, int c, T) { return (r == c) ? keep(a, r, c) + T(1) : keep(a, r, c); });
    if (K == 1) {
      __syncthreads();
      return 2;
    }
    acc_store<T, NP, NW>(W, acc, amap, [=](T a, int r, int c, T) { return keep(a, r, c); });
    __syncthreads();
    int cur = 1;  // W = E^cur, V = sum_{k < 2 cur} E^k
    for (;;) {
      acc.zero();
      mm_ll<T, NP, NW>(acc, W, W, Kend);  // E^(2 cur)
      cur *= 2;
      if (K == cur) {  // close with "+ E^cur"
        acc_store<T, NP, NW>(V, acc, amap, [](T a, int, int, T old) { return old + a; });
        __syncthreads();
        break;
      }
      __syncthreads();
      acc_store<T, NP, NW>(W, acc, amap, [](T a, int, int, T) { return a; });
      __syncthreads();
      acc.zero();
      mm_ll<T, NP, NW>(acc, V, W, Kend);  // V * E^cur
      __syncthreads();
      acc_store<T, NP, NW>(V, acc, amap, [](T a, int, int, T old) { return old + a; });
      __syncthreads();
      if (K == 2 * cur - 1) break;
    }
    return 1 + K;
  }
  acc_store<T, NP, NW>(V, acc, amap, [=](T a, int r, int c, T) { return (r == c) ? T(1) - keep(a, r, c) : -keep(a, r, c); });
  __syncthreads();
  gj_lds<T, NP, NW>(V, N, &sm.gj);
  return 1;
}

// y1 = M*x1, y2 = M*x2: TPR consecutive lanes share a row; each lane walks a statically unrolled
// strided column range (all LDS reads in flight at once), then a shuffle reduction.  All lanes of a
// row group receive the sums.  M's and x's padding must be zero.
template <typename T, int NP, int NW>
__device__ __forceinline__ void matvec2(const T* M, const T* x1, const T* x2, T& y1, T& y2) {
  constexpr int TPR = fcfg<NP, NW>::TPR;
  constexpr int CNT = NP / TPR;
  const int row = threadIdx.x / TPR, q = threadIdx.x % TPR;
  T s1 = 0, s2 = 0;
  if (row < NP) {
    T mv[CNT], v1[CNT], v2[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int j = q + TPR * c;
      mv[c] = M[lidx<NP>(row, j)];
      v1[c] = x1[j];
      v2[c] = x2[j];
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      s1 += mv[c] * v1[c];
      s2 += mv[c] * v2[c];
    }
  }
#pragma unroll
  for (int off = 1; off < TPR; off <<= 1) {
    s1 += __shfl_xor(s1, off);
    s2 += __shfl_xor(s2, off);
  }
  y1 = s1;
  y2 = s2;
}

// ---------------------------------------------------------------------------
// elemental! + doubling! + apply_D!
// ---------------------------------------------------------------------------
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_elemental_doubling(
    quad<T> q, int m, int ndoubl, const T* __restrict__ dtau, const T* __restrict__ varpi,
    const T* __restrict__ tau_sum, const T* __restrict__ F0, const T* __restrict__ Zpp, const T* __restrict__ Zmp,
    long long zs, added<T> out) {
  using C = fcfg<NP, NW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  T* R = sm.L[0];
  T* Tm = sm.L[1];
  T* W = sm.L[2];
  T* V = sm.L[3];
  T* jp = sm.vec[0];
  T* jm = sm.vec[1];
  T* j1p = sm.vec[2];
  T* j1m = sm.vec[3];
  T* uu = sm.vec[4];   // matvec path: u ;  spare-column path: tmp*j0+
  T* vv = sm.vec[5];   //              v ;                      tmp*j1-
  T* mus = sm.vec[6];
  T* wcs = sm.vec[7];
  T* xa = sm.vec[8];   // spare-column path: tt*j0+
  T* xb = sm.vec[9];   //                    tt*j1-

  VSM_STAMP_DECL;
  const int s = blockIdx.x;
  const int N = q.N, ns = q.n_stokes;
  const int tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  // Two free padded columns (>= Kend, never read by any k-loop) carry j0+ and j1- through the last
  // two products of the step, so the source update costs no extra pass (rt_helpers.jl:128-134).
  const bool spare = (Kend + 2 <= NP);
  const int c1 = Kend, c2 = Kend + 1;
  const T d = dtau[s], w = varpi[s];
  const T* Zp = Zpp + (long long)s * zs;
  const T* Zm = Zmp + (long long)s * zs;

  if (tid < NP) {
    mus[tid] = (tid < N) ? q.mu[tid] : T(1);
    const T wt = (tid < N) ? q.wt[tid] : T(0);
    wcs[tid] = (m == 0) ? wt / T(2) : wt / T(4);
  }
  __syncthreads();

  // ---- elemental (elemental.jl:289-334) -------------------------------------
  // Per-stream factors once per workgroup: x_i = dtau/mu_i, e_i = exp(-x_i), em_i = expm1(-x_i).
  // Then   -expm1(-(x_i+x_j)) = -(em_i + em_j + em_i em_j)        (exact identity, no cancellation)
  //        exp(-x_i)-exp(-x_j) = em_i - em_j  when the arguments are small and well separated
  //                              (relative error <= 16 eps), else the reference's expdiff_neg.
  T* xs = j1p;   // these three vectors are not used before the doubling loop
  T* es = j1m;
  T* ems = uu;
  if (tid < NP) {
    const T x = d / mus[tid];
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
  }
  __syncthreads();
  {
    constexpr int CNT = NP * NP / C::NT;
    T zp[CNT], zm[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {  // raw loads from clamped addresses: all in flight together
      const int e = tid + C::NT * c;
      const long long zo = min(e % NP, N - 1) + (long long)N * min(e / NP, N - 1);
      zp[c] = Zp[zo];
      zm[c] = Zm[zo];
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = tid + C::NT * c;
      const int i = e % NP, j = e / NP;
      T r = T(0), t = T(0);
      if (i < N && j < N) {
        const T mi = mus[i], mj = mus[j], wct = wcs[j];
        const T xi = xs[i], xj = xs[j];
        if (wct > num<T>::eps()) {
          const T emi = ems[i], emj = ems[j];
          r = w * zm[c] * (mj / (mi + mj)) * wct * (-(emi + emj + emi * emj));
          if (mi == mj) {
            if (i == j)
              t = es[i] * (T(1) + w * zp[c] * xi * wct);
            else
              t = es[j] * (w * zp[c] * xi * wct);
          } else {
            const T xm = fmax(xi, xj);
            const T ediff = (xm < T(0.5) && fabs(xi - xj) > T(0.125) * xm) ? (emi - emj) : expdiff_neg<T>(xi, xj);
            t = w * zp[c] * (mj / (mi - mj)) * wct * ediff;
          }
        } else {
          t = (i == j) ? es[i] : T(0);
        }
        if (ndoubl >= 1 && is_uv_row(i, ns)) r = -r;  // starred R* = D R (apply_D_elemental!, elemental.jl:403-422)
      }
      const int ix = lidx<NP>(i, j);
      R[ix] = r;
      Tm[ix] = t;
    }
  }
  // ---- SFI source (elemental.jl:348-392) ---------------------------------------
  if (tid < NP) {
    T vjp = T(0), vjm = T(0);
    if (tid < N) {
      const int i = tid;
      const int i_start = ns * q.i_mu0;
      const T wct02 = (m == 0) ? T(0.5) : T(0.25);
      T zp = 0, zm = 0;
      for (int qq = 0; qq < ns; ++qq) {
        const long long zo = i + (long long)N * (i_start + qq);
        const T f = F0[qq + (long long)ns * s];
        zp += Zp[zo] * f;
        zm += Zm[zo] * f;
      }
      const T mi = mus[i], ms = mus[i_start];
      if (i >= i_start && i < i_start + ns)
        vjp = wct02 * w * zp * (d / mi) * exp(-d / mi);
      else
        vjp = wct02 * w * zp * (ms / (mi - ms)) * expdiff_neg<T>(d / mi, d / ms);
      vjm = wct02 * w * zm * (ms / (mi + ms)) * (-expm1(-d * ((T(1) / mi) + (T(1) / ms))));
      const T att = exp(-tau_sum[s] / ms);
      vjp *= att;
      vjm *= att;
      if (ndoubl >= 1 && is_uv_row(i, ns)) vjm = -vjm;
    }
    jp[tid] = vjp;
    jm[tid] = vjm;
  }
  __syncthreads();

  // ---- doubling (rt_helpers.jl:102-166) -----------------------------------------
  T expk = exp(-d / q.mu0);
  int slot = 0;
  acc_block<T, NP, NW> acc, acc2;
  acc_map<NP, NW> amap;
  amap.template init<T>();
  VSM_STAMP(0);  // elemental
  for (int n = 0; n < ndoubl; ++n) {
    // G = (I - r r)^-1  -> V
    acc.zero();
    mm_ll<T, NP, NW>(acc, R, R, Kend);
    VSM_STAMP(1);  // r*r
    invert_one_minus<T, NP, NW>(acc, V, W, N, Kend, sm, slot, 0, &amap);
    VSM_STAMP(2);  // inverse
    // tt = t G -> W   (W is free: its readers finished before the barrier that ended the inverse)
    acc.zero();
    mm_ll<T, NP, NW>(acc, Tm, V, Kend);
    acc_store<T, NP, NW>(W, acc, amap, [](T a, int, int, T) { return a; });
    if (tid < NP) {
      const T a1 = jp[tid] * expk, a2 = jm[tid] * expk;
      j1p[tid] = a1;
      j1m[tid] = a2;
      if (spare) {  // columns c1/c2 of t are outside every k-range: safe to write while t is being read
        Tm[lidx<NP>(tid, c1)] = jp[tid];
        Tm[lidx<NP>(tid, c2)] = a2;
      }
    }
    __syncthreads();  // tt complete; V (=G) no longer read
    VSM_STAMP(3);  // tt = t G + store + barrier
    if (!spare) {
      // sources by mat-vec: u = j1- + r j0+ ; v = j0+ + r j1-
      T y1, y2;
      matvec2<T, NP, NW>(R, jp, j1m, y1, y2);
      const int row = tid / C::TPR;
      if (row < NP && (tid % C::TPR) == 0) {
        uu[row] = j1m[row] + y1;
        vv[row] = jp[row] + y2;
      }
      __syncthreads();
      matvec2<T, NP, NW>(W, uu, vv, y1, y2);
      if (row < NP && (tid % C::TPR) == 0) {
        jm[row] = jm[row] + y1;   // j0- <- j0- + tt (j1- + r j0+)
        jp[row] = j1p[row] + y2;  // j0+ <- j1+ + tt (j0+ + r j1-)
      }
    }
    // tmp = tt r -> V
    acc.zero();
    mm_ll<T, NP, NW>(acc, W, R, Kend);
    acc_store<T, NP, NW>(V, acc, amap, [](T a, int, int, T) { return a; });
    __syncthreads();
    VSM_STAMP(4);  // (matvec +) tmp = tt r + store + barrier
    // r <- r + tmp t ; t <- tt t   (+ columns c1,c2: tmp*j0+, tmp*j1-, tt*j0+, tt*j1-)
    acc.zero();
    acc2.zero();
    mm_ll2<T, NP, NW>(acc, acc2, V, W, Tm, Kend);
    VSM_STAMP(5);  // two products
    // r is not an operand of this product: update it right away
    acc_store<T, NP, NW>(R, acc, amap, [=](T a, int r, int c, T old) {
      if (spare && c == c1) uu[r] = a;
      if (spare && c == c2) vv[r] = a;
      return (c < N) ? old + a : T(0);
    });
    __syncthreads();  // everybody finished reading t
    acc_store<T, NP, NW>(Tm, acc2, amap, [=](T a, int r, int c, T) {
      if (spare && c == c1) xa[r] = a;
      if (spare && c == c2) xb[r] = a;
      return (c < N) ? a : T(0);
    });
    expk = expk * expk;
    __syncthreads();
    if (spare && tid < NP) {
      const T njm = jm[tid] + xb[tid] + uu[tid];   // j0- + tt j1- + (tt r) j0+
      const T njp = j1p[tid] + xa[tid] + vv[tid];  // j1+ + tt j0+ + (tt r) j1-
      jm[tid] = (tid < N) ? njm : T(0);
      jp[tid] = (tid < N) ? njp : T(0);
    }
    // (jp/jm are next read after the barriers of the following inverse)
    VSM_STAMP(6);  // stores + 2 barriers + source combine
  }
  __syncthreads();

  // ---- apply_D (doubling.jl:178-252) + write the added layer -------------------------
  T* g_rmp = out.r_mp + (long long)s * out.mat_stride;
  T* g_tpp = out.t_pp + (long long)s * out.mat_stride;
  T* g_rpm = out.r_pm + (long long)s * out.mat_stride;
  T* g_tmm = out.t_mm + (long long)s * out.mat_stride;
  for (int e = tid; e < N * N; e += C::NT) {
    const int i = e % N, j = e / N;
    const int ix = lidx<NP>(i, j);
    T r = R[ix];
    const T t = Tm[ix];
    const bool ui = is_uv_row(i, ns), uj = is_uv_row(j, ns);
    if (ndoubl >= 1 && ui) r = -r;
    g_rmp[e] = r;
    g_tpp[e] = t;
    if (!out.d_symmetric) {
      g_rpm[e] = (ui == uj) ? r : -r;
      g_tmm[e] = (ui == uj) ? t : -t;
    }
  }
  if (tid < N) {
    T vjm = jm[tid];
    if (ndoubl >= 1 && is_uv_row(tid, ns)) vjm = -vjm;
    out.j0_p[(long long)s * N + tid] = jp[tid];
    out.j0_m[(long long)s * N + tid] = vjm;
  }
  VSM_STAMP(7);  // write-out
}

// ---------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11)  (interaction.jl:207-266)
// ---------------------------------------------------------------------------
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_interaction11(int N, composite<T> c, added<T> a) {
  using C = fcfg<NP, NW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  T* L1 = sm.L[0];  // R+-  (resident)
  T* L2 = sm.L[1];  // r-+  (resident)
  T* L3 = sm.L[2];
  T* L4 = sm.L[3];
  T* vJp = sm.vec[0];
  T* vJm = sm.vec[1];
  T* vjp = sm.vec[2];
  T* vjm = sm.vec[3];
  T* vu = sm.vec[4];
  T* vz = sm.vec[5];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  const long long NN = (long long)N * N;
  T* R_mp = c.R_mp + s * NN;
  T* R_pm = c.R_pm + s * NN;
  T* T_pp = c.T_pp + s * NN;
  T* T_mm = c.T_mm + s * NN;
  T* J0_p = c.J0_p + (long long)s * N;
  T* J0_m = c.J0_m + (long long)s * N;
  const T* r_mp = a.r_mp + s * a.mat_stride;
  const T* r_pm = a.r_pm + s * a.mat_stride;
  const T* t_pp = a.t_pp + s * a.mat_stride;
  const T* t_mm = a.t_mm + s * a.mat_stride;
  const T* j0_p = a.j0_p + (long long)s * N;
  const T* j0_m = a.j0_m + (long long)s * N;
  const int mrow = tid / C::TPR;
  const bool mlead = (mrow < NP) && (tid % C::TPR == 0);
  int slot = 0;

  VSM_STAMP_DECL;
  acc_block<T, NP, NW> acc;
  flat_regs<T, NP, NW> oldRmp, addrpm;  // R-+ (accumulated into) and r+- (added at the very end)
  {
    stage_regs<T, NP, NW> s1, s2;
    s1.load(R_pm, N);
    s2.load(r_mp, N);
    oldRmp.load(R_mp, N);
    if (a.d_symmetric) addrpm.load_dsym(r_mp, N, a.d_symmetric); else addrpm.load(r_pm, N);
    if (tid < NP) {
      const bool in = tid < N;
      vJp[tid] = in ? J0_p[tid] : T(0);
      vJm[tid] = in ? J0_m[tid] : T(0);
      vjp[tid] = in ? j0_p[tid] : T(0);
      vjm[tid] = in ? j0_m[tid] : T(0);
    }
    s1.store(L1);
    s2.store(L2);
  }
  gl_operand_a<T, NP, NW> opA;   // A operands streamed from global: T--, later t++
  gl_operand_b<T, NP, NW> opB;   // B operands streamed from global: T++, t--
  opA.prefetch(T_mm, N);     // lands while G1 is being formed
  __syncthreads();
  VSM_STAMP(8);   // stage
  auto ident = [](T x, int, int, T) { return x; };
  // ---- G1 = (I - r-+ R+-)^-1 -> L3 -------------------------------------------------------
  acc.zero();
  mm_ll<T, NP, NW>(acc, L2, L1, Kend);
  VSM_STAMP(9);   // r R
  invert_one_minus<T, NP, NW>(acc, L3, L4, N, Kend, sm, slot, 0);
  VSM_STAMP(10);  // inverse 1
  opB.prefetch(T_pp, N);
  // H = G1 r-+ -> L4 (scratch of the inverse, free after its final barrier).  H serves twice:
  //   T01_inv r-+ = T-- H, and the second inverse of the reference (interaction.jl:243) by the
  //   push-through identity (I - R+- r-+)^-1 = I + R+- (I - r-+ R+-)^-1 r-+ = I + R+- H,
  // which replaces a second series / Gauss-Jordan by one product.
  acc.zero();
  mm_ll<T, NP, NW>(acc, L3, L2, Kend);
  acc_store<T, NP, NW>(L4, acc, ident);
  // T01_inv = T-- G1 (kept in registers until every wave is done reading G1)
  acc_block<T, NP, NW> acc2;
  acc2.zero();
  opA.run(acc2, L3);
  // u = r-+ J0+ + j0-
  {
    T y1, y2;
    matvec2<T, NP, NW>(L2, vJp, vJp, y1, y2);
    if (mlead) vu[mrow] = y1 + vjm[mrow];
  }
  __syncthreads();  // G1 (L3) and r-+ (L2) no longer read; H and u complete
  VSM_STAMP(11);  // H = G1 r, T01 = T-- G1, matvec u
  acc_store<T, NP, NW>(L3, acc2, ident);  // T01_inv -> L3
  // T01_inv r-+ = T-- H -> L2
  acc.zero();
  opA.run(acc, L4);
  acc_store<T, NP, NW>(L2, acc, ident);
  opA.prefetch(t_pp, N);  // needed only for T21_inv; lands during the next products
  __syncthreads();        // T01_inv (L3) and T01_inv r-+ (L2) complete
  VSM_STAMP(12);  // T-- H
  // J0- += T01_inv u
  {
    T y1, y2;
    matvec2<T, NP, NW>(L3, vu, vu, y1, y2);
    if (mlead && mrow < N) J0_m[mrow] = vJm[mrow] + y1;
  }
  // R-+ += (T01_inv r-+) T++
  acc.zero();
  opB.run(acc, L2);
  if (a.d_symmetric) opB.prefetch_dsym(t_pp, N, a.d_symmetric); else opB.prefetch(t_mm, N);
  __syncthreads();  // everybody finished reading L2 (as A operand)
  acc_store<T, NP, NW>(L2, acc, ident);
  __syncthreads();
  oldRmp.add_lds_store(R_mp, L2, N);
  VSM_STAMP(13);  // (..) T++ -> R-+ update
  // T-- = T01_inv t--
  acc.zero();
  opB.run(acc, L3);
  opB.prefetch(T_pp, N);  // pre-update T++ again, for T++ = T21_inv T++
  // ---- G2 = (I - R+- r-+)^-1 = I + R+- H -> L2 --------------------------------------------
  acc2.zero();
  mm_ll<T, NP, NW>(acc2, L1, L4, Kend);
  __syncthreads();  // T01_inv (L3), H (L4) no longer read; R-+ write-out finished reading L2
  acc_store<T, NP, NW>(L3, acc, ident);  // new T-- -> L3
  acc_store<T, NP, NW>(L2, acc2, [=](T x, int r, int c, T) {
    const T v = (r < N && c < N) ? x : T(0);
    return (r == c) ? v + T(1) : v;
  });
  __syncthreads();
  lds_to_global<T, NP, NW>(T_mm, L3, N);
  VSM_STAMP(14);  // T-- = T01 t--, G2 = I + R H, write T--
  // T21_inv = t++ G2 -> L4
  acc.zero();
  opA.run(acc, L2);
  acc_store<T, NP, NW>(L4, acc, ident);
  // z = J0+ + R+- j0-
  {
    T y1, y2;
    matvec2<T, NP, NW>(L1, vjm, vjm, y1, y2);
    if (mlead) vz[mrow] = vJp[mrow] + y1;
  }
  __syncthreads();  // T21_inv and z complete; G2 (L2) no longer read; T-- write-out finished reading L3
  VSM_STAMP(16);  // T21 = t++ G2, matvec z
  // J0+ = j0+ + T21_inv z
  {
    T y1, y2;
    matvec2<T, NP, NW>(L4, vz, vz, y1, y2);
    if (mlead && mrow < N) J0_p[mrow] = vjp[mrow] + y1;
  }
  // T++ = T21_inv T++   and   tmp = T21_inv R+-
  acc.zero();
  opB.run(acc, L4);
  if (a.d_symmetric) opB.prefetch_dsym(t_pp, N, a.d_symmetric); else opB.prefetch(t_mm, N);
  acc2.zero();
  mm_ll<T, NP, NW>(acc2, L4, L1, Kend);
  acc_store<T, NP, NW>(L3, acc2, ident);  // tmp -> L3
  acc_store<T, NP, NW>(L2, acc, ident);   // new T++ -> L2 (G2 is dead)
  __syncthreads();
  lds_to_global<T, NP, NW>(T_pp, L2, N);
  VSM_STAMP(17);  // T21 T++, T21 R+-, write T++
  // R+- = r+- + tmp t--
  acc.zero();
  opB.run(acc, L3);
  acc_store<T, NP, NW>(L4, acc, ident);  // T21_inv (L4) dead since the barrier above
  __syncthreads();
  addrpm.add_lds_store(R_pm, L4, N);
  VSM_STAMP(18);  // (..) t-- -> R+-
}

// ---------------------------------------------------------------------------
// diagnostics: LDS tile product and LDS inverse on plain inputs
// ---------------------------------------------------------------------------
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_test_mm(int N, const T* A, const T* B, T* Cout) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  const long long o = (long long)blockIdx.x * N * N;
  const int Kend = ((N + 3) >> 2) << 2;
  stage<T, NP, NW>(sm.L[0], A + o, N);
  stage<T, NP, NW>(sm.L[1], B + o, N);
  __syncthreads();
  acc_block<T, NP, NW> acc;
  acc.zero();
  mm_ll<T, NP, NW>(acc, sm.L[0], sm.L[1], Kend);
  acc_store<T, NP, NW>(sm.L[2], acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  lds_to_global<T, NP, NW>(Cout + o, sm.L[2], N);
}
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_test_inv(int N, const T* A, T* X, int mode, int* path_out) {
  using C = fcfg<NP, NW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  const long long o = (long long)blockIdx.x * N * N;
  const int Kend = ((N + 3) >> 2) << 2;
  // E = I - A, formed as a product so that it arrives in accumulator layout: E = (I - A) * I
  for (int e = threadIdx.x; e < NP * NP; e += C::NT) {
    const int i = e % NP, j = e / NP;
    const T av = (i < N && j < N) ? A[o + i + (long long)N * j] : ((i == j) ? T(1) : T(0));
    sm.L[0][lidx<NP>(i, j)] = ((i == j) ? T(1) : T(0)) - av;
    sm.L[1][lidx<NP>(i, j)] = (i == j) ? T(1) : T(0);
  }
  __syncthreads();
  acc_block<T, NP, NW> acc;
  acc.zero();
  mm_ll<T, NP, NW>(acc, sm.L[0], sm.L[1], NP);
  __syncthreads();
  int slot = 0;
  const int path = invert_one_minus<T, NP, NW>(acc, sm.L[2], sm.L[3], N, Kend, sm, slot, mode);
  lds_to_global<T, NP, NW>(X + o, sm.L[2], N);
  if (path_out && threadIdx.x == 0) path_out[blockIdx.x] = path;
}

// X = (I - A B)^-1 in one launch (replaces the `I_static .- A ⊠ B` product + batch_inv! pair of the operator-level
// paths: rt_helpers.jl:103-104, interaction.jl:220-222,243-244): product, series inverse or Gauss-Jordan, all in LDS.
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_inv_one_minus_ab(int N, const T* __restrict__ A, long long sa,
                                                              const T* __restrict__ B, long long sb, T* X) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  const long long s = blockIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  stage<T, NP, NW>(sm.L[0], A + s * sa, N);
  stage<T, NP, NW>(sm.L[1], B + s * sb, N);
  __syncthreads();
  acc_block<T, NP, NW> acc;
  acc.zero();
  mm_ll<T, NP, NW>(acc, sm.L[0], sm.L[1], Kend);
  __syncthreads();
  int slot = 0;
  invert_one_minus<T, NP, NW>(acc, sm.L[2], sm.L[3], N, Kend, sm, slot, 0);
  lds_to_global<T, NP, NW>(X + s * (long long)N * N, sm.L[2], N);
}

// ---------------------------------------------------------------------------
// Rotational-Raman doubling, ELASTIC part of one doubling step around the line kernels (doubling_inelastic.jl:36-61 and
// :132-164): per spectral point, everything the inelastic recurrences read and the update of the elastic operators
// afterwards.  The operator-level chain spends 18 launches per step on N x N products; here it is two, LDS-resident.
//   pre :  gp = (I - r r)^-1 ; ttg = t gp ; gt = gp t ; gr = gp r ; grt = gr t ; J1+- = J0+- expk ;
//          u = J0+ + r J1- ; u2 = J1- + r J0+ ; tmp1 = gp u ; tmp2 = gp u2
//   post:  J0- += ttg u2 ; J0+ = J1+ + ttg u ; expk <- expk^2 ; r <- r + (ttg r) t ; t <- ttg t
// ---------------------------------------------------------------------------
template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_raman_elastic_pre(int N, const T* __restrict__ r, const T* __restrict__ t,
                                                               const T* __restrict__ j0p, const T* __restrict__ j0m,
                                                               const T* __restrict__ expk, T* ttg, T* gt, T* gr, T* grt,
                                                               T* j1p, T* j1m, T* u, T* u2, T* tmp1, T* tmp2) {
  using C = fcfg<NP, NW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  const long long s = blockIdx.x, NN = (long long)N * N;
  const int tid = threadIdx.x, Kend = ((N + 3) >> 2) << 2;
  T *R = sm.L[0], *Tt = sm.L[1], *G = sm.L[2], *W = sm.L[3];
  T *vjp = sm.vec[0], *vj1m = sm.vec[1], *vu = sm.vec[2], *vu2 = sm.vec[3];
  stage<T, NP, NW>(R, r + s * NN, N);
  stage<T, NP, NW>(Tt, t + s * NN, N);
  if (tid < NP) {
    const T e = expk[s];
    const T a = (tid < N) ? j0p[s * N + tid] : T(0), b = (tid < N) ? j0m[s * N + tid] : T(0);
    vjp[tid] = a;
    vj1m[tid] = b * e;
    if (tid < N) {
      j1p[s * N + tid] = a * e;
      j1m[s * N + tid] = b * e;
    }
  }
  __syncthreads();
  acc_block<T, NP, NW> acc;
  acc.zero();
  mm_ll<T, NP, NW>(acc, R, R, Kend);
  __syncthreads();
  int slot = 0;
  invert_one_minus<T, NP, NW>(acc, G, W, N, Kend, sm, slot, 0);
  __syncthreads();
  const int row = tid / C::TPR;
  const bool lead = row < NP && (tid % C::TPR) == 0;
  {
    T y1, y2;
    matvec2<T, NP, NW>(R, vj1m, vjp, y1, y2);
    if (lead) {
      const T a = (row < N) ? vjp[row] + y1 : T(0), b = (row < N) ? vj1m[row] + y2 : T(0);
      vu[row] = a;
      vu2[row] = b;
      if (row < N) {
        u[s * N + row] = a;
        u2[s * N + row] = b;
      }
    }
  }
  const auto ident = [](T a, int, int, T) { return a; };
  // ttg = t gp
  acc.zero();
  mm_ll<T, NP, NW>(acc, Tt, G, Kend);
  acc_store<T, NP, NW>(W, acc, ident);
  __syncthreads();   // vu, vu2 and the image of ttg complete
  {
    T y1, y2;
    matvec2<T, NP, NW>(G, vu, vu2, y1, y2);
    if (lead && row < N) {
      tmp1[s * N + row] = y1;
      tmp2[s * N + row] = y2;
    }
  }
  lds_to_global<T, NP, NW>(ttg + s * NN, W, N);
  // gt = gp t
  acc.zero();
  mm_ll<T, NP, NW>(acc, G, Tt, Kend);
  __syncthreads();   // ttg's image has been copied out
  acc_store<T, NP, NW>(W, acc, ident);
  __syncthreads();
  lds_to_global<T, NP, NW>(gt + s * NN, W, N);
  // gr = gp r
  acc.zero();
  mm_ll<T, NP, NW>(acc, G, R, Kend);
  __syncthreads();   // gt's image has been copied out; gp has been read by every wave
  acc_store<T, NP, NW>(W, acc, ident);
  __syncthreads();
  lds_to_global<T, NP, NW>(gr + s * NN, W, N);
  // grt = gr t   (gp's image is free)
  acc.zero();
  mm_ll<T, NP, NW>(acc, W, Tt, Kend);
  acc_store<T, NP, NW>(G, acc, ident);
  __syncthreads();
  lds_to_global<T, NP, NW>(grt + s * NN, G, N);
}

template <typename T, int NP, int NW>
__global__ __launch_bounds__(64 * NW) void k_raman_elastic_post(int N, T* r, T* t, const T* __restrict__ ttg,
                                                                const T* __restrict__ u, const T* __restrict__ u2,
                                                                const T* __restrict__ j1p, T* j0p, T* j0m, T* expk) {
  using C = fcfg<NP, NW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP, NW>& sm = *reinterpret_cast<fsmem<T, NP, NW>*>(smem_raw);
  const long long s = blockIdx.x, NN = (long long)N * N;
  const int tid = threadIdx.x, Kend = ((N + 3) >> 2) << 2;
  T *TG = sm.L[0], *R = sm.L[1], *Tt = sm.L[2], *W = sm.L[3];
  T *vu = sm.vec[0], *vu2 = sm.vec[1];
  stage<T, NP, NW>(TG, ttg + s * NN, N);
  stage<T, NP, NW>(R, r + s * NN, N);
  stage<T, NP, NW>(Tt, t + s * NN, N);
  if (tid < NP) {
    vu[tid] = (tid < N) ? u[s * N + tid] : T(0);
    vu2[tid] = (tid < N) ? u2[s * N + tid] : T(0);
  }
  if (tid == 0) {
    const T e = expk[s];
    expk[s] = e * e;
  }
  __syncthreads();
  {
    const int row = tid / C::TPR;
    T y1, y2;
    matvec2<T, NP, NW>(TG, vu2, vu, y1, y2);
    if (row < N && (tid % C::TPR) == 0) {
      j0m[s * N + row] += y1;
      j0p[s * N + row] = j1p[s * N + row] + y2;
    }
  }
  const auto ident = [](T a, int, int, T) { return a; };
  acc_block<T, NP, NW> accM, accT;
  accM.zero();
  mm_ll<T, NP, NW>(accM, TG, R, Kend);    // ttg r
  acc_store<T, NP, NW>(W, accM, ident);
  accT.zero();
  mm_ll<T, NP, NW>(accT, TG, Tt, Kend);   // t' = ttg t
  __syncthreads();
  accM.zero();
  mm_ll<T, NP, NW>(accM, W, Tt, Kend);    // (ttg r) t
  __syncthreads();   // ttg, t and ttg r have been read by every wave
  acc_store<T, NP, NW>(TG, accM, [=](T a, int rr, int c, T) { return a + R[lidx<NP>(rr, c)]; });
  acc_store<T, NP, NW>(W, accT, ident);
  __syncthreads();
  lds_to_global<T, NP, NW>(r + s * NN, TG, N);
  lds_to_global<T, NP, NW>(t + s * NN, W, N);
}

// ---------------------------------------------------------------------------
// Rotational-Raman doubling, inelastic part of ONE doubling step for all Raman lines of one recipient point
// (doubling_inelastic.jl:62-123: the two `for dn` loops), N <= 30.
// One workgroup per recipient point n1 walks the lines dn; per line the ten N^3 products
//     X = ier r0 + r1 ier ;  X gt0, X gr0, iet gt0, iet grt0, r1 iet, (ier + X gr0) t0, ttg1 (..), ttg1 (..)
// run on LDS-resident operands (NP = 32: 16 buffers of 8 KB in FP64), and EVERY matrix-vector product of the source
// recurrences rides along as a spare column (N, N+1) of a B operand -- ier j1-, ier j0+ with r0; X tmp1 / iet tmp1
// with gt0; X tmp2 with gr0; iet tmp2 with grt0; r1 iej1-, r1 iej0+ with iet; ttg1 a3 / ttg1 a4 with the last two
// products.  The operator-level path streams the 4-D arrays through ~30 launches per step; here each block of
// ier / iet / ieJ is read once and written once.  (subscripts: 1 = recipient point n1, 0 = donor n0 = n1 + shift[dn])
// ---------------------------------------------------------------------------
template <typename T>
struct rdsmem {
  T B[9][32 * 32];
  T v[12][32];
};
// element-wise visit of the accumulator tiles: f(value, row, col)
template <typename T, int NP, int NW, typename F>
__device__ __forceinline__ void acc_visit(const acc_block<T, NP, NW>& acc, F f) {
  using C = fcfg<NP, NW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = 16 * ((wave / C::WC) * C::TMR), c0 = 16 * ((wave % C::WC) * C::TMC) + (lane & 15);
#pragma unroll
  for (int a = 0; a < C::TMR; ++a)
#pragma unroll
    for (int b = 0; b < C::TMC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) f(acc.v[a][b][r], r0 + 16 * a + mfma<T>::crow(lane, r), c0 + 16 * b);
}
// One workgroup per recipient point n1, all Raman lines of it in a loop (doubling_inelastic.jl:56-131 per line):
//   X   = ier r0 + r1 ier                         W1 = iet + X gt0        WA = ier + X gr0        W3 = WA t0 + r1 iet
//   iet' = ttg1 W1 + iet gt0                      ier' = ier + iet grt0 + ttg1 W3
// with the six source mat-vecs riding in the spare columns N, N+1 of the right operands.  Nine 32 x 32 LDS images
// (72 KB: TWO workgroups per CU -- the kernel is latency-bound: ten barriers around 32 x 32 x 24 products): inputs that
// are only right operands give their image to the intermediate that replaces them (r0 -> X -> iet', gt0 -> W1, gr0 -> WA,
// grt0 -> W3, t0 -> ier'), addend-only intermediates (r1 iet, iet gt0, iet grt0) stay in accumulator registers, and the
// next in-band line's inputs are fetched into registers while the current line is computed.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_raman_doubling_lines(
    int N, int S, int K, const int* __restrict__ shift, const T* __restrict__ r, const T* __restrict__ t,
    const T* __restrict__ ttg, const T* __restrict__ gt, const T* __restrict__ gr, const T* __restrict__ grt,
    const T* __restrict__ jp, const T* __restrict__ j1m, const T* __restrict__ tmp1, const T* __restrict__ tmp2,
    const T* __restrict__ expk, T* ier, T* iet, T* ieJp, T* ieJm) {
  constexpr int NP = 32, NW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  rdsmem<T>& sm = *reinterpret_cast<rdsmem<T>*>(smem_raw);
  T* R1 = sm.B[0];
  T* TTG = sm.B[1];
  T* IER = sm.B[2];
  T* IET = sm.B[3];    // + columns N, N+1: iej1- , iej0+
  T* R0X = sm.B[4];    // r0 (+ columns N, N+1: j1-[n0], j0+[n0]) -> X -> image of iet'
  T* GTW1 = sm.B[5];   // gt0 (+ column N: tmp1[n0]) -> W1 (+ column N: a3)
  T* GRWA = sm.B[6];   // gr0 (+ column N: tmp2[n0]) -> WA
  T* GRTW3 = sm.B[7];  // grt0 (+ column N: tmp2[n0]) -> W3 (+ column N: a4)
  T* T0O = sm.B[8];    // t0 -> image of ier'
  T* v1 = sm.v[0];     // ier j1-
  T* v2 = sm.v[1];     // ier j0+
  T* v4 = sm.v[3];     // X tmp2
  T* v5 = sm.v[4];     // iet tmp1
  T* v6 = sm.v[5];     // iet tmp2
  T* v7 = sm.v[6];     // r1 iej1-
  T* v8 = sm.v[7];     // r1 iej0+
  T* vJp = sm.v[8];    // iej0+
  T* vJ1m = sm.v[9];   // iej1-
  T* vJm = sm.v[10];   // iej0-
  const int n1 = blockIdx.x, tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  const long long NN = (long long)N * N;
  const int cA = N, cB = N + 1;  // spare columns (N <= 30)
  stage<T, NP, NW>(R1, r + n1 * NN, N);
  stage<T, NP, NW>(TTG, ttg + n1 * NN, N);
  stage_regs<T, NP, NW> pf[7];
  T pv[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0)};   // ieJ0+, ieJ0-, j1-[n0], j0+[n0], tmp1[n0], tmp2[n0], expk[n0]
  auto next_line = [&](int from) {
    int d = from;
    while (d < K) {
      const int n0 = n1 + shift[d];
      if (n0 >= 0 && n0 < S) break;
      ++d;
    }
    return d;
  };
  auto prefetch = [&](int d) {
    if (d >= K) return;
    const int n0 = n1 + shift[d];
    const long long o4 = ((long long)n1 + (long long)S * d) * NN, o4v = ((long long)n1 + (long long)S * d) * N;
    pf[0].load(ier + o4, N);
    pf[1].load(iet + o4, N);
    pf[2].load(r + n0 * NN, N);
    pf[3].load(gt + n0 * NN, N);
    pf[4].load(gr + n0 * NN, N);
    pf[5].load(grt + n0 * NN, N);
    pf[6].load(t + n0 * NN, N);
    pv[6] = expk[n0];
    if (tid < N) {
      pv[0] = ieJp[o4v + tid];
      pv[1] = ieJm[o4v + tid];
      pv[2] = j1m[(long long)n0 * N + tid];
      pv[3] = jp[(long long)n0 * N + tid];
      pv[4] = tmp1[(long long)n0 * N + tid];
      pv[5] = tmp2[(long long)n0 * N + tid];
    }
  };
  int dn = next_line(0);
  prefetch(dn);
  for (; dn < K;) {
    const long long o4 = ((long long)n1 + (long long)S * dn) * NN, o4v = ((long long)n1 + (long long)S * dn) * N;
    __syncthreads();  // previous line's readers of the images are done
    pf[0].store(IER);
    pf[1].store(IET);
    pf[2].store(R0X);
    pf[3].store(GTW1);
    pf[4].store(GRWA);
    pf[5].store(GRTW3);
    pf[6].store(T0O);
    const T e0 = pv[6];
    const T a_jp = pv[0], b_jm = pv[1], x_j1m = pv[2], x_jp = pv[3], x1 = pv[4], x2 = pv[5];
    const int dn_next = next_line(dn + 1);
    prefetch(dn_next);
    __syncthreads();
    if (tid < N) {
      vJp[tid] = a_jp;
      vJm[tid] = b_jm;
      vJ1m[tid] = b_jm * e0;
      IET[lidx<NP>(tid, cA)] = b_jm * e0;
      IET[lidx<NP>(tid, cB)] = a_jp;
      R0X[lidx<NP>(tid, cA)] = x_j1m;
      R0X[lidx<NP>(tid, cB)] = x_jp;
      GTW1[lidx<NP>(tid, cA)] = x1;
      GRWA[lidx<NP>(tid, cA)] = x2;
      GRTW3[lidx<NP>(tid, cA)] = x2;
    }
    __syncthreads();
    // X = ier r0 + r1 ier  (+ v1 = ier j1-, v2 = ier j0+) ;  r1 iet stays in registers (+ v7 = r1 iej1-, v8 = r1 iej0+)
    acc_block<T, NP, NW> accX, accR1IET;
    accX.zero();
    mm_ll<T, NP, NW>(accX, IER, R0X, Kend);
    mm_ll<T, NP, NW>(accX, R1, IER, Kend);       // (ier's image is zero in the columns >= N: the riders are untouched)
    accR1IET.zero();
    mm_ll<T, NP, NW>(accR1IET, R1, IET, Kend);
    acc_visit<T, NP, NW>(accR1IET, [=](T a, int rr, int c) {
      if (c == cA) v7[rr] = a;
      if (c == cB) v8[rr] = a;
    });
    __syncthreads();  // r0, and iet / ier as right operands, have been read by every wave
    acc_store<T, NP, NW>(R0X, accX, [=](T a, int rr, int c, T) {
      if (c == cA) v1[rr] = a;
      if (c == cB) v2[rr] = a;
      return (c < N) ? a : T(0);
    });
    __syncthreads();
    // X gt0 (+ v3), X gr0 (+ v4), iet gt0 (+ v5), iet grt0 (+ v6)
    acc_block<T, NP, NW> accW1, accWA, accIETGT, accIETGRT;
    accW1.zero();
    mm_ll<T, NP, NW>(accW1, R0X, GTW1, Kend);
    accWA.zero();
    mm_ll<T, NP, NW>(accWA, R0X, GRWA, Kend);
    accIETGT.zero();
    mm_ll<T, NP, NW>(accIETGT, IET, GTW1, Kend);
    accIETGRT.zero();
    mm_ll<T, NP, NW>(accIETGRT, IET, GRTW3, Kend);
    __syncthreads();  // X, gt0, gr0, grt0 have been read by every wave
    // W1 = iet + X gt0, its column N = a3 = iej0+ + r1 iej1- + ier j1- + X tmp1 ;  WA = ier + X gr0
    acc_store<T, NP, NW>(GTW1, accW1, [=](T a, int rr, int c, T) {
      if (c == cA) return (rr < N) ? vJp[rr] + v7[rr] + v1[rr] + a : T(0);   // (rows >= N of the vectors are never written)
      return (c < N) ? a + IET[lidx<NP>(rr, c)] : T(0);
    });
    acc_store<T, NP, NW>(GRWA, accWA, [=](T a, int rr, int c, T) {
      if (c == cA) v4[rr] = a;
      return (c < N) ? a + IER[lidx<NP>(rr, c)] : T(0);
    });
    acc_visit<T, NP, NW>(accIETGT, [=](T a, int rr, int c) {
      if (c == cA) v5[rr] = a;
    });
    acc_visit<T, NP, NW>(accIETGRT, [=](T a, int rr, int c) {
      if (c == cA) v6[rr] = a;
    });
    __syncthreads();
    // W3 = WA t0 + r1 iet, its column N = a4 = iej1- + ier j0+ + r1 iej0+ + X tmp2   (grt0's image is free)
    {
      acc_block<T, NP, NW> acc;
      acc.zero();
      mm_ll<T, NP, NW>(acc, GRWA, T0O, Kend);
      acc.v[0][0] += accR1IET.v[0][0];
      acc_store<T, NP, NW>(GRTW3, acc, [=](T a, int rr, int c, T) {
        if (c == cA) return (rr < N) ? vJ1m[rr] + v2[rr] + v8[rr] + v4[rr] : T(0);
        return (c < N) ? a : T(0);
      });
    }
    __syncthreads();
    // iet' = ttg1 W1 + iet gt0 (+ ieJ0+' from column N) ;  ier' = ier + iet grt0 + ttg1 W3 (+ ieJ0-')
    {
      acc_block<T, NP, NW> acc;
      acc.zero();
      mm_ll<T, NP, NW>(acc, TTG, GTW1, Kend);
      const acc_block<T, NP, NW> add = accIETGT;
      int q = 0;
      acc_store<T, NP, NW>(R0X, acc, [&](T a, int rr, int c, T) {   // X is dead: image of iet'
        const T g = add.v[0][0][q++];
        if (c == cA && rr < N) ieJp[o4v + rr] = vJp[rr] * e0 + a + v5[rr];
        return a + g;
      });
    }
    {
      acc_block<T, NP, NW> acc;
      acc.zero();
      mm_ll<T, NP, NW>(acc, TTG, GRTW3, Kend);
      const acc_block<T, NP, NW> add = accIETGRT;
      int q = 0;
      acc_store<T, NP, NW>(T0O, acc, [&](T a, int rr, int c, T) {   // t0 is dead: image of ier'
        const T g = add.v[0][0][q++];
        if (c == cA && rr < N) ieJm[o4v + rr] = vJm[rr] + a + v6[rr];
        return a + IER[lidx<NP>(rr, c)] + g;
      });
    }
    __syncthreads();
    lds_to_global<T, NP, NW>(iet + o4, R0X, N);
    lds_to_global<T, NP, NW>(ier + o4, T0O, N);
    dn = dn_next;
  }
}

// element-wise load of an N x N global block into the accumulator layout (addend-only operands)
template <typename T, int NP, int NW>
__device__ __forceinline__ void acc_load_global(acc_block<T, NP, NW>& acc, const T* __restrict__ g, int N) {
  using C = fcfg<NP, NW>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = 16 * ((wave / C::WC) * C::TMR), c0 = 16 * ((wave % C::WC) * C::TMC) + (lane & 15);
#pragma unroll
  for (int a = 0; a < C::TMR; ++a)
#pragma unroll
    for (int b = 0; b < C::TMC; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * a + mfma<T>::crow(lane, r), col = c0 + 16 * b;
        acc.v[a][b][r] = (row < N && col < N) ? g[row + (long long)N * col] : T(0);
      }
}
// One pass of interaction_helper!(::RRS, ::ScatteringInterface_11) (interaction_inelastic.jl:319-521) for every line of
// one recipient point (see rs_ia_pass in vsm_internal.h for the operand roles):
//   W1 = L1 E0[n0] + L2 I1 ;  Y = TI W1 + YA ;  W3 = L1 E3[n0] + L2 I3
//   OUTA = ACCA + TI W3 + Y GX[n0] ;  OUTB = TI I4 + Y GY[n0]
//   V1 = L1 VE0[n0] + L2 VI1 + VADD ;  VOUT = VACC + TI V1 + Y VV[n0]        (riding in spare column N)
// Same scheme as k_raman_doubling_lines: nine 32 x 32 images (two workgroups per CU), dead right operands give their
// image to the intermediate that replaces them (E0 -> W1, I1 -> Y, E3 -> W3, I3 -> OUTA, I4 -> OUTB, L1 -> GY), the next
// line's operands are prefetched into registers.
template <typename T>
__global__ __launch_bounds__(256, 2) void k_raman_interaction_lines(int N, int S, int K, const int* __restrict__ shift,
                                                                    rs_ia_pass<T> h) {
  constexpr int NP = 32, NW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  rdsmem<T>& sm = *reinterpret_cast<rdsmem<T>*>(smem_raw);
  T* L2 = sm.B[0];
  T* TI = sm.B[1];
  T* L1GY = sm.B[2];   // L1 -> GY
  T* E0W1 = sm.B[3];   // E0 (+ column N: VE0[n0]) -> W1 (+ column N: V1)
  T* I1Y = sm.B[4];    // I1 (+ column N: VI1) -> Y
  T* E3W3 = sm.B[5];   // E3 -> W3
  T* I3OA = sm.B[6];   // I3 -> image of OUTA
  T* GX = sm.B[7];     // (+ column N: VV[n0])
  T* I4OB = sm.B[8];   // I4 -> image of OUTB
  T* vadd = sm.v[0];
  T* vacc = sm.v[1];
  T* vt = sm.v[2];     // TI V1
  const int n1 = blockIdx.x, tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  const long long NN = (long long)N * N;
  const int cA = N;
  stage<T, NP, NW>(L2, h.L2 + n1 * h.sL2, N);
  stage<T, NP, NW>(TI, h.TI + n1 * NN, N);
  stage_regs<T, NP, NW> pf[8];   // L1, E0, I1, E3, I3, GX, I4, GY
  acc_block<T, NP, NW> pYA, pACC;
  T pv[5] = {T(0), T(0), T(0), T(0), T(0)};   // VE0[n0], VI1, VV[n0], VADD, VACC
  auto next_line = [&](int from) {
    int d = from;
    while (d < K) {
      const int n0 = n1 + shift[d];
      if (n0 >= 0 && n0 < S) break;
      ++d;
    }
    return d;
  };
  auto prefetch = [&](int d) {
    if (d >= K) return;
    const int n0 = n1 + shift[d];
    const long long o4 = ((long long)n1 + (long long)S * d) * NN, o4v = ((long long)n1 + (long long)S * d) * N;
    pf[0].load(h.L1 + o4, N);
    pf[1].load(h.E0 + n0 * h.sE0, N);
    pf[2].load(h.I1 + o4, N);
    pf[3].load(h.E3 + n0 * h.sE3, N);
    pf[4].load(h.I3 + o4, N);
    pf[5].load(h.GX + n0 * NN, N);
    pf[6].load(h.I4 + o4, N);
    pf[7].load(h.GY + n0 * NN, N);
    acc_load_global<T, NP, NW>(pYA, h.YA + o4, N);
    acc_load_global<T, NP, NW>(pACC, h.ACCA + o4, N);
    if (tid < N) {
      pv[0] = h.VE0[(long long)n0 * N + tid];
      pv[1] = h.VI1[o4v + tid];
      pv[2] = h.VV[(long long)n0 * N + tid];
      pv[3] = h.VADD[o4v + tid];
      pv[4] = h.VACC[o4v + tid];
    }
  };
  int dn = next_line(0);
  prefetch(dn);
  for (; dn < K;) {
    const long long o4 = ((long long)n1 + (long long)S * dn) * NN, o4v = ((long long)n1 + (long long)S * dn) * N;
    __syncthreads();  // previous line's readers of the images are done
    pf[0].store(L1GY);
    pf[1].store(E0W1);
    pf[2].store(I1Y);
    pf[3].store(E3W3);
    pf[4].store(I3OA);
    pf[5].store(GX);
    pf[6].store(I4OB);
    const stage_regs<T, NP, NW> gy = pf[7];   // stored into L1's image once L1 is dead
    const acc_block<T, NP, NW> aYA = pYA, aACC = pACC;
    const T x_ve0 = pv[0], x_vi1 = pv[1], x_vv = pv[2], x_vadd = pv[3], x_vacc = pv[4];
    const int dn_next = next_line(dn + 1);
    prefetch(dn_next);
    __syncthreads();
    if (tid < N) {
      E0W1[lidx<NP>(tid, cA)] = x_ve0;
      I1Y[lidx<NP>(tid, cA)] = x_vi1;
      GX[lidx<NP>(tid, cA)] = x_vv;
      vadd[tid] = x_vadd;
      vacc[tid] = x_vacc;
    }
    __syncthreads();
    acc_block<T, NP, NW> accW1, accW3;
    accW1.zero();
    mm_ll<T, NP, NW>(accW1, L1GY, E0W1, Kend);   // (+ L1 VE0)
    mm_ll<T, NP, NW>(accW1, L2, I1Y, Kend);      // (+ L2 VI1)
    accW3.zero();
    mm_ll<T, NP, NW>(accW3, L1GY, E3W3, Kend);
    mm_ll<T, NP, NW>(accW3, L2, I3OA, Kend);
    __syncthreads();  // L1, E0, I1, E3, I3 have been read by every wave
    acc_store<T, NP, NW>(E0W1, accW1, [=](T a, int rr, int c, T) {
      if (c == cA) return (rr < N) ? a + vadd[rr] : T(0);   // V1
      return (c < N) ? a : T(0);
    });
    acc_store<T, NP, NW>(E3W3, accW3, [=](T a, int, int c, T) { return (c < N) ? a : T(0); });
    gy.store(L1GY);
    __syncthreads();
    acc_block<T, NP, NW> accA, accB;
    {
      acc_block<T, NP, NW> accY;
      accY.zero();
      mm_ll<T, NP, NW>(accY, TI, E0W1, Kend);    // TI W1 (+ TI V1)
      accA.zero();
      mm_ll<T, NP, NW>(accA, TI, E3W3, Kend);    // TI W3
      accB.zero();
      mm_ll<T, NP, NW>(accB, TI, I4OB, Kend);    // TI I4
      int q = 0;
      acc_store<T, NP, NW>(I1Y, accY, [&](T a, int rr, int c, T) {   // I1 is dead since the barrier above
        const T ya = aYA.v[0][0][q++];
        if (c == cA) vt[rr] = a;
        return (c < N) ? a + ya : T(0);
      });
    }
    __syncthreads();  // Y complete; W1, W3, I4 have been read
    mm_ll<T, NP, NW>(accA, I1Y, GX, Kend);       // + Y GX (+ Y VV)
    mm_ll<T, NP, NW>(accB, I1Y, L1GY, Kend);     // + Y GY
    {
      int q = 0;
      acc_store<T, NP, NW>(I3OA, accA, [&](T a, int rr, int c, T) {   // I3 is dead: image of OUTA
        const T g = aACC.v[0][0][q++];
        if (c == cA && rr < N) h.VOUT[o4v + rr] = vacc[rr] + vt[rr] + a;
        return a + g;
      });
    }
    acc_store<T, NP, NW>(I4OB, accB, [=](T a, int, int, T) { return a; });   // I4 is dead: image of OUTB
    __syncthreads();
    lds_to_global<T, NP, NW>(h.OUTA + o4, I3OA, N);
    lds_to_global<T, NP, NW>(h.OUTB + o4, I4OB, N);
    dn = dn_next;
  }
}

// ---------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------
template <typename T>
int fused_max_n() {
  return sizeof(T) == 8 ? 64 : 96;
}
template int fused_max_n<double>();
template int fused_max_n<float>();

// waves per workgroup of the NP = 64 instantiations (build-time tunables)
#ifndef ED_WAVES_64
#define ED_WAVES_64 8
#endif
#ifndef IA_WAVES_64
#define IA_WAVES_64 4
#endif
// waves per workgroup of the NP = 96 (FP32) instantiations: 4 (2x2 grid, 3x3 tiles); 6 (3x2 grid, 2x3 tiles) loads the
// SIMDs unevenly and measured 24 % slower on C4
#ifndef ED_WAVES_96
#define ED_WAVES_96 4
#endif
#ifndef IA_WAVES_96
#define IA_WAVES_96 4
#endif

template <typename K>
static int enable_lds(K kern, size_t bytes) {   // once per (device, kernel)
  return ensure_dyn_lds(reinterpret_cast<const void*>(kern), bytes, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
}

// NP = N rounded up to 32; f64 fits up to 64, f32 up to 96 (4 N x N buffers in 160 KB of LDS).
template <int V>
struct np_tag {
  static constexpr int value = V;
};
template <typename T, typename F>
static int dispatch_np(int N, F f) {
  if (N <= 32) return f(np_tag<32>{});
  if (N <= 64) return f(np_tag<64>{});
  if constexpr (sizeof(T) == 4) {
    if (N <= 96) return f(np_tag<96>{});
  }
  set_error("fused kernels: N=%d exceeds the LDS-resident limit (%d)", N, fused_max_n<T>());
  return VSM_ERR_UNSUPPORTED;
}

template <typename T>
int fused_elemental_doubling(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi,
                             const T* tau_sum, const T* F0, const T* Zpp, const T* Zmp, long long zs,
                             const added<T>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if constexpr (sizeof(T) == 8) {
    static const bool no_strip = ab_switch("VSM_NO_STRIP");   // A/B switch for benchmarking
    if (!no_strip && strip_supported(q.N))
      return strip_elemental_doubling(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, zsrc<double>{Zpp, Zmp, zs, 0, nullptr}, a, st);
  }
  return dispatch_np<T>(q.N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = (NP == 64) ? ED_WAVES_64 : (NP == 96) ? ED_WAVES_96 : 4;
    auto kern = k_elemental_doubling<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, q, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp,
                       zs, a);
    VSM_LAUNCH_CHECK("k_elemental_doubling");
    return (int)VSM_OK;
  });
}

template <typename T>
int fused_interaction(int iface, int N, int S, const composite<T>& c, const added<T>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (iface != VSM_IFACE_11) {
    set_error("fused_interaction: only ScatteringInterface_11 is fused");
    return VSM_ERR_UNSUPPORTED;
  }
  if constexpr (sizeof(T) == 8) {
    static const bool no_native = ab_switch("VSM_NO_NATIVE_IA");
    if (!no_native && N <= 64) return native_interaction11(N, S, c, a, st);
    static const bool no_strip = ab_switch("VSM_NO_STRIP") || ab_switch("VSM_NO_STRIP_IA");
    if (!no_strip && strip_layer_supported(N)) return strip_interaction11(N, S, c, a, st);
  }
  if constexpr (sizeof(T) == 4) {
    if (strip32_supported(N)) return strip32_interaction11(N, S, c, a, st);
  }
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = (NP == 64) ? IA_WAVES_64 : (NP == 96) ? IA_WAVES_96 : 4;
    auto kern = k_interaction11<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, c, a);
    VSM_LAUNCH_CHECK("k_interaction11");
    return (int)VSM_OK;
  });
}

template <typename T>
int test_lds_mm(int N, int S, const T* A, const T* B, T* Cout, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = (NP == 64) ? ED_WAVES_64 : (NP == 96) ? ED_WAVES_96 : 4;
    auto kern = k_test_mm<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, A, B, Cout);
    VSM_LAUNCH_CHECK("k_test_mm");
    return (int)VSM_OK;
  });
}

template <typename T>
int test_lds_inv(int N, int S, const T* A, T* X, int mode, int* path_out, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = (NP == 64) ? ED_WAVES_64 : (NP == 96) ? ED_WAVES_96 : 4;
    auto kern = k_test_inv<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, A, X, mode, path_out);
    VSM_LAUNCH_CHECK("k_test_inv");
    return (int)VSM_OK;
  });
}

// (I - A B)^-1 ; VSM_ERR_UNSUPPORTED when N exceeds the on-chip limit (callers then use gemm + batch_inv)
template <typename T>
int inv_one_minus_product(int N, int S, const T* A, long long sa, const T* B, long long sb, T* X, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (N > fused_max_n<T>()) return VSM_ERR_UNSUPPORTED;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = (NP == 64) ? ED_WAVES_64 : (NP == 96) ? ED_WAVES_96 : 4;
    auto kern = k_inv_one_minus_ab<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, A, sa, B, sb, X);
    VSM_LAUNCH_CHECK("k_inv_one_minus_ab");
    return (int)VSM_OK;
  });
}

// inelastic part of one Raman doubling step (all lines); VSM_ERR_UNSUPPORTED for N > 30 (callers use the operator-level chain)
template <typename T>
int raman_elastic_pre(int N, int S, const T* r, const T* t, const T* j0p, const T* j0m, const T* expk, T* ttg, T* gt, T* gr,
                      T* grt, T* j1p, T* j1m, T* u, T* u2, T* tmp1, T* tmp2, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  static const bool off = ab_switch("VSM_NO_RAMAN_FUSION") || ab_switch("VSM_NO_RAMAN_ELASTIC_FUSION");
  if (off || N > fused_max_n<T>()) return VSM_ERR_UNSUPPORTED;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = 4;
    auto kern = k_raman_elastic_pre<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, r, t, j0p, j0m, expk, ttg, gt, gr, grt, j1p, j1m,
                       u, u2, tmp1, tmp2);
    VSM_LAUNCH_CHECK("k_raman_elastic_pre");
    return (int)VSM_OK;
  });
}
template <typename T>
int raman_elastic_post(int N, int S, T* r, T* t, const T* ttg, const T* u, const T* u2, const T* j1p, T* j0p, T* j0m,
                       T* expk, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  static const bool off = ab_switch("VSM_NO_RAMAN_FUSION") || ab_switch("VSM_NO_RAMAN_ELASTIC_FUSION");
  if (off || N > fused_max_n<T>()) return VSM_ERR_UNSUPPORTED;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    constexpr int NW = 4;
    auto kern = k_raman_elastic_post<T, NP, NW>;
    const size_t bytes = sizeof(fsmem<T, NP, NW>);
    const int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(fcfg<NP, NW>::NT), bytes, st, N, r, t, ttg, u, u2, j1p, j0p, j0m, expk);
    VSM_LAUNCH_CHECK("k_raman_elastic_post");
    return (int)VSM_OK;
  });
}

template <typename T>
int raman_doubling_lines(int N, int S, int K, const int* shift, const T* r, const T* t, const T* ttg, const T* gt, const T* gr,
                         const T* grt, const T* jp, const T* j1m, const T* tmp1, const T* tmp2, const T* expk, T* ier, T* iet,
                         T* ieJp, T* ieJm, hipStream_t st) {
  if (S <= 0 || K <= 0) return VSM_OK;
  static const bool off = ab_switch("VSM_NO_RAMAN_FUSION");
  if (N > 30 || off) return VSM_ERR_UNSUPPORTED;
  auto kern = k_raman_doubling_lines<T>;
  const size_t bytes = sizeof(rdsmem<T>);
  const int prepared = enable_lds(kern, bytes);
  if (prepared) return prepared;
  hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, N, S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1, tmp2, expk, ier,
                     iet, ieJp, ieJm);
  VSM_LAUNCH_CHECK("k_raman_doubling_lines");
  return VSM_OK;
}

template <typename T>
int raman_interaction_lines(int N, int S, int K, const int* shift, const rs_ia_pass<T>& h, hipStream_t st) {
  if (S <= 0 || K <= 0) return VSM_OK;
  static const bool off = ab_switch("VSM_NO_RAMAN_FUSION") || ab_switch("VSM_NO_RAMAN_IA_FUSION");
  if (N > 30 || off) return VSM_ERR_UNSUPPORTED;
  auto kern = k_raman_interaction_lines<T>;
  const size_t bytes = sizeof(rdsmem<T>);
  const int prepared = enable_lds(kern, bytes);
  if (prepared) return prepared;
  hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, N, S, K, shift, h);
  VSM_LAUNCH_CHECK("k_raman_interaction_lines");
  return VSM_OK;
}

#define VSM_INST_F(T)                                                                                               \
  template int fused_elemental_doubling<T>(const quad<T>&, int, int, int, const T*, const T*, const T*, const T*,  \
                                           const T*, const T*, long long, const added<T>&, hipStream_t);           \
  template int fused_interaction<T>(int, int, int, const composite<T>&, const added<T>&, hipStream_t);              \
  template int inv_one_minus_product<T>(int, int, const T*, long long, const T*, long long, T*, hipStream_t);       \
  template int raman_elastic_pre<T>(int, int, const T*, const T*, const T*, const T*, const T*, T*, T*, T*, T*, T*, T*, T*, \
                                    T*, T*, T*, hipStream_t);                                                        \
  template int raman_elastic_post<T>(int, int, T*, T*, const T*, const T*, const T*, const T*, T*, T*, T*, hipStream_t); \
  template int raman_doubling_lines<T>(int, int, int, const int*, const T*, const T*, const T*, const T*, const T*, \
                                       const T*, const T*, const T*, const T*, const T*, const T*, T*, T*, T*, T*,  \
                                       hipStream_t);                                                                \
  template int raman_interaction_lines<T>(int, int, int, const int*, const rs_ia_pass<T>&, hipStream_t);            \
  template int test_lds_mm<T>(int, int, const T*, const T*, T*, hipStream_t);                                       \
  template int test_lds_inv<T>(int, int, const T*, T*, int, int*, hipStream_t);
VSM_INST_F(double)
VSM_INST_F(float)

}  // namespace vsm

#ifdef VSM_PHASE_TIMING
extern "C" int vsm_debug_phase_cycles(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_phase_cycles), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_phase_cycles), z, sizeof(z));
  }
  return 0;
}
#endif
