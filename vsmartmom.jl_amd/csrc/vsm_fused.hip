// Fused, LDS-resident kernels: ONE 256-thread workgroup owns ONE spectral point and keeps
// its N x N operators on-chip for a whole layer step.
//
//   k_elemental_doubling : elemental! + ndoubl x doubling step + apply_D!   (1 launch / layer)
//   k_interaction11      : interaction_helper!(::ScatteringInterface_11)      (1 launch / layer)
//
// Data layout in LDS: column-major NP x NP (NP = N rounded up to 32), row index XOR-swizzled
// by column so that BOTH MFMA operand fetch patterns are bank-conflict free:
//   A-fragment  (16 consecutive rows  x 2 consecutive k-columns per half-wave)
//   B-fragment  (2 consecutive k-rows x 16 consecutive columns  per half-wave)
// MFMA: v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, 4 waves in a 2x2 grid, each wave
// owning a (NP/32 x NP/32) block of 16x16 accumulator tiles.
//
// The matrix inverse (I - E)^-1 that the reference obtains from batched getrf/getri is
// produced on-chip either by a truncated Neumann series whose truncation error is bounded
// below rounding by ||E||_F (thin layers: 0-2 extra MFMA products), or by the pivoted
// Gauss-Jordan of vsm_inverse.h (general case).
#include "vsm_internal.h"
#include "vsm_inverse.h"

namespace vsm {

template <int NP>
__device__ __forceinline__ int lidx(int a, int b) {
  return (a ^ (((b & 1) << 4) | (((b >> 1) & 7) << 1))) + NP * b;
}

template <typename T, int NP>
struct fsmem {
  T L[4][NP * NP];
  T vec[8][NP];
  T red[8];
  int flag[4];
  gj_scratch<T, NP> gj;
};

template <typename T, int NP>
struct acc_block {
  static constexpr int TM = NP / 32;
  typename mfma<T>::acc_t v[TM][TM];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) v[a][b] = acc_zero<T>();
  }
};

// ---- MFMA k-loops ---------------------------------------------------------------------
// All loops are software-pipelined by hand: the fragments of k-step (k0+4) are requested
// before the MFMAs of k-step k0 are issued, so LDS (or L2) latency hides behind the
// 64-cycle f64 / 32-cycle f32 MFMAs.  (hipcc does not pipeline a runtime-trip-count loop.)
template <typename T, int NP>
struct frag_set {
  static constexpr int TM = NP / 32;
  T a[TM], b[TM];
};

template <typename T, int NP>
__device__ __forceinline__ void load_ll(frag_set<T, NP>& f, const T* A, const T* B, int k, int rowA, int colB) {
  constexpr int TM = NP / 32;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    f.a[t] = A[lidx<NP>(rowA + 16 * t, k)];
    f.b[t] = B[lidx<NP>(k, colB + 16 * t)];
  }
}
template <typename T, int NP>
__device__ __forceinline__ void mma_all(acc_block<T, NP>& acc, const frag_set<T, NP>& f) {
  constexpr int TM = NP / 32;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b) acc.v[a][b] = mfma<T>::mma(f.a[a], f.b[b], acc.v[a][b]);
}

// acc += A * B with A, B in (swizzled) LDS.  Kend: multiple of 4 covering N.
template <typename T, int NP>
__device__ __forceinline__ void mm_ll(acc_block<T, NP>& acc, const T* A, const T* B, int Kend) {
  constexpr int TM = NP / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowA = 16 * ((wave >> 1) * TM) + (lane & 15);
  const int colB = 16 * ((wave & 1) * TM) + (lane & 15);
  const int kq = lane >> 4;
  frag_set<T, NP> f0, f1;
  load_ll<T, NP>(f0, A, B, kq, rowA, colB);
  int k0 = 0;
  for (; k0 + 8 <= Kend; k0 += 8) {
    load_ll<T, NP>(f1, A, B, k0 + 4 + kq, rowA, colB);
    mma_all<T, NP>(acc, f0);
    if (k0 + 8 < Kend) load_ll<T, NP>(f0, A, B, k0 + 8 + kq, rowA, colB);
    mma_all<T, NP>(acc, f1);
  }
  if (k0 < Kend) mma_all<T, NP>(acc, f0);  // odd number of k-steps
}

// two products sharing the B operand: acc1 += A1*B, acc2 += A2*B
template <typename T, int NP>
__device__ __forceinline__ void mm_ll2(acc_block<T, NP>& acc1, acc_block<T, NP>& acc2, const T* A1, const T* A2,
                                       const T* B, int Kend) {
  constexpr int TM = NP / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rowA = 16 * ((wave >> 1) * TM) + (lane & 15);
  const int colB = 16 * ((wave & 1) * TM) + (lane & 15);
  const int kq = lane >> 4;
  T a1[2][TM], a2[2][TM], bf[2][TM];
  auto load = [&](int buf, int k) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int ia = lidx<NP>(rowA + 16 * t, k);
      a1[buf][t] = A1[ia];
      a2[buf][t] = A2[ia];
      bf[buf][t] = B[lidx<NP>(k, colB + 16 * t)];
    }
  };
  auto mma = [&](int buf) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TM; ++b) {
        acc1.v[a][b] = mfma<T>::mma(a1[buf][a], bf[buf][b], acc1.v[a][b]);
        acc2.v[a][b] = mfma<T>::mma(a2[buf][a], bf[buf][b], acc2.v[a][b]);
      }
  };
  load(0, kq);
  int k0 = 0;
  for (; k0 + 8 <= Kend; k0 += 8) {
    load(1, k0 + 4 + kq);
    mma(0);
    if (k0 + 8 < Kend) load(0, k0 + 8 + kq);
    mma(1);
  }
  if (k0 < Kend) mma(0);
}

// A read straight from global memory (column-major N x N), B in LDS.  The A fragments of the
// whole k-range are requested up front (Kend/4 * TM values per lane) so the L2/HBM latency is
// paid once, overlapped with whatever the caller does between `prefetch` and `run`.
template <typename T, int NP>
struct gl_operand {
  static constexpr int TM = NP / 32;
  static constexpr int KS = NP / 4;
  T a[KS][TM];
  __device__ __forceinline__ void prefetch(const T* __restrict__ Ag, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowA = 16 * ((wave >> 1) * TM) + (lane & 15);
    const int kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int row = rowA + 16 * t, k = 4 * ks + kq;
        a[ks][t] = (row < N && k < N) ? Ag[row + (long long)N * k] : T(0);
      }
  }
  // acc += A * B
  __device__ __forceinline__ void run(acc_block<T, NP>& acc, const T* B) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colB = 16 * ((wave & 1) * TM) + (lane & 15);
    const int kq = lane >> 4;
    T bf[2][TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) bf[0][t] = B[lidx<NP>(kq, colB + 16 * t)];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < TM; ++t) bf[(ks + 1) & 1][t] = B[lidx<NP>(4 * (ks + 1) + kq, colB + 16 * t)];
      }
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TM; ++y) acc.v[x][y] = mfma<T>::mma(a[ks][x], bf[ks & 1][y], acc.v[x][y]);
    }
  }
};
// B read straight from global memory (column-major N x N), A in LDS.
template <typename T, int NP>
struct gl_operand_b {
  static constexpr int TM = NP / 32;
  static constexpr int KS = NP / 4;
  T b[KS][TM];
  __device__ __forceinline__ void prefetch(const T* __restrict__ Bg, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int colB = 16 * ((wave & 1) * TM) + (lane & 15);
    const int kq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        const int col = colB + 16 * t, k = 4 * ks + kq;
        b[ks][t] = (col < N && k < N) ? Bg[k + (long long)N * col] : T(0);
      }
  }
  // acc += A * B
  __device__ __forceinline__ void run(acc_block<T, NP>& acc, const T* A) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rowA = 16 * ((wave >> 1) * TM) + (lane & 15);
    const int kq = lane >> 4;
    T af[2][TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) af[0][t] = A[lidx<NP>(rowA + 16 * t, kq)];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < TM; ++t) af[(ks + 1) & 1][t] = A[lidx<NP>(rowA + 16 * t, 4 * (ks + 1) + kq)];
      }
#pragma unroll
      for (int x = 0; x < TM; ++x)
#pragma unroll
        for (int y = 0; y < TM; ++y) acc.v[x][y] = mfma<T>::mma(af[ks & 1][x], b[ks][y], acc.v[x][y]);
    }
  }
};

// global -> LDS staging split in two halves so that the global latency overlaps other work:
// load() issues the reads into registers, store() writes the swizzled LDS image.
template <typename T, int NP>
struct stage_regs {
  static constexpr int CNT = NP * NP / 256;
  T v[CNT];
  __device__ __forceinline__ void load(const T* __restrict__ src, int N) {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + 256 * c;
      const int i = e % NP, j = e / NP;
      v[c] = (i < N && j < N) ? src[i + (long long)N * j] : T(0);
    }
  }
  __device__ __forceinline__ void store(T* dst) const {
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int e = threadIdx.x + 256 * c;
      dst[lidx<NP>(e % NP, e / NP)] = v[c];
    }
  }
};

template <typename T, int NP>
__device__ __forceinline__ void mm_gl(acc_block<T, NP>& acc, const T* __restrict__ Ag, int N, const T* B, int Kend) {
  gl_operand<T, NP> op;
  op.prefetch(Ag, N);
  op.run(acc, B);
}

// dst(row,col) = f(acc(row,col), row, col) for every accumulator element of this wave.
template <typename T, int NP, typename F>
__device__ __forceinline__ void acc_store(T* dst, const acc_block<T, NP>& acc, F f) {
  constexpr int TM = NP / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * (wr * TM + a) + mfma<T>::crow(lane, r);
        const int col = 16 * (wc * TM + b) + (lane & 15);
        const int ix = lidx<NP>(row, col);
        dst[ix] = f(acc.v[a][b][r], row, col, dst[ix]);
      }
}

// global (column-major N x N) -> swizzled LDS, zero padded
template <typename T, int NP>
__device__ __forceinline__ void stage(T* dst, const T* __restrict__ src, int N) {
  for (int e = threadIdx.x; e < NP * NP; e += 256) {
    const int i = e % NP, j = e / NP;
    dst[lidx<NP>(i, j)] = (i < N && j < N) ? src[i + (long long)N * j] : T(0);
  }
}

// Frobenius norm of the wave-distributed accumulator block (all threads get the result).
template <typename T, int NP>
__device__ __forceinline__ T acc_fro(const acc_block<T, NP>& acc, T* red) {
  constexpr int TM = NP / 32;
  T s = 0;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TM; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc.v[a][b][r] * acc.v[a][b][r];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
  __syncthreads();  // protect red[] from the previous use
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return sqrt(red[0] + red[1] + red[2] + red[3]);
}

// In-place pivoted Gauss-Jordan of the swizzled LDS matrix V (identity padded).  Ends with a barrier.
template <typename T, int NP>
__device__ __noinline__ void gj_lds(T* V, int N, gj_scratch<T, NP>* sc) {
  using C = gj_cfg<NP>;
  const int tr = threadIdx.x % C::TR, tc = threadIdx.x / C::TR;
  T g[C::RB][C::CB];
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < C::CB; ++cb) g[rb][cb] = V[lidx<NP>(tr + C::TR * rb, tc * C::CB + cb)];
  gj_invert<T, NP>(g, N, *sc);
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < C::CB; ++cb) V[lidx<NP>(tr + C::TR * rb, sc->dst[tc * C::CB + cb])] = g[rb][cb];
  __syncthreads();
}

// G = (I - E)^-1 with E given in the accumulators.  Result -> V.  W is scratch.
//
// Series path: G = sum_{k<=K} E^k, built by repeated squaring
//     (I+E)(I+E^2)(I+E^4)... = sum_{k < 2^p} E^k          (2 MFMA products per doubling of the order)
// optionally closed with "+ E^(2^p)" (1 product).  Available orders K = 1,2,3,4,7,8,15,16,31 cost
// 0,1,2,3,4,5,6,7,8 products.  K is the smallest order whose truncation error
// ||E||^(K+1)/(1-||E||) (Frobenius norm, an upper bound of the 2-norm) stays below eps/4, i.e. the
// result is the inverse to working precision -- the same contract as the LU of the reference.
// General path (||E||_F >= 0.3): pivoted Gauss-Jordan.
// mode 0 = automatic, 1 = force Gauss-Jordan, 2 = force series (order 31 if the bound fails).
// Returns 1 for Gauss-Jordan, 1+K for a series of order K.  Ends with a barrier.
template <typename T, int NP>
__device__ __forceinline__ int invert_one_minus(acc_block<T, NP>& acc, T* V, T* W, int N, int Kend,
                                                fsmem<T, NP>& sm, int mode) {
  const T nrm = acc_fro<T, NP>(acc, sm.red);
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
  if (K > 0) {
    // W = E, V = I + E
    acc_store<T, NP>(W, acc, [](T a, int, int, T) { return a; });
    acc_store<T, NP>(V, acc, [](T a, int r, int c, T) { return (r == c) ? a + T(1) : a; });
    __syncthreads();
    int cur = 1;  // W = E^cur, V = sum_{k < 2 cur} E^k
    while (K > 2 * cur - 1) {
      acc.zero();
      mm_ll<T, NP>(acc, W, W, Kend);  // E^(2 cur)
      cur *= 2;
      if (K == cur) {  // close with "+ E^cur"
        acc_store<T, NP>(V, acc, [](T a, int, int, T old) { return old + a; });
        __syncthreads();
        break;
      }
      __syncthreads();
      acc_store<T, NP>(W, acc, [](T a, int, int, T) { return a; });
      __syncthreads();
      acc.zero();
      mm_ll<T, NP>(acc, V, W, Kend);  // V * E^cur
      __syncthreads();
      acc_store<T, NP>(V, acc, [](T a, int, int, T old) { return old + a; });
      __syncthreads();
    }
    return 1 + K;
  }
  // general case: V = I - E, pivoted Gauss-Jordan in registers (out of line: keeps its register
  // footprint out of the MFMA loops' allocation)
  acc_store<T, NP>(V, acc, [](T a, int r, int c, T) { return (r == c) ? T(1) - a : -a; });
  __syncthreads();
  gj_lds<T, NP>(V, N, &sm.gj);
  return 1;
}

// y1 = M*x1, y2 = M*x2: TPR consecutive lanes share a row; each lane walks a statically unrolled
// strided column range (all LDS reads in flight at once), then a shuffle reduction.  All lanes of a
// row group receive the sums.
template <typename T, int NP>
struct mv_map {
  static constexpr int TPR = (NP <= 64) ? 4 : 2;  // threads per row
};
template <typename T, int NP>
__device__ __forceinline__ void matvec2(const T* M, const T* x1, const T* x2, int N, T& y1, T& y2) {
  constexpr int TPR = mv_map<T, NP>::TPR;
  constexpr int CNT = NP / TPR;
  const int row = threadIdx.x / TPR, q = threadIdx.x % TPR;
  T s1 = 0, s2 = 0;
  if (row < NP) {
    T mv[CNT], v1[CNT], v2[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
      const int j = q + TPR * c;  // padded columns hold zeros in M and in x
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
template <typename T, int NP>
__global__ __launch_bounds__(256) void k_elemental_doubling(quad<T> q, int m, int ndoubl, const T* __restrict__ dtau,
                                                            const T* __restrict__ varpi,
                                                            const T* __restrict__ tau_sum, const T* __restrict__ F0,
                                                            const T* __restrict__ Zpp, const T* __restrict__ Zmp,
                                                            long long zs, added<T> out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP>& sm = *reinterpret_cast<fsmem<T, NP>*>(smem_raw);
  T* R = sm.L[0];
  T* Tm = sm.L[1];
  T* W = sm.L[2];
  T* V = sm.L[3];
  T* jp = sm.vec[0];
  T* jm = sm.vec[1];
  T* j1p = sm.vec[2];
  T* j1m = sm.vec[3];
  T* uu = sm.vec[4];
  T* vv = sm.vec[5];
  T* mus = sm.vec[6];
  T* wcs = sm.vec[7];

  const int s = blockIdx.x;
  const int N = q.N, ns = q.n_stokes;
  const int tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
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
  for (int e = tid; e < NP * NP; e += 256) {
    const int i = e % NP, j = e / NP;
    T r = T(0), t = T(0);
    if (i < N && j < N) {
      const T mi = mus[i], mj = mus[j], wct = wcs[j];
      const long long zo = i + (long long)N * j;
      if (wct > num<T>::eps()) {
        r = w * Zm[zo] * (mj / (mi + mj)) * wct * (-expm1(-d * ((T(1) / mi) + (T(1) / mj))));
        if (mi == mj) {
          if (i == j)
            t = exp(-d / mi) * (T(1) + w * Zp[zo] * (d / mi) * wct);
          else
            t = exp(-d / mj) * (w * Zp[zo] * (d / mi) * wct);
        } else {
          t = w * Zp[zo] * (mj / (mi - mj)) * wct * expdiff_neg<T>(d / mi, d / mj);
        }
      } else {
        t = (i == j) ? exp(-d / mi) : T(0);
      }
      if (ndoubl >= 1 && is_uv_row(i, ns)) r = -r;  // starred R* = D R (apply_D_elemental!, elemental.jl:403-422)
    }
    const int ix = lidx<NP>(i, j);
    R[ix] = r;
    Tm[ix] = t;
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
  acc_block<T, NP> acc, acc2;
  for (int n = 0; n < ndoubl; ++n) {
    // G = (I - r r)^-1  -> V
    acc.zero();
    mm_ll<T, NP>(acc, R, R, Kend);
    invert_one_minus<T, NP>(acc, V, W, N, Kend, sm, 0);
    // tt = t G -> W
    acc.zero();
    mm_ll<T, NP>(acc, Tm, V, Kend);
    __syncthreads();
    acc_store<T, NP>(W, acc, [](T a, int, int, T) { return a; });
    if (tid < NP) {
      j1p[tid] = jp[tid] * expk;
      j1m[tid] = jm[tid] * expk;
    }
    __syncthreads();
    // sources: u = j1- + r j0+ ; v = j0+ + r j1-
    {
      T y1, y2;
      matvec2<T, NP>(R, jp, j1m, N, y1, y2);
      constexpr int TPR = mv_map<T, NP>::TPR;
      const int row = tid / TPR;
      if (row < NP && (tid % TPR) == 0) {
        uu[row] = j1m[row] + y1;
        vv[row] = jp[row] + y2;
      }
    }
    __syncthreads();
    {
      T y1, y2;
      matvec2<T, NP>(W, uu, vv, N, y1, y2);
      constexpr int TPR = mv_map<T, NP>::TPR;
      const int row = tid / TPR;
      if (row < NP && (tid % TPR) == 0) {
        jm[row] = jm[row] + y1;   // j0- <- j0- + tt (j1- + r j0+)
        jp[row] = j1p[row] + y2;  // j0+ <- j1+ + tt (j0+ + r j1-)
      }
    }
    // tmp = tt r -> V
    acc.zero();
    mm_ll<T, NP>(acc, W, R, Kend);
    __syncthreads();
    acc_store<T, NP>(V, acc, [](T a, int, int, T) { return a; });
    __syncthreads();
    // r <- r + tmp t ; t <- tt t
    acc.zero();
    acc2.zero();
    mm_ll2<T, NP>(acc, acc2, V, W, Tm, Kend);
    __syncthreads();
    acc_store<T, NP>(R, acc, [](T a, int, int, T old) { return old + a; });
    acc_store<T, NP>(Tm, acc2, [](T a, int, int, T) { return a; });
    expk = expk * expk;
    __syncthreads();
  }

  // ---- apply_D (doubling.jl:178-252) + write the added layer -------------------------
  T* g_rmp = out.r_mp + (long long)s * out.mat_stride;
  T* g_tpp = out.t_pp + (long long)s * out.mat_stride;
  T* g_rpm = out.r_pm + (long long)s * out.mat_stride;
  T* g_tmm = out.t_mm + (long long)s * out.mat_stride;
  for (int e = tid; e < N * N; e += 256) {
    const int i = e % N, j = e / N;
    const int ix = lidx<NP>(i, j);
    T r = R[ix];
    const T t = Tm[ix];
    const bool ui = is_uv_row(i, ns), uj = is_uv_row(j, ns);
    if (ndoubl >= 1 && ui) r = -r;
    g_rmp[e] = r;
    g_tpp[e] = t;
    g_rpm[e] = (ui == uj) ? r : -r;
    g_tmm[e] = (ui == uj) ? t : -t;
  }
  if (tid < N) {
    T vjm = jm[tid];
    if (ndoubl >= 1 && is_uv_row(tid, ns)) vjm = -vjm;
    out.j0_p[(long long)s * N + tid] = jp[tid];
    out.j0_m[(long long)s * N + tid] = vjm;
  }
}

// ---------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11)  (interaction.jl:207-266)
// ---------------------------------------------------------------------------
template <typename T, int NP>
__device__ __forceinline__ void lds_to_global(T* __restrict__ dst, const T* L, int N) {
  for (int e = threadIdx.x; e < N * N; e += 256) dst[e] = L[lidx<NP>(e % N, e / N)];
}

template <typename T, int NP>
__global__ __launch_bounds__(256) void k_interaction11(int N, composite<T> c, added<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP>& sm = *reinterpret_cast<fsmem<T, NP>*>(smem_raw);
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
  constexpr int TPR = mv_map<T, NP>::TPR;
  const int mrow = tid / TPR;
  const bool mlead = (mrow < NP) && (tid % TPR == 0);

  acc_block<T, NP> acc;
  {
    stage_regs<T, NP> s1, s2;
    s1.load(R_pm, N);
    s2.load(r_mp, N);
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
  gl_operand<T, NP> opA;     // A operands streamed from global: T--, later t++
  gl_operand_b<T, NP> opB;   // B operands streamed from global: T++, t--
  opA.prefetch(T_mm, N);     // lands while G1 is being formed
  __syncthreads();
  // ---- G1 = (I - r-+ R+-)^-1 -> L3 -------------------------------------------------------
  acc.zero();
  mm_ll<T, NP>(acc, L2, L1, Kend);
  invert_one_minus<T, NP>(acc, L3, L4, N, Kend, sm, 0);
  // T01_inv = T-- G1 -> L4
  opB.prefetch(T_pp, N);
  acc.zero();
  opA.run(acc, L3);
  __syncthreads();
  acc_store<T, NP>(L4, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  // J0- += T01_inv (r-+ J0+ + j0-)
  {
    T y1, y2;
    matvec2<T, NP>(L2, vJp, vJp, N, y1, y2);
    if (mlead) vu[mrow] = y1 + vjm[mrow];
  }
  __syncthreads();
  {
    T y1, y2;
    matvec2<T, NP>(L4, vu, vu, N, y1, y2);
    if (mlead && mrow < N) J0_m[mrow] = vJm[mrow] + y1;
  }
  // R-+ += (T01_inv r-+) T++
  acc.zero();
  mm_ll<T, NP>(acc, L4, L2, Kend);
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });  // G1 is dead (barrier above)
  __syncthreads();
  acc.zero();
  opB.run(acc, L3);
  opB.prefetch(t_mm, N);
  __syncthreads();
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  for (int e = tid; e < N * N; e += 256) R_mp[e] += L3[lidx<NP>(e % N, e / N)];
  // T-- = T01_inv t--
  acc.zero();
  opB.run(acc, L4);
  opA.prefetch(t_pp, N);
  __syncthreads();  // R-+ update finished reading L3
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  lds_to_global<T, NP>(T_mm, L3, N);
  // ---- G2 = (I - R+- r-+)^-1 -> L3 -------------------------------------------------------
  acc.zero();
  mm_ll<T, NP>(acc, L1, L2, Kend);
  __syncthreads();  // T-- write-out finished reading L3; T01_inv (L4) is dead
  invert_one_minus<T, NP>(acc, L3, L4, N, Kend, sm, 0);
  // T21_inv = t++ G2 -> L4
  opB.prefetch(T_pp, N);  // pre-update T++
  acc.zero();
  opA.run(acc, L3);
  __syncthreads();
  acc_store<T, NP>(L4, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  // J0+ = j0+ + T21_inv (J0+ + R+- j0-)
  {
    T y1, y2;
    matvec2<T, NP>(L1, vjm, vjm, N, y1, y2);
    if (mlead) vz[mrow] = vJp[mrow] + y1;
  }
  __syncthreads();
  {
    T y1, y2;
    matvec2<T, NP>(L4, vz, vz, N, y1, y2);
    if (mlead && mrow < N) J0_p[mrow] = vjp[mrow] + y1;
  }
  // T++ = T21_inv T++
  acc.zero();
  opB.run(acc, L4);
  opB.prefetch(t_mm, N);
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });  // G2 is dead (barrier above)
  __syncthreads();
  lds_to_global<T, NP>(T_pp, L3, N);
  // R+- = r+- + (T21_inv R+-) t--
  acc.zero();
  mm_ll<T, NP>(acc, L4, L1, Kend);
  __syncthreads();  // T++ write-out finished reading L3
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  acc.zero();
  opB.run(acc, L3);
  __syncthreads();
  acc_store<T, NP>(L3, acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  for (int e = tid; e < N * N; e += 256) R_pm[e] = r_pm[e] + L3[lidx<NP>(e % N, e / N)];
}

// ---------------------------------------------------------------------------
// diagnostics: LDS tile product and LDS inverse on plain inputs
// ---------------------------------------------------------------------------
template <typename T, int NP>
__global__ __launch_bounds__(256) void k_test_mm(int N, const T* A, const T* B, T* C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP>& sm = *reinterpret_cast<fsmem<T, NP>*>(smem_raw);
  const long long o = (long long)blockIdx.x * N * N;
  const int Kend = ((N + 3) >> 2) << 2;
  stage<T, NP>(sm.L[0], A + o, N);
  stage<T, NP>(sm.L[1], B + o, N);
  __syncthreads();
  acc_block<T, NP> acc;
  acc.zero();
  mm_ll<T, NP>(acc, sm.L[0], sm.L[1], Kend);
  acc_store<T, NP>(sm.L[2], acc, [](T x, int, int, T) { return x; });
  __syncthreads();
  lds_to_global<T, NP>(C + o, sm.L[2], N);
}
template <typename T, int NP>
__global__ __launch_bounds__(256) void k_test_inv(int N, const T* A, T* X, int mode, int* path_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem<T, NP>& sm = *reinterpret_cast<fsmem<T, NP>*>(smem_raw);
  const long long o = (long long)blockIdx.x * N * N;
  const int Kend = ((N + 3) >> 2) << 2;
  // E = I - A, formed as a product so that it arrives in accumulator layout: E = (I - A) * I
  for (int e = threadIdx.x; e < NP * NP; e += 256) {
    const int i = e % NP, j = e / NP;
    const T av = (i < N && j < N) ? A[o + i + (long long)N * j] : ((i == j) ? T(1) : T(0));
    sm.L[0][lidx<NP>(i, j)] = ((i == j) ? T(1) : T(0)) - av;
    sm.L[1][lidx<NP>(i, j)] = (i == j) ? T(1) : T(0);
  }
  __syncthreads();
  acc_block<T, NP> acc;
  acc.zero();
  mm_ll<T, NP>(acc, sm.L[0], sm.L[1], NP);
  __syncthreads();
  const int path = invert_one_minus<T, NP>(acc, sm.L[2], sm.L[3], N, Kend, sm, mode);
  lds_to_global<T, NP>(X + o, sm.L[2], N);
  if (path_out && threadIdx.x == 0) path_out[blockIdx.x] = path;
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

template <typename K>
static int enable_lds(K kern, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  return VSM_OK;
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
  return dispatch_np<T>(q.N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    auto kern = k_elemental_doubling<T, NP>;
    const size_t bytes = sizeof(fsmem<T, NP>);
    static int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, q, m, ndoubl, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, a);
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
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    auto kern = k_interaction11<T, NP>;
    const size_t bytes = sizeof(fsmem<T, NP>);
    static int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, N, c, a);
    VSM_LAUNCH_CHECK("k_interaction11");
    return (int)VSM_OK;
  });
}

template <typename T>
int test_lds_mm(int N, int S, const T* A, const T* B, T* C, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    auto kern = k_test_mm<T, NP>;
    const size_t bytes = sizeof(fsmem<T, NP>);
    static int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, N, A, B, C);
    VSM_LAUNCH_CHECK("k_test_mm");
    return (int)VSM_OK;
  });
}

template <typename T>
int test_lds_inv(int N, int S, const T* A, T* X, int mode, int* path_out, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  return dispatch_np<T>(N, [&](auto tag) {
    constexpr int NP = decltype(tag)::value;
    auto kern = k_test_inv<T, NP>;
    const size_t bytes = sizeof(fsmem<T, NP>);
    static int prepared = enable_lds(kern, bytes);
    if (prepared) return prepared;
    hipLaunchKernelGGL(kern, dim3(S), dim3(256), bytes, st, N, A, X, mode, path_out);
    VSM_LAUNCH_CHECK("k_test_inv");
    return (int)VSM_OK;
  });
}

#define VSM_INST_F(T)                                                                                               \
  template int fused_elemental_doubling<T>(const quad<T>&, int, int, int, const T*, const T*, const T*, const T*,  \
                                           const T*, const T*, long long, const added<T>&, hipStream_t);           \
  template int fused_interaction<T>(int, int, int, const composite<T>&, const added<T>&, hipStream_t);              \
  template int test_lds_mm<T>(int, int, const T*, const T*, T*, hipStream_t);                                       \
  template int test_lds_inv<T>(int, int, const T*, T*, int, int*, hipStream_t);
VSM_INST_F(double)
VSM_INST_F(float)

}  // namespace vsm
