// Column-strip kernels, FP32 edition (64 < N <= 96): the fused layer step of vsm_strip.hip for the hyperspectral
// configuration (N = 96, Float32).  Every matrix lives as 16-column strips in the accumulator registers of the wave that
// owns the strip, B operands come straight from those registers, A operands from one of TWO LDS buffers.
//
// Shape of a workgroup: 6 waves (6 strips of 16 columns; 6 row tiles of 4 f32 = 24 VGPRs per strip), built for THREE waves
// per SIMD (<= 168 registers): two workgroups per CU = 12 waves = 3 per SIMD.  (One 6-wave workgroup per CU leaves the four
// SIMDs with 2, 2, 1, 1 waves -- at most 75 % of the MFMA rate -- and every barrier and LDS round trip exposed.)
//
// LDS layout ("A-form", k-major): element (row, k) of an A operand at word  k + LDK * row,  LDK = 104.
//   * v_mfma_f32_16x16x4_f32 keeps accumulator element r of a tile at row 4 (lane>>4) + r, so a strip tile used as B operand
//     of "k-step r" covers k = 16 tb + 4 kq + r (kq = lane>>4): the four k-steps of a block need A[row][16 tb + 4 kq + 0..3] --
//     16 contiguous bytes: ONE ds_read_b128 feeds four MFMAs (the earlier layout: four ds_read_b32);
//   * every LDS address is "one per-lane base + immediate":  fragment (row 16 t + l15)  = afrag + 16 tb + 16 LDK t,
//     strip element (row 16 ta + 4 kq + r, column col) = sbase + LDK (16 ta + r) -- no XOR swizzle, nothing for the
//     compiler to hoist;
//   * LDK = 104: the b128 fragment reads are conflict-free in the hardware's 4 x 16 lane groups (checked exhaustively for
//     LDK in 96..128; 100 / 108 are 2-way), the strip stores are 2-way on ds_write_b32, which is free (the instruction's
//     VGPR transfer costs as much);
//   * the 8 padding words of each row of a buffer are 8 strided vector slots; the Gauss-Jordan fallback keeps its scratch
//     there (in the padding of the very matrix it inverts), so two buffers + five contiguous vectors = 81 860 B: two
//     workgroups per CU.
// N = 96 leaves no spare column for the source vectors: their products  tt j, tmp j, r J0+, T01 u, R+- j0-, T21 z  are VALU
// mat-vecs over the A-form (the A-fragment pattern of the wave's own row tile: conflict-free; two shuffles).
// Rows / columns >= N of every strip and every A-form are exactly zero BY CONSTRUCTION (the elemental step and the global
// loads mask them, products of zero-padded operands stay zero, the identity is only added where row < N), so no store needs
// a mask.
#include <stdlib.h>

#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_lds.h"
#include "vsm_elemental.h"

namespace vsm {

#ifdef VSM_PHASE_TIMING
__device__ unsigned long long vsm_phase_cycles_strip32[32];
#define VSM_STAMP_DECL unsigned long long _t_prev = __builtin_readcyclecounter()
#define VSM_STAMP(i)                                                     \
  do {                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                           \
      const unsigned long long _t = __builtin_readcyclecounter();        \
      vsm_phase_cycles_strip32[i] += _t - _t_prev;                       \
      _t_prev = _t;                                                      \
    }                                                                    \
  } while (0)
#else
#define VSM_STAMP_DECL
#define VSM_STAMP(i)
#endif

namespace {

constexpr int FNP = 96;        // padded matrix size
constexpr int FNW = 6;         // waves = column strips
constexpr int FNT = 64 * FNW;
constexpr int FTL = FNP / 16;  // row tiles per strip
constexpr int LDK = 104;       // words per row of an A-form
constexpr int NVEC = 5;
// Added layer of the elemental pre-pass (k_elemental_img32) in global memory, per (moment, point): the A-form images [r-+*], [t++]
// exactly as the doubling loop wants them in LDS (96 rows of LDK words, zero padding), then j0+[96], j0-[96], aux[96]
// (aux[0] = exp(-dtau / mu_0)).  The layer kernel copies the images with global_load_lds_dwordx4.
constexpr int PRE32_IMG = FNP * LDK;
constexpr int PRE32_STRIDE = 2 * PRE32_IMG + 3 * FNP;

struct fstrip {
  f4_t v[FTL];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < FTL; ++a) v[a] = acc_zero<float>();
  }
};

struct __attribute__((aligned(16))) fsmem32 {   // LDS of ONE spectral point (one half of a workgroup); 16-byte multiple: b128 reads
  float P[FNP * LDK];
  float Q[FNP * LDK];
  float vec[NVEC][FNP];
  float red[2][8];
  int gj_info;
  unsigned bar;    // arrival counter of the half's software barrier
};
static_assert(2 * sizeof(fsmem32) <= 163840 && sizeof(fsmem32) % 16 == 0, "two spectral points per CU, both 16-byte aligned");

// Gauss-Jordan scratch in the padding words (k = 96 .. 103) of the rows of the matrix being inverted
struct pad_vec {
  float* b;
  __device__ __forceinline__ float& operator[](int i) const { return b[LDK * i]; }
};
struct pad_vec2 {
  float* b;
  __device__ __forceinline__ pad_vec operator[](int par) const { return pad_vec{b + par}; }
};
struct pad_ivec {
  int* b;
  __device__ __forceinline__ int& operator[](int i) const { return b[LDK * i]; }
};
struct gj_pad_scratch {
  pad_vec2 col, rowP, rowK;
  pad_ivec piv, dst;
  int& info;
};

struct fpos {
  int s;                               // spectral point of this half
  int tid, lane, wave, l15, kq, col;   // tid / wave: within the half (0..383 / 0..5)
  unsigned* bar;
  mutable unsigned epoch;
  bool active;                         // false: the odd point out of a workgroup (lockstep build: computes, never stores)
  const float* red_other;              // lockstep build: the other half's norm partials
  int afrag;   // 4 kq + LDK l15
  int sbase;   // col + 4 LDK kq
  int dr;      // l15 - 4 kq: the diagonal of the wave's strip is element r == dr of tile ta == wave (when 0 <= dr < 4)
  __device__ __forceinline__ fpos(int s_, int tid_, unsigned* bar_) {
    s = s_;
    active = true;
    red_other = nullptr;
    tid = tid_;
    bar = bar_;
    epoch = 0;
    lane = tid & 63;
    wave = tid >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    afrag = 4 * kq + LDK * l15;
    sbase = col + 4 * LDK * kq;
    dr = l15 - 4 * kq;
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + 4 * kq + r; }
};

__device__ __forceinline__ f4_t lds4(const float* p) { return *reinterpret_cast<const f4_t*>(p); }

// Barriers.  A 6-wave workgroup lands on the four SIMDs as 2, 2, 1, 1 waves and the dispatcher starts every workgroup on the
// same SIMD, so a second 6-wave workgroup never fits beside the first at three waves per SIMD (residency census:
// tools/occ_probe.hip).  The kernels therefore run TWO spectral points per 12-wave workgroup: 3 waves on every SIMD.
// The two points walk ONE barrier sequence in lockstep (hardware s_barrier): in every product phase each SIMD then carries
// three equally loaded waves, where independent halves leave the 2-2-1-1 imbalance of whichever half is alone in its MFMA
// phase (measured on C4: independent halves with an LDS arrival-counter barrier 11.9k points/s, lockstep 12.8k).  The only
// data-dependent barrier count is the inverse (series order / Gauss-Jordan): after the shared norm reduction every wave knows
// BOTH points' orders and the point that needs fewer barriers pads its sequence (invert_strip) -- the arithmetic of a point
// never depends on its neighbour, so results are independent of the pairing.  The fences order LDS traffic only: global
// loads and stores are not drained at a barrier (no wave reads global data another wave wrote after the start of the kernel:
// each wave loads and stores its own 16 columns of the composite; J0+- are read at the start, many barriers before they are
// written).
__device__ __forceinline__ void half_barrier(const fpos& p) {
  // lockstep build: both points of the workgroup walk the same barrier sequence (the hardware barrier; LDS-only fences)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
struct half_sync {
  const fpos* p;
  __device__ __forceinline__ void operator()() const { half_barrier(*p); }
};

// acc += A * B   (A: A-form in LDS, B: strip in registers); KB = ceil(N / 16) blocks of four MFMA k-steps.
// Row tiles go in pairs (two independent accumulators alternate: the 40-cycle dependent latency of the f32 MFMA never
// stalls its 32-cycle issue); the fragments of the next pair are requested before the MFMAs of the current one.
template <int KB>
__device__ __forceinline__ void mm_ab(fstrip& acc, const float* A, const fstrip& B, const fpos& p) {
  const float* a0 = A + p.afrag;
#ifndef VSM_MM_PAIRS
  // variant: the six fragments of a k-block are requested together, then 24 MFMAs walk r-major over the six accumulators
#pragma unroll
  for (int tb = 0; tb < KB; ++tb) {
    f4_t fb[FTL];
#pragma unroll
    for (int t = 0; t < FTL; ++t) fb[t] = lds4(a0 + 16 * tb + 16 * LDK * t);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int t = 0; t < FTL; ++t) acc.v[t] = mfma<float>::mma(fb[t][r], B.v[tb][r], acc.v[t]);
  }
  return;
#endif
  constexpr int NP = 3 * KB;
  f4_t fa[2][2];
  fa[0][0] = lds4(a0);
  fa[0][1] = lds4(a0 + 16 * LDK);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i + 1 < NP) {
      const int tb = (i + 1) / 3, t0 = 2 * ((i + 1) % 3);
      fa[(i + 1) & 1][0] = lds4(a0 + 16 * tb + 16 * LDK * t0);
      fa[(i + 1) & 1][1] = lds4(a0 + 16 * tb + 16 * LDK * (t0 + 1));
    }
    const int tb = i / 3, t0 = 2 * (i % 3);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc.v[t0] = mfma<float>::mma(fa[i & 1][0][r], B.v[tb][r], acc.v[t0]);
      acc.v[t0 + 1] = mfma<float>::mma(fa[i & 1][1][r], B.v[tb][r], acc.v[t0 + 1]);
    }
  }
}
// acc1 += A * B1 ; acc2 += A * B2   (shared A fragments: eight MFMAs per ds_read_b128)
template <int KB>
__device__ __forceinline__ void mm_ab2(fstrip& acc1, fstrip& acc2, const float* A, const fstrip& B1, const fstrip& B2,
                                       const fpos& p) {
  const float* a0 = A + p.afrag;
  constexpr int NP = 6 * KB;
  f4_t fa[2];
  fa[0] = lds4(a0);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    if (i + 1 < NP) {
      const int tb = (i + 1) / 6, t = (i + 1) % 6;
      fa[(i + 1) & 1] = lds4(a0 + 16 * tb + 16 * LDK * t);
    }
    const int tb = i / 6, t = i % 6;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc1.v[t] = mfma<float>::mma(fa[i & 1][r], B1.v[tb][r], acc1.v[t]);
      acc2.v[t] = mfma<float>::mma(fa[i & 1][r], B2.v[tb][r], acc2.v[t]);
    }
  }
}

__device__ __forceinline__ void store_strip(float* dst, const fstrip& s, const fpos& p) {
  float* d0 = dst + p.sbase;
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) d0[LDK * (16 * ta + r)] = s.v[ta][r];
}
__device__ __forceinline__ void load_strip(fstrip& s, const float* src, const fpos& p) {
  const float* s0 = src + p.sbase;
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = s0[LDK * (16 * ta + r)];
}
// Strip <-> global column-major N x N.  A lane's four elements of a tile are four consecutive rows of one column: one
// 16-byte access when N % 4 == 0 (then the rows of a tile are all inside or all outside the matrix).
// (decided at run time although AL is known: with a compile-time branch the scheduler hoists every global load to the top of
// its phase and pays for it in spills -- measured 8.4k vs 9.7k points/s on C4)
#if defined(VSM_AB_SWITCHES) && defined(VSM_AL_STATIC)   // (an ablation of the diagnostic build only)
#define VSM_ALIGNED(AL, N) (AL)
#else
#define VSM_ALIGNED(AL, N) (((N) & 3) == 0)
#endif
template <bool AL>   // AL: N % 4 == 0
__device__ __forceinline__ void load_strip_global(fstrip& s, const float* __restrict__ g, int N, const fpos& p) {
  const int cc = min(p.col, N - 1);
  const float* g0 = g + (long long)N * cc + 4 * p.kq;
  if (VSM_ALIGNED(AL, N)) {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta) {
      const bool in = p.col < N && 16 * ta + 4 * p.kq < N;
      const f4_t v = *reinterpret_cast<const f4_t*>(in ? g0 + 16 * ta : g);
      s.v[ta] = in ? v : acc_zero<float>();
    }
  } else {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        const float v = g[min(row, N - 1) + (long long)N * cc];
        s.v[ta][r] = (row < N && p.col < N) ? v : 0.0f;
      }
  }
}
template <bool AL>
__device__ __forceinline__ void store_strip_global(float* __restrict__ g, const fstrip& s, int N, const fpos& p) {
  if (p.col >= N || !p.active) return;
  float* g0 = g + (long long)N * p.col + 4 * p.kq;
  if (VSM_ALIGNED(AL, N)) {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta)
      if (16 * ta + 4 * p.kq < N) *reinterpret_cast<f4_t*>(g0 + 16 * ta) = s.v[ta];
  } else {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (p.row(ta, r) < N) g0[16 * ta + r] = s.v[ta][r];
  }
}
// D X D: sign (+) where row and column have the same U/V parity (doubling.jl:178-201)
__device__ __forceinline__ void dsym_strip(fstrip& d, const fstrip& x, int ns, const fpos& p) {
  const bool uc = is_uv_row(p.col, ns);
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) d.v[ta][r] = (is_uv_row(p.row(ta, r), ns) == uc) ? x.v[ta][r] : -x.v[ta][r];
}
// y1 = A x1, y2 = scale2 * A x2 over the A-form in LDS: lane (kq, l15) of wave w walks row 16 w + l15 over the columns
// k = 16 tb + 4 kq + 0..3 -- the A-fragment reads of row tile w -- and the four kq groups are summed with two shuffles
// (every lane receives the sums).  x1 / x2: contiguous vectors, entries >= N zero.
template <int KB>
__device__ __forceinline__ void matvec2(const float* A, const float* x1, const float* x2, float scale2, float& y1, float& y2,
                                        const fpos& p) {
  const float* a0 = A + p.afrag + 16 * LDK * p.wave;
  const float* u0 = x1 + 4 * p.kq;
  const float* v0 = x2 + 4 * p.kq;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int tb = 0; tb < KB; ++tb) {
    const f4_t a = lds4(a0 + 16 * tb), u = lds4(u0 + 16 * tb), v = lds4(v0 + 16 * tb);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1 += a[r] * u[r];
      s2 += a[r] * v[r];
    }
  }
  s2 *= scale2;
  s1 += __shfl_xor(s1, 16);
  s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  y1 = s1;
  y2 = s2;
}
template <int KB>
__device__ __forceinline__ float matvec1(const float* A, const float* x, const fpos& p) {
  const float* a0 = A + p.afrag + 16 * LDK * p.wave;
  const float* u0 = x + 4 * p.kq;
  float s1 = 0.f;
#pragma unroll
  for (int tb = 0; tb < KB; ++tb) {
    const f4_t a = lds4(a0 + 16 * tb), u = lds4(u0 + 16 * tb);
#pragma unroll
    for (int r = 0; r < 4; ++r) s1 += a[r] * u[r];
  }
  s1 += __shfl_xor(s1, 16);
  s1 += __shfl_xor(s1, 32);
  return s1;
}

// Frobenius-norm bound of the block whose strips the waves hold (padding is zero).  Contains ONE barrier.
__device__ __forceinline__ float strip_norm_bound(const fstrip& e, fsmem32& sm, int& slot, const fpos& p, float& nrm_other) {
  float ss = 0;
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss += e.v[ta][r] * e.v[ta][r];
  const float ws = wave_sum(ss * 1.0001f);
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  half_barrier(p);
  float tot = 0.f, tot2 = 0.f;
#pragma unroll
  for (int w = 0; w < FNW; ++w) tot += sm.red[slot][w];
#pragma unroll
  for (int w = 0; w < FNW; ++w) tot2 += p.red_other[8 * slot + w];
  slot ^= 1;
  nrm_other = sqrtf(tot2) * 1.001f;
  return sqrtf(tot) * 1.001f;
}
// Series order for a norm bound (0: Gauss-Jordan) and the number of barriers invert_strip executes after the norm reduction
__device__ __forceinline__ int series_order(float nrm) {
  const float tol = num<float>::eps() * 0.25f;
  int K = 0;
  if (nrm < 0.3f) {
    const float lim = tol * (1.0f - nrm);
    const float n2 = nrm * nrm, n4 = n2 * n2, n8 = n4 * n4, n16 = n8 * n8;
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
  return K;
}
__device__ __forceinline__ int inverse_barriers(int K, int N) {
  if (K == 0) return 2 * N + 4;   // store | 2 per pivot step | permutation, end of gj_invert | end of gj_lds_strip
  if (K == 1) return 0;
  if (K == 2) return 1;
  if (K <= 4) return 3;
  if (K <= 8) return 5;
  if (K <= 16) return 7;
  return 9;
}

// In-place pivoted Gauss-Jordan of the A-form matrix V (N x N block, identity-padded), 384 threads; ends with a barrier.
__device__ __forceinline__ void gj_lds_strip(float* V, int N, fsmem32& sm, const fpos& p) {
  using G = gj_cfg<FNP, FNT>;
  const int tr = p.tid % G::TR, tc = p.tid / G::TR;
  float g[G::RB][G::CB];
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      g[rb][cb] = (i < N && j < N) ? V[j + LDK * i] : ((i == j) ? 1.0f : 0.0f);
    }
  float* pad = V + FNP;
  gj_pad_scratch sc{pad_vec2{pad}, pad_vec2{pad + 2}, pad_vec2{pad + 4}, pad_ivec{reinterpret_cast<int*>(pad + 6)},
                    pad_ivec{reinterpret_cast<int*>(pad + 7)}, sm.gj_info};
  gj_invert<float, FNP, FNT, false>(g, N, sc, p.tid, half_sync{&p});
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      if (i < N && j < N) V[sc.dst[j] + LDK * i] = g[rb][cb];
    }
  half_barrier(p);
}

// x + I on the rows < N of the strip
__device__ __forceinline__ void add_identity(fstrip& x, int N, const fpos& p) {
  const bool d = p.dr >= 0 && p.dr < 4 && p.col < N;
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
    if (ta == p.wave) {   // (wave-uniform)
#pragma unroll
      for (int r = 0; r < 4; ++r) x.v[ta][r] += (d && r == p.dr) ? 1.0f : 0.0f;
    }
}

// G_s = strip of (I - E)^-1, E given as strips (see invert_strip in vsm_strip_dev.h).  W: A-form scratch.  Returns after
// a point where other waves may still be READING W: barrier before overwriting it.
template <int KB>
__device__ __forceinline__ int invert_strip_own(fstrip& E, fstrip& G, float* W, int N, fsmem32& sm, const fpos& p, int K);
template <int KB>
__device__ __forceinline__ int invert_strip(fstrip& E, fstrip& G, float* W, int N, fsmem32& sm, int& slot, const fpos& p) {
  float nrm_other;
  const float nrm = strip_norm_bound(E, sm, slot, p, nrm_other);
  const int K = series_order(nrm);
  const int rc = invert_strip_own<KB>(E, G, W, N, sm, p, K);
  // the other point of the workgroup may need more barriers for its inverse: keep the two barrier sequences equal
  const int own = inverse_barriers(K, N), oth = inverse_barriers(series_order(nrm_other), N);
  for (int i = own; i < oth; ++i) half_barrier(p);
  return rc;
}
template <int KB>
__device__ __forceinline__ int invert_strip_own(fstrip& E, fstrip& G, float* W, int N, fsmem32& sm, const fpos& p, int K) {
  if (K == 0) {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta) G.v[ta] = -E.v[ta];
    add_identity(G, N, p);
    store_strip(W, G, p);
    half_barrier(p);
    gj_lds_strip(W, N, sm, p);
    load_strip(G, W, p);
    return 1;
  }
  G = E;
  add_identity(G, N, p);
  if (K == 1) return 2;
  store_strip(W, E, p);
  half_barrier(p);
  int cur = 1;  // W = E^cur (A-form), E = its strip, G = strip of sum_{k < 2 cur} E^k
  for (;;) {
    fstrip W2;
    W2.zero();
    mm_ab<KB>(W2, W, E, p);  // E^(2 cur)
    cur *= 2;
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < FTL; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    half_barrier(p);
    store_strip(W, W2, p);
    half_barrier(p);
    {
      fstrip& T = W2;        // (its value now lives in W; the strip is re-read from there if another squaring follows)
      T.zero();
      mm_ab<KB>(T, W, G, p);   // E^cur * G   (powers of E commute)
#pragma unroll
      for (int ta = 0; ta < FTL; ++ta) G.v[ta] += T.v[ta];
    }
    if (K == 2 * cur - 1) break;
    load_strip(E, W, p);
  }
  return 1 + K;
}

// ---------------------------------------------------------------------------
// elemental! + doubling! + apply_D!  (see ed_body in vsm_strip.hip; sources by mat-vec)
// On return: r_s = strip of r-+, t_s = strip of t++, sm.vec[2 jpair] = j0+, sm.vec[2 jpair + 1] = j0-, all waves past a
// barrier.
// ---------------------------------------------------------------------------
// THERMAL: the `:thermal` per-source slot instead of the solar beam (see vsm_strip.hip): F0 = B[S], expk = 1
// LDS DMA copy of one A-form image (PRE32_IMG words = 2496 x 16 B) by the 6 waves of a half: 6.5 rounds of 6 x 1 KB
__device__ __forceinline__ void copy_image_to_lds32(float* L, const float* __restrict__ g, const fpos& p) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int blk = (FNW * i + p.wave) * 256;   // words
    if (blk < PRE32_IMG)                        // (wave-uniform; false only for waves 3..5 of the last round)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blk + 4 * p.lane),
                                       (__attribute__((address_space(3))) void*)(L + blk), 16, 0, 0);
  }
}

// PRE: the elemental layer comes from the pre-pass (k_elemental_img32) as two A-form images + vectors at `img`.
template <int KB, bool MIX, bool THERMAL = false, bool PRE = false>
__device__ __forceinline__ void ed_body(fsmem32& sm, fpos& p, const quad<float>& q, int m, int ndoubl,
                                        const float* __restrict__ dtau, const float* __restrict__ varpi,
                                        const float* __restrict__ tau_sum, const float* __restrict__ F0,
                                        const zsrc<float>& z, fstrip& r_s, fstrip& t_s, int& jpair,
                                        const float* __restrict__ img = nullptr) {
  float* P = sm.P;
  float* Q = sm.Q;
  float expk0 = 1.0f;
  if constexpr (PRE) {
    const int N = q.N, tid = p.tid;
    copy_image_to_lds32(P, img, p);
    copy_image_to_lds32(Q, img + PRE32_IMG, p);
    float vjp = 0.0f, vjm = 0.0f;
    if (tid < FNP) {
      vjp = img[2 * PRE32_IMG + tid];
      vjm = img[2 * PRE32_IMG + FNP + tid];
    }
    expk0 = img[2 * PRE32_IMG + 2 * FNP];
    (void)N;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA writes have landed in LDS
    if (tid < FNP) {
      sm.vec[0][tid] = vjp;
      sm.vec[1][tid] = vjm;
    }
    jpair = 0;
    half_barrier(p);
    load_strip(r_s, P, p);
    load_strip(t_s, Q, p);
  } else {
    float* mus = sm.vec[0];
    float* wcs = sm.vec[1];
    float* xs = sm.vec[2];
    float* es = sm.vec[3];
    float* ems = sm.vec[4];
    const int s = p.s;
    const int N = q.N, ns = q.n_stokes;
    const int tid = p.tid;
    const float d = dtau[s], w = varpi[s];
    const int ncomp = MIX ? z.ncomp : 0;
    const long long NNz = (long long)q.N * q.N;
    const float* Zp = z.Zpp + (ncomp ? 0 : (long long)s * z.zs);
    const float* Zm = z.Zmp + (ncomp ? 0 : (long long)s * z.zs);
    float fk[4] = {0.f, 0.f, 0.f, 0.f};
  #pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ncomp) fk[k] = z.fcomp[(long long)s * ncomp + k];
    auto zget = [&](const float* Z, long long zo) {
      if (ncomp == 0) return Z[zo];
      float acc = 0.f;
  #pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < ncomp) acc += fk[k] * Z[k * NNz + zo];
      return acc;
    };

    if (tid < FNP) {
      mus[tid] = (tid < N) ? q.mu[tid] : 1.0f;
      const float wt = (tid < N) ? q.wt[tid] : 0.0f;
      wcs[tid] = (m == 0) ? wt / 2.0f : wt / 4.0f;
      const float x = d / mus[tid];
      xs[tid] = x;
      es[tid] = exp(-x);
      ems[tid] = expm1(-x);
    }
    half_barrier(p);

    // ---- elemental (elemental.jl:289-334) ------------------------------------------------------------------------
    {
      const int j = p.col;
      const int jc = min(j, N - 1);
      const float mj = mus[j], wct = wcs[j], xj = xs[j], emj = ems[j], ej = es[j];
  #pragma unroll
      for (int ta = 0; ta < FTL; ++ta) {
        float zp[4], zm[4];
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long long zo = min(p.row(ta, r), N - 1) + (long long)N * jc;
          zp[r] = zget(Zp, zo);
          zm[r] = zget(Zm, zo);
        }
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = p.row(ta, r);
          float rr = 0.0f, tt = 0.0f;
          if (i < N && j < N) {
            const float mi = mus[i], xi = xs[i];
            if (wct > num<float>::eps()) {
              const float emi = ems[i];
              rr = w * zm[r] * (mj / (mi + mj)) * wct * (-(emi + emj + emi * emj));
              if (mi == mj) {
                if (i == j)
                  tt = es[i] * (1.0f + w * zp[r] * xi * wct);
                else
                  tt = ej * (w * zp[r] * xi * wct);
              } else {
                const float xm = fmax(xi, xj);
                const float ediff =
                    (xm < 0.5f && fabs(xi - xj) > 0.125f * xm) ? (emi - emj) : expdiff_neg<float>(xi, xj);
                tt = w * zp[r] * (mj / (mi - mj)) * wct * ediff;
              }
            } else {
              tt = (i == j) ? es[i] : 0.0f;
            }
            if (ndoubl >= 1 && is_uv_row(i, ns)) rr = -rr;
          }
          r_s.v[ta][r] = rr;
          t_s.v[ta][r] = tt;
        }
      }
    }
    // ---- SFI source (elemental.jl:348-392) -----------------------------------------------------------------------
    float vjp = 0.0f, vjm = 0.0f;
    if (THERMAL) {
      if (tid < N && tid % ns == 0 && mus[tid] > num<float>::eps())
        vjp = vjm = 6.283185307179586476925286766559f * (1.0f - w) * F0[s] * (-expm1(-d / mus[tid]));
    } else if (tid < N) {
      const int i = tid;
      const int i_start = ns * q.i_mu0;
      const float wct02 = (m == 0) ? 0.5f : 0.25f;
      float zp = 0, zm = 0;
      for (int qq = 0; qq < ns; ++qq) {
        const long long zo = i + (long long)N * (i_start + qq);
        const float f = F0[qq + (long long)ns * s];
        zp += zget(Zp, zo) * f;
        zm += zget(Zm, zo) * f;
      }
      const float mi = mus[i], ms = mus[i_start];
      if (i >= i_start && i < i_start + ns)
        vjp = wct02 * w * zp * (d / mi) * exp(-d / mi);
      else
        vjp = wct02 * w * zp * (ms / (mi - ms)) * expdiff_neg<float>(d / mi, d / ms);
      vjm = wct02 * w * zm * (ms / (mi + ms)) * (-expm1(-d * ((1.0f / mi) + (1.0f / ms))));
      const float att = exp(-tau_sum[s] / ms);
      vjp *= att;
      vjm *= att;
      if (ndoubl >= 1 && is_uv_row(i, ns)) vjm = -vjm;
    }
    if (ndoubl > 0) {
      store_strip(P, r_s, p);
      store_strip(Q, t_s, p);
    }
    half_barrier(p);  // (the helper vectors mus.. are dead from here on)
    jpair = 0;
    if (tid < FNP) {
      sm.vec[0][tid] = vjp;
      sm.vec[1][tid] = vjm;
    }
    half_barrier(p);
    expk0 = THERMAL ? 1.0f : exp(-d / q.mu0);
  }

  // ---- doubling (rt_helpers.jl:102-166) ------------------------------------------------------------------------
  const int N = q.N, ns = q.n_stokes;
  const int tid = p.tid;
  float expk = expk0;
  int slot = 0;
  const int mrow = 16 * p.wave + p.l15;   // row of the mat-vecs
  const bool mlead = p.kq == 0;
  VSM_STAMP_DECL;
  VSM_STAMP(0);
  for (int n = 0; n < ndoubl; ++n) {
    // on entry: P = r, Q = t (A-form), r_s in registers, all waves past a barrier
    const float* jp = sm.vec[2 * jpair];
    const float* jm = sm.vec[2 * jpair + 1];
    float* njp = sm.vec[2 * (jpair ^ 1)];
    float* njm = sm.vec[2 * (jpair ^ 1) + 1];
    fstrip G;
    {
      fstrip E;
      E.zero();
      mm_ab<KB>(E, P, r_s, p);
      VSM_STAMP(1);
      invert_strip<KB>(E, G, P, N, sm, slot, p);
      VSM_STAMP(2);
    }
    fstrip tt;
    tt.zero();
    mm_ab<KB>(tt, Q, G, p);
    load_strip(t_s, Q, p);  // t's strip is not kept in registers across the inverse
    half_barrier(p);        // P (series powers) and Q (t) no longer read
    store_strip(P, tt, p);
    half_barrier(p);        // tt complete in P
    VSM_STAMP(3);
    float a1, a2;           // tt j0+ , tt j1-  (j1- = j0- expk)
    matvec2<KB>(P, jp, jm, expk, a1, a2, p);
    VSM_STAMP(4);
    fstrip tn;
    {
      fstrip tmp;
      tmp.zero();
      tn.zero();
      mm_ab2<KB>(tmp, tn, P, r_s, t_s, p);
      store_strip(Q, tmp, p);
    }
    half_barrier(p);        // tmp complete in Q
    VSM_STAMP(5);
    float b1, b2;           // tmp j0+ , tmp j1-
    matvec2<KB>(Q, jp, jm, expk, b1, b2, p);
    VSM_STAMP(6);
    mm_ab<KB>(r_s, Q, t_s, p);  // r' = r + tmp t
    t_s = tn;
    // j0- <- j0- + tt j1- + tmp j0+ ; j0+ <- j1+ + tt j0+ + tmp j1-   (rt_helpers.jl:128-134), into the other pair
    if (mlead) {
      njm[mrow] = (mrow < N) ? jm[mrow] + a2 + b1 : 0.0f;
      njp[mrow] = (mrow < N) ? jp[mrow] * expk + a1 + b2 : 0.0f;
    }
    jpair ^= 1;
    expk = expk * expk;
    half_barrier(p);  // everybody finished reading P (tt), Q (tmp), the old sources
    VSM_STAMP(7);
    if (n + 1 < ndoubl) {
      store_strip(P, r_s, p);
      store_strip(Q, t_s, p);
      half_barrier(p);
    }
    VSM_STAMP(8);
  }

  // ---- apply_D (doubling.jl:178-252) -----------------------------------------------------------------------------
  if (ndoubl >= 1) {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (is_uv_row(p.row(ta, r), ns)) r_s.v[ta][r] = -r_s.v[ta][r];
    float* jm = sm.vec[2 * jpair + 1];
    if (tid < FNP && is_uv_row(tid, ns)) jm[tid] = -jm[tid];
  }
  half_barrier(p);
}

// ---------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11)  (interaction.jl:207-266) with ONE inverse G2 = (I - R+- r-+)^-1 and the
// push-through identities -- the algebra of ia_body in vsm_strip.hip (ten products + the series, seven barriers + those of the
// series; every composite matrix crosses the memory system once in each direction):
//   [E2 | Z] = R+- [r-+ | t--] ; [S | V] = T-- [r-+ | t--] ; T21 = t++ G2 ; Y = S G2
//   [R+- | T++] = [r+- | 0] + T21 [Z | T++] ;  [R-+ | T--] = [R-+ | V] + Y [T++ | Z]
//   z = J0+ + R+- j0- ; J0+ = j0+ + T21 z ; J0- = J0- + T-- j0- + Y z          (N = 96 has no spare column: VALU mat-vecs)
// On entry: r_s / t_s = strips of the added layer's r-+ / t++, sm.vec[2 jpair] / [2 jpair + 1] = its j0+ / j0-, all waves
// past a barrier, P and Q free.  ns > 0: r+- = D r-+ D, t-- = D t++ D; ns == 0: read from r_pm / t_mm (surface layers).
// ---------------------------------------------------------------------------
template <int KB, bool AL>
__device__ __forceinline__ void ia_body(fsmem32& sm, fpos& p, int N, int ns, const composite<float>& c, fstrip& r_s,
                                        fstrip& t_s, const float* __restrict__ r_pm, const float* __restrict__ t_mm,
                                        int jpair) {
  float* P = sm.P;
  float* Q = sm.Q;
  const float* vjp = sm.vec[2 * jpair];
  const float* vjm = sm.vec[2 * jpair + 1];
  float* vz = sm.vec[2 * (jpair ^ 1)];
  const int s = p.s;
  const long long NN = (long long)N * N;
  float* R_mp = c.R_mp + s * NN;
  float* R_pm = c.R_pm + s * NN;
  float* T_pp = c.T_pp + s * NN;
  float* T_mm = c.T_mm + s * NN;
  float* J0_p = c.J0_p + (long long)s * N;
  float* J0_m = c.J0_m + (long long)s * N;
  const int mrow = 16 * p.wave + p.l15;   // row of the mat-vecs
  const bool mlead = p.kq == 0;
  int slot = 0;

  const float Jp_old = (mlead && mrow < N) ? J0_p[mrow] : 0.0f;
  const float Jm_old = (mlead && mrow < N) ? J0_m[mrow] : 0.0f;
  VSM_STAMP_DECL;
  // ---- stage: [R+-] -> P, [T--] -> Q (each wave moves its own 16 columns: global strip -> A-form) -----------------------
  {
    fstrip Y1, Y2;
    load_strip_global<AL>(Y1, R_pm, N, p);
    load_strip_global<AL>(Y2, T_mm, N, p);
    store_strip(P, Y1, p);
    store_strip(Q, Y2, p);
  }
  half_barrier(p);                                                                                       // (a)
  VSM_STAMP(10);
  // z = J0+ + R+- j0- ; vs = T-- j0-
  float vs_keep;
  {
    const float y1 = matvec1<KB>(P, vjm, p);
    vs_keep = matvec1<KB>(Q, vjm, p);
    if (mlead) vz[mrow] = (mrow < N) ? Jp_old + y1 : 0.0f;
  }
  fstrip Z, V, G;
  {
    fstrip E, Sx;
    if (ns) {   // t-- = D t++ D formed in place and undone afterwards (D is an involution)
      dsym_strip(t_s, t_s, ns, p);
      E.zero();
      Z.zero();
      mm_ab2<KB>(E, Z, P, r_s, t_s, p);
      Sx.zero();
      V.zero();
      mm_ab2<KB>(Sx, V, Q, r_s, t_s, p);
      dsym_strip(t_s, t_s, ns, p);
      dsym_strip(r_s, r_s, ns, p);      // r_s <- r+- (the accumulator of the R+- update)
    } else {
      fstrip tmm;
      load_strip_global<AL>(tmm, t_mm, N, p);
      E.zero();
      Z.zero();
      mm_ab2<KB>(E, Z, P, r_s, tmm, p);
      Sx.zero();
      V.zero();
      mm_ab2<KB>(Sx, V, Q, r_s, tmm, p);
      load_strip_global<AL>(r_s, r_pm, N, p);
    }
    VSM_STAMP(11);
    // ---- G2 = (I - E2)^-1: norm (barrier (b): every wave is done reading [R+-], [T--]), [S] -> Q, series on P -------
    float nrm_other;
    const float nrm = strip_norm_bound(E, sm, slot, p, nrm_other);
    store_strip(Q, Sx, p);
    const int K = series_order(nrm);
    invert_strip_own<KB>(E, G, P, N, sm, p, K);
    // the other point of the workgroup may need more barriers for its inverse: keep the two barrier sequences equal
    const int own = inverse_barriers(K, N), oth = inverse_barriers(series_order(nrm_other), N);
    for (int i = own; i < oth; ++i) half_barrier(p);
  }
  VSM_STAMP(12);
  half_barrier(p);            // (d): the series' powers in P no longer read
  store_strip(P, t_s, p);     // [t++] -> P
  half_barrier(p);            // (e)
  VSM_STAMP(13);
  {
    fstrip X, Y;
    X.zero();
    mm_ab<KB>(X, P, G, p);    // T21 = t++ G2
    Y.zero();
    mm_ab<KB>(Y, Q, G, p);    // Y = S G2 = T01 r-+
    VSM_STAMP(14);
    half_barrier(p);          // (f): [t++], [S] no longer read
    store_strip(P, X, p);     // [T21] -> P
    store_strip(Q, Y, p);     // [Y]   -> Q
  }
  fstrip Tpp, Rmp;
  load_strip_global<AL>(Tpp, T_pp, N, p);
  load_strip_global<AL>(Rmp, R_mp, N, p);
  half_barrier(p);            // (g)
  VSM_STAMP(15);
  // J0+ = j0+ + T21 z ; J0- = J0- + T-- j0- + Y z
  {
    const float y1 = matvec1<KB>(P, vz, p);
    const float y2 = matvec1<KB>(Q, vz, p);
    if (mlead && mrow < N && p.active) {
      J0_p[mrow] = vjp[mrow] + y1;
      J0_m[mrow] = Jm_old + vs_keep + y2;
    }
  }
  VSM_STAMP(16);
  {
    fstrip acc;
    acc.zero();
    mm_ab2<KB>(r_s, acc, P, Z, Tpp, p);   // R+- = r+- + T21 Z ; T++ = T21 T++
    VSM_STAMP(17);
    store_strip_global<AL>(R_pm, r_s, N, p);
    store_strip_global<AL>(T_pp, acc, N, p);
  }
  VSM_STAMP(18);
  mm_ab2<KB>(Rmp, V, Q, Tpp, Z, p);       // R-+ = R-+ + Y T++ ; T-- = V + Y Z
  VSM_STAMP(19);
  store_strip_global<AL>(R_mp, Rmp, N, p);
  store_strip_global<AL>(T_mm, V, N, p);
  VSM_STAMP(20);
}

// Common prologue: the two halves of the 12-wave workgroup (two spectral points), their LDS and barrier counters.
#define VSM_HALF_PROLOGUE()                                                              \
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];               \
  const int half = threadIdx.x / FNT;                                                    \
  fsmem32& sm = reinterpret_cast<fsmem32*>(smem_raw)[half];                              \
  fpos p(min(2 * (int)blockIdx.x + half, S - 1), threadIdx.x % FNT, &sm.bar);            \
  p.active = 2 * (int)blockIdx.x + half < S;                                             \
  p.red_other = &reinterpret_cast<fsmem32*>(smem_raw)[half ^ 1].red[0][0]

template <int KB, bool AL>
__global__ __launch_bounds__(2 * FNT, 3) void k_ia_strip32(int N, int S, composite<float> c, added<float> a) {
  VSM_HALF_PROLOGUE();
  const int s = p.s, tid = p.tid;
  if (tid < FNP) {
    const bool in = tid < N;
    sm.vec[0][tid] = in ? a.j0_p[(long long)s * N + tid] : 0.0f;
    sm.vec[1][tid] = in ? a.j0_m[(long long)s * N + tid] : 0.0f;
  }
  fstrip r_s, t_s;
  load_strip_global<AL>(r_s, a.r_mp + s * a.mat_stride, N, p);
  load_strip_global<AL>(t_s, a.t_pp + s * a.mat_stride, N, p);
  half_barrier(p);
  ia_body<KB, AL>(sm, p, N, a.d_symmetric, c, r_s, t_s, a.d_symmetric ? nullptr : a.r_pm + s * a.mat_stride,
                  a.d_symmetric ? nullptr : a.t_mm + s * a.mat_stride, 0);
}

template <int KB, bool MIX, bool AL, bool THERMAL, bool PRE = false>
__device__ __forceinline__ void layer_body32(fsmem32& sm, fpos& p, const quad<float>& q, int S, int m, int ndoubl,
                                             const float* __restrict__ dtau, const float* __restrict__ varpi,
                                             const float* __restrict__ tau_sum, const float* __restrict__ F0,
                                             const zsrc<float>& z, int toa, const composite<float>& c,
                                             const float* __restrict__ img = nullptr) {
  fstrip r_s, t_s;
  int jpair;
  ed_body<KB, MIX, THERMAL, PRE>(sm, p, q, m, ndoubl, dtau, varpi, tau_sum, F0, z, r_s, t_s, jpair, img);
  const int N = q.N, ns = q.n_stokes;
  if (toa) {
    const int s = p.s, tid = p.tid;
    const long long NN = (long long)N * N;
    store_strip_global<AL>(c.R_mp + s * NN, r_s, N, p);
    store_strip_global<AL>(c.T_pp + s * NN, t_s, N, p);
    fstrip d;
    dsym_strip(d, r_s, ns, p);
    store_strip_global<AL>(c.R_pm + s * NN, d, N, p);
    dsym_strip(d, t_s, ns, p);
    store_strip_global<AL>(c.T_mm + s * NN, d, N, p);
    if (tid < N && p.active) {
      c.J0_p[(long long)s * N + tid] = sm.vec[2 * jpair][tid];
      c.J0_m[(long long)s * N + tid] = sm.vec[2 * jpair + 1][tid];
    }
    return;
  }
  ia_body<KB, AL>(sm, p, N, ns, c, r_s, t_s, nullptr, nullptr, jpair);
}


template <int KB, bool MIX, bool AL, bool THERMAL = false>
__global__ __launch_bounds__(2 * FNT, 3) void k_layer_strip32(quad<float> q, int S, int m, int ndoubl,
                                                              const float* __restrict__ dtau, const float* __restrict__ varpi,
                                                              const float* __restrict__ tau_sum, const float* __restrict__ F0,
                                                              zsrc<float> z, int toa, composite<float> c) {
  VSM_HALF_PROLOGUE();
  layer_body32<KB, MIX, AL, THERMAL>(sm, p, q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c);
}
// The same for several Fourier moments at once: blockIdx.y picks the moment's composite; the elemental layers come from the
// pre-pass k_elemental_img32 as A-form images (see k_layer_strip_mm in vsm_strip.hip: in lockstep the elemental step of both
// points idles the MFMA pipe of the whole CU -- a tenth of the layer step at C4's two doubling steps).
struct layer_mm_comps32 {
  composite<float> c[VSM_MM_MAX];
};
template <int KB, bool AL>
__global__ __launch_bounds__(2 * FNT, 3) void k_layer_strip32_mm(quad<float> q, int S, int ndoubl, layer_mm_comps32 a, int toa,
                                                                 const float* __restrict__ pre) {
  VSM_HALF_PROLOGUE();
  const int im = blockIdx.y;
  const float* img = pre + ((long long)im * S + p.s) * PRE32_STRIDE;
  layer_body32<KB, false, AL, false, true>(sm, p, q, S, 0, ndoubl, nullptr, nullptr, nullptr, nullptr, zsrc<float>{}, toa, a.c[im], img);
}

// Elemental pre-pass of k_layer_strip32_mm: elemental! incl. the SFI source (elemental.jl:289-392) for every (point, moment) of a
// layer into A-form images (PRE32_STRIDE).  One 384-thread workgroup per (point, moment): thread = (column j, quarter of the rows),
// so that a wave writes 64 consecutive words of an image row.
template <bool MIX>
__global__ __launch_bounds__(FNT) void k_elemental_img32(quad<float> q, int ndoubl, const float* __restrict__ dtau,
                                                         const float* __restrict__ varpi, const float* __restrict__ tau_sum,
                                                         const float* __restrict__ F0, layer_mm_args<float> a,
                                                         float* __restrict__ pre) {
  __shared__ float mus[FNP], xs[FNP], es[FNP], ems[FNP], sgs[FNP];
  __shared__ int thick_flag;
  const int s = blockIdx.x, im = blockIdx.y, tid = threadIdx.x;
  const int N = q.N, ns = q.n_stokes, m = a.m[im];
  const zsrc<float> z = a.z[im];
  const float d = dtau[s], w = varpi[s];
  const int ncomp = MIX ? z.ncomp : 0;
  const long long NNz = (long long)N * N;
  const float* Zp = z.Zpp + (ncomp ? 0 : (long long)s * z.zs);
  const float* Zm = z.Zmp + (ncomp ? 0 : (long long)s * z.zs);
  float fk[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < ncomp) fk[k] = z.fcomp[(long long)s * ncomp + k];
  auto zget = [&](const float* Z, long long zo) {
    if (ncomp == 0) return Z[zo];
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ncomp) acc += fk[k] * Z[k * NNz + zo];
    return acc;
  };
  if (tid == 0) thick_flag = 0;
  __syncthreads();
  if (tid < FNP) {
    const bool in = tid < N;
    const float mu = in ? q.mu[tid] : 1.0f;
    const float x = d / mu;
    mus[tid] = mu;
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
    sgs[tid] = (ndoubl >= 1 && is_uv_row(tid, ns)) ? -1.0f : 1.0f;   // starred R* = D R (elemental.jl:403-422)
    if (in && x >= 0.5f) thick_flag = 1;
  }
  __syncthreads();
  const bool thick = thick_flag != 0;
  float* out = pre + ((long long)im * gridDim.x + s) * PRE32_STRIDE;
  float* R = out;
  float* T = out + PRE32_IMG;
  // thread = (quad of columns 4 jq .. 4 jq + 3, group of 6 rows): one 16-byte store per image and row, the row's table entries
  // read once per four elements
  const int jq = tid % (FNP / 4), rg = tid / (FNP / 4);
  float mj[4], xj[4], aj[4], ej[4], wct[4];
  bool act[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = 4 * jq + c;
    const float wt = (j < N) ? q.wt[min(j, N - 1)] : 0.0f;
    wct[c] = (m == 0) ? wt / 2.0f : wt / 4.0f;
    act[c] = wct[c] > num<float>::eps();
    mj[c] = mus[j];
    xj[c] = xs[j];
    aj[c] = ems[j];
    ej[c] = es[j];
  }
#pragma unroll 1
  for (int k = 0; k < FNP / 16; ++k) {
    const int i = rg * (FNP / 16) + k;
    const float mi = mus[i], xi = xs[i], ai = ems[i], ei = es[i], sg = sgs[i];
    f4_t rv = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
    if (i < N) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = 4 * jq + c;
        if (j < N) {
          const long long zo = i + (long long)N * j;
          float rr, tt;
          elemental_pair<float>(w, zget(Zp, zo), zget(Zm, zo), mi, xi, ai, ei, mj[c], xj[c], aj[c], ej[c], wct[c], i == j, thick, rr, tt);
          rv[c] = act[c] ? rr * sg : 0.0f;
          tv[c] = act[c] ? tt : ((i == j) ? ei : 0.0f);
        }
      }
    }
    const unsigned o = (unsigned)(i * LDK + 4 * jq);
    *reinterpret_cast<f4_t*>(R + o) = rv;
    *reinterpret_cast<f4_t*>(T + o) = tv;
    if (jq < (LDK - FNP) / 4) {   // the padding words of the row
      const f4_t zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f4_t*>(R + o + FNP) = zero;
      *reinterpret_cast<f4_t*>(T + o + FNP) = zero;
    }
  }
  if (tid < FNP) {   // SFI source of the solar beam: the same formulas with the solar column (vsm_elemental.h)
    const int i = tid, ic = min(i, N - 1);
    const int i0 = ns * q.i_mu0;
    float zp = 0.0f, zm = 0.0f;
    for (int qq = 0; qq < ns; ++qq) {
      const long long zo = ic + (long long)N * (i0 + qq);
      const float f = F0[qq + (long long)ns * s];
      zp += zget(Zp, zo) * f;
      zm += zget(Zm, zo) * f;
    }
    float rr, tt;
    elemental_pair<float>(w, zp, zm, mus[i], xs[i], ems[i], es[i], mus[i0], xs[i0], ems[i0], es[i0], (m == 0) ? 0.5f : 0.25f, false,
                          thick, rr, tt);
    const float att = exp(-tau_sum[s] / mus[i0]);
    out[2 * PRE32_IMG + i] = (i < N) ? tt * att : 0.0f;
    out[2 * PRE32_IMG + FNP + i] = (i < N) ? rr * att * sgs[i] : 0.0f;
    out[2 * PRE32_IMG + 2 * FNP + i] = exp(-d / q.mu0);
  }
}

template <typename K>
static int enable_lds32(K kern, const char* what) {
  return ensure_dyn_lds(reinterpret_cast<const void*>(kern), 2 * sizeof(fsmem32), what);   // once per (device, kernel)
}

// (the thermal slot only: a solar layer step goes through the pre-pass pair, strip32_layer_forward)
template <int KB, bool AL>
static int launch_layer32(const quad<float>& q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const zsrc<float>& z, int toa, const composite<float>& c,
                          hipStream_t st, int thermal) {
  if (!thermal) {
    set_error("launch_layer32: the solar layer step is the pre-pass pair");
    return VSM_ERR_UNSUPPORTED;
  }
  const int prepared_th = enable_lds32(k_layer_strip32<KB, false, AL, true>, "hipFuncSetAttribute(k_layer_strip32 th)");
  const int prepared_thm = enable_lds32(k_layer_strip32<KB, true, AL, true>, "hipFuncSetAttribute(k_layer_strip32 thm)");
  if (prepared_th) return prepared_th;
  if (prepared_thm) return prepared_thm;
  const dim3 grid((S + 1) / 2), block(2 * FNT);
  const size_t lds = 2 * sizeof(fsmem32);
  if (z.ncomp > 0)   // F0 = B[S]
    hipLaunchKernelGGL((k_layer_strip32<KB, true, AL, true>), grid, block, lds, st, q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z,
                       toa, c);
  else
    hipLaunchKernelGGL((k_layer_strip32<KB, false, AL, true>), grid, block, lds, st, q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z,
                       toa, c);
  VSM_LAUNCH_CHECK("k_layer_strip32(thermal)");
  return VSM_OK;
}
template <int KB, bool AL>
static int launch_layer32_mm(const quad<float>& q, int S, int nm, int ndoubl, const float* dtau, const float* varpi,
                             const float* tau_sum, const float* F0, const layer_mm_args<float>& a, int toa, hipStream_t st) {
  const int prepared = enable_lds32(k_layer_strip32_mm<KB, AL>, "hipFuncSetAttribute(k_layer_strip32_mm)");
  if (prepared) return prepared;
  float* pre = static_cast<float*>(scratch((size_t)nm * S * PRE32_STRIDE * sizeof(float), 3, st));
  if (!pre) return VSM_ERR_HIP;
  if (a.z[0].ncomp > 0)
    hipLaunchKernelGGL((k_elemental_img32<true>), dim3(S, nm), dim3(FNT), 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  else
    hipLaunchKernelGGL((k_elemental_img32<false>), dim3(S, nm), dim3(FNT), 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  VSM_LAUNCH_CHECK("k_elemental_img32");
  layer_mm_comps32 cc;
  for (int i = 0; i < VSM_MM_MAX; ++i) cc.c[i] = a.c[i];
  const dim3 grid((S + 1) / 2, nm), block(2 * FNT);
  hipLaunchKernelGGL((k_layer_strip32_mm<KB, AL>), grid, block, 2 * sizeof(fsmem32), st, q, S, ndoubl, cc, toa, pre);
  VSM_LAUNCH_CHECK("k_layer_strip32_mm");
  return VSM_OK;
}
template <int KB, bool AL>
static int launch_ia32(int N, int S, const composite<float>& c, const added<float>& a, hipStream_t st) {
  const int prepared = enable_lds32(k_ia_strip32<KB, AL>, "hipFuncSetAttribute(k_ia_strip32)");
  if (prepared) return prepared;
  hipLaunchKernelGGL((k_ia_strip32<KB, AL>), dim3((S + 1) / 2), dim3(2 * FNT), 2 * sizeof(fsmem32), st, N, S, c, a);
  VSM_LAUNCH_CHECK("k_ia_strip32");
  return VSM_OK;
}

}  // namespace

bool strip32_supported(int N) {
  static const bool off = ab_switch("VSM_NO_STRIP") || ab_switch("VSM_NO_STRIP32");
  return !off && N > 64 && N <= FNP;
}

int strip32_layer_forward_mm(const quad<float>& q, int S, int nm, int ndoubl, const float* dtau, const float* varpi,
                             const float* tau_sum, const float* F0, const layer_mm_args<float>& a, int toa, hipStream_t st);
int strip32_layer_forward(const quad<float>& q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const zsrc<float>& z, int toa, const composite<float>& c,
                          hipStream_t st, int thermal) {
  if (S <= 0) return VSM_OK;
  if (!thermal) {   // a solar layer step of one moment: the multi-moment pair (pre-pass + layer kernel) with nm = 1
    layer_mm_args<float> a;
    for (int i = 0; i < VSM_MM_MAX; ++i) {
      a.m[i] = m;
      a.z[i] = z;
      a.c[i] = c;
    }
    return strip32_layer_forward_mm(q, S, 1, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st);
  }
  // N % 4 != 0: element-wise global accesses, one instantiation (KB = 6) for all such N
  if (q.N & 3) return launch_layer32<6, false>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c, st, thermal);
  if (q.N > 80) return launch_layer32<6, true>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c, st, thermal);
  return launch_layer32<5, true>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c, st, thermal);
}

int strip32_layer_forward_mm(const quad<float>& q, int S, int nm, int ndoubl, const float* dtau, const float* varpi,
                             const float* tau_sum, const float* F0, const layer_mm_args<float>& a, int toa, hipStream_t st) {
  if (S <= 0 || nm <= 0) return VSM_OK;
  if (q.N & 3) return launch_layer32_mm<6, false>(q, S, nm, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st);
  if (q.N > 80) return launch_layer32_mm<6, true>(q, S, nm, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st);
  return launch_layer32_mm<5, true>(q, S, nm, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st);
}

int strip32_interaction11(int N, int S, const composite<float>& c, const added<float>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (N & 3) return launch_ia32<6, false>(N, S, c, a, st);
  if (N > 80) return launch_ia32<6, true>(N, S, c, a, st);
  return launch_ia32<5, true>(N, S, c, a, st);
}

}  // namespace vsm

#ifdef VSM_PHASE_TIMING
extern "C" int vsm_debug_phase_cycles_strip32(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_phase_cycles_strip32), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_phase_cycles_strip32), z, sizeof(z));
  }
  return 0;
}
#endif
