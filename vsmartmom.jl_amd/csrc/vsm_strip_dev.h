// Device-side building blocks of the FP64 column-strip kernels (vsm_strip.hip, vsm_strip_lin.hip): the strip /
// A-form layouts, LDS-operand products, the series / Gauss-Jordan inverse.  See the header of vsm_strip.hip.
#pragma once
#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_lds.h"
#include "vsm_elemental.h"

namespace vsm {
namespace {

constexpr int SNP = 64;    // padded matrix size
constexpr int SNT = 256;   // threads per workgroup (4 waves)

struct sstrip {
  d4_t v[4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < 4; ++a) v[a] = acc_zero<double>();
  }
};

struct ssmem {
  double P[SNP * SNP];
  double Q[SNP * SNP];
  double vec[8][SNP];
  float red[2][4];
  int flags[4];
  double usg[SNP];   // -1.0 on the U/V rows (i mod n_stokes >= 2), +1.0 elsewhere
  gj_scratch<double, SNP> gj;
  double xw[4][16 * 10];   // wave-private transposer tiles of load/store_strip_global_c8
};

// Per-lane addressing of the swizzled A-form (lidx of vsm_lds.h) in BYTES, split into per-lane bases and compile-time offsets
// so that an LDS access of a product needs no address arithmetic at all:
//   A fragment (row 16 t + l15, column 4 ks + kq):  ab8[ks & 3][t] + 2048 ks   (one base per (ks & 3, t); the 2048 ks and the
//   distance of the A-form from the bound base go into the instruction's 16-bit offset field)
//   strip element (row 16 ta + kq + 4 r, column col): ((128 ta + 32 r) ^ cm_hi8) + c_lo8    (one v_xad_u32)
// bind(base) adds the LDS address of `base` (the first A-form of the kernel's LDS block) to the per-lane bases; the helpers
// then address an A-form at `A` by the compile-time distance A - base.  Unbound: the distance is the full LDS address.
using lds_d = __attribute__((address_space(3))) double;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
struct spos {
  int lane, wave, l15, kq, col;
  unsigned ab8[4][4];
  unsigned cm_hi8, c_lo8;
  const char* lbase;
  bool bound;
  __device__ __forceinline__ spos() {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    const int L = l15 ^ ((kq >> 1) << 1), pq = kq & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) ab8[j][t] = 8u * (unsigned)(64 * kq + (L ^ (4 * j)) + 16 * (t ^ pq));
    const int m = ((col & 1) << 4) | (((col >> 1) & 7) << 1);
    cm_hi8 = 8u * (unsigned)(m & 0x3C);
    c_lo8 = 8u * (unsigned)((kq ^ (m & 3)) + SNP * col);
    lbase = nullptr;
    bound = false;
  }
  __device__ __forceinline__ void bind(const void* base) {
    const unsigned L = lds_addr(base);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) ab8[j][t] += L;
    c_lo8 += L;
    lbase = static_cast<const char*>(base);
    bound = true;
  }
  __device__ __forceinline__ unsigned delta(const void* A) const {
    return bound ? (unsigned)(static_cast<const char*>(A) - lbase) : lds_addr(A);
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + kq + 4 * r; }
  // A fragment / strip element of the A-form at distance dA (see delta)
  __device__ __forceinline__ const lds_d* aptr(unsigned dA, int t, int ks) const {
    // (pointer + constant, not integer + constant: the constant then lands in the instruction's offset field also when dA
    // is a run-time value, and ab8 + dA is formed once per base)
    return reinterpret_cast<const lds_d*>((unsigned long long)(ab8[ks & 3][t] + dA)) + 256 * ks;
  }
  __device__ __forceinline__ lds_d* sptr(unsigned dA, int ta, int r) const {
    return reinterpret_cast<lds_d*>((unsigned long long)((((unsigned)(128 * ta + 32 * r)) ^ cm_hi8) + (c_lo8 + dA)));
  }
  // hide the loop invariance of the bases from LICM (hoisting every derived address costs > 100 VGPRs)
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(ab8[j][t]));
    asm volatile("" : "+v"(cm_hi8));
    asm volatile("" : "+v"(c_lo8));
  }
};

// The products keep FOUR independent accumulator chains in flight (the four row tiles of a k-step): a dependent f64 MFMA
// waits for its predecessor's result, and the register-pressure-minimising scheduler, left alone, turns the loop tile-major
// (one chain of KS dependent MFMAs per tile: 1.45x the issue time of the products with one wave per SIMD).  A scheduling
// fence after every k-step keeps the written order: fragments of step ks + 1 requested, then the MFMAs of step ks.
#if defined(VSM_AB_SWITCHES) && defined(VSM_NO_KSTEP_FENCE)   // (an ablation of the diagnostic build only)
#define VSM_KSTEP_FENCE()
#else
#define VSM_KSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// value of the neighbour lane (lane ^ 1): one DPP quad permutation per 32-bit half, no LDS round trip (__shfl_xor is ds_bpermute)
__device__ __forceinline__ double dpp_swap1(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// Added layer of the elemental pre-pass (k_elemental_img) in global memory, per (moment, point): the two A-form IMAGES [r-+*],
// [t++] exactly as the doubling loop wants them in LDS (swizzled 64 x 64 incl. zero padding and the source vectors in the
// spare columns), then j0+[64], j0-[64] and aux[64] (aux[0] = exp(-dtau / mu_0)).  The layer kernel copies the images with
// global_load_lds_dwordx4: 16 instructions per lane, no registers, no address arithmetic.
constexpr int PRE_IMG = SNP * SNP;
constexpr int PRE_STRIDE = 2 * PRE_IMG + 3 * SNP;
__device__ __forceinline__ void copy_image_to_lds(double* L, const double* __restrict__ g, const spos& p) {
#pragma unroll
  for (int i = 0; i < PRE_IMG / (SNT * 2); ++i) {   // 8 x (256 lanes x 16 B)
    const int blk = (4 * i + p.wave) * 128;         // doubles: one wave moves 1 KB per instruction
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blk + 2 * p.lane),
                                     (__attribute__((address_space(3))) void*)(L + blk), 16, 0, 0);
  }
}

// Source vectors without spare columns (N = 61..64: the strips are full): y = A x as VALU mat-vecs over the A-form.  A wave takes
// a quarter of the columns (lane = row: a column of the swizzled image is a permutation of 64 consecutive words, conflict-free)
// and leaves its partial sums of two products in two 64-word slots; after a barrier mv_sum adds the four waves' slots.
__device__ __forceinline__ void mv_part(const double* A, const double* x, double s, double* slot, const spos& p) {
  double acc = 0.0;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int k = 16 * p.wave + kk;
    acc = fma(A[lidx<SNP>(p.lane, k)], x[k], acc);
  }
  slot[p.lane] = acc * s;
}
__device__ __forceinline__ double mv_sum(double* const* slots, int i) {
  return (slots[0][i] + slots[1][i]) + (slots[2][i] + slots[3][i]);
}

// acc += A * B   (A: A-form in LDS, B: strip in registers).  Software-pipelined by one k-step.
template <int KS>
__device__ __forceinline__ void mm_ab(sstrip& acc, const double* A, const sstrip& B, spos& p) {
  p.opaque();
  const unsigned dA = p.delta(A);
  double a[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc.v[t] = mfma<double>::mma(a[ks & 1][t], b, acc.v[t]);
    VSM_KSTEP_FENCE();
  }
}
// out = C0 + A * B without copying C0 into the accumulators first (the first k-step takes C0 as its addend)
template <int KS>
__device__ __forceinline__ void mm_ab_c(sstrip& out, const sstrip& C0, const double* A, const sstrip& B, spos& p) {
  p.opaque();
  const unsigned dA = p.delta(A);
  double a[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < 4; ++t) out.v[t] = mfma<double>::mma(a[ks & 1][t], b, ks == 0 ? C0.v[t] : out.v[t]);
    VSM_KSTEP_FENCE();
  }
}
// acc1 += A * B1 ; acc2 += A * B2   (shared A fragments)
template <int KS>
__device__ __forceinline__ void mm_ab2(sstrip& acc1, sstrip& acc2, const double* A, const sstrip& B1, const sstrip& B2,
                                       spos& p) {
  p.opaque();
  const unsigned dA = p.delta(A);
  double a[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b1 = B1.v[ks >> 2][ks & 3], b2 = B2.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc1.v[t] = mfma<double>::mma(a[ks & 1][t], b1, acc1.v[t]);
      acc2.v[t] = mfma<double>::mma(a[ks & 1][t], b2, acc2.v[t]);
    }
    VSM_KSTEP_FENCE();
  }
}

// strip -> A-form in LDS, through f(value, row, col)
template <typename F>
__device__ __forceinline__ void store_strip(double* dst, const sstrip& s, const spos& p, F f) {
  const unsigned dA = p.delta(dst);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      *p.sptr(dA, ta, r) = f(s.v[ta][r], row, p.col);
    }
}
__device__ __forceinline__ void load_strip(sstrip& s, const double* src, const spos& p) {
  const unsigned dA = p.delta(src);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = *p.sptr(dA, ta, r);
}

// Frobenius-norm bound of the N x N block whose strips the waves hold (deterministic; see vsm_fused.hip).
// Contains ONE barrier: on return every wave has finished whatever it did before the call.
template <typename SM>
__device__ __forceinline__ double strip_norm_bound(const sstrip& e, int N, SM& sm, int& slot, const spos& p) {
  double ss = 0;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double v = e.v[ta][r];
      if (p.row(ta, r) < N && p.col < N) ss += v * v;
    }
  const float ws = wave_sum(to_float_up(ss));
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  __syncthreads();
  const float tot = (sm.red[slot][0] + sm.red[slot][1]) + (sm.red[slot][2] + sm.red[slot][3]);
  slot ^= 1;
  return (double)(sqrtf(tot) * 1.001f);
}

// The same for a strip whose padding rows (>= N) are zero by construction (the doubling loop's products): no per-element
// mask, one select for the lanes of the padding / rider columns.
template <typename SM>
__device__ __forceinline__ double strip_norm_bound_clean(const sstrip& e, int N, SM& sm, int& slot, const spos& p) {
  double ss = 0;
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss = fma(e.v[ta][r], e.v[ta][r], ss);
  ss = (p.col < N) ? ss : 0.0;
  const float ws = wave_sum(to_float_up(ss));
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  __syncthreads();
  const float tot = (sm.red[slot][0] + sm.red[slot][1]) + (sm.red[slot][2] + sm.red[slot][3]);
  slot ^= 1;
  return (double)(sqrtf(tot) * 1.001f);
}

// In-place pivoted Gauss-Jordan of the A-form matrix V (N x N block), 256 threads.  Ends with a barrier.
__device__ __forceinline__ void gj_lds_strip(double* V, int N, gj_scratch<double, SNP>* sc) {
  using G = gj_cfg<SNP, SNT>;
  const int tr = threadIdx.x % G::TR, tc = threadIdx.x / G::TR;
  double g[G::RB][G::CB];
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      g[rb][cb] = (i < N && j < N) ? V[lidx<SNP>(i, j)] : ((i == j) ? 1.0 : 0.0);
    }
  gj_invert<double, SNP, SNT, false>(g, N, *sc);
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      if (i < N && j < N) V[lidx<SNP>(i, sc->dst[j])] = g[rb][cb];
    }
  __syncthreads();
}

// Order K of the Neumann series sum_{k <= K} E^k that reaches (I - E)^-1 to working precision, from a bound nrm >= ||E||_2:
// the smallest of {1,2,3,4,7,8,15,16,31} with nrm^(K+1) / (1 - nrm) <= eps / 4; 0 = no series (nrm >= 0.3: Gauss-Jordan).
__device__ __forceinline__ int series_order(double nrm) {
  const double tol = num<double>::eps() * 0.25;
  int K = 0;
  if (nrm < 0.3) {
    const double lim = tol * (1.0 - nrm);
    const double n2 = nrm * nrm, n4 = n2 * n2, n8 = n4 * n4, n16 = n8 * n8;
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

// G_s = strip of (I - E)^-1, E given as strips.  W (LDS, A-form scratch) must not be read by anybody once the
// first barrier inside has been passed (the norm reduction), which the callers guarantee.  On return other
// waves may still be READING W: barrier before overwriting it.  Returns 1 (Gauss-Jordan) or 1 + series order.
template <int KS, typename SM>
__device__ __forceinline__ int invert_strip_k(int K, sstrip& E, sstrip& G, double* W, int N, SM& sm, spos& p, int mode);
template <int KS, typename SM>
__device__ __forceinline__ int invert_strip(sstrip& E, sstrip& G, double* W, int N, SM& sm, int& slot, spos& p,
                                            int mode) {
  const double nrm = strip_norm_bound(E, N, sm, slot, p);
  return invert_strip_k<KS>(series_order(nrm), E, G, W, N, sm, p, mode);
}
// the same after the norm reduction (all waves past its barrier), for a given order
template <int KS, typename SM>
__device__ __forceinline__ int invert_strip_k(int K, sstrip& E, sstrip& G, double* W, int N, SM& sm, spos& p, int mode) {
  if (mode == 1) K = 0;
  if (mode == 2 && K == 0) K = 31;
  auto keep = [N](double a, int r, int c) { return (r < N && c < N) ? a : 0.0; };
  if (K == 0) {
    store_strip(W, E, p, [=](double a, int r, int c) { return (r == c) ? 1.0 - keep(a, r, c) : -keep(a, r, c); });
    __syncthreads();
    gj_lds_strip(W, N, &sm.gj);
    load_strip(G, W, p);
    return 1;
  }
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const double e = keep(E.v[ta][r], row, p.col);
      E.v[ta][r] = e;
      G.v[ta][r] = (row == p.col && row < N) ? e + 1.0 : e;
    }
  if (K == 1) return 2;
  store_strip(W, E, p, [](double a, int, int) { return a; });
  __syncthreads();
  int cur = 1;  // W = E^cur (A-form), E = its strip, G = strip of sum_{k < 2 cur} E^k
  for (;;) {
    sstrip W2;
    W2.zero();
    mm_ab<KS>(W2, W, E, p);  // E^(2 cur)
    cur *= 2;
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    __syncthreads();  // everybody finished reading W
    store_strip(W, W2, p, [](double a, int, int) { return a; });
    __syncthreads();
    sstrip T;
    T.zero();
    mm_ab<KS>(T, W, G, p);  // E^cur * G   (powers of E commute)
#pragma unroll
    for (int ta = 0; ta < 4; ++ta) G.v[ta] += T.v[ta];
    if (K == 2 * cur - 1) break;
    E = W2;
  }
  return 1 + K;
}

// The general inverse out of line: the doubling loop of the layer kernels calls it only for orders the Horner path below
// does not take (K > 8, Gauss-Jordan) -- rare, and its four live strips and the Gauss-Jordan register block stay out of the
// loop's register allocation.
template <int KS, typename SM>
__device__ __attribute__((noinline)) void invert_strip_slow(int K, sstrip& E, sstrip& G, double* W, int N, SM& sm, spos& p) {
  invert_strip_k<KS>(K, E, G, W, N, sm, p, 0);
}
// (I - E)^-1 for the doubling loop: after the norm reduction, orders 1..8 by Horner's rule with ONE A-form store,
//   X_0 = E,  X_{j+1} = E + E X_j  ->  X_{K-1} = E + E^2 + ... + E^K,  G = I + X_{K-1}
// (A = [E] for every product: no barrier between the K - 1 products, three live strips; for K <= 4 as many products as the
// squaring scheme of invert_strip, whose every level costs an A-form store and two barriers more).  On return other waves may
// still be reading W.
// The series part for a given order K (all waves past the norm reduction's barrier, so nobody reads W any more).  From the
// third term on, E is re-read from its A-form (16 LDS reads per term) instead of being kept: two live strips.
template <int KS, typename SM>
__device__ __forceinline__ void invert_strip_horner_k(int K, sstrip& E, sstrip& G, double* W, int N, SM& sm, spos& p) {
  if (K < 1 || K > 8) {   // (copies: only these objects have their address taken, and only on this path)
    sstrip Es = E, Gs;
    spos ps = p;
    invert_strip_slow<KS>(K, Es, Gs, W, N, sm, ps);
    G = Gs;
    return;
  }
  // The rider columns (>= N) of E are NOT cleared: every product of the series (and every product that takes G as its right
  // operand afterwards) maps a column of B to the same column of the result, so whatever rides there stays there -- in columns
  // that are never read as a contraction index and never stored to the composite.  (Rows >= N are zero by construction.)
  if (K == 1) {
    G = E;
  } else {
    store_strip(W, E, p, [](double a, int, int) { return a; });
    __syncthreads();
    mm_ab_c<KS>(G, E, W, E, p);              // X1 = E + E E
    for (int j = 2; j < K; ++j) {            // X_j = E + E X_{j-1}
      sstrip X;
      load_strip(X, W, p);
      mm_ab<KS>(X, W, G, p);
      G = X;
    }
  }
  // + I: a lane owns at most one diagonal element, in row tile ta = wave: (r = l15 >> 2, kq = l15 & 3)
  {
    const bool dl = p.kq == (p.l15 & 3) && p.col < N;
    const int dr = p.l15 >> 2;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
      if (ta == p.wave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) G.v[ta][r] += (dl && dr == r) ? 1.0 : 0.0;
      }
  }
}
template <int KS, typename SM>
__device__ __forceinline__ void invert_strip_horner(sstrip& E, sstrip& G, double* W, int N, SM& sm, int& slot, spos& p) {
  const double nrm = strip_norm_bound_clean(E, N, sm, slot, p);
  invert_strip_horner_k<KS>(series_order(nrm), E, G, W, N, sm, p);
}

__device__ __forceinline__ void load_strip_global(sstrip& s, const double* __restrict__ g, int N, const spos& p) {
  const int cc = min(p.col, N - 1);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const double v = g[min(row, N - 1) + (long long)N * cc];   // clamped address, masked value
      s.v[ta][r] = (row < N && p.col < N) ? v : 0.0;
    }
}
__device__ __forceinline__ void store_strip_global(double* __restrict__ g, const sstrip& s, int N, const spos& p) {
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      if (row < N && p.col < N) g[row + (long long)N * p.col] = s.v[ta][r];
    }
}
// D X D: sign (+) where row and column have the same U/V parity (doubling.jl:178-201)
__device__ __forceinline__ void dsym_strip(sstrip& d, const sstrip& x, int ns, const spos& p) {
  const bool uc = is_uv_row(p.col, ns);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) d.v[ta][r] = (is_uv_row(p.row(ta, r), ns) == uc) ? x.v[ta][r] : -x.v[ta][r];
}
// The same with the parities of the lane's 16 rows and of its column evaluated once (bit 4 ta + r of `rows`, `uc`): the `%`
// by the run-time n_stokes costs ~20 instructions per row, and the layer kernels form D X D six times.
struct dpar {
  unsigned rows;
  bool uc;
  // from a table usg[i] = -1.0 on the U/V rows, +1.0 elsewhere (one `%` per row of the matrix instead of 17 per lane)
  __device__ __forceinline__ dpar(const double* usg, const spos& p) {
    rows = 0;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) rows |= (usg[p.row(ta, r)] < 0.0 ? 1u : 0u) << (4 * ta + r);
    uc = usg[p.col] < 0.0;
    if (uc) rows = ~rows;   // bit set = sign flip
  }
  __device__ __forceinline__ dpar(int ns, const spos& p) {
    rows = 0;
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) rows |= (is_uv_row(p.row(ta, r), ns) ? 1u : 0u) << (4 * ta + r);
    uc = is_uv_row(p.col, ns);
    if (uc) rows = ~rows;   // bit set = sign flip
  }
};
__device__ __forceinline__ void dsym_strip(sstrip& d, const sstrip& x, const dpar& dp) {
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // flip the sign bit: shift, mask, xor on the high word
      const int n = 4 * ta + r;
      const unsigned sb = (dp.rows << (31 - n)) & 0x80000000u;
      d.v[ta][r] = __hiloint2double(__double2hiint(x.v[ta][r]) ^ (int)sb, __double2loint(x.v[ta][r]));
    }
}
// global column-major N x N -> A-form in LDS (zero padded); one column per wave and pass, coalesced reads
__device__ __forceinline__ void stage_aform(double* L, const double* __restrict__ g, int N, const spos& p) {
#pragma unroll 4
  for (int j = p.wave; j < SNP; j += 4) {
    const double v = (p.lane < N && j < N) ? g[p.lane + (long long)N * j] : 0.0;
    L[lidx<SNP>(p.lane, j)] = v;
  }
}

// Strip <-> global through a wave-private 16 x 8 tile of LDS (XS8 = 10 doubles per column: conflict-free).  The
// direct versions above touch 16 columns x 32 B per instruction -- sixteen quarter-used cache lines that the next
// instruction fetches from L2 again.  Here a lane moves 16 contiguous bytes of one column (one dwordx4; a wave covers
// 16 columns x 64 B) and the permutation to the MFMA layout happens in LDS, in place in the strip's registers.
// Wave-private, so no workgroup barrier: LDS operations of one wave complete in order.
constexpr int XS8 = 10;
struct d2_t {
  double a, b;
} __attribute__((aligned(8)));
// (two halves: the request leaves 16 contiguous bytes of a column per register pair; the permutation can wait until the
// strip is needed, so that the round trip hides behind whatever lies between)
__device__ __forceinline__ void load_strip_global_c8_issue(sstrip& x, const double* __restrict__ g, int N, const spos& p) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  const double* src = g + (long long)N * min(col, N - 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 8 * j + 2 * q;
    double a = 0.0, b = 0.0;
    if (col < N && r + 1 < N) {
      const d2_t t = *reinterpret_cast<const d2_t*>(src + r);
      a = t.a;
      b = t.b;
    } else if (col < N && r < N) {
      a = src[r];
    }
    x.v[j >> 1][2 * (j & 1)] = a;
    x.v[j >> 1][2 * (j & 1) + 1] = b;
  }
}
__device__ __forceinline__ void load_strip_global_c8_finish(sstrip& x, const spos& p, double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __builtin_amdgcn_wave_barrier();
    xw[c * XS8 + 2 * q] = x.v[j >> 1][2 * (j & 1)];
    xw[c * XS8 + 2 * q + 1] = x.v[j >> 1][2 * (j & 1) + 1];
    __builtin_amdgcn_wave_barrier();
    x.v[j >> 1][2 * (j & 1)] = xw[p.l15 * XS8 + p.kq];
    x.v[j >> 1][2 * (j & 1) + 1] = xw[p.l15 * XS8 + p.kq + 4];
  }
}
__device__ __forceinline__ void load_strip_global_c8(sstrip& x, const double* __restrict__ g, int N, const spos& p,
                                                     double* __restrict__ xw) {
  load_strip_global_c8_issue(x, g, N, p);
  load_strip_global_c8_finish(x, p, xw);
}
__device__ __forceinline__ void store_strip_global_c8(double* __restrict__ g, const sstrip& x, int N, const spos& p,
                                                      double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  double* dst = g + (long long)N * min(col, N - 1);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __builtin_amdgcn_wave_barrier();
    xw[p.l15 * XS8 + p.kq] = x.v[j >> 1][2 * (j & 1)];
    xw[p.l15 * XS8 + p.kq + 4] = x.v[j >> 1][2 * (j & 1) + 1];
    __builtin_amdgcn_wave_barrier();
    d2_t t;
    t.a = xw[c * XS8 + 2 * q];
    t.b = xw[c * XS8 + 2 * q + 1];
    const int r = 8 * j + 2 * q;
    if (col < N && r + 1 < N)
      *reinterpret_cast<d2_t*>(dst + r) = t;
    else if (col < N && r < N)
      dst[r] = t.a;
  }
}

// global column-major N x N -> A-form in LDS with ALL loads in flight before the first LDS write (one wave per SIMD:
// nothing else hides a round trip, and stage_aform's batches of four make four of them)
__device__ __forceinline__ void stage_aform_full(double* L, const double* __restrict__ g, int N, const spos& p) {
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = p.wave + 4 * i;
    v[i] = (p.lane < N && j < N) ? g[p.lane + (long long)N * j] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) L[lidx<SNP>(p.lane, p.wave + 4 * i)] = v[i];
}
__device__ __forceinline__ void stage_aform_full2(double* L1, const double* __restrict__ g1, double* L2,
                                                  const double* __restrict__ g2, int N, const spos& p) {
  double v[16], w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = p.wave + 4 * i;
    const bool ok = p.lane < N && j < N;
    v[i] = ok ? g1[p.lane + (long long)N * j] : 0.0;
    w[i] = ok ? g2[p.lane + (long long)N * j] : 0.0;
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    L1[lidx<SNP>(p.lane, p.wave + 4 * i)] = v[i];
    L2[lidx<SNP>(p.lane, p.wave + 4 * i)] = w[i];
  }
}

}  // namespace
}  // namespace vsm
