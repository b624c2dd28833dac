// Device building blocks of the FP64 column-strip kernels for 64 < N <= 128 (vsm_strip128.hip: forward; vsm_strip128lin.hip:
// linearized): strips as accumulator tiles, the one A-form in LDS, parked strips, norm / series / pivoted inverse.  See the
// header of vsm_strip128.hip for the scheme.
#pragma once
#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_lds.h"

namespace vsm {
namespace {

using lds_d = __attribute__((address_space(3))) double;
__device__ __forceinline__ unsigned lds_addr128(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ double dpp_swap1_128(double x) {   // value of lane ^ 1
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

constexpr int B_MAXW = 8;   // waves per workgroup at most
// A pointer that went through an empty asm (to keep LICM from hoisting every derived address) comes back as a GENERIC pointer:
// its accesses would be flat_load / flat_store, which count on lgkmcnt as well -- every LDS wait of the products and every
// barrier would then wait for the parked strips and the operand loads.  Cast back to the global address space.
using gd4_p = __attribute__((address_space(1))) d4_t*;
using gcd4_p = const __attribute__((address_space(1))) d4_t*;
using gd_p = __attribute__((address_space(1))) double*;
using gcd_p = const __attribute__((address_space(1))) double*;

template <int RT>
struct bstrip {
  d4_t v[RT];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < RT; ++a) v[a] = acc_zero<double>();
  }
};

template <int RT>
struct bpos {
  int lane, wave, l15, kq, col;
  unsigned ab[4][2];   // fragment bases (bytes) by (ks & 3, ks >= 16)
  unsigned sb[4];      // strip element bases by r
  bool mat_wave;       // the wave's columns are columns of the A-form (wave < RT)
  int ksn;             // k-steps that hold columns < N: ceil(N / 4) (the products skip the all-padding tail, up to three of 4 RT)
  __device__ __forceinline__ bpos(unsigned af, int N) {
    ksn = (N + 3) >> 2;
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    mat_wave = wave < RT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ab[j][0] = af + 8u * (unsigned)((kq << 4) | (l15 ^ (kq | (j << 2))));
      ab[j][1] = ab[j][0] + 8u * 64u * 16u * RT;
    }
    const int j = l15 >> 2, ww = mat_wave ? wave : 0;   // (a wave without matrix columns gets valid addresses it never stores to)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sb[r] = af + 8u * (unsigned)(((4 * ww + j) * RT * 64) + (((l15 & 3) << 4) | (kq ^ (l15 & 3)) | ((r ^ j) << 2)));
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + kq + 4 * r; }
  __device__ __forceinline__ const lds_d* aptr(int t, int ks) const {
    const int h = ks >= 16 ? 1 : 0;
    return reinterpret_cast<const lds_d*>((unsigned long long)ab[ks & 3][h]) + 64 * ((ks - 16 * h) * RT + t);
  }
  __device__ __forceinline__ lds_d* sptr(int ta, int r) const {
    return reinterpret_cast<lds_d*>((unsigned long long)sb[r]) + 64 * ta;
  }
  // hide the loop invariance of the bases from LICM (it would hoist every derived address into a register)
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("" : "+v"(ab[j][0]));
      asm volatile("" : "+v"(ab[j][1]));
    }
  }
};

// acc += [A] * B.  One register per row tile for the fragments: the fragment of tile t for step ks + 1 is requested right
// behind the MFMA that consumed its predecessor (RT MFMAs = 64 RT cycles ahead of its use).
template <int RT>
__device__ __forceinline__ void mm128(bstrip<RT>& acc, const bstrip<RT>& B, bpos<RT>& p) {
  constexpr int KS = 4 * RT;
  p.opaque();
  double a[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) a[t] = *p.aptr(t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks >= KS - 3 && ks >= p.ksn) break;   // (uniform) N > 16 (RT - 1): only the last three k-steps can be all padding
    const double b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      acc.v[t] = mfma<double>::mma(a[t], b, acc.v[t]);
      if (ks + 1 < KS) a[t] = *p.aptr(t, ks + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The source vectors where the strips have no spare column (N = 127, 128): wave w forms row tile w of [A] (x_0 | x_1) for the two
// vectors of the LDS table at byte address xb (lane: (l15 & 1) NP + kq; every even / odd column of the tile holds the same
// product) -- 4 RT MFMAs more behind the product of the same [A], two accumulation chains.
template <int RT>
__device__ __forceinline__ void rider_tile(d4_t& yr, unsigned xb, bpos<RT>& p) {
  constexpr int KS = 4 * RT;
  p.opaque();
  const unsigned wo = 512u * (unsigned)p.wave;
  auto afr = [&](int ks) {
    const int h = ks >= 16 ? 1 : 0;
    return *(reinterpret_cast<const lds_d*>((unsigned long long)(p.ab[ks & 3][h] + wo)) + 64 * ((ks - 16 * h) * RT));
  };
  auto xfr = [&](int ks) { return *(reinterpret_cast<const lds_d*>((unsigned long long)xb) + 4 * ks); };
  d4_t y1 = acc_zero<double>();
#pragma unroll
  for (int ks = 0; ks < KS; ks += 2) {
    yr = mfma<double>::mma(afr(ks), xfr(ks), yr);
    y1 = mfma<double>::mma(afr(ks + 1), xfr(ks + 1), y1);
  }
  yr += y1;
}
template <int RT>
__device__ __forceinline__ void mm128r(bstrip<RT>& acc, const bstrip<RT>& B, d4_t& yr, unsigned xb, bpos<RT>& p) {
  mm128(acc, B, p);
  rider_tile(yr, xb, p);
}

// strip -> A-form (columns >= N, i.e. the riders and the padding, as zeros); waves without matrix columns stay out
template <int RT>
__device__ __forceinline__ void store_af(const bstrip<RT>& s, int N, const bpos<RT>& p) {
  if (!p.mat_wave) return;
  const bool keep = p.col < N;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) *p.sptr(ta, r) = keep ? s.v[ta][r] : 0.0;
}
template <int RT>
__device__ __forceinline__ void load_af(bstrip<RT>& s, const bpos<RT>& p) {
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = *p.sptr(ta, r);
}
// strip <-> the wave's slot of the workgroup's global scratch (lane-linear 32-byte records)
template <int RT>
__device__ __forceinline__ void spill(d4_t* g, const bstrip<RT>& s, const bpos<RT>& p) {
  asm volatile("" : "+v"(g));   // (the per-tile addresses are formed here, not hoisted out of the doubling loop into registers)
  gd4_p gg = (gd4_p)g;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) gg[64 * ta] = s.v[ta];
}
template <int RT>
__device__ __forceinline__ void fill(bstrip<RT>& s, const d4_t* g, const bpos<RT>& p) {
  asm volatile("" : "+v"(g));
  gcd4_p gg = (gcd4_p)g;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) s.v[ta] = gg[64 * ta];
}

// Frobenius-norm bound of the N x N block (rows >= N are zero by construction).  ONE barrier inside.
template <int RT>
__device__ __forceinline__ double norm128(const bstrip<RT>& e, int N, int nw, float* red, int& slot, const bpos<RT>& p) {
  double ss = 0;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss = fma(e.v[ta][r], e.v[ta][r], ss);
  ss = (p.col < N) ? ss : 0.0;
  const float ws = wave_sum(to_float_up(ss));
  if (p.lane == 0) red[16 * slot + p.wave] = ws;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < nw; ++w) tot += red[16 * slot + w];
  slot ^= 1;
  return (double)(sqrtf(tot) * 1.001f);
}

// order of the Neumann series (vsm_strip_dev.h series_order); 0 = the norm bound is not below 0.3 (or not a number): pivoted inverse
__device__ __forceinline__ int series_order128(double nrm) {
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

template <int RT>
__device__ __forceinline__ void add_identity(bstrip<RT>& G, int N, const bpos<RT>& p) {
  // a lane owns at most one diagonal element, in row tile ta = wave: r = l15 >> 2, kq = l15 & 3
  const bool dl = p.kq == (p.l15 & 3) && p.col < N;
  const int dr = p.l15 >> 2;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
    if (ta == p.wave) {
#pragma unroll
      for (int r = 0; r < 4; ++r) G.v[ta][r] += (dl && dr == r) ? 1.0 : 0.0;
    }
}

// ---- pivoted inverse (the contract of the reference's LU: batch_inv!, cpu_batched.jl:32-47, ext/gpu_batched_cuda.jl:149-179) ----
// Where the norm bound gives no series order (||E||_F >= 0.3: the last doublings of thick near-conservative layers, bright
// surfaces -- and anything a caller of the public entry points hands in, spectral radius >= 1 included) the inverse is an
// in-place Gauss-Jordan elimination with partial pivoting (the pivot rule of getrf: largest |m_ik|, first occurrence) on
// M = I - E in the A-form's LDS (plain column-major, pitch NP).  LDS-resident on purpose: a dozen registers per lane, so the
// cold path does not shape the register allocation of the products around it (an out-of-line register-resident elimination
// cost the product loops 2.5 %).  Lane = row (rows lane, lane + 64), wave w = columns w, w + nw, ...: per pivot step every wave
// reads column k and finds the pivot row redundantly (DPP maximum + ballot), a barrier, every wave updates its own columns
// (two broadcast reads, the row interchange folded into the update), a barrier.  The cost does not depend on the spectral
// radius: 2 N barriers and N^2 LDS read-modify-writes per step (a squaring level is two products, and 1 - rho = 1e-3 would take
// fifteen of them).  The row interchanges come back as a column permutation `src` that the strip read-back applies.
// status (device words, vsm_device_status): [0] |= VSM_DEVSTAT_SINGULAR on an exactly zero pivot, [1] += 1 per pivoted inverse.
using lds_i = __attribute__((address_space(3))) int;
constexpr size_t GJS_BYTES = 2 * 128 * sizeof(int);   // piv[128], src[128]: LDS behind the kernels' own tables
template <int RT>
__device__ __forceinline__ void gj128_lds(int N, lds_d* M, lds_i* piv, lds_i* src, int nw, int* status, const bpos<RT>& p) {
  constexpr int NP = 16 * RT;
  const int i0 = p.lane, i1 = p.lane + 64;
  const bool ok0 = i0 < N, ok1 = i1 < N;
  bool singular = false;
  for (int k = 0; k < N; ++k) {
    const lds_d* ck = M + k * NP;
    double f0 = ok0 ? ck[i0] : 0.0, f1 = ok1 ? ck[i1] : 0.0;
    const double v0 = (ok0 && i0 >= k) ? fabs(f0) : -1.0, v1 = (ok1 && i1 >= k) ? fabs(f1) : -1.0;
    const double best = wave_max(fmax(v0, v1));
    const unsigned long long m0 = __ballot(v0 >= 0.0 && v0 == best), m1 = __ballot(v1 >= 0.0 && v1 == best);
    const int pr = m0 ? (__ffsll((long long)m0) - 1) : (m1 ? 64 + __ffsll((long long)m1) - 1 : k);
    const double ckk = ck[k], pv = ck[pr];   // (broadcast reads)
    const double d = 1.0 / pv;
    singular |= pv == 0.0;
    f0 = (i0 == pr) ? ckk : f0;              // the pivot column after the interchange of rows k and pr
    f1 = (i1 == pr) ? ckk : f1;
    if (threadIdx.x == 0) piv[k] = pr;
    __syncthreads();                         // every wave holds column k
    for (int j = p.wave; j < N; j += nw) {
      lds_d* cj = M + j * NP;
      const bool isk = j == k;
      const double a = cj[pr], b = cj[k];
      const double u = isk ? d : a * d;
      double x0 = ok0 ? cj[i0] : 0.0, x1 = ok1 ? cj[i1] : 0.0;
      x0 = isk ? 0.0 : ((i0 == pr) ? b : x0);
      x1 = isk ? 0.0 : ((i1 == pr) ? b : x1);
      x0 = (i0 == k) ? u : fma(-f0, u, x0);
      x1 = (i1 == k) ? u : fma(-f1, u, x1);
      if (ok0) cj[i0] = x0;
      if (ok1) cj[i1] = x1;
    }
    __syncthreads();
  }
  // undo the row interchanges as a column permutation of the inverse: for k = N-1 .. 0 swap columns k and piv[k];
  // src[x] = the column of M that is column x of the inverse (vsm_inverse.h)
  if (p.wave == 0) {
    const int lane = p.lane;
    int s0 = lane, s1 = lane + 64;
    const int p0 = (lane < N) ? piv[lane] : lane;
    const int p1 = (lane + 64 < N) ? piv[lane + 64] : lane + 64;
    for (int k = N - 1; k >= 0; --k) {
      const int ku = __builtin_amdgcn_readfirstlane(k);
      const int q = (ku < 64) ? __builtin_amdgcn_readlane(p0, ku) : __builtin_amdgcn_readlane(p1, ku - 64);
      if (q != ku) {
        const int sk = (ku < 64) ? __builtin_amdgcn_readlane(s0, ku) : __builtin_amdgcn_readlane(s1, ku - 64);
        const int sq = (q < 64) ? __builtin_amdgcn_readlane(s0, q) : __builtin_amdgcn_readlane(s1, q - 64);
        if (ku < 64) {
          if (lane == ku) s0 = sq;
        } else {
          if (lane == ku - 64) s1 = sq;
        }
        if (q < 64) {
          if (lane == q) s0 = sk;
        } else {
          if (lane == q - 64) s1 = sk;
        }
      }
    }
    src[lane] = s0;
    src[lane + 64] = s1;
    if (lane == 0) {
      if (singular) atomicOr(&status[0], (int)VSM_DEVSTAT_SINGULAR);
      atomicAdd(&status[1], 1);
    }
  }
  __syncthreads();
}

struct inv128_ctx {   // what the pivoted path needs beside the strips
  double* AF;         // the A-form's LDS (>= NP * NP doubles)
  double* gjs;        // GJS_BYTES of LDS (pivot rows, column permutation)
  int* status;
};

// G_s = strip of (I - E)^-1 from E's strips; every wave is past a barrier behind the last read of the A-form.
//   K = 1..4: Horner off ONE A-form [E] (K - 1 products, no further barrier, two live strips)
//   K = 7 .. 31: G <- (I + E^(2^l)) G level by level (an A-form store and a product more per level) to the series order K
//   K = 0 (no order from the norm bound): Gauss-Jordan with partial pivoting (above)
// On return other waves may still be reading the A-form.
template <int RT>
__device__ __forceinline__ void invert128(int K, bstrip<RT>& E, bstrip<RT>& G, int N, const inv128_ctx& cx, bpos<RT>& p) {
  if (K == 1) {
    G = E;
    add_identity(G, N, p);
    return;
  }
  if (K == 0) {
    constexpr int NP = 16 * RT;
    if (p.mat_wave && p.col < N) {       // M = I - E, plain column-major (only the N x N block is read back)
      double* mc = cx.AF + p.col * NP + p.kq;
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) mc[16 * ta + 4 * r] = (p.row(ta, r) == p.col ? 1.0 : 0.0) - E.v[ta][r];
    }
    __syncthreads();
    lds_d* M = reinterpret_cast<lds_d*>((unsigned long long)lds_addr128(cx.AF));
    lds_i* piv = reinterpret_cast<lds_i*>((unsigned long long)lds_addr128(cx.gjs));
    gj128_lds<RT>(N, M, piv, piv + 128, blockDim.x >> 6, cx.status, p);
    const bool cok = p.mat_wave && p.col < N;
    const lds_d* gc = M + (cok ? piv[128 + p.col] : 0) * NP + p.kq;
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = gc[16 * ta + 4 * r];
        G.v[ta][r] = (cok && p.row(ta, r) < N) ? v : 0.0;
      }
    return;
  }
  store_af(E, N, p);
  __syncthreads();
  if (K >= 2 && K <= 4) {
    G = E;
    mm128(G, E, p);                  // X1 = E + E E
    for (int j = 2; j < K; ++j) {    // X_j = E + E X_{j-1}
      bstrip<RT> X;
      load_af(X, p);
      mm128(X, G, p);
      G = X;
    }
    add_identity(G, N, p);
    return;
  }
  G = E;
  add_identity(G, N, p);             // sum_{k < 2} E^k
  int cur = 1;                       // [A] = E^cur, E = its strip, G = sum_{k < 2 cur} E^k
  for (int lvl = 0; lvl < 5; ++lvl) {   // (K <= 31: at most four levels)
    bstrip<RT> W2;
    W2.zero();
    mm128(W2, E, p);                 // E^(2 cur)
    cur *= 2;
    __syncthreads();
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    store_af(W2, N, p);              // (everybody finished reading the A-form: the barrier above)
    __syncthreads();
    {
      bstrip<RT> T;
      T.zero();
      mm128(T, G, p);                // E^(2 cur) G = sum_{2 cur <= k < 4 cur} E^k
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += T.v[ta];
    }
    if (K == 2 * cur - 1) break;
    E = W2;
  }
}
// the series order from the norm bound; a norm that is not finite (NaN / Inf input) is flagged and takes the pivoted path
__device__ __forceinline__ int inv_order128(double nrm, int* status) {
  if (!(nrm < 1e300) && threadIdx.x == 0) atomicOr(&status[0], (int)VSM_DEVSTAT_NONFINITE);
  return series_order128(nrm);
}

template <int RT>
__device__ __forceinline__ void load_global128(bstrip<RT>& s, const double* __restrict__ g, int N, const bpos<RT>& p) {
  const bool cok = p.col < N;
  const double* gc0 = g + (long long)N * min(p.col, N - 1) + p.kq;
  asm volatile("" : "+v"(gc0));
  gcd_p gc = (gcd_p)gc0;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ta < RT - 1) {   // N > 16 (RT - 1): these rows exist
        const double v = gc[16 * ta + 4 * r];
        s.v[ta][r] = cok ? v : 0.0;
      } else {
        const int row = p.row(ta, r);
        const double v = gc[min(row, N - 1) - p.kq];
        s.v[ta][r] = (row < N && cok) ? v : 0.0;
      }
    }
}


// The same from / to FP32 arrays (the reference's Float32 runs on these kernels: storage in single, arithmetic in double -- the
// FP64 MFMA rate of these shapes is above what the FP32 operator chains reach)
// 4 x 4 transposition of (register r, 16-lane row q) in two swap stages (v_permlane32_swap on (r0, r2), (r1, r3), then
// v_permlane16_swap on (r0, r1), (r2, r3); tools/permlane_probe.hip): X[r][q] -> X[q][r].  With r_r = accumulator register r (row
// q + 4 r of the tile in lane-row q) it leaves lane-row q with the four CONSECUTIVE rows 4 q .. 4 q + 3 -- one 16-byte access per
// row tile and lane for single-precision arrays, 16 columns x 64 contiguous bytes per instruction instead of sixteen 16-byte
// pieces per request and four requests.  Its own inverse.
typedef float f4u_t __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes, 4-byte aligned (any N)
__device__ __forceinline__ void transpose4_f32(float& r0, float& r1, float& r2, float& r3) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  r0 = __uint_as_float(c[0]);
  r1 = __uint_as_float(c[1]);
  r2 = __uint_as_float(d[0]);
  r3 = __uint_as_float(d[1]);
}
template <int RT>
__device__ __forceinline__ void load_global128(bstrip<RT>& s, const float* g, int N, const bpos<RT>& p) {
  const bool cok = p.col < N;
  const float* gc = g + (long long)N * min(p.col, N - 1) + 4 * p.kq;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) {
    const int r0 = 16 * ta + 4 * p.kq;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (ta < RT - 1 || r0 + 3 < N) {     // (N > 16 (RT - 1): the rows of the tiles before the last exist)
      const f4u_t t = *reinterpret_cast<const f4u_t*>(gc + 16 * ta);
      v0 = t.x;
      v1 = t.y;
      v2 = t.z;
      v3 = t.w;
    } else {
      if (r0 < N) v0 = gc[16 * ta];
      if (r0 + 1 < N) v1 = gc[16 * ta + 1];
      if (r0 + 2 < N) v2 = gc[16 * ta + 2];
    }
    v0 = cok ? v0 : 0.f;
    v1 = cok ? v1 : 0.f;
    v2 = cok ? v2 : 0.f;
    v3 = cok ? v3 : 0.f;
    transpose4_f32(v0, v1, v2, v3);
    s.v[ta][0] = (double)v0;
    s.v[ta][1] = (double)v1;
    s.v[ta][2] = (double)v2;
    s.v[ta][3] = (double)v3;
  }
}
template <int RT>
__device__ __forceinline__ void store_global128(float* g, const bstrip<RT>& s, int N, const bpos<RT>& p) {
  float* gc = g + (long long)N * min(p.col, N - 1) + 4 * p.kq;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) {
    float v0 = (float)s.v[ta][0], v1 = (float)s.v[ta][1], v2 = (float)s.v[ta][2], v3 = (float)s.v[ta][3];
    transpose4_f32(v0, v1, v2, v3);
    const int r0 = 16 * ta + 4 * p.kq;
    if (p.col < N) {
      if (ta < RT - 1 || r0 + 3 < N) {
        f4u_t t;
        t.x = v0;
        t.y = v1;
        t.z = v2;
        t.w = v3;
        *reinterpret_cast<f4u_t*>(gc + 16 * ta) = t;
      } else {
        if (r0 < N) gc[16 * ta] = v0;
        if (r0 + 1 < N) gc[16 * ta + 1] = v1;
        if (r0 + 2 < N) gc[16 * ta + 2] = v2;
      }
    }
  }
}

// global column-major N x N -> the A-form (zero padded): a wave takes whole columns, lane = row (512-byte requests; the 16-lane
// groups of the LDS store are 16 rows of one column: (m ^ const) -- conflict-free)
template <int RT, typename G>
__device__ __forceinline__ void stage_af(double* AF, const G* __restrict__ g, int N, int nw, const bpos<RT>& p) {
  constexpr int NP = 16 * RT, CB = 8, NH = (NP + 63) / 64;   // CB columns x NH row blocks of a wave in flight together
  for (int c0 = p.wave; c0 < NP; c0 += CB * nw) {
    double v[CB][NH];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int col = c0 + i * nw;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int row = 64 * h + p.lane;
        v[i][h] = (row < N && col < N) ? (double)g[row + (long long)N * col] : 0.0;
      }
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int col = c0 + i * nw;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const int row = 64 * h + p.lane;
        if (row < NP && col < NP)
          AF[(col >> 2) * (RT * 64) + (row >> 4) * 64 + ((col & 3) << 4) + ((row & 15) ^ (col & 15))] = v[i][h];
      }
    }
  }
}
template <int RT>
__device__ __forceinline__ void store_global128(double* __restrict__ g, const bstrip<RT>& s, int N, const bpos<RT>& p) {
  const bool cok = p.col < N;
  double* gc0 = g + (long long)N * min(p.col, N - 1) + p.kq;
  asm volatile("" : "+v"(gc0));
  gd_p gc = (gd_p)gc0;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rok = ta < RT - 1 || p.row(ta, r) < N;
      if (rok && cok) gc[16 * ta + 4 * r] = s.v[ta][r];
    }
}

// ---- strip <-> global through a wave-private 16 x 8 tile of LDS (the c8 scheme of vsm_strip_dev.h for RT row tiles) --------------
// load_global128 / store_global128 touch 16 columns x 32 B per instruction -- sixteen quarter-used cache lines per request, and
// sixteen requests per strip tile.  Here a lane moves 16 contiguous bytes of one column (lane = (column c = lane >> 2, q = lane & 3):
// a wave-instruction covers 16 columns x 64 B) and the permutation to the accumulator layout happens in a 16 x 8 tile private to
// the wave (pitch 10 doubles: conflict-free both ways) -- no workgroup barrier: LDS operations of one wave complete in order.
// `issue` only requests (the strip's registers hold the raw doubles meanwhile), `finish` permutes: whatever lies between hides
// the round trip.  xw: 160 doubles of LDS per wave.
constexpr int XS8B = 10;
typedef double d2b_t __attribute__((ext_vector_type(2), aligned(8)));   // 16 bytes, 8-byte aligned (N may be odd)
template <int RT>
__device__ __forceinline__ void load_c8_issue(bstrip<RT>& x, const double* __restrict__ g, int N, const bpos<RT>& p) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  const double* src0 = g + (long long)N * min(col, N - 1);
  asm volatile("" : "+v"(src0));
  gcd_p src = (gcd_p)src0;
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    const int r = 8 * j + 2 * q;
    double a = 0.0, b = 0.0;
    if (col < N && r + 1 < N) {
      const d2b_t t = *reinterpret_cast<const __attribute__((address_space(1))) d2b_t*>(src + r);
      a = t.x;
      b = t.y;
    } else if (col < N && r < N) {
      a = src[r];
    }
    x.v[j >> 1][2 * (j & 1)] = a;
    x.v[j >> 1][2 * (j & 1) + 1] = b;
  }
}
template <int RT>
__device__ __forceinline__ void load_c8_finish(bstrip<RT>& x, const bpos<RT>& p, double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    __builtin_amdgcn_wave_barrier();
    xw[c * XS8B + 2 * q] = x.v[j >> 1][2 * (j & 1)];
    xw[c * XS8B + 2 * q + 1] = x.v[j >> 1][2 * (j & 1) + 1];
    __builtin_amdgcn_wave_barrier();
    x.v[j >> 1][2 * (j & 1)] = xw[p.l15 * XS8B + p.kq];
    x.v[j >> 1][2 * (j & 1) + 1] = xw[p.l15 * XS8B + p.kq + 4];
  }
}
template <int RT>
__device__ __forceinline__ void load_c8(bstrip<RT>& x, const double* __restrict__ g, int N, const bpos<RT>& p, double* __restrict__ xw) {
  load_c8_issue(x, g, N, p);
  load_c8_finish(x, p, xw);
}
template <int RT>
__device__ __forceinline__ void store_c8(double* __restrict__ g, const bstrip<RT>& x, int N, const bpos<RT>& p, double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  double* dst0 = g + (long long)N * min(col, N - 1);
  asm volatile("" : "+v"(dst0));
  gd_p dst = (gd_p)dst0;
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    __builtin_amdgcn_wave_barrier();
    xw[p.l15 * XS8B + p.kq] = x.v[j >> 1][2 * (j & 1)];
    xw[p.l15 * XS8B + p.kq + 4] = x.v[j >> 1][2 * (j & 1) + 1];
    __builtin_amdgcn_wave_barrier();
    d2b_t t;
    t.x = xw[c * XS8B + 2 * q];
    t.y = xw[c * XS8B + 2 * q + 1];
    const int r = 8 * j + 2 * q;
    if (col < N && r + 1 < N)
      *reinterpret_cast<__attribute__((address_space(1))) d2b_t*>(dst + r) = t;
    else if (col < N && r < N)
      dst[r] = t.x;
  }
}

// ---- strip <-> global with v_permlane16_swap (round 5): the access geometry of the c8 scheme without its LDS round trips -------
// A lane holds rows kq + 4 r of a row tile.  v_permlane16_swap trades the odd 16-lane rows of register r = 2h with the even ones of
// r = 2h + 1 (tools/permlane_probe.hip); afterwards lane-row kq holds the ADJACENT rows rb, rb + 1 with rb = 4 (kq & 1) + 2 (kq >> 1)
// + 8 h of its column: one 16-byte access per pair, 16 columns x 64 contiguous bytes per instruction, two instructions per row tile
// instead of four.  The swap is its own inverse: loads request the pairs (`issue`) and swap when they are needed (`finish`: four
// VALU instructions per pair, no LDS, no barrier), stores swap first.
__device__ __forceinline__ void swap16_128(double& a, double& b) {
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  a = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
  b = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}
template <int RT>
__device__ __forceinline__ void load_sw_issue(bstrip<RT>& x, const double* __restrict__ g, int N, const bpos<RT>& p) {
  const double* src0 = g + (long long)N * min(p.col, N - 1) + 4 * (p.kq & 1) + 2 * (p.kq >> 1);
  asm volatile("" : "+v"(src0));
  gcd_p src = (gcd_p)src0;
  const bool cok = p.col < N;
  const int rb = 4 * (p.kq & 1) + 2 * (p.kq >> 1);
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    double a = 0.0, b = 0.0;
    if (j < 2 * (RT - 1)) {   // N > 16 (RT - 1): these rows exist
      const d2b_t t = *reinterpret_cast<const __attribute__((address_space(1))) d2b_t*>(src + 8 * j);
      a = cok ? t.x : 0.0;
      b = cok ? t.y : 0.0;
    } else {
      const int r = 8 * j + rb;
      if (cok && r + 1 < N) {
        const d2b_t t = *reinterpret_cast<const __attribute__((address_space(1))) d2b_t*>(src + 8 * j);
        a = t.x;
        b = t.y;
      } else if (cok && r < N) {
        a = src[8 * j];
      }
    }
    x.v[j >> 1][2 * (j & 1)] = a;
    x.v[j >> 1][2 * (j & 1) + 1] = b;
  }
}
template <int RT>
__device__ __forceinline__ void load_sw_finish(bstrip<RT>& x) {
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    double a = x.v[j >> 1][2 * (j & 1)], b = x.v[j >> 1][2 * (j & 1) + 1];
    swap16_128(a, b);
    x.v[j >> 1][2 * (j & 1)] = a;
    x.v[j >> 1][2 * (j & 1) + 1] = b;
  }
}
template <int RT>
__device__ __forceinline__ void store_sw(double* __restrict__ g, const bstrip<RT>& x, int N, const bpos<RT>& p) {
  double* dst0 = g + (long long)N * min(p.col, N - 1) + 4 * (p.kq & 1) + 2 * (p.kq >> 1);
  asm volatile("" : "+v"(dst0));
  gd_p dst = (gd_p)dst0;
  const bool cok = p.col < N;
  const int rb = 4 * (p.kq & 1) + 2 * (p.kq >> 1);
#pragma unroll
  for (int j = 0; j < 2 * RT; ++j) {
    double a = x.v[j >> 1][2 * (j & 1)], b = x.v[j >> 1][2 * (j & 1) + 1];
    swap16_128(a, b);
    d2b_t t;
    t.x = a;
    t.y = b;
    const int r = 8 * j + rb;
    if (cok && (j < 2 * (RT - 1) || r + 1 < N))
      *reinterpret_cast<__attribute__((address_space(1))) d2b_t*>(dst + 8 * j) = t;
    else if (cok && r < N)
      dst[8 * j] = a;
  }
}

// Which of the three a kernel uses is a matter of measurement.  Round 4 (direct 8-byte accesses against the c8 tiles, forward and
// linearized runs, 2000 points): five row tiles and fewer 3 ... 4 % faster direct, six and more 0 ... 4.5 % faster through the tiles.
// Round 5: the permlane pairs replace both (VSM_S128_OLD_IO restores the round-4 choice for A/B builds).
template <int RT>
struct use_c8 {
#if defined(VSM_AB_SWITCHES) && defined(VSM_S128_OLD_IO)
  static constexpr bool value = RT >= 6, sw = false;
#else
  static constexpr bool value = false, sw = true;
#endif
};
template <int RT>
__device__ __forceinline__ void ldg_issue(bstrip<RT>& x, const double* __restrict__ g, int N, const bpos<RT>& p) {
  if constexpr (use_c8<RT>::sw) load_sw_issue(x, g, N, p);
  else if constexpr (use_c8<RT>::value) load_c8_issue(x, g, N, p);
  else load_global128(x, g, N, p);
}
// (ST: the element type the matching ldg_issue read -- FP32 arrays are always read directly, in the final layout)
template <typename ST = double, int RT>
__device__ __forceinline__ void ldg_finish(bstrip<RT>& x, const bpos<RT>& p, double* __restrict__ xw) {
  if constexpr (sizeof(ST) == 8) {
    if constexpr (use_c8<RT>::sw) load_sw_finish(x);
    else if constexpr (use_c8<RT>::value) load_c8_finish(x, p, xw);
  }
}
template <int RT>
__device__ __forceinline__ void ldg(bstrip<RT>& x, const double* __restrict__ g, int N, const bpos<RT>& p, double* __restrict__ xw) {
  ldg_issue(x, g, N, p);
  ldg_finish(x, p, xw);
}
template <int RT>
__device__ __forceinline__ void stg(double* __restrict__ g, const bstrip<RT>& x, int N, const bpos<RT>& p, double* __restrict__ xw) {
  if constexpr (use_c8<RT>::sw) store_sw(g, x, N, p);
  else if constexpr (use_c8<RT>::value) store_c8(g, x, N, p, xw);
  else store_global128(g, x, N, p);
}
template <int RT>
__device__ __forceinline__ void ldg(bstrip<RT>& x, const float* __restrict__ g, int N, const bpos<RT>& p, double*) {
  load_global128(x, g, N, p);
}
template <int RT>
__device__ __forceinline__ void ldg_issue(bstrip<RT>& x, const float* __restrict__ g, int N, const bpos<RT>& p) {
  load_global128(x, g, N, p);
}
template <int RT>
__device__ __forceinline__ void stg(float* __restrict__ g, const bstrip<RT>& x, int N, const bpos<RT>& p, double*) {
  store_global128(g, x, N, p);
}

}  // namespace
}  // namespace vsm
