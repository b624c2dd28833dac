// Device building blocks of the FP32 NATIVE-LAYOUT layer kernels (vsm_native32.hip): the scheme of vsm_native_dev.h -- wave w owns
// the 16-column strip w of every matrix in accumulator registers that ARE the B operands of the next product, left operands as
// A-forms in LDS, the composite in strip records between the layer steps -- in single precision on v_mfma_f32_16x16x4_f32, for
// sub-problems of n <= 128 rows in RT = 1..8 row tiles.
//
// Accumulator layout.  v_mfma_f32_16x16x4 leaves element (M index i, column j) of a tile in lane 16 (i >> 2) + j, register i & 3
// (the FP64 instruction: lane 16 (i & 3) + j, register i >> 2).  The kernels feed the A operand with the rows of a tile PERMUTED:
// A-lane m supplies matrix row pi(m) = 4 (m & 3) + (m >> 2) of the tile, so that M index 4 kq + r of the result is matrix row
// 4 r + kq -- lane group kq, register r hold row 16 ta + 4 r + kq of the lane's column: EXACTLY the FP64 family's strip layout.
// Register (ta, r) of a strip then is the B operand of k-step 4 ta + r (k = 4 ks + kq, contiguous), a block of n rows needs
// KS = ceil(n / 4) k-steps as in FP64, the rider columns, the diagonal lanes and the D-symmetry masks are the same; the
// permutation lives in the A-form's index function alone.
//
// Workgroups.  A workgroup of six waves lands on the four SIMDs as 2-2-1-1 and the dispatcher never puts a second one beside it
// (tools/occ_probe.hip): five and six row tiles run TWO spectral points per workgroup (PP = 2: 10 / 12 waves,
// three per SIMD).  The two points share the workgroup's barriers and nothing else; the only data-dependent barrier count -- the
// order of the series inverse -- is made uniform by running the inverse once per DISTINCT order of the pair, each point keeping
// the result of its own order (n32invert_pair): the arithmetic of a point never depends on its neighbour.
#pragma once
#include "vsm_internal.h"
#include "vsm_lds.h"
#include "vsm_inverse.h"
#include "vsm_elemental.h"

#ifndef VSM_N32_WPS4
#define VSM_N32_WPS4 4
#endif
#ifndef VSM_N32_SINGLE_FRAG_RT
#define VSM_N32_SINGLE_FRAG_RT 9
#endif

namespace vsm {
namespace {

using n32lds_f = __attribute__((address_space(3))) float;
using n32lds_i = __attribute__((address_space(3))) int;
__device__ __forceinline__ unsigned n32lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}

template <int RT>
struct n32geo {
  static constexpr int NP = 16 * RT;                  // padded size of a sub-problem
  static constexpr int NTP = 64 * RT;                 // threads per spectral point (RT waves)
  static constexpr int PP = (RT == 5 || RT == 6) ? 2 : 1;   // spectral points per workgroup (7 / 8 row tiles: 7 / 8 waves fill the SIMDs by themselves)
  static constexpr int NT = NTP * PP;                 // threads per workgroup
  static constexpr int AF = NP * NP;                  // floats per A-form / per pre-pass image / per native matrix
  static constexpr int PRE_STRIDE = 2 * AF + 3 * NP;  // pre-pass record: [r-+*] image, [t++] image, j0+, j0-, aux (aux[0] = expk)
  static constexpr int COMP_STRIDE = 4 * AF + 2 * NP; // native composite of one point: R-+, R+-, T++, T--, J0+, J0-
  // minimum waves per SIMD the register budget is set for (launch bounds): a strip is 4 RT registers
  static constexpr int WPS = RT <= 3 ? 4 : (RT == 4 ? VSM_N32_WPS4 : (RT <= 6 ? 3 : 2));
};
enum { N32_RMP = 0, N32_RPM = 1, N32_TPP = 2, N32_TMM = 3 };

// A-form: 64-float blocks per (k-step ks = k >> 2, row tile t = row >> 4), block index ks RT + t; inside a block element
// (row-local rl = 4 rr + kqr, k' = k & 3) sits at word  16 k' + 4 kqr + (rr ^ (ks & 3)).  The fragment read of lane (m = l15, kq)
// -- matrix row pi(m): rr = m & 3, kqr = m >> 2; k' = kq -- is word 16 kq + 4 (m >> 2) + ((m & 3) ^ (ks & 3)): the 64 lanes cover
// the block's 64 banks once.  A strip store of register (ta, r) -- lane (l15, kq): row 16 ta + 4 r + kq, k = 16 w + l15, i.e.
// block (4 w + q, ta) with q = l15 >> 2, k' = l15 & 3 -- writes word 16 k' + 4 kq + (r ^ q): over the lanes (k', kq, q) again
// every bank once.
template <int RT>
__host__ __device__ __forceinline__ int n32af_idx(int row, int k) {
  const int ks = k >> 2, kk = k & 3, rl = row & 15;
  return (ks * RT + (row >> 4)) * 64 + ((kk << 4) | ((rl & 3) << 2) | ((rl >> 2) ^ (ks & 3)));
}
// element (i, j) of a native matrix: wave j >> 4 owns the column; RT units of 64 lanes x 16 bytes (the lane's four rows of a tile)
template <int RT>
__host__ __device__ __forceinline__ int n32nat_idx(int i, int j) {
  const int w = j >> 4, l15 = j & 15, ta = i >> 4, mm = i & 15, kq = mm & 3, r = mm >> 2;
  return ((w * RT + ta) * 64 + ((kq << 4) | l15)) * 4 + r;
}

template <int RT>
struct n32strip {
  f4_t v[RT];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < RT; ++a) v[a] = acc_zero<float>();
  }
};

// Per-lane byte addresses: fragment bases by (ks & 3), strip-element bases by r; everything else is an instruction offset.
template <int RT>
struct n32pos {
  int lane, wave, l15, kq, col, tid, pt;   // wave / tid: inside the spectral point; pt: the point's slot in the workgroup
  bool live;                               // false: the filler point of a workgroup of two (odd batch): no global store, no status
  unsigned fb[4];
  unsigned sb[4];
  __device__ __forceinline__ n32pos(const void* base) {
    lane = threadIdx.x & 63;
    const int wg_wave = threadIdx.x >> 6;
    pt = n32geo<RT>::PP == 1 ? 0 : wg_wave / RT;
    wave = wg_wave - pt * RT;
    tid = (int)threadIdx.x - pt * n32geo<RT>::NTP;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    live = true;
    const unsigned L = n32lds_addr(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = L + 4u * (unsigned)((kq << 4) | ((l15 >> 2) << 2) | ((l15 & 3) ^ j));
    const int q = l15 >> 2, kk = l15 & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) sb[r] = L + 4u * (unsigned)(((4 * wave + q) * RT * 64) + ((kk << 4) | (kq << 2) | (r ^ q)));
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + kq + 4 * r; }
  __device__ __forceinline__ const n32lds_f* aptr(unsigned dA, int t, int ks) const {
    return reinterpret_cast<const n32lds_f*>((unsigned long long)(fb[ks & 3] + dA)) + 64 * (ks * RT + t);
  }
  __device__ __forceinline__ n32lds_f* sptr(unsigned dA, int ta, int r) const {
    return reinterpret_cast<n32lds_f*>((unsigned long long)(sb[r] + dA)) + 64 * ta;
  }
  // hide the loop invariance of the bases from LICM (hoisting every derived address costs registers)
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[j]));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(sb[r]));
  }
};

#define VSM_N32KSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)

// value of the neighbour lane (lane ^ 1): DPP quad permutation
__device__ __forceinline__ float n32dpp_swap1(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
}

// ---- products: acc += [A] B, A-form at byte distance dA, B a strip in registers ------------------------------------------------
// One fragment register per row tile: the fragment of the next k-step is requested right behind the MFMA that has consumed the
// current one (RT - 1 MFMAs = 32 (RT - 1) cycles of latency cover below it from three row tiles on); one and two row tiles keep
// two sets (requested a whole k-step ahead).
// (Z / Z1 / Z2: the accumulator starts from zero -- the first k-step takes the constant 0 as its addend: no register is cleared)
template <int RT>
struct n32frag {
  static constexpr int SETS = (RT >= VSM_N32_SINGLE_FRAG_RT) ? 1 : 2;
  float a[SETS][RT];
  __device__ __forceinline__ void first(const n32pos<RT>& p, unsigned dA) {
#pragma unroll
    for (int t = 0; t < RT; ++t) a[0][t] = *p.aptr(dA, t, 0);
  }
  template <int KS>
  __device__ __forceinline__ void ahead(const n32pos<RT>& p, unsigned dA, int ks) {   // SETS == 2: before the MFMAs of k-step ks
    if (SETS == 2 && ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < RT; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
  }
  __device__ __forceinline__ float get(int ks, int t) const { return a[SETS == 2 ? (ks & 1) : 0][t]; }
  template <int KS>
  __device__ __forceinline__ void behind(const n32pos<RT>& p, unsigned dA, int ks, int t) {   // SETS == 1: behind the MFMA (ks, t)
    if (SETS == 1 && ks + 1 < KS) a[0][t] = *p.aptr(dA, t, ks + 1);
  }
};
template <int RT, int KS, bool Z = false>
__device__ __forceinline__ void n32mm(n32strip<RT>& acc, unsigned dA, const n32strip<RT>& B, n32pos<RT>& p) {
  p.opaque();
  n32frag<RT> f;
  f.first(p, dA);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    f.template ahead<KS>(p, dA, ks);
    const float b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      acc.v[t] = mfma<float>::mma(f.get(ks, t), b, (Z && ks == 0) ? acc_zero<float>() : acc.v[t]);
      f.template behind<KS>(p, dA, ks, t);
    }
    VSM_N32KSTEP_FENCE();
  }
}
// out = C0 + [A] B
template <int RT, int KS>
__device__ __forceinline__ void n32mm_c(n32strip<RT>& out, const n32strip<RT>& C0, unsigned dA, const n32strip<RT>& B, n32pos<RT>& p) {
  p.opaque();
  n32frag<RT> f;
  f.first(p, dA);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    f.template ahead<KS>(p, dA, ks);
    const float b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      out.v[t] = mfma<float>::mma(f.get(ks, t), b, ks == 0 ? C0.v[t] : out.v[t]);
      f.template behind<KS>(p, dA, ks, t);
    }
    VSM_N32KSTEP_FENCE();
  }
}
// acc1 += [A] B1 ; acc2 += [A] B2  (shared fragments)
template <int RT, int KS, bool Z1 = false, bool Z2 = false>
__device__ __forceinline__ void n32mm2(n32strip<RT>& acc1, n32strip<RT>& acc2, unsigned dA, const n32strip<RT>& B1,
                                       const n32strip<RT>& B2, n32pos<RT>& p) {
  p.opaque();
  n32frag<RT> f;
  f.first(p, dA);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    f.template ahead<KS>(p, dA, ks);
    const float b1 = B1.v[ks >> 2][ks & 3], b2 = B2.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      acc1.v[t] = mfma<float>::mma(f.get(ks, t), b1, (Z1 && ks == 0) ? acc_zero<float>() : acc1.v[t]);
      acc2.v[t] = mfma<float>::mma(f.get(ks, t), b2, (Z2 && ks == 0) ? acc_zero<float>() : acc2.v[t]);
      f.template behind<KS>(p, dA, ks, t);
    }
    VSM_N32KSTEP_FENCE();
  }
}

// ---- strip <-> A-form ---------------------------------------------------------------------------------------------------------
template <int RT>
__device__ __forceinline__ void n32store(unsigned dA, const n32strip<RT>& s, const n32pos<RT>& p) {
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) *p.sptr(dA, ta, r) = s.v[ta][r];
}
template <int RT>
__device__ __forceinline__ void n32load(n32strip<RT>& s, unsigned dA, const n32pos<RT>& p) {
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = *p.sptr(dA, ta, r);
}

// ---- strip <-> native matrix in global memory: RT coalesced 16-byte accesses per lane ------------------------------------------
template <int RT>
__device__ __forceinline__ void n32ld_native(n32strip<RT>& s, const float* __restrict__ mat, const n32pos<RT>& p) {
  const f4_t* g = reinterpret_cast<const f4_t*>(mat) + (p.wave * RT) * 64 + p.lane;
#pragma unroll
  for (int u = 0; u < RT; ++u) s.v[u] = g[u * 64];
}
template <int RT>
__device__ __forceinline__ void n32st_native(float* __restrict__ mat, const n32strip<RT>& s, const n32pos<RT>& p, bool live) {
  f4_t* g = reinterpret_cast<f4_t*>(mat) + (p.wave * RT) * 64 + p.lane;
  if (live) {
#pragma unroll
    for (int u = 0; u < RT; ++u) g[u * 64] = s.v[u];
  }
}

// ---- image (pre-pass record in global memory) -> A-form by LDS DMA: RT instructions of 1 KB per wave ----------------------------
template <int RT>
__device__ __forceinline__ void n32copy_image(float* L, const float* __restrict__ g, const n32pos<RT>& p) {
#pragma unroll
  for (int i = 0; i < RT; ++i) {
    const int blk = (i * RT + p.wave) * 256;   // floats
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blk + 4 * p.lane),
                                     (__attribute__((address_space(3))) void*)(L + blk), 16, 0, 0);
  }
}

// ---- D X D with the parities of the lane's rows and of its column as bits (doubling.jl:178-201) ---------------------------------
template <int RT>
struct n32dpar {
  unsigned rows;   // bit 4 ta + r set = sign flip of that element
  __device__ __forceinline__ n32dpar(const float* usg, const n32pos<RT>& p) {
    rows = 0;
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) rows |= (usg[p.row(ta, r)] < 0.f ? 1u : 0u) << (4 * ta + r);
    if (usg[p.col] < 0.f) rows = ~rows;
  }
};
template <int RT>
__device__ __forceinline__ void n32dsym(n32strip<RT>& d, const n32strip<RT>& x, const n32dpar<RT>& dp) {
  unsigned rows = dp.rows;   // (opaque: the sign masks are two VALU operations each -- not 4 RT registers kept from call to call)
  asm volatile("" : "+v"(rows));
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * ta + r;
      const unsigned sbit = (rows << (31 - n)) & 0x80000000u;
      d.v[ta][r] = __int_as_float(__float_as_int(x.v[ta][r]) ^ (int)sbit);
    }
}

// ---- norm bound / series order / inverses -------------------------------------------------------------------------------------
// Frobenius-norm bound of the n x n block whose strips the waves of a point hold (rows >= n are zero by construction; the rider
// and padding columns are excluded).  ONE barrier inside.  other: the same bound for the other point of the workgroup (PP = 2;
// its partial sums sit `other_bytes` away in LDS), else the point's own.
template <int RT, typename SM>
__device__ __forceinline__ float n32norm(const n32strip<RT>& e, int n, SM& sm, int& slot, const n32pos<RT>& p, float& other) {
  float ss = 0.f;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss = fmaf(e.v[ta][r], e.v[ta][r], ss);
  ss = (p.col < n) ? ss : 0.f;
  const float ws = wave_sum(ss);
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < RT; ++w) tot += sm.red[slot][w];
  const float own = sqrtf(tot) * 1.001f;
  other = own;
  if constexpr (n32geo<RT>::PP == 2) {
    const SM& so = (&sm)[p.pt == 0 ? 1 : -1];
    float to = 0.f;
#pragma unroll
    for (int w = 0; w < RT; ++w) to += so.red[slot][w];
    other = sqrtf(to) * 1.001f;
  }
  slot ^= 1;
  return own;
}
// smallest order K of {1,2,3,4,7,8,15,16,31} with nrm^(K+1) / (1 - nrm) <= eps(Float32) / 4; 0: no series (pivoted inverse)
__device__ __forceinline__ int n32series_order(float nrm) {
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
template <int RT>
__device__ __forceinline__ void n32add_identity(n32strip<RT>& G, int n, const n32pos<RT>& p) {
  // a lane owns at most one diagonal element, in row tile ta = wave: r = l15 >> 2, kq = l15 & 3
  const bool dl = p.kq == (p.l15 & 3) && p.col < n;
  const int dr = p.l15 >> 2;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
    if (ta == p.wave) {
#pragma unroll
      for (int r = 0; r < 4; ++r) G.v[ta][r] += (dl && dr == r) ? 1.0f : 0.0f;
    }
}

// In-place Gauss-Jordan with partial pivoting (the pivot rule of getrf: the contract of the reference's LU,
// cpu_batched.jl:32-47) of the n x n matrix M (plain column-major, pitch NP) in LDS; lanes = rows (two per lane beyond 64 rows),
// wave w = columns w, w + RT, ...  Per pivot step every wave reads column k and finds the pivot redundantly (DPP maximum +
// ballot), a barrier, every wave updates its columns with the row interchange folded in, a barrier.  src[x] = column of M that is
// column x of the inverse.  status (vsm_device_status): [0] |= VSM_DEVSTAT_SINGULAR on an exactly zero pivot, [1] += 1 per
// pivoted inverse.  `mine`: false = the result of this pass is discarded (the other point's order, n32invert_pair): no status.
template <int RT>
__device__ __forceinline__ void n32gj_lds(int n, n32lds_f* M, n32lds_i* piv, n32lds_i* src, int* status, bool mine, const n32pos<RT>& p) {
  constexpr int NP = 16 * RT;
  constexpr bool TWO = NP > 64;
  const int i0 = p.lane, i1 = p.lane + 64;
  const bool ok0 = i0 < n, ok1 = TWO && i1 < n;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    const n32lds_f* ck = M + k * NP;
    float f0 = ok0 ? ck[i0] : 0.f, f1 = ok1 ? ck[i1] : 0.f;
    const float v0 = (ok0 && i0 >= k) ? fabsf(f0) : -1.f, v1 = (ok1 && i1 >= k) ? fabsf(f1) : -1.f;
    const float best = wave_max(v0 > v1 ? v0 : v1);
    const unsigned long long m0 = __ballot(v0 >= 0.f && v0 == best);
    const unsigned long long m1 = __ballot(v1 >= 0.f && v1 == best);
    const int pr = m0 ? (__ffsll((long long)m0) - 1) : (m1 ? 64 + __ffsll((long long)m1) - 1 : k);
    const float ckk = ck[k], pv = ck[pr];
    const float d = 1.0f / pv;
    singular |= pv == 0.f;
    f0 = (i0 == pr) ? ckk : f0;
    f1 = (i1 == pr) ? ckk : f1;
    if (p.tid == 0) piv[k] = pr;
    __syncthreads();
    for (int j = p.wave; j < n; j += RT) {
      n32lds_f* cj = M + j * NP;
      const bool isk = j == k;
      const float a = cj[pr], b = cj[k];
      const float u = isk ? d : a * d;
      float x0 = ok0 ? cj[i0] : 0.f;
      x0 = isk ? 0.f : ((i0 == pr) ? b : x0);
      x0 = (i0 == k) ? u : fmaf(-f0, u, x0);
      if (ok0) cj[i0] = x0;
      if constexpr (TWO) {
        float x1 = ok1 ? cj[i1] : 0.f;
        x1 = isk ? 0.f : ((i1 == pr) ? b : x1);
        x1 = (i1 == k) ? u : fmaf(-f1, u, x1);
        if (ok1) cj[i1] = x1;
      }
    }
    __syncthreads();
  }
  if (p.wave == 0) {   // the column permutation that undoes the row interchanges: src = P applied right to left (one thread: rare path)
    if (p.lane == 0) {
      for (int x = 0; x < n; ++x) src[x] = x;
      for (int k = n - 1; k >= 0; --k) {
        const int q = piv[k];
        if (q != k) {
          const int t = src[k];
          src[k] = src[q];
          src[q] = t;
        }
      }
      if (mine) {
        if (singular) atomicOr(&status[0], (int)VSM_DEVSTAT_SINGULAR);
        atomicAdd(&status[1], 1);
      }
    }
  }
  __syncthreads();
}

struct n32inv_ctx {
  unsigned dW;     // byte distance of the scratch A-form (= P: 0)
  float* W;        // the same A-form as a pointer (plain column-major use by the pivoted path)
  int* gjs;        // 2 NP ints of LDS: piv[NP], src[NP]
  int* status;
};

// G = (I - E)^-1 of order K.  K = 1..8: Horner's rule with ONE A-form store
//   X_0 = E,  X_{j+1} = E + E X_j  ->  G = I + X_{K-1}
// (A = [E] for every product: no barrier between the K - 1 products; from the third term on E is re-read from its A-form);
// K = 15, 16, 31: G <- (I + E^(2^l)) G level by level; K = 0: pivoted Gauss-Jordan of I - E in LDS.  The rider columns (>= n) of
// E are NOT cleared on the Horner path (every product maps a column of B to the same column of the result); the other paths
// lose them -- the callers restore what they need.  Every wave is past the norm reduction's barrier; on return other waves may
// still be reading W.  E is left untouched.
template <int RT, int KS>
__device__ __forceinline__ void n32invert(int K, const n32strip<RT>& E, n32strip<RT>& G, int n, const n32inv_ctx& cx, bool mine,
                                          n32pos<RT>& p) {
  constexpr int NP = 16 * RT;
  if (K >= 1 && K <= 8) {
    if (K == 1) {
      G = E;
    } else {
      n32store(cx.dW, E, p);
      __syncthreads();
      n32mm_c<RT, KS>(G, E, cx.dW, E, p);   // X1 = E + E E
      for (int j = 2; j < K; ++j) {          // X_j = E + E X_{j-1}
        n32strip<RT> X;
        n32load(X, cx.dW, p);
        n32mm<RT, KS>(X, cx.dW, G, p);
        G = X;
      }
    }
    n32add_identity(G, n, p);
    return;
  }
  if (K == 0) {
    if (p.col < n) {     // M = I - E, plain column-major (only the n x n block is read back)
      float* mc = cx.W + p.col * NP + p.kq;
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) mc[16 * ta + 4 * r] = (p.row(ta, r) == p.col ? 1.0f : 0.0f) - E.v[ta][r];
    }
    __syncthreads();
    n32lds_f* M = reinterpret_cast<n32lds_f*>((unsigned long long)n32lds_addr(cx.W));
    n32lds_i* piv = reinterpret_cast<n32lds_i*>((unsigned long long)n32lds_addr(cx.gjs));
    n32gj_lds<RT>(n, M, piv, piv + NP, cx.status, mine, p);
    const bool cok = p.col < n;
    const n32lds_f* gc = M + (cok ? piv[NP + p.col] : 0) * NP + p.kq;
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = gc[16 * ta + 4 * r];
        G.v[ta][r] = (cok && p.row(ta, r) < n) ? v : 0.0f;
      }
    return;
  }
  n32strip<RT> Ec;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) Ec.v[ta][r] = (p.col < n) ? E.v[ta][r] : 0.0f;
  G = Ec;
  n32add_identity(G, n, p);
  n32store(cx.dW, Ec, p);
  __syncthreads();
  int cur = 1;   // W = E^cur (A-form), Ec = its strip, G = strip of sum_{k < 2 cur} E^k
  for (int lvl = 0; lvl < 5; ++lvl) {
    n32strip<RT> W2;
    W2.zero();
    n32mm<RT, KS>(W2, cx.dW, Ec, p);   // E^(2 cur)
    cur *= 2;
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    __syncthreads();   // everybody finished reading W
    n32store(cx.dW, W2, p);
    __syncthreads();
    {
      n32strip<RT> T;
      T.zero();
      n32mm<RT, KS>(T, cx.dW, G, p);   // E^cur G  (powers of E commute)
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += T.v[ta];
    }
    if (K == 2 * cur - 1) break;
    Ec = W2;
  }
}
// The inverse of a point given its own order and the order of the other point of the workgroup (PP = 2; K_other = K_own when the
// point is alone).  The two points share the workgroup's barriers, and the barrier sequence of an inverse depends on its order:
//   K = 1: none;  K = 2..8 (Horner): ONE, behind the A-form store, however many products follow;  the long orders and the pivoted
//   path: level- / size-dependent.
// Both orders in 1..8 (practically always): each point runs its own order, a point of order 1 beside one of order >= 2 adds that
// barrier.  Otherwise both points walk the out-of-line inverse of point 0's order and then -- when the orders differ -- that of
// point 1's, each keeping the result of its own: the arithmetic of a point never depends on its neighbour.  Out of line: those
// paths are rare and their live strips stay out of the hot path's register allocation.
template <int RT, int KS>
__device__ __attribute__((noinline)) void n32invert_rare(int K0, int K1, const n32strip<RT>& E_in, n32strip<RT>& G_out, int n,
                                                         const n32inv_ctx& cx_in, n32pos<RT>& p_in) {
  n32strip<RT> E = E_in, G;
  n32pos<RT> p = p_in;
  const n32inv_ctx cx = cx_in;
  const int npass = (K0 != K1) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const bool mine = npass == 1 || pass == p.pt;
    if (pass) __syncthreads();   // the first pass's readers of W are done
    n32strip<RT> Gt;
    n32invert<RT, KS>(pass == 0 ? K0 : K1, E, Gt, n, cx, mine && p.live, p);
    if (mine) G = Gt;
  }
  G_out = G;
}
template <int RT, int KS>
__device__ __forceinline__ void n32invert_pair(int K_own, int K_other, const n32strip<RT>& E, n32strip<RT>& G, int n,
                                               const n32inv_ctx& cx, n32pos<RT>& p) {
  if (K_own >= 1 && K_own <= 8 && K_other >= 1 && K_other <= 8) {
    if (K_own == 1) {
      G = E;
      if (n32geo<RT>::PP == 2 && K_other > 1) __syncthreads();   // (the other point's barrier behind its A-form store)
    } else {
      n32store(cx.dW, E, p);
      __syncthreads();
      n32mm_c<RT, KS>(G, E, cx.dW, E, p);   // X1 = E + E E
      for (int j = 2; j < K_own; ++j) {      // X_j = E + E X_{j-1}
        n32strip<RT> X;
        n32load(X, cx.dW, p);
        n32mm<RT, KS>(X, cx.dW, G, p);
        G = X;
      }
    }
    n32add_identity(G, n, p);
    return;
  }
#if !(defined(VSM_N32_NO_RARE) && defined(VSM_AB_SWITCHES))   // (A/B build only: the hot path without the call site of the rare one)
  n32strip<RT> Es = E, Gs;
  n32pos<RT> ps = p;
  n32invert_rare<RT, KS>(p.pt == 0 ? K_own : K_other, p.pt == 0 ? K_other : K_own, Es, Gs, n, cx, ps);
  G = Gs;
#endif
}
__device__ __forceinline__ int n32inv_order(float nrm, int* status, bool report) {
  if (!(nrm < 1e30f) && report) atomicOr(&status[0], (int)VSM_DEVSTAT_NONFINITE);
  return n32series_order(nrm);
}

// (mat-vec source path: blocks without spare columns, 4 KS + 2 > 16 RT, i.e. n = 13..16, 29..32, ... 93..96)
// LDS block of one spectral point (six row tiles with the mat-vec slots, two points: 163 712 of the CU's 163 840 bytes)
template <int RT, bool MV>
struct n32smem;
template <int RT>
struct n32smem<RT, false> {
  float P[n32geo<RT>::AF];
  float Q[n32geo<RT>::AF];
  float vec[8][n32geo<RT>::NP];
  float usg[n32geo<RT>::NP];   // -1 on the U/V rows of the sub-problem, +1 elsewhere
  float red[2][8];
  int gjs_[2 * n32geo<RT>::NP];
  __device__ __forceinline__ int* gjs() { return gjs_; }
};
template <int RT>
struct n32smem<RT, true> {
  float P[n32geo<RT>::AF];
  float Q[n32geo<RT>::AF];
  float mv[2][16 * RT];
  float vec[6][n32geo<RT>::NP];
  float usg[n32geo<RT>::NP];
  float red[2][8];
  int gjs_[2 * n32geo<RT>::NP];
  __device__ __forceinline__ int* gjs() { return gjs_; }
};
static_assert(sizeof(n32smem<6, true>) * 2 <= 163840 && sizeof(n32smem<6, false>) * 2 <= 163840, "six row tiles: two points per CU");
static_assert(sizeof(n32smem<4, true>) * 4 <= 163840, "four row tiles: four workgroups per CU");
static_assert(sizeof(n32smem<8, true>) <= 163840 && sizeof(n32smem<8, false>) <= 163840, "eight row tiles: one workgroup per CU");

// Source vectors without spare columns: y = sc [A] x as a VALU mat-vec over the A-form.  Wave w takes the rows of row tile w
// through the fragment pattern of the products (lane (m, kq): row pi(m), k = 4 ks + kq -- conflict-free, per-lane bases +
// instruction offsets), reduces over the four lane groups and writes its sixteen finished rows: no partial sums.
template <int RT, int KS>
__device__ __forceinline__ void n32mv_rows(unsigned dA, const float* x, float sc, float* out, n32pos<RT>& p) {
  unsigned ab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ab[j] = p.fb[j] + dA + 256u * (unsigned)p.wave;
    asm volatile("" : "+v"(ab[j]));
  }
  unsigned xb = n32lds_addr(x) + 4u * (unsigned)p.kq;
  asm volatile("" : "+v"(xb));
  float acc = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float a = *(reinterpret_cast<const n32lds_f*>((unsigned long long)ab[ks & 3]) + 64 * ks * RT);
    const float xv = *(reinterpret_cast<const n32lds_f*>((unsigned long long)xb) + 4 * ks);
    acc = fmaf(a, xv, acc);
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if (p.kq == 0) out[16 * p.wave + 4 * (p.l15 & 3) + (p.l15 >> 2)] = acc * sc;
}

}  // namespace
}  // namespace vsm
