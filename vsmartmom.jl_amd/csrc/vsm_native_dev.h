// Device building blocks of the NATIVE-LAYOUT layer kernels (vsm_native.hip): FP64 column strips for sub-problems of n <= 60
// rows in RT = 1..4 row tiles, TWO A-forms in LDS, the composite layer kept in the kernels' own strip layout between the layer
// steps of a run.  The scheme is that of vsm_strip.hip (wave w owns the 16-column strip w of every matrix in the accumulator
// layout of v_mfma_f64_16x16x4, which IS its B-operand layout; only left operands live in LDS), re-dimensioned by RT:
//   RT waves per workgroup, strips of 8 RT registers, A-forms of (16 RT)^2 doubles, as many workgroups per CU as give
//   three to four waves per SIMD below four row tiles.
#pragma once
#include "vsm_internal.h"
#include "vsm_lds.h"
#include "vsm_inverse.h"
#include "vsm_elemental.h"

namespace vsm {
namespace {

using nlds_d = __attribute__((address_space(3))) double;
using nlds_i = __attribute__((address_space(3))) int;
__device__ __forceinline__ unsigned nlds_addr(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
typedef double nd2_t __attribute__((ext_vector_type(2)));

template <int RT>
struct ngeo {
  static constexpr int NP = 16 * RT;                  // padded size of a sub-problem
  static constexpr int NT = 64 * RT;                  // threads per workgroup (RT waves)
  static constexpr int AF = NP * NP;                  // doubles per A-form / per pre-pass image / per native matrix
  static constexpr int UNITS = 2 * RT;                // 16-byte units per lane and strip
  static constexpr int PRE_STRIDE = 2 * AF + 3 * NP;  // pre-pass record: [r-+*] image, [t++] image, j0+, j0-, aux (aux[0] = expk)
  static constexpr int COMP_STRIDE = 4 * AF + 2 * NP; // native composite of one point: R-+, R+-, T++, T--, J0+, J0-
  // minimum waves per SIMD the register budget is set for (launch bounds): 128 / 128 / 168 / 256 registers
  static constexpr int WPS = RT <= 2 ? 4 : (RT == 3 ? 3 : 2);
};
// order of the matrices inside a native composite record
enum { NC_RMP = 0, NC_RPM = 1, NC_TPP = 2, NC_TMM = 3 };

// A-form: 64-double blocks per (k-step ks = k >> 2, row tile t = row >> 4), block index ks RT + t; inside a block element
// (m = row & 15, k' = k & 3) sits at word  k' << 4 | (m ^ (k' | (ks & 3) << 2)):  an A-fragment read covers a block linearly
// (conflict-free ds_read_b64), the 16-lane groups of a strip store hit 16 distinct 8-byte slots of a bank row (the layout of
// vsm_strip128_dev.h, checked exhaustively there).
template <int RT>
__host__ __device__ __forceinline__ int naf_idx(int row, int k) {
  const int ks = k >> 2, kk = k & 3;
  return (ks * RT + (row >> 4)) * 64 + ((kk << 4) | ((row & 15) ^ (kk | ((ks & 3) << 2))));
}
// element (i, j) of a native matrix (wave = j >> 4 owns the column; 16-byte units of two rows r, r + 1 per lane, unit-major)
template <int RT>
__host__ __device__ __forceinline__ int nnat_idx(int i, int j) {
  const int w = j >> 4, l15 = j & 15, ta = i >> 4, mm = i & 15, kq = mm & 3, r = mm >> 2;
  const int lane = (kq << 4) | l15, u = 2 * ta + (r >> 1);
  return ((w * 2 * RT + u) * 64 + lane) * 2 + (r & 1);
}

template <int RT>
struct nstrip {
  d4_t v[RT];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < RT; ++a) v[a] = acc_zero<double>();
  }
};

// Per-lane byte addresses: fragment bases by (ks & 3), strip-element bases by r; everything else is an instruction offset
// (block index and the distance dA of the A-form from the bound base `P`).
template <int RT>
struct npos {
  int lane, wave, l15, kq, col;
  unsigned fb[4];
  unsigned sb[4];
  __device__ __forceinline__ npos(const void* base) {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    const unsigned L = nlds_addr(base);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = L + 8u * (unsigned)((kq << 4) | (l15 ^ (kq | (j << 2))));
    const int q = l15 >> 2, kk = l15 & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      sb[r] = L + 8u * (unsigned)(((4 * wave + q) * RT * 64) + ((kk << 4) | (kq ^ kk) | ((r ^ q) << 2)));
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + kq + 4 * r; }
  __device__ __forceinline__ const nlds_d* aptr(unsigned dA, int t, int ks) const {
    return reinterpret_cast<const nlds_d*>((unsigned long long)(fb[ks & 3] + dA)) + 64 * (ks * RT + t);
  }
  __device__ __forceinline__ nlds_d* sptr(unsigned dA, int ta, int r) const {
    return reinterpret_cast<nlds_d*>((unsigned long long)(sb[r] + dA)) + 64 * ta;
  }
  // hide the loop invariance of the bases from LICM (hoisting every derived address costs registers)
  __device__ __forceinline__ void opaque() {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(fb[j]));
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(sb[r]));
  }
};

#define VSM_NKSTEP_FENCE() __builtin_amdgcn_sched_barrier(0)

// value of the neighbour lane (lane ^ 1): DPP quad permutation
__device__ __forceinline__ double ndpp_swap1(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ---- products: acc += [A] B, A-form at byte distance dA, B a strip in registers; pipelined by one k-step -------------------
// (Z / Z1 / Z2: the accumulator starts from zero -- the first k-step takes the constant 0 as its addend: no register is cleared)
template <int RT, int KS, bool Z = false>
__device__ __forceinline__ void nmm(nstrip<RT>& acc, unsigned dA, const nstrip<RT>& B, npos<RT>& p) {
  p.opaque();
  double a[2][RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < RT; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc.v[t] = mfma<double>::mma(a[ks & 1][t], b, (Z && ks == 0) ? acc_zero<double>() : acc.v[t]);
    VSM_NKSTEP_FENCE();
  }
}
// out = C0 + [A] B
template <int RT, int KS>
__device__ __forceinline__ void nmm_c(nstrip<RT>& out, const nstrip<RT>& C0, unsigned dA, const nstrip<RT>& B, npos<RT>& p) {
  p.opaque();
  double a[2][RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < RT; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) out.v[t] = mfma<double>::mma(a[ks & 1][t], b, ks == 0 ? C0.v[t] : out.v[t]);
    VSM_NKSTEP_FENCE();
  }
}
// acc1 += [A] B1 ; acc2 += [A] B2  (shared fragments)
template <int RT, int KS, bool Z1 = false, bool Z2 = false>
__device__ __forceinline__ void nmm2(nstrip<RT>& acc1, nstrip<RT>& acc2, unsigned dA, const nstrip<RT>& B1,
                                     const nstrip<RT>& B2, npos<RT>& p) {
  p.opaque();
  double a[2][RT];
#pragma unroll
  for (int t = 0; t < RT; ++t) a[0][t] = *p.aptr(dA, t, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + 1 < KS) {
#pragma unroll
      for (int t = 0; t < RT; ++t) a[(ks + 1) & 1][t] = *p.aptr(dA, t, ks + 1);
    }
    const double b1 = B1.v[ks >> 2][ks & 3], b2 = B2.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
      acc1.v[t] = mfma<double>::mma(a[ks & 1][t], b1, (Z1 && ks == 0) ? acc_zero<double>() : acc1.v[t]);
      acc2.v[t] = mfma<double>::mma(a[ks & 1][t], b2, (Z2 && ks == 0) ? acc_zero<double>() : acc2.v[t]);
    }
    VSM_NKSTEP_FENCE();
  }
}

// ---- strip <-> A-form ---------------------------------------------------------------------------------------------------------
template <int RT>
__device__ __forceinline__ void nstore(unsigned dA, const nstrip<RT>& s, const npos<RT>& p) {
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) *p.sptr(dA, ta, r) = s.v[ta][r];
}
template <int RT>
__device__ __forceinline__ void nload(nstrip<RT>& s, unsigned dA, const npos<RT>& p) {
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = *p.sptr(dA, ta, r);
}

// ---- strip <-> native matrix in global memory: 2 RT coalesced 16-byte accesses per lane, no permutation ------------------------
template <int RT>
__device__ __forceinline__ void nld_native(nstrip<RT>& s, const double* __restrict__ mat, const npos<RT>& p) {
  const nd2_t* g = reinterpret_cast<const nd2_t*>(mat) + (p.wave * 2 * RT) * 64 + p.lane;
#pragma unroll
  for (int u = 0; u < 2 * RT; ++u) {
    const nd2_t t = g[u * 64];
    s.v[u >> 1][2 * (u & 1)] = t.x;
    s.v[u >> 1][2 * (u & 1) + 1] = t.y;
  }
}
template <int RT>
__device__ __forceinline__ void nst_native(double* __restrict__ mat, const nstrip<RT>& s, const npos<RT>& p) {
  nd2_t* g = reinterpret_cast<nd2_t*>(mat) + (p.wave * 2 * RT) * 64 + p.lane;
#pragma unroll
  for (int u = 0; u < 2 * RT; ++u) {
    nd2_t t;
    t.x = s.v[u >> 1][2 * (u & 1)];
    t.y = s.v[u >> 1][2 * (u & 1) + 1];
    g[u * 64] = t;
  }
}

// ---- image (pre-pass record in global memory) -> A-form by LDS DMA: 2 RT instructions of 1 KB per wave --------------------------
template <int RT>
__device__ __forceinline__ void ncopy_image(double* L, const double* __restrict__ g, const npos<RT>& p) {
#pragma unroll
  for (int i = 0; i < 2 * RT; ++i) {
    const int blk = (i * RT + p.wave) * 128;   // doubles
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + blk + 2 * p.lane),
                                     (__attribute__((address_space(3))) void*)(L + blk), 16, 0, 0);
  }
}

// ---- D X D with the parities of the lane's rows and of its column as bits (doubling.jl:178-201) ---------------------------------
template <int RT>
struct ndpar {
  unsigned rows;   // bit 4 ta + r set = sign flip of that element
  __device__ __forceinline__ ndpar(const double* usg, const npos<RT>& p) {
    rows = 0;
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) rows |= (usg[p.row(ta, r)] < 0.0 ? 1u : 0u) << (4 * ta + r);
    if (usg[p.col] < 0.0) rows = ~rows;
  }
};
template <int RT>
__device__ __forceinline__ void ndsym(nstrip<RT>& d, const nstrip<RT>& x, const ndpar<RT>& dp) {
  unsigned rows = dp.rows;   // (opaque: the sign masks are two VALU operations each -- not 4 RT registers kept from call to call)
  asm volatile("" : "+v"(rows));
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = 4 * ta + r;
      const unsigned sbit = (rows << (31 - n)) & 0x80000000u;
      d.v[ta][r] = __hiloint2double(__double2hiint(x.v[ta][r]) ^ (int)sbit, __double2loint(x.v[ta][r]));
    }
}

// ---- norm bound / series order / inverses -------------------------------------------------------------------------------------
// Frobenius-norm bound of the n x n block whose strips the waves hold (rows >= n are zero by construction; the rider and
// padding columns are excluded).  ONE barrier inside: on return every wave has finished whatever it did before the call.
template <int RT, typename SM>
__device__ __forceinline__ double nnorm(const nstrip<RT>& e, int n, SM& sm, int& slot, const npos<RT>& p) {
  double ss = 0;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) ss = fma(e.v[ta][r], e.v[ta][r], ss);
  ss = (p.col < n) ? ss : 0.0;
  const float ws = wave_sum(to_float_up(ss));
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < RT; ++w) tot += sm.red[slot][w];
  slot ^= 1;
  return (double)(sqrtf(tot) * 1.001f);
}
// smallest order K of {1,2,3,4,7,8,15,16,31} with nrm^(K+1) / (1 - nrm) <= eps / 4; 0: no series (pivoted inverse)
__device__ __forceinline__ int nseries_order(double nrm) {
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
__device__ __forceinline__ void nadd_identity(nstrip<RT>& G, int n, const npos<RT>& p) {
  // a lane owns at most one diagonal element, in row tile ta = wave: r = l15 >> 2, kq = l15 & 3
  const bool dl = p.kq == (p.l15 & 3) && p.col < n;
  const int dr = p.l15 >> 2;
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
    if (ta == p.wave) {
#pragma unroll
      for (int r = 0; r < 4; ++r) G.v[ta][r] += (dl && dr == r) ? 1.0 : 0.0;
    }
}

// In-place Gauss-Jordan with partial pivoting (the pivot rule of getrf: the contract of the reference's LU,
// cpu_batched.jl:32-47) of the n x n matrix M (plain column-major, pitch NP) in LDS; lane = row, wave w = columns w, w + RT, ...
// Per pivot step every wave reads column k and finds the pivot redundantly (DPP maximum + ballot), a barrier, every wave
// updates its columns with the row interchange folded in, a barrier.  src[x] = column of M that is column x of the inverse.
// status (vsm_device_status): [0] |= VSM_DEVSTAT_SINGULAR on an exactly zero pivot, [1] += 1 per pivoted inverse.
template <int RT>
__device__ __forceinline__ void ngj_lds(int n, nlds_d* M, nlds_i* piv, nlds_i* src, int* status, const npos<RT>& p) {
  constexpr int NP = 16 * RT;
  constexpr bool TWO = NP > 64;     // (five and six row tiles: a lane carries rows lane and lane + 64)
  const int i0 = p.lane, i1 = p.lane + 64;
  const bool ok0 = i0 < n, ok1 = TWO && i1 < n;
  bool singular = false;
  for (int k = 0; k < n; ++k) {
    const nlds_d* ck = M + k * NP;
    double f0 = ok0 ? ck[i0] : 0.0, f1 = ok1 ? ck[i1] : 0.0;
    const double v0 = (ok0 && i0 >= k) ? fabs(f0) : -1.0, v1 = (ok1 && i1 >= k) ? fabs(f1) : -1.0;
    const double best = wave_max(v0 > v1 ? v0 : v1);
    const unsigned long long m0 = __ballot(v0 >= 0.0 && v0 == best);
    const unsigned long long m1 = __ballot(v1 >= 0.0 && v1 == best);
    const int pr = m0 ? (__ffsll((long long)m0) - 1) : (m1 ? 64 + __ffsll((long long)m1) - 1 : k);
    const double ckk = ck[k], pv = ck[pr];
    const double d = 1.0 / pv;
    singular |= pv == 0.0;
    f0 = (i0 == pr) ? ckk : f0;
    f1 = (i1 == pr) ? ckk : f1;
    if (threadIdx.x == 0) piv[k] = pr;
    __syncthreads();
    for (int j = p.wave; j < n; j += RT) {
      nlds_d* cj = M + j * NP;
      const bool isk = j == k;
      const double a = cj[pr], b = cj[k];
      const double u = isk ? d : a * d;
      double x0 = ok0 ? cj[i0] : 0.0;
      x0 = isk ? 0.0 : ((i0 == pr) ? b : x0);
      x0 = (i0 == k) ? u : fma(-f0, u, x0);
      if (ok0) cj[i0] = x0;
      if constexpr (TWO) {
        double x1 = ok1 ? cj[i1] : 0.0;
        x1 = isk ? 0.0 : ((i1 == pr) ? b : x1);
        x1 = (i1 == k) ? u : fma(-f1, u, x1);
        if (ok1) cj[i1] = x1;
      }
    }
    __syncthreads();
  }
  if (p.wave == 0) {
    if constexpr (TWO) {   // the column permutation that undoes the row interchanges, by one thread (a rare path)
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
      }
    } else {
      const int lane = p.lane;
      int s0 = lane;
      const int p0 = (lane < n) ? piv[lane] : lane;
      for (int k = n - 1; k >= 0; --k) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
        const int q = __builtin_amdgcn_readlane(p0, ku);
        if (q != ku) {
          const int sk = __builtin_amdgcn_readlane(s0, ku);
          const int sq = __builtin_amdgcn_readlane(s0, q);
          if (lane == ku) s0 = sq;
          if (lane == q) s0 = sk;
        }
      }
      src[lane] = s0;
    }
    if (p.lane == 0) {
      if (singular) atomicOr(&status[0], (int)VSM_DEVSTAT_SINGULAR);
      atomicAdd(&status[1], 1);
    }
  }
  __syncthreads();
}

struct ninv_ctx {
  unsigned dW;     // byte distance of the scratch A-form (= P: 0)
  double* W;       // the same A-form as a pointer (plain column-major use by the pivoted path)
  int* gjs;        // LDS: piv[max(NP, 64)], src[max(NP, 64)]
  int* status;
};

// The orders the Horner path does not take, out of line (K = 0: pivoted Gauss-Jordan; K = 15, 16, 31: G <- (I + E^(2^l)) G
// level by level): rare, and their live strips stay out of the hot path's register allocation.  Every wave is past the norm
// reduction's barrier (nobody reads W any more); on return other waves may still be reading W.
template <int RT, int KS>
__device__ __forceinline__ void ninvert_slow_body(int K, nstrip<RT>& E, nstrip<RT>& G, int n, const ninv_ctx& cx, npos<RT>& p);
// REGS: copy the arguments (the caller's stack objects) into registers once -- working through the references puts a flat load
// in front of every k-step (the standalone interaction's long orders: 0.21 -> 0.37 of the peak).  The layer kernels keep the
// by-reference form: a callee with ~250 VGPRs makes their register allocator spill around the call site in the hot path
// (measured: C2 -1.8 %), and their atmospheres hardly ever take this path.
template <int RT, int KS, bool REGS>
__device__ __attribute__((noinline)) void ninvert_slow(int K, nstrip<RT>& E_in, nstrip<RT>& G_out, int n, const ninv_ctx& cx_in,
                                                       npos<RT>& p_in) {
  if constexpr (REGS) {
    nstrip<RT> E = E_in, G;
    npos<RT> p = p_in;
    const ninv_ctx cx = cx_in;
    ninvert_slow_body<RT, KS>(K, E, G, n, cx, p);
    G_out = G;
  } else {
    ninvert_slow_body<RT, KS>(K, E_in, G_out, n, cx_in, p_in);
  }
}
template <int RT, int KS>
__device__ __forceinline__ void ninvert_slow_body(int K, nstrip<RT>& E, nstrip<RT>& G, int n, const ninv_ctx& cx, npos<RT>& p) {
  constexpr int NP = 16 * RT;
  if (K == 0) {
    if (p.col < n) {     // M = I - E, plain column-major (only the n x n block is read back)
      double* mc = cx.W + p.col * NP + p.kq;
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) mc[16 * ta + 4 * r] = (p.row(ta, r) == p.col ? 1.0 : 0.0) - E.v[ta][r];
    }
    __syncthreads();
    nlds_d* M = reinterpret_cast<nlds_d*>((unsigned long long)nlds_addr(cx.W));
    nlds_i* piv = reinterpret_cast<nlds_i*>((unsigned long long)nlds_addr(cx.gjs));
    constexpr int PV = NP > 64 ? NP : 64;
    ngj_lds<RT>(n, M, piv, piv + PV, cx.status, p);
    const bool cok = p.col < n;
    const nlds_d* gc = M + (cok ? piv[PV + p.col] : 0) * NP + p.kq;
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = gc[16 * ta + 4 * r];
        G.v[ta][r] = (cok && p.row(ta, r) < n) ? v : 0.0;
      }
    return;
  }
  // (columns >= n of E may carry riders: cleared here, this path is not the riders' -- see ninvert)
#pragma unroll
  for (int ta = 0; ta < RT; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) E.v[ta][r] = (p.col < n) ? E.v[ta][r] : 0.0;
  G = E;
  nadd_identity(G, n, p);
  nstore(cx.dW, E, p);
  __syncthreads();
  int cur = 1;   // W = E^cur (A-form), E = its strip, G = strip of sum_{k < 2 cur} E^k
  for (int lvl = 0; lvl < 5; ++lvl) {
    nstrip<RT> W2;
    W2.zero();
    nmm<RT, KS>(W2, cx.dW, E, p);   // E^(2 cur)
    cur *= 2;
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    __syncthreads();   // everybody finished reading W
    nstore(cx.dW, W2, p);
    __syncthreads();
    {
      nstrip<RT> T;
      T.zero();
      nmm<RT, KS>(T, cx.dW, G, p);   // E^cur G  (powers of E commute)
#pragma unroll
      for (int ta = 0; ta < RT; ++ta) G.v[ta] += T.v[ta];
    }
    if (K == 2 * cur - 1) break;
    E = W2;
  }
}

// G = (I - E)^-1 for a given order K after the norm reduction: orders 1..8 by Horner's rule with ONE A-form store
//   X_0 = E,  X_{j+1} = E + E X_j  ->  G = I + X_{K-1}
// (A = [E] for every product: no barrier between the K - 1 products; from the third term on E is re-read from its A-form).
// The rider columns (>= n) of E are NOT cleared on this path: every product maps a column of B to the same column of the
// result, so whatever rides there stays in columns that are never a contraction index.  On the slow path the riders of G are
// lost -- RIDERS tells whether the caller needs them (then it restores them itself: see the callers).
// On return other waves may still be reading W.
template <int RT, int KS, bool REGS = false>
__device__ __forceinline__ void ninvert(int K, nstrip<RT>& E, nstrip<RT>& G, int n, const ninv_ctx& cx, npos<RT>& p) {
  if (K < 1 || K > 8) {
    nstrip<RT> Es = E, Gs;
    npos<RT> ps = p;
    ninvert_slow<RT, KS, REGS>(K, Es, Gs, n, cx, ps);
    G = Gs;
    return;
  }
  if (K == 1) {
    G = E;
  } else {
    nstore(cx.dW, E, p);
    __syncthreads();
    nmm_c<RT, KS>(G, E, cx.dW, E, p);     // X1 = E + E E
    for (int j = 2; j < K; ++j) {          // X_j = E + E X_{j-1}
      nstrip<RT> X;
      nload(X, cx.dW, p);
      nmm<RT, KS>(X, cx.dW, G, p);
      G = X;
    }
  }
  nadd_identity(G, n, p);
}
// Order 7 without riders in G, in four products instead of Horner's six:  (I + E)(I + E^2)(I + E^4)  with
//   W2 = E E,  T = E W2,  W4 = E T  (A = [E]),  S = E + W2 + T,  G = I + S + W4 + W4 S  (A = [W4]: one more A-form store).
// The interaction's inverse: its G is only ever a B operand whose rider columns nobody reads.  Three live strips.
template <int RT, int KS>
__device__ __forceinline__ void ninvert7(nstrip<RT>& E, nstrip<RT>& G, int n, const ninv_ctx& cx, npos<RT>& p) {
  nstore(cx.dW, E, p);
  __syncthreads();
  nstrip<RT> X, Y;
  nmm_c<RT, KS>(X, E, cx.dW, E, p);          // X = E + E^2
  nload(Y, cx.dW, p);
  nmm<RT, KS>(Y, cx.dW, X, p);               // Y = E + E X = S
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) X.v[ta] = Y.v[ta] - X.v[ta];   // E^3
  nmm<RT, KS, true>(G, cx.dW, X, p);         // W4 = E E^3
  __syncthreads();                           // [E] no longer read
  nstore(cx.dW, G, p);
#pragma unroll
  for (int ta = 0; ta < RT; ++ta) G.v[ta] += Y.v[ta];
  __syncthreads();
  nmm<RT, KS>(G, cx.dW, Y, p);               // S + W4 + W4 S
  nadd_identity(G, n, p);
}
__device__ __forceinline__ int ninv_order(double nrm, int* status) {
  if (!(nrm < 1e300) && threadIdx.x == 0) atomicOr(&status[0], (int)VSM_DEVSTAT_NONFINITE);
  return nseries_order(nrm);
}

// Partial sums of the mat-vec source path (blocks without spare columns: 4 KS + 2 > 16 RT, i.e. n = 13..16, 29..32, 45..48, 61..64): the
// other blocks must not pay for them (three row tiles with rider columns sit exactly at four workgroups per CU).
template <int RT, bool ON>
struct nmv_slots {};
template <int RT>
struct nmv_slots<RT, true> {
  double mv[2][16 * RT];
};
// LDS block of a workgroup
template <int RT, bool MV = false>
struct nsmem : nmv_slots<RT, MV> {
  double P[ngeo<RT>::AF];
  double Q[ngeo<RT>::AF];
  double vec[8][ngeo<RT>::NP];
  double usg[ngeo<RT>::NP];   // -1.0 on the U/V rows of the sub-problem, +1.0 elsewhere
  float red[2][8];
  int flags[4];
  int gjs[2 * (ngeo<RT>::NP > 64 ? ngeo<RT>::NP : 64)];
};

// Source vectors without spare columns: y = sc [A] x as a VALU mat-vec over the A-form.  Wave w takes the rows of row tile w
// through the fragment pattern of the products (lane (m, kq): row 16 w + m, k = 4 ks + kq -- conflict-free, four per-lane bases +
// instruction offsets: nothing for LICM to hoist into registers, which is what the column-wise form of round 5 paid its 160
// spilled registers for), reduces over the four lane groups and writes its sixteen finished rows: no partial sums.
template <int RT, int KS>
__device__ __forceinline__ void nmv_rows(unsigned dA, const double* x, double sc, double* out, npos<RT>& p) {
  unsigned ab[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ab[j] = p.fb[j] + dA + 512u * (unsigned)p.wave;
    asm volatile("" : "+v"(ab[j]));
  }
  unsigned xb = nlds_addr(x) + 8u * (unsigned)p.kq;
  asm volatile("" : "+v"(xb));
  double acc = 0.0;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const double a = *(reinterpret_cast<const nlds_d*>((unsigned long long)ab[ks & 3]) + 64 * ks * RT);
    const double xv = *(reinterpret_cast<const nlds_d*>((unsigned long long)xb) + 4 * ks);
    acc = fma(a, xv, acc);
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  if (p.kq == 0) out[16 * p.wave + p.l15] = acc * sc;
}

}  // namespace
}  // namespace vsm
