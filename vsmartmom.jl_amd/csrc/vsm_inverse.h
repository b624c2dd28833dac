// Register-resident in-place Gauss-Jordan inversion with partial pivoting for one
// N x N matrix per 256- or 512-thread workgroup (N <= NPAD <= 128).
//
// Replaces the reference's batched getrf+getri pair (ext/gpu_batched_cuda.jl:97-182;
// CPU: src/CoreRT/tools/cpu_batched.jl:32-47 `A \ I`).  Same pivot rule as getrf
// (largest |a_ik|, first occurrence).  The matrix lives in VGPRs for the whole
// elimination: thread (tr, tc) owns rows i = tr + TR*rb and columns j = tc*CB + cb.
// Per pivot step the pivot column and the pivot row are published through LDS
// (double-buffered by step parity -> two barriers per step).
#pragma once
#include "vsm_common.h"

namespace vsm {

// ---- wave maximum (all 64 lanes active): quad_perm xor 1, xor 2, row_half_mirror, row_mirror, then the four row maxima ----
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_value(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ double lane_value(double x, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
template <typename T>
__device__ __forceinline__ T wave_max(T x) {
  x = fmax(x, dpp_mov<0xB1>(x));    // quad_perm [1,0,3,2]
  x = fmax(x, dpp_mov<0x4E>(x));    // quad_perm [2,3,0,1]
  x = fmax(x, dpp_mov<0x141>(x));   // row_half_mirror
  x = fmax(x, dpp_mov<0x140>(x));   // row_mirror: every lane of a 16-lane row holds the row maximum
  return fmax(fmax(lane_value(x, 0), lane_value(x, 16)), fmax(lane_value(x, 32), lane_value(x, 48)));
}

template <int NPAD, int NT = 256>
struct gj_cfg {
  static_assert(NPAD == 32 || NPAD == 64 || NPAD == 96 || NPAD == 128, "NPAD must be 32/64/96/128");
  static_assert(NT % 64 == 0 && NT >= 128 && NT <= 512, "workgroup of 2..8 waves");
  static constexpr int TR = (NPAD % 64 == 0) ? 64 : 32;
  static constexpr int TC = NT / TR;
  static constexpr int RB = NPAD / TR;
  static constexpr int CB = NPAD / TC;
  static_assert(CB * TC == NPAD && RB * TR == NPAD, "thread grid must tile the matrix");
};

template <typename T, int NPAD>
struct gj_scratch {
  T col[2][NPAD];
  T rowP[2][NPAD];
  T rowK[2][NPAD];
  int piv[NPAD];
  int dst[NPAD];
  int info;
};

// a[rb][cb] holds element (tr + TR*rb, tc*CB + cb) of the identity-padded matrix.
// On exit a holds the inverse with its columns permuted: the caller must store
// element a[rb][cb] at column sc.dst[tc*CB + cb].  sc.info = 0 or (k+1) of the first
// exactly-zero pivot.  All NT threads must call this (contains barriers).
template <typename T, int NPAD, int NT>
using gj_regs = T[gj_cfg<NPAD, NT>::RB][gj_cfg<NPAD, NT>::CB];

// SC: gj_scratch<T, NPAD> or any type with the same members (col / rowP / rowK indexable as [parity][i], piv / dst as
// [i], info assignable) -- the FP32 strip kernels keep the scratch in the padding columns of an LDS matrix.
// SYNC: barrier over the NT threads that share the matrix (default: the whole workgroup); `tid`: index within them.
struct wg_sync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
template <typename T, int NPAD, int NT = 256, bool DPP_SEARCH = true, typename SC, typename SYNC = wg_sync>
__device__ __forceinline__ void gj_invert(gj_regs<T, NPAD, NT>& a, int N, SC& sc, int tid = threadIdx.x, SYNC sync = SYNC()) {
  using C = gj_cfg<NPAD, NT>;
  const int lane = tid & 63;
  const int tr = tid % C::TR;
  const int tc = tid / C::TR;
  if (tid == 0) sc.info = 0;

  for (int tck = 0; tck < C::TC; ++tck) {
#pragma unroll
    for (int cbk = 0; cbk < C::CB; ++cbk) {
      const int k = tck * C::CB + cbk;
      if (k < N) {  // uniform
        const int par = k & 1;
        // (a) publish column k
        if (tc == tck) {
#pragma unroll
          for (int rb = 0; rb < C::RB; ++rb) sc.col[par][tr + C::TR * rb] = a[rb][cbk];
        }
        sync();
        int bi;
        if constexpr (DPP_SEARCH) {
          // (b) pivot search, redundantly in every wave (no second barrier needed): the wave maximum of |a_ik| by DPP (no LDS
          // round trips), then the FIRST row that attains it from two ballots (row i = lane, lane + 64)
          const bool in0 = lane >= k && lane < N, in1 = NPAD > 64 && lane + 64 >= k && lane + 64 < N;
          T v0 = T(-1), v1 = T(-1);
          if (in0) v0 = fabs(sc.col[par][lane]);
          if (in1) v1 = fabs(sc.col[par][(lane + 64) % NPAD]);
          const T best = wave_max(v0 > v1 ? v0 : v1);
          const unsigned long long m0 = __ballot(in0 && v0 == best), m1 = __ballot(in1 && v1 == best);
          bi = m0 ? (__ffsll((long long)m0) - 1) : (m1 ? 64 + __ffsll((long long)m1) - 1 : k);
        } else {
          // the same search by butterfly shuffles (LDS round trips; kept for the strip layer kernels, whose register
          // allocation -- 1.8 % (C2) / 4 % (C4) of the layer time -- is tuned around this form)
          T best = T(-1);
          bi = k;
          for (int i = lane; i < NPAD; i += 64) {
            if (i >= k && i < N) {
              T v = fabs(sc.col[par][i]);
              if (v > best) {
                best = v;
                bi = i;
              }
            }
          }
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            T ov = __shfl_xor(best, off);
            int oi = __shfl_xor(bi, off);
            if (ov > best || (ov == best && oi < bi)) {
              best = ov;
              bi = oi;
            }
          }
        }
        const int p = __builtin_amdgcn_readfirstlane(bi);
        // (c) publish pivot row (old row p) and, if swapping, old row k
#pragma unroll
        for (int rb = 0; rb < C::RB; ++rb) {
          const int i = tr + C::TR * rb;
          if (i == p) {
#pragma unroll
            for (int cb = 0; cb < C::CB; ++cb) sc.rowP[par][tc * C::CB + cb] = a[rb][cb];
          }
          if (i == k && p != k) {
#pragma unroll
            for (int cb = 0; cb < C::CB; ++cb) sc.rowK[par][tc * C::CB + cb] = a[rb][cb];
          }
        }
        sync();
        // (d) eliminate
        const T pv = sc.rowP[par][k];
        const T d = T(1) / pv;
        const T colk = sc.col[par][k];
        if (tid == 0) {
          sc.piv[k] = p;
          if (pv == T(0) && sc.info == 0) sc.info = k + 1;
        }
        T u[C::CB];
#pragma unroll
        for (int cb = 0; cb < C::CB; ++cb) {
          const bool jk = (cb == cbk) && (tc == tck);
          const T rp = sc.rowP[par][tc * C::CB + cb];
          u[cb] = jk ? d : rp * d;
        }
#pragma unroll
        for (int rb = 0; rb < C::RB; ++rb) {
          const int i = tr + C::TR * rb;
          if (i == p && p != k) {  // row p now holds the old row k
#pragma unroll
            for (int cb = 0; cb < C::CB; ++cb) a[rb][cb] = sc.rowK[par][tc * C::CB + cb];
          }
          const T f = (i == p) ? colk : sc.col[par][i];
          const bool isk = (i == k);
#pragma unroll
          for (int cb = 0; cb < C::CB; ++cb) {
            const bool jk = (cb == cbk) && (tc == tck);
            const T base = jk ? T(0) : a[rb][cb];
            a[rb][cb] = isk ? u[cb] : base - f * u[cb];
          }
        }
      }
    }
  }
  sync();
  // Undo the row interchanges as a column permutation of the inverse:
  // for k = N-1..0 swap columns k and piv[k].  src[x] = source column of final column x.
  if (tid < 64) {
    int s0 = lane, s1 = lane + 64;
    int p0 = (lane < N) ? sc.piv[lane] : lane;
    int p1 = (lane + 64 < N) ? sc.piv[(lane + 64) % NPAD] : lane + 64;
    for (int k = N - 1; k >= 0; --k) {
      const int ku = __builtin_amdgcn_readfirstlane(k);
      const int p = (ku < 64) ? __builtin_amdgcn_readlane(p0, ku) : __builtin_amdgcn_readlane(p1, ku - 64);
      if (p != ku) {
        const int sk = (ku < 64) ? __builtin_amdgcn_readlane(s0, ku) : __builtin_amdgcn_readlane(s1, ku - 64);
        const int sp = (p < 64) ? __builtin_amdgcn_readlane(s0, p) : __builtin_amdgcn_readlane(s1, p - 64);
        if (ku < 64) {
          if (lane == ku) s0 = sp;
        } else {
          if (lane == ku - 64) s1 = sp;
        }
        if (p < 64) {
          if (lane == p) s0 = sk;
        } else {
          if (lane == p - 64) s1 = sk;
        }
      }
    }
    if (lane < NPAD) sc.dst[s0] = lane;
    if (lane + 64 < NPAD) sc.dst[s1] = lane + 64;
  }
  sync();
}

}  // namespace vsm
