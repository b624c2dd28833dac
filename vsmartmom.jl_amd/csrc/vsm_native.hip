// Native-layout layer kernels (FP64, sub-problems of n <= 60 rows): rt_kernel!'s scattering branch (rt_kernel.jl:175-250:
// elemental! -> doubling! -> interaction!(::ScatteringInterface_11) | TOA copy) for a whole run with
//   (1) the CompositeLayer kept in the kernels' own strip layout between the layer steps (vsm_run_*: the composite is converted
//       to the reference's [N,N,S] arrays ONCE, when the surface step needs it, rt_run.jl:455-517) -- no transposer tiles, no
//       A-form staging arithmetic, every composite matrix moves as 16-byte coalesced records;
//   (2) Stokes blocks that do not couple run as independent sub-problems: for the Fourier moment m = 0 the phase matrices have
//       exactly zero (I,Q) x (U,V) blocks (compute_Z_matrices.jl:26-110: the functions T_l^m carry a factor m), so r, t and every
//       product of the doubling / adding recurrences are block diagonal after a row permutation -- a run with n_stokes = 3 and
//       N = 60 walks a 40 x 40 and a 20 x 20 problem instead of one 60 x 60 one (products with exact zeros are not formed;
//       results identical);
//   (3) RT = ceil((n + 2) / 16) = 1..4 row tiles as a template parameter: RT waves per workgroup, as many workgroups per CU as
//       give three to four waves per SIMD below four row tiles -- the same code serves every n <= 60.
// Doubling step: rt_helpers.jl:102-166 in the [E | W] = r [r | t] form of vsm_strip.hip; interaction: interaction.jl:207-266 with
// ONE inverse (push-through identities, vsm_strip.hip) and the product phases ordered so that at most six strips are live.
#include <stdlib.h>

#include <vector>

#include "vsm_native_dev.h"
#include "vsm_native_run.h"

#ifndef VSM_NATIVE_PARK_RT
#define VSM_NATIVE_PARK_RT 3
#endif

namespace vsm {

struct nlayer_comps {
  double* c[NSUB_MAX];
};

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// elemental (from the pre-pass images) + ndoubl x doubling step + apply_D
// ---------------------------------------------------------------------------------------------------------------------------
// On return (all waves past a barrier): r_s = strip of the final r-+ (row signs of apply_D applied), t_s = strip of t++ (rider
// columns cleared), sm.vec[0] = j0+, sm.vec[1] = j0- (final sign), sm.usg filled.
template <int RT, int KS>
__device__ __forceinline__ void ned_body(nsmem<RT, (4 * KS + 2 > 16 * RT)>& sm, npos<RT>& p, int n, int gsz, unsigned uvmask, int ndoubl,
                                         const double* __restrict__ img, nstrip<RT>& r_s, nstrip<RT>& t_s, int* status) {
  using G = ngeo<RT>;
  constexpr unsigned dP = 0, dQ = G::AF * 8;
  double* jp = sm.vec[0];
  double* jm = sm.vec[1];
  double* rsg = sm.vec[3];   // row sign of apply_D
  const int tid = threadIdx.x;
  constexpr int c1 = 4 * KS, c2 = 4 * KS + 1;   // spare columns (>= n, never read as k) carry j0+ / j1- through a step
  constexpr bool RID = c2 < G::NP;              // KS = 4 RT (n = 13..16, 29..32, 45..48, 61..64): no spare column, the source vectors by VALU mat-vecs
  static_assert(RID || KS == 4 * RT, "no spare columns for the source vectors");
  const bool own_wave = RID && p.wave == (c1 >> 4);
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2), laneAB = laneA || laneB;
  ncopy_image<RT>(sm.P, img, p);
  ncopy_image<RT>(sm.Q, img + G::AF, p);
  if (tid < G::NP) {
    jp[tid] = img[2 * G::AF + tid];
    jm[tid] = img[2 * G::AF + G::NP + tid];
    const bool uv = (uvmask >> (tid % gsz)) & 1u;
    sm.usg[tid] = uv ? -1.0 : 1.0;
    rsg[tid] = (ndoubl >= 1 && uv) ? -1.0 : 1.0;
  }
  const double expk0 = img[2 * G::AF + 2 * G::NP];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA writes have landed in LDS
  __syncthreads();
  nload(r_s, dP, p);
  nload(t_s, dQ, p);
  // ---- doubling (rt_helpers.jl:102-166) -----------------------------------------------------------------------------------
  //   [E | W]   = r [r | t]                 A = P = [r]
  //   G         = (I - E)^-1                Horner series on [E] in P
  //   tt        = t G                       A = Q = [t]
  //   [r' | t'] = [r | 0] + tt [W | t]      A = P = [tt]
  // Sources (rt_helpers.jl:128-134: j0- += tt (j1- + r j0+), j0+ = j1+ + tt (j0+ + r j1-)) ride in the spare columns:
  //   t_s[c1] = j0+, t_s[c2] = j1- = j0- expk  ->  W[c1] = r j0+, W[c2] = r j1-
  //   r_s[c1] = j0-, r_s[c2] = j0+             ->  W[c1] += r_s[c1] expk, W[c2] += r_s[c2], r_s[c2] *= expk: lane-local;
  //                                                r'[c1] = j0- + tt (j1- + r j0+), r'[c2] = j1+ + tt (j0+ + r j1-)
  double expk = expk0;
  int slot = 0;
  const ninv_ctx cx{dP, sm.P, sm.gjs, status};
  for (int it = 0; it < ndoubl; ++it) {
    nstrip<RT> W, tt;
    const double fW = laneA ? expk : (laneB ? 1.0 : 0.0), fR = laneB ? expk : 1.0;
    if constexpr (!RID) {   // r j0+ , r j1-   (P = [r])
      nmv_rows<RT, KS>(dP, jp, 1.0, sm.mv[0], p);
      nmv_rows<RT, KS>(dP, jm, expk, sm.mv[1], p);
    }
    {
      nstrip<RT> Gs;
      {
        nstrip<RT> E;
        nmm2<RT, KS, true, true>(E, W, dP, r_s, t_s, p);
        if (own_wave) {
#pragma unroll
          for (int ta = 0; ta < RT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              W.v[ta][r] = fma(r_s.v[ta][r], fW, W.v[ta][r]);
              r_s.v[ta][r] *= fR;
            }
        }
        const double nrm = nnorm(E, n, sm, slot, p);   // (its barrier: [r] is free)
        if constexpr (!RID) {   // u1 = j1- + r j0+ , u2 = j0+ + r j1-   (every wave is past the norm reduction's barrier)
          if (tid < G::NP) {
            sm.vec[4][tid] = jm[tid] * expk + sm.mv[0][tid];
            sm.vec[5][tid] = jp[tid] + sm.mv[1][tid];
          }
        }
        ninvert<RT, KS>(ninv_order(nrm, status), E, Gs, n, cx, p);
      }
      nmm<RT, KS, true>(tt, dQ, Gs, p);    // tt = t G
    }
    nload(t_s, dQ, p);                     // t's strip (with its riders) is not kept in registers across the inverse
    __syncthreads();                       // P ([E]) and Q ([t]) no longer read
    nstore(dP, tt, p);
    __syncthreads();
    if constexpr (!RID) {   // tt u1 , tt u2   (the partial sums of r j were consumed two barriers ago)
      nmv_rows<RT, KS>(dP, sm.vec[4], 1.0, sm.mv[0], p);
      nmv_rows<RT, KS>(dP, sm.vec[5], 1.0, sm.mv[1], p);
    }
    {
      nstrip<RT> tn;
      nmm2<RT, KS, false, true>(r_s, tn, dP, W, t_s, p);   // r' = r + tt W (riders: the new j0-, j0+) ; t' = tt t
      t_s = tn;
    }
    const double expk_step = expk;
    expk = expk * expk;
    if (own_wave) {
      const double ft = laneB ? expk : 1.0;   // t_s[c1] = j0+', t_s[c2] = j1-' = j0-' expk'
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = ndpp_swap1(r_s.v[ta][r]) * ft;
          t_s.v[ta][r] = laneAB ? u : t_s.v[ta][r];
        }
    }
    if (it + 1 < ndoubl || !RID) __syncthreads();   // everybody finished reading P ([tt])
    if constexpr (!RID) {   // j0- += tt u1 ; j0+ = j0+ expk + tt u2   (visible after the barrier below / after the loop)
      if (tid < G::NP) {
        jm[tid] += sm.mv[0][tid];
        jp[tid] = jp[tid] * expk_step + sm.mv[1][tid];
      }
    }
    if (it + 1 < ndoubl) {
      nstore(dP, r_s, p);
      nstore(dQ, t_s, p);
      __syncthreads();
    }
  }
  if (ndoubl > 0 && own_wave) {   // the riders go back to the LDS vectors; the strips leave the loop with clean padding
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        double* dpv = laneA ? jp : sm.vec[7];   // (the other lanes write to a dummy vector)
        double* dmv = laneA ? jm : sm.vec[7];
        dpv[row] = t_s.v[ta][r];          // lane A: j0+
        dmv[row] = r_s.v[ta][r];          // lane A: j0-
        t_s.v[ta][r] = laneAB ? 0.0 : t_s.v[ta][r];
        r_s.v[ta][r] = laneAB ? 0.0 : r_s.v[ta][r];
      }
  }
  __syncthreads();
  // ---- apply_D (doubling.jl:178-252): r-+ = D r*, j0- = D j0-* ------------------------------------------------------------
  if (ndoubl >= 1) {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] *= rsg[p.row(ta, r)];
    if (tid < G::NP) jm[tid] *= rsg[tid];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266) on the run's native records: the layer kernel's copy.  The
// steps are those of nia_body below (which documents them) in the plain order -- every composite strip is a coalesced 16-byte
// record here and nothing of what nia_body does about exposed loads pays (each measured: DESIGN.md 4.0b).  Kept as a function of
// its own: routed through nia_body's I/O policy the same statements came out with 14 spilled VGPRs instead of 8.
// ---------------------------------------------------------------------------------------------------------------------------
template <int RT, int KS>
__device__ __forceinline__ void nia_body_native(nsmem<RT, (4 * KS + 2 > 16 * RT)>& sm, npos<RT>& p, int n, double* __restrict__ comp, nstrip<RT>& r_s,
                                         nstrip<RT>& t_s, int* status) {
  using G = ngeo<RT>;
  constexpr unsigned dP = 0, dQ = G::AF * 8;
  double* vjp = sm.vec[0];
  double* vjm = sm.vec[1];
  double* vJp = sm.vec[2];
  double* vJm = sm.vec[3];
  double* vs = sm.vec[4];
  double* vz = sm.vec[5];
  const int tid = threadIdx.x;
  double* R_mp = comp + NC_RMP * G::AF;
  double* R_pm = comp + NC_RPM * G::AF;
  double* T_pp = comp + NC_TPP * G::AF;
  double* T_mm = comp + NC_TMM * G::AF;
  double* J0_p = comp + 4 * G::AF;
  double* J0_m = J0_p + G::NP;
  constexpr int c1 = 4 * KS, c2 = 4 * KS + 1;
  constexpr bool RID = c2 < G::NP;
  // Three row tiles (168 registers, strips of 24) and five / six (256 registers, strips of 40 / 48): at most FIVE live strips.  Z = R+- t-- is not needed between its product and
  // the closing pairs: it waits in the R+- record of the composite (consumed into P at the entry, overwritten by the new R+- at
  // the end; a lane reads back exactly the addresses it wrote), and R-+ is requested after R+- / T++ have left.
  constexpr bool PARK = VSM_NATIVE_PARK_RT > 0 && (RT == VSM_NATIVE_PARK_RT || RT >= 5);
  const bool own_wave = RID && p.wave == (c1 >> 4);
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2);
  int slot = 0;
  const ndpar<RT> dp(sm.usg, p);
  const ninv_ctx cx{dP, sm.P, sm.gjs, status};
  // ---- stage: composite vectors, [R+-] -> P, [T--] -> Q ---------------------------------------------------------------------
  if (tid < G::NP) {
    vJp[tid] = J0_p[tid];
    vJm[tid] = J0_m[tid];
  }
  nstrip<RT> Z, Gs;
  {
    nstrip<RT> A1, A2;
    nld_native(A1, R_pm, p);
    nld_native(A2, T_mm, p);
    nstore(dP, A1, p);
    nstore(dQ, A2, p);
  }
  if (own_wave) {  // j0- rides in the spare column c2 of r-+:  E2[:, c2] = R+- j0-, S[:, c2] = T-- j0-
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] = laneB ? vjm[p.row(ta, r)] : r_s.v[ta][r];
  }
  ndsym(t_s, t_s, dp);   // t-- = D t++ D in place (undone below: D is an involution)
  __syncthreads();                                                                                       // (a)
  if constexpr (!RID) {   // R+- j0- , T-- j0-
    nmv_rows<RT, KS>(dP, vjm, 1.0, sm.mv[0], p);
    nmv_rows<RT, KS>(dQ, vjm, 1.0, sm.mv[1], p);
  }
  {
    nstrip<RT> E;
    nmm2<RT, KS, true, true>(E, Z, dP, r_s, t_s, p);
    if (own_wave) {
      double* zd = laneB ? vz : sm.vec[7];   // (the other lanes write to a dummy vector)
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          zd[row] = vJp[row] + E.v[ta][r];
        }
    }
    if constexpr (PARK) nst_native(R_pm, Z, p);   // (its old contents are in P; the new R+- overwrites it at the end)
    const double nrm = nnorm(E, n, sm, slot, p);   // (b): every wave is done reading [R+-]
    if constexpr (!RID) {   // z = J0+ + R+- j0- ; vs = T-- j0-
      if (tid < G::NP) {
        vz[tid] = vJp[tid] + sm.mv[0][tid];
        vs[tid] = sm.mv[1][tid];
      }
    }
    ninvert<RT, KS>(ninv_order(nrm, status), E, Gs, n, cx, p);   // [E2] -> P, barrier (c), series
  }
  nstrip<RT> V;
  {
    nstrip<RT> S;
    nmm2<RT, KS, true, true>(S, V, dQ, r_s, t_s, p);   // (Q = [T--] has not been touched since (a))
    if (own_wave) {
      double* sd = laneB ? vs : sm.vec[7];
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) sd[p.row(ta, r)] = S.v[ta][r];
    }
    // r_s <- r+- (the accumulator of the R+- update; its rider column is never used), t_s <- t++
    ndsym(r_s, r_s, dp);
    ndsym(t_s, t_s, dp);
    __syncthreads();                      // (d): [E2] (series) and [T--] no longer read
    nstore(dQ, S, p);                     // [S]   -> Q
    nstore(dP, t_s, p);                   // [t++] -> P
  }
  __syncthreads();                        // (e)
  {
    nstrip<RT> X, Y;
    nmm<RT, KS, true>(X, dP, Gs, p);      // T21 = t++ G2
    nmm<RT, KS, true>(Y, dQ, Gs, p);      // Y = S G2 = T01 r-+
    __syncthreads();                      // (f): [t++], [S] no longer read
    nstore(dP, X, p);                     // [T21] -> P
    nstore(dQ, Y, p);                     // [Y]   -> Q
  }
  __builtin_amdgcn_sched_barrier(0);      // (the composite strips are requested once X and Y are dead, not above their stores)
  nstrip<RT> Tpp, Rmp;
  nld_native(Tpp, T_pp, p);
  if constexpr (PARK) nld_native(Z, R_pm, p); else nld_native(Rmp, R_mp, p);
  __syncthreads();                        // (g)
  if constexpr (!RID) {   // T21 z , Y z
    nmv_rows<RT, KS>(dP, vz, 1.0, sm.mv[0], p);
    nmv_rows<RT, KS>(dQ, vz, 1.0, sm.mv[1], p);
  }
  if (own_wave) {  // z rides in the spare column c1 of T++:  (T21 T++)[:, c1] = T21 z, (Y T++)[:, c1] = Y z
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Tpp.v[ta][r] = laneA ? vz[p.row(ta, r)] : Tpp.v[ta][r];
        if constexpr (!PARK) Rmp.v[ta][r] = laneA ? 0.0 : Rmp.v[ta][r];   // (the native record keeps the rider column of the previous layer step)
      }
  }
  {
    nstrip<RT> acc;
    nmm2<RT, KS, false, true>(r_s, acc, dP, Z, Tpp, p);   // R+- = r+- + T21 Z ; T++ = T21 T++
    nst_native(R_pm, r_s, p);
    nst_native(T_pp, acc, p);
    if (laneA) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          J0_p[row] = vjp[row] + acc.v[ta][r];
        }
    }
  }
  if constexpr (PARK) {
    __builtin_amdgcn_sched_barrier(0);      // (requested once r_s and acc have left)
    nld_native(Rmp, R_mp, p);
    if (own_wave) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rmp.v[ta][r] = laneA ? 0.0 : Rmp.v[ta][r];
    }
  }
  nmm2<RT, KS>(Rmp, V, dQ, Tpp, Z, p);       // R-+ = R-+ + Y T++ ; T-- = V + Y Z
  nst_native(R_mp, Rmp, p);
  nst_native(T_mm, V, p);
  if (laneA) {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        J0_m[row] = vJm[row] + vs[row] + Rmp.v[ta][r];
      }
  }
  if constexpr (!RID) {   // J0+ = j0+ + T21 z ; J0- = J0- + T-- j0- + Y z
    __syncthreads();
    if (tid < G::NP) {
      J0_p[tid] = vjp[tid] + sm.mv[0][tid];
      J0_m[tid] = vJm[tid] + vs[tid] + sm.mv[1][tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// I/O of the standalone interaction (k_ia_native: vsm_interaction_f64 for N <= 64 -- the surface interaction of every run, the
// callers of the operator API): the composite and the added layer in the reference's [N,N,S] arrays.
// ---------------------------------------------------------------------------------------------------------------------------
template <int RT>
struct nio_ref {
  int N;
  double* m[4];          // R-+, R+-, T++, T-- of this point (NC_* order)
  double* J[2];          // J0+, J0-
  const double* a[4];    // the added layer's r-+, r+-, t++, t-- of this point (r+- / t-- only read when it is not D-symmetric)
  double jpre[2];        // J0+[tid], J0-[tid], requested at the entry of the kernel
  bool al16;             // the matrices start on 16-byte boundaries (launcher: N even and every array 16-byte aligned)
  static constexpr bool REF = true;
  // Direct accesses in the accumulator layout.  A lane holds rows kq + 4 r of a row tile: 8-byte accesses would touch 32-byte
  // pieces of 16 lines per instruction (measured: ~ 350 cycles of issue each while the memory pipeline is busy).
  // v_permlane16_swap trades the odd 16-lane rows of register r = 2h with the even ones of r = 2h + 1, which leaves lane-row kq
  // with the ADJACENT rows (4 (kq & 1) + 2 (kq >> 1), + 1) + 8 h of its column -- one 16-byte access, 64 contiguous bytes per
  // column and instruction, half the instructions.  (The swap is its own inverse: loads apply it after, stores before.)
  static __device__ __forceinline__ void swap16(double& a, double& b) {
    const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    a = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
    b = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
  }
  // (pair types: 16-byte aligned where the launcher found every matrix on a 16-byte boundary and N even -- one dwordx4 access;
  //  8-byte aligned otherwise -- a column of an odd N, or a caller's array off a 16-byte boundary: still one request per pair)
  typedef double npair16_t __attribute__((ext_vector_type(2)));
  typedef double npair8_t __attribute__((ext_vector_type(2), aligned(8)));
  template <typename PT>
  __device__ __forceinline__ void ldg_even(nstrip<RT>& x, const double* __restrict__ gc, bool cok, int rb) const {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)   // (N even: a pair never straddles the last row -- unconditional loads, clamped addresses)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 16 * ta + 8 * h + rb;
        const PT v = *reinterpret_cast<const PT*>(gc + min(row, N - 2));
        const bool ok = cok && row < N;
        double a = ok ? v.x : 0.0, b = ok ? v.y : 0.0;
        swap16(a, b);
        x.v[ta][2 * h] = a;
        x.v[ta][2 * h + 1] = b;
      }
  }
  __device__ __forceinline__ void ldg(nstrip<RT>& x, const double* __restrict__ g, const npos<RT>& p) const {
    const bool cok = p.col < N;
    const int rb = 4 * (p.kq & 1) + 2 * (p.kq >> 1);
    const double* gc = g + (long long)N * min(p.col, N - 1);
    if (al16) {
      ldg_even<npair16_t>(x, gc, cok, rb);
      return;
    }
    if ((N & 1) == 0) {
      ldg_even<npair8_t>(x, gc, cok, rb);
      return;
    }
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 16 * ta + 8 * h + rb;
        double a = 0.0, b = 0.0;
        if (cok && row + 1 < N) {
          const npair8_t v = *reinterpret_cast<const npair8_t*>(gc + row);
          a = v.x;
          b = v.y;
        } else if (cok && row < N) {
          a = gc[row];
        }
        swap16(a, b);
        x.v[ta][2 * h] = a;
        x.v[ta][2 * h + 1] = b;
      }
  }
  __device__ __forceinline__ void ld(nstrip<RT>& x, int which, const npos<RT>& p) const { ldg(x, m[which], p); }
  __device__ __forceinline__ void ld_added(nstrip<RT>& x, int which, const npos<RT>& p) const { ldg(x, a[which], p); }
  template <typename PT>
  __device__ __forceinline__ void st_even(double* gc, const nstrip<RT>& x, bool cok, int rb) const {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double a = x.v[ta][2 * h], b = x.v[ta][2 * h + 1];
        swap16(a, b);
        const int row = 16 * ta + 8 * h + rb;
        if (cok && row < N) {
          PT v;
          v.x = a;
          v.y = b;
          *reinterpret_cast<PT*>(gc + row) = v;
        }
      }
  }
  __device__ __forceinline__ void st(int which, const nstrip<RT>& x, const npos<RT>& p) const {
    const int rb = 4 * (p.kq & 1) + 2 * (p.kq >> 1);
    double* gc = m[which] + (long long)N * min(p.col, N - 1);
    const bool cok = p.col < N;
    if (al16) {
      st_even<npair16_t>(gc, x, cok, rb);
      return;
    }
    if ((N & 1) == 0) {
      st_even<npair8_t>(gc, x, cok, rb);
      return;
    }
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        double a = x.v[ta][2 * h], b = x.v[ta][2 * h + 1];
        swap16(a, b);
        const int row = 16 * ta + 8 * h + rb;
        if (cok && row + 1 < N) {
          npair8_t v;
          v.x = a;
          v.y = b;
          *reinterpret_cast<npair8_t*>(gc + row) = v;
        } else if (cok && row < N) {
          gc[row] = a;
        }
      }
  }
  __device__ __forceinline__ double ldJ(int pm, int) const { return jpre[pm]; }   // (only ever asked for i = threadIdx.x)
  __device__ __forceinline__ void stJ(int pm, int i, double v) const {
    if (i < N) J[pm][i] = v;
  }
  // Whole matrix -> LDS as it lies in memory (column-major, leading dimension N) by LDS DMA: 1 KB per wave instruction, every
  // line fetched once; ld_raw then picks the lane's accumulator elements out of the image.  (N * N even: 16-B pieces.)
  __device__ __forceinline__ void dma_raw(const double* __restrict__ g, double* L, const npos<RT>& p) const {
    if (al16) {           // 16-B pieces: every matrix of the batch starts on a 16-B boundary (N even, the arrays 16-byte aligned)
      const int chunks = (N * N) >> 1;
      for (int c0 = 64 * p.wave; c0 < chunks; c0 += ngeo<RT>::NT) {
        if (c0 + p.lane < chunks)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2 * (c0 + p.lane)),
                                           (__attribute__((address_space(3))) void*)(L + 2 * c0), 16, 0, 0);
      }
    } else {              // 4-B pieces
      const int chunks = 2 * N * N;
      const float* g4 = reinterpret_cast<const float*>(g);
      float* L4 = reinterpret_cast<float*>(L);
      for (int c0 = 64 * p.wave; c0 < chunks; c0 += ngeo<RT>::NT) {
        if (c0 + p.lane < chunks)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g4 + c0 + p.lane),
                                           (__attribute__((address_space(3))) void*)(L4 + c0), 4, 0, 0);
      }
    }
  }
  __device__ __forceinline__ void ld_raw(nstrip<RT>& x, const double* L, const npos<RT>& p) const {
    const bool cok = p.col < N;
    const double* Lc = L + N * min(p.col, N - 1);
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        const double v = Lc[min(row, N - 1)];
        x.v[ta][r] = (cok && row < N) ? v : 0.0;
      }
  }
  // Ask for a composite matrix ahead of its use: one dword of every 128-B line by LDS DMA into a junk vector (no register is
  // tied up, the lines wait in the L2 / MALL) -- the strips that are read just before the closing products would otherwise be
  // fetched with nothing to overlap them, the six live strips leave no room to request them earlier.
  // junk: >= 256 bytes of LDS nobody reads while the requests are in flight (one dword per lane lands there) -- the callers hand in
  // sm.gjs (512 B: touched by the pivoted inverse only, which never runs between a request and the wait that covers it); a
  // vec[] row is too small below two row tiles (NP * 8 = 128 B).
  __device__ __forceinline__ void prefetch(int which, int* junk) const {
    const int lines = (N * N * 8 + 127) >> 7;
    const char* g = reinterpret_cast<const char*>(m[which]);
    for (int k0 = 0; k0 < lines; k0 += ngeo<RT>::NT) {
      const int k = min(k0 + (int)threadIdx.x, lines - 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 128 * k),
                                       (__attribute__((address_space(3))) void*)junk, 4, 0, 0);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266), native composite
// ---------------------------------------------------------------------------------------------------------------------------
// On entry: r_s / t_s = strips of the added layer's r-+ / t++ (columns >= n zero), sm.vec[0] / vec[1] = its j0+ / j0-, all waves
// past a barrier, P and Q free.  r+- = D r-+ D, t-- = D t++ D (added layers from doubling).  With G2 = (I - R+- r-+)^-1 and
// the push-through identities (I - r R)^-1 r = r G2, (I - r R)^-1 = I + r G2 R:
//   [E2 | Z] = R+- [r-+ | t--]          A = P = [R+-]   z  = J0+ + R+- j0-   (j0- rides in a spare column of r-+)
//   G2       = (I - E2)^-1              A = P = [E2]    BEFORE the products of [T--]: V is not live across the series
//   [S  | V] = T-- [r-+ | t--]          A = Q = [T--]   vs = T-- j0-
//   T21 = t++ G2 ;  Y = S G2            A = P = [t++], Q = [S]
//   [R+- | T++] = [r+- | 0] + T21 [Z | T++]      A = P = [T21]   J0+ = j0+ + T21 z   (z rides in a spare column of T++)
//   [R-+ | T--] = [R-+ | V] + Y [T++ | Z]        A = Q = [Y]     J0- = J0- + vs + Y z
// Ten products and the inverse (order 7, the usual one: four products, ninvert7), at most six live strips.
// This is the body of the STANDALONE kernel (k_ia_native): the composite lives in the reference's [N,N,S] arrays (IO = nio_ref),
// where a lane's accumulator elements are 32-byte pieces of 16 different lines, and a workgroup that waits for such loads has
// nothing to overlap them with -- stamped (tools/ia_phases.py), the loads were 55 % of a workgroup's life.  So (a) every matrix
// that is read while P or Q is free comes in as a raw image by LDS DMA (whole lines, no registers) and the lanes pick their
// elements out of LDS: r-+ / t++, then R+- / T--, later T++ through P before [T21] moves in; (b) what cannot (R-+: P and Q hold
// [T21], [Y]) is asked into the L2 a few products ahead by a one-dword-per-line DMA into a junk vector and loaded as a late addend
// UNDER the last products, its request issued ahead of the stores of R+- / T++ (the memory pipeline is in order); (c) the kernel's
// entry issues every request it can before its first wait; (d) direct accesses move 16 bytes (permlane pairs).  0.34 -> 0.46 of
// the FP64 MFMA peak at N = 60.  The layer kernel keeps nia_body_native: the same steps measured -0.4 ... -1.8 %
// on coalesced native records.
#ifdef VSM_IA_PHASES   // diagnostic build (tools/ia_phases.py): cycles of wave 0 between the stamps, summed over the workgroups
__device__ unsigned long long vsm_ia_phase_cycles[16];
#define VSM_IA_STAMP(k)                                                  \
  do {                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                   \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();        \
    if (threadIdx.x == 0) atomicAdd(&vsm_ia_phase_cycles[k], now_ - t_); \
    t_ = now_;                                                           \
    __builtin_amdgcn_sched_barrier(0);                                   \
  } while (0)
#else
#define VSM_IA_STAMP(k)
#endif
// DSYM: r+- = D r-+ D, t-- = D t++ D (added layers from doubling); else they are read through io.ld_added (surface layers).
template <int RT, int KS, bool DSYM = true, typename IO>
__device__ __forceinline__ void nia_body(nsmem<RT, (4 * KS + 2 > 16 * RT)>& sm, npos<RT>& p, int n, const IO& io, nstrip<RT>& r_s,
                                         nstrip<RT>& t_s, int* status) {
  using G = ngeo<RT>;
  constexpr unsigned dP = 0, dQ = G::AF * 8;
  double* vjp = sm.vec[0];
  double* vjm = sm.vec[1];
  double* vJp = sm.vec[2];
  double* vJm = sm.vec[3];
  double* vs = sm.vec[4];
  double* vz = sm.vec[5];
  const int tid = threadIdx.x;
  constexpr int c1 = 4 * KS, c2 = 4 * KS + 1;
  constexpr bool RID = c2 < G::NP;
  const bool own_wave = RID && p.wave == (c1 >> 4);
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2);
  int slot = 0;
  const ndpar<RT> dp(sm.usg, p);
  const ninv_ctx cx{dP, sm.P, sm.gjs, status};
#ifdef VSM_IA_PHASES
  unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
  // ---- stage: composite vectors, [R+-] -> P, [T--] -> Q ---------------------------------------------------------------------
  if (tid < G::NP) {
    vJp[tid] = io.ldJ(0, tid);
    vJm[tid] = io.ldJ(1, tid);
  }
  nstrip<RT> Z, Gs;
  nstrip<RT> A2;
  {
    nstrip<RT> A1;
    __builtin_amdgcn_s_waitcnt(0x0F70);   // their raw images are on the way to Q / P (k_ia_native)
    __syncthreads();
    io.ld_raw(A1, sm.Q, p);
    io.ld_raw(A2, sm.P, p);
    __syncthreads();
    nstore(dP, A1, p);                    // ([T--] goes to Q after the products of [R+-]: it is first read after barrier (c))
  }
  if (own_wave) {  // j0- rides in the spare column c2 of r-+:  E2[:, c2] = R+- j0-, S[:, c2] = T-- j0-
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] = laneB ? vjm[p.row(ta, r)] : r_s.v[ta][r];
  }
  if constexpr (DSYM) ndsym(t_s, t_s, dp);   // t-- = D t++ D in place (undone below: D is an involution)
  __syncthreads();                                                                                       // (a)
  VSM_IA_STAMP(0);
  if constexpr (!RID) nmv_rows<RT, KS>(dP, vjm, 1.0, sm.mv[0], p);   // R+- j0-
  {
    nstrip<RT> E;
    if constexpr (DSYM) {
      nmm2<RT, KS, true, true>(E, Z, dP, r_s, t_s, p);
    } else {   // (a surface layer: t-- from memory, here and again for the products of [T--] -- it is not kept across the inverse)
      nstrip<RT> tb;
      io.ld_added(tb, NC_TMM, p);
      nmm2<RT, KS, true, true>(E, Z, dP, r_s, tb, p);
    }
    if (own_wave) {
      double* zd = laneB ? vz : sm.vec[7];   // (the other lanes write to a dummy vector)
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          zd[row] = vJp[row] + E.v[ta][r];
        }
    }
    VSM_IA_STAMP(1);
    nstore(dQ, A2, p);                             // [T--] -> Q (free since the entry; read after barrier (c))
    const double nrm = nnorm(E, n, sm, slot, p);   // (b): every wave is done reading [R+-]
    if constexpr (!RID) {   // z = J0+ + R+- j0-
      if (tid < G::NP) vz[tid] = vJp[tid] + sm.mv[0][tid];
    }
    VSM_IA_STAMP(2);
    const int K = ninv_order(nrm, status);                       // [E2] -> P, barrier (c), series
    if (K == 7) ninvert7<RT, KS>(E, Gs, n, cx, p);               // (the usual order, in four products)
    else ninvert<RT, KS, true>(K, E, Gs, n, cx, p);
    VSM_IA_STAMP(3);
  }
  if constexpr (!RID) nmv_rows<RT, KS>(dQ, vjm, 1.0, sm.mv[1], p);   // T-- j0- (summed after barrier (d))
  io.prefetch(NC_TPP, sm.gjs);         // (four products ahead of its use: a line lives some tens of microseconds in the L2)
  nstrip<RT> V;
  {
    nstrip<RT> S;
    if constexpr (DSYM) {
      nmm2<RT, KS, true, true>(S, V, dQ, r_s, t_s, p);   // (Q = [T--] has not been touched since (a))
    } else {
      nstrip<RT> tb;
      io.ld_added(tb, NC_TMM, p);
      nmm2<RT, KS, true, true>(S, V, dQ, r_s, tb, p);
    }
    if (own_wave) {
      double* sd = laneB ? vs : sm.vec[7];
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) sd[p.row(ta, r)] = S.v[ta][r];
    }
    // r_s <- r+- (the accumulator of the R+- update; its rider column is never used), t_s <- t++
    if constexpr (DSYM) {
      ndsym(r_s, r_s, dp);
      ndsym(t_s, t_s, dp);
    } else {
      io.ld_added(r_s, NC_RPM, p);
    }
    VSM_IA_STAMP(4);
    __syncthreads();                      // (d): [E2] (series) and [T--] no longer read
    if constexpr (!RID) {
      if (tid < G::NP) vs[tid] = sm.mv[1][tid];
    }
    nstore(dQ, S, p);                     // [S]   -> Q
    nstore(dP, t_s, p);                   // [t++] -> P
  }
  __syncthreads();                        // (e)
  io.prefetch(NC_RMP, sm.gjs);
  VSM_IA_STAMP(5);
  nstrip<RT> Tpp;
  {
    nstrip<RT> X, Y;
    nmm<RT, KS, true>(X, dP, Gs, p);      // T21 = t++ G2
    nmm<RT, KS, true>(Y, dQ, Gs, p);      // Y = S G2 = T01 r-+
    VSM_IA_STAMP(6);
    __syncthreads();                      // (f): [t++], [S] no longer read
    io.dma_raw(io.m[NC_TPP], sm.P, p);    // T++ as a raw image through P (its lines wait in the L2), before [T21] moves in
    nstore(dQ, Y, p);                     // [Y]   -> Q
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    io.ld_raw(Tpp, sm.P, p);
    __syncthreads();
    nstore(dP, X, p);                     // [T21] -> P
  }
  __builtin_amdgcn_sched_barrier(0);      // (the composite strips are requested once X and Y are dead, not above their stores)
  nstrip<RT> Rmp;
  __syncthreads();                        // (g)
  VSM_IA_STAMP(7);
  if constexpr (!RID) {   // T21 z , Y z
    nmv_rows<RT, KS>(dP, vz, 1.0, sm.mv[0], p);
    nmv_rows<RT, KS>(dQ, vz, 1.0, sm.mv[1], p);
  }
  if (own_wave) {  // z rides in the spare column c1 of T++:  (T21 T++)[:, c1] = T21 z, (Y T++)[:, c1] = Y z
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        Tpp.v[ta][r] = laneA ? vz[p.row(ta, r)] : Tpp.v[ta][r];
      }
  }
  {
    nstrip<RT> acc;
    nmm2<RT, KS, false, true>(r_s, acc, dP, Z, Tpp, p);   // R+- = r+- + T21 Z ; T++ = T21 T++
    VSM_IA_STAMP(8);
    io.ld(Rmp, NC_RMP, p);                   // (ahead of the stores in the memory pipeline; it arrives under the last products)
    io.st(NC_RPM, r_s, p);
    io.st(NC_TPP, acc, p);
    if (laneA) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          io.stJ(0, row, vjp[row] + acc.v[ta][r]);
        }
    }
  }
  VSM_IA_STAMP(9);
  {   // R-+ as a late addend (its strip was still on the way)
    nstrip<RT> YT;
    nmm2<RT, KS, true, false>(YT, V, dQ, Tpp, Z, p);       // Y T++ ; T-- = V + Y Z
    VSM_IA_STAMP(10);
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) Rmp.v[ta][r] += YT.v[ta][r];
  }
  io.st(NC_RMP, Rmp, p);
  io.st(NC_TMM, V, p);
  if (laneA) {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        io.stJ(1, row, vJm[row] + vs[row] + Rmp.v[ta][r]);
      }
  }
  if constexpr (!RID) {   // J0+ = j0+ + T21 z ; J0- = J0- + T-- j0- + Y z
    __syncthreads();
    if (tid < G::NP) {
      io.stJ(0, tid, vjp[tid] + sm.mv[0][tid]);
      io.stJ(1, tid, vJm[tid] + vs[tid] + sm.mv[1][tid]);
    }
  }
#ifdef VSM_IA_PHASES
  __builtin_amdgcn_s_waitcnt(0);   // (the stores have left)
  VSM_IA_STAMP(11);
  if (threadIdx.x == 0) atomicAdd(&vsm_ia_phase_cycles[15], 1ull);
#endif
}

// interaction!(::ScatteringInterface_11) on the reference's arrays (vsm_interaction_f64, N <= 64): one workgroup per point
template <int RT, int KS, bool DSYM>
__global__ __launch_bounds__(ngeo<RT>::NT, ngeo<RT>::WPS) void k_ia_native(int N, int ns, int al16, composite<double> c,
                                                                            added<double> a, int* status) {
  using G = ngeo<RT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using SM = nsmem<RT, (4 * KS + 2 > 16 * RT)>;
  SM& sm = *reinterpret_cast<SM*>(smem_raw);
  npos<RT> p(sm.P);
  const long long s = blockIdx.x, NN = (long long)N * N;
  const int tid = threadIdx.x;
  nio_ref<RT> io;
  io.N = N;
  io.al16 = al16 != 0;
  io.m[NC_RMP] = c.R_mp + s * NN;
  io.m[NC_RPM] = c.R_pm + s * NN;
  io.m[NC_TPP] = c.T_pp + s * NN;
  io.m[NC_TMM] = c.T_mm + s * NN;
  io.J[0] = c.J0_p + s * N;
  io.J[1] = c.J0_m + s * N;
  io.a[NC_RMP] = a.r_mp + s * a.mat_stride;
  io.a[NC_TPP] = a.t_pp + s * a.mat_stride;
  io.a[NC_RPM] = DSYM ? nullptr : a.r_pm + s * a.mat_stride;
  io.a[NC_TMM] = DSYM ? nullptr : a.t_mm + s * a.mat_stride;
#ifdef VSM_IA_PHASES
  unsigned long long t_ = __builtin_amdgcn_s_memtime();
#endif
  // every request the entry can make goes out before the first wait: the four source vectors into registers, r-+ / t++ as raw
  // images by DMA, and the lines of R+- / T-- into the L2 (their images follow as soon as P and Q have been read)
  const bool in = tid < N;
  const double j0p = in ? a.j0_p[s * N + tid] : 0.0, j0m = in ? a.j0_m[s * N + tid] : 0.0;
  io.jpre[0] = in ? io.J[0][tid] : 0.0;
  io.jpre[1] = in ? io.J[1][tid] : 0.0;
  nstrip<RT> r_s, t_s;
  io.dma_raw(io.a[NC_RMP], sm.P, p);
  io.dma_raw(io.a[NC_TPP], sm.Q, p);
  io.prefetch(NC_RPM, sm.gjs);
  io.prefetch(NC_TMM, sm.gjs);
  if (tid < G::NP) {
    sm.vec[0][tid] = j0p;
    sm.vec[1][tid] = j0m;
    sm.usg[tid] = (DSYM && (tid % ns) >= 2) ? -1.0 : 1.0;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  io.ld_raw(r_s, sm.P, p);
  io.ld_raw(t_s, sm.Q, p);
  __syncthreads();
  io.dma_raw(io.m[NC_RPM], sm.Q, p);   // (nia_body waits for them)
  io.dma_raw(io.m[NC_TMM], sm.P, p);
#ifdef VSM_IA_PHASES
  __builtin_amdgcn_s_waitcnt(0);
  VSM_IA_STAMP(12);
#endif
  nia_body<RT, KS, DSYM>(sm, p, N, io, r_s, t_s, status);
}

// the LDS block decides the workgroups per CU the register budgets (ngeo::WPS) are set for
static_assert(sizeof(nsmem<2>) * 8 <= 163840, "two row tiles: eight workgroups per CU");
static_assert(sizeof(nsmem<3>) * 4 <= 163840, "three row tiles: four workgroups per CU");
static_assert(sizeof(nsmem<3, true>) * 3 <= 163840, "three row tiles, mat-vec sources: three workgroups per CU");
static_assert(sizeof(nsmem<4, true>) * 2 <= 163840, "four row tiles: two workgroups per CU");
static_assert(sizeof(nsmem<6, true>) <= 163840 && sizeof(nsmem<6>) <= 163840, "six row tiles: one workgroup per CU");

template <int RT, int KS>
__global__ __launch_bounds__(ngeo<RT>::NT, ngeo<RT>::WPS) void k_layer_native(int n, int gsz, unsigned uvmask, int ndoubl,
                                                                                int toa, const double* __restrict__ pre,
                                                                                nlayer_comps a, int* status) {
  using G = ngeo<RT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using SM = nsmem<RT, (4 * KS + 2 > 16 * RT)>;
  SM& sm = *reinterpret_cast<SM*>(smem_raw);
  npos<RT> p(sm.P);
  const int s = blockIdx.x, isub = blockIdx.y;
  const double* img = pre + ((long long)isub * gridDim.x + s) * G::PRE_STRIDE;
  double* comp = a.c[isub] + (long long)s * G::COMP_STRIDE;
  nstrip<RT> r_s, t_s;
  ned_body<RT, KS>(sm, p, n, gsz, uvmask, ndoubl, img, r_s, t_s, status);
  if (toa) {   // copy_added_to_composite! (rt_helpers.jl:188-200)
    const ndpar<RT> dp(sm.usg, p);
    nst_native(comp + NC_RMP * G::AF, r_s, p);
    nst_native(comp + NC_TPP * G::AF, t_s, p);
    nstrip<RT> d;
    ndsym(d, r_s, dp);
    nst_native(comp + NC_RPM * G::AF, d, p);
    ndsym(d, t_s, dp);
    nst_native(comp + NC_TMM * G::AF, d, p);
    if (threadIdx.x < G::NP) {
      comp[4 * G::AF + threadIdx.x] = sm.vec[0][threadIdx.x];
      comp[4 * G::AF + G::NP + threadIdx.x] = sm.vec[1][threadIdx.x];
    }
    return;
  }
  nia_body_native<RT, KS>(sm, p, n, comp, r_s, t_s, status);
}

}  // namespace

// one object per k-step count (parallel build): vsm_native_<KS>.o is this file built with -DVSM_NATIVE_KS=<KS>
#define VSM_NCAT2(a, b) a##b
#define VSM_NCAT(a, b) VSM_NCAT2(a, b)
#define VSM_NATIVE_DECL(KS) \
  int VSM_NCAT(launch_layer_native_, KS)(int, int, int, unsigned, int, int, int, const double*, const nlayer_comps&, int*, hipStream_t); \
  int VSM_NCAT(launch_ia_native_, KS)(int, int, const composite<double>&, const added<double>&, int*, hipStream_t);

#ifdef VSM_NATIVE_KS
VSM_NATIVE_DECL(VSM_NATIVE_KS)
int VSM_NCAT(launch_layer_native_, VSM_NATIVE_KS)(int S, int nsub, int n, unsigned uvmask, int gsz, int ndoubl, int toa,
                                                  const double* pre, const nlayer_comps& comps, int* status, hipStream_t st) {
  constexpr int KS = VSM_NATIVE_KS;
  // KS = 4, 8, 12, 16 (n = 13..16, 29..32, 45..48, 61..64): KS / 4 row tiles without spare columns (mat-vec source path) instead of one row
  // tile more for the two rider columns
  constexpr int RT = (KS % 4 == 0 && KS >= 4) ? KS / 4 : (4 * KS + 2 + 15) / 16;
  static_assert(RT >= 1 && RT <= 6, "n <= 96");
  using SM = nsmem<RT, (4 * KS + 2 > 16 * RT)>;
  auto kern = k_layer_native<RT, KS>;
  const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kern), sizeof(SM), "hipFuncSetAttribute(k_layer_native)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(kern, dim3(S, nsub), dim3(ngeo<RT>::NT), sizeof(SM), st, n, gsz, uvmask, ndoubl, toa, pre, comps, status);
  VSM_LAUNCH_CHECK("k_layer_native");
  return VSM_OK;
}
#if VSM_NATIVE_KS <= 16
int VSM_NCAT(launch_ia_native_, VSM_NATIVE_KS)(int N, int S, const composite<double>& c, const added<double>& a, int* status,
                                               hipStream_t st) {
  constexpr int KS = VSM_NATIVE_KS;
  constexpr int RT = (KS % 4 == 0 && KS >= 4) ? KS / 4 : (4 * KS + 2 + 15) / 16;
  using SM = nsmem<RT, (4 * KS + 2 > 16 * RT)>;
  auto kd = k_ia_native<RT, KS, true>;
  auto kg = k_ia_native<RT, KS, false>;
  int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kd), sizeof(SM), "hipFuncSetAttribute(k_ia_native)");
  if (!prepared) prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kg), sizeof(SM), "hipFuncSetAttribute(k_ia_native general)");
  if (prepared) return prepared;
  // 16-byte DMA pieces need every matrix of the batch on a 16-byte boundary: N even (a matrix is N N 8 bytes) and aligned arrays
  unsigned long long bits = (unsigned long long)c.R_mp | (unsigned long long)c.R_pm | (unsigned long long)c.T_pp |
                            (unsigned long long)c.T_mm | (unsigned long long)a.r_mp | (unsigned long long)a.t_pp |
                            (unsigned long long)(a.mat_stride * 8);
  if (!a.d_symmetric) bits |= (unsigned long long)a.r_pm | (unsigned long long)a.t_mm;
  const int al16 = ((N & 1) == 0 && (bits & 15ull) == 0) ? 1 : 0;
  if (a.d_symmetric)   // (d_symmetric carries n_stokes)
    hipLaunchKernelGGL(kd, dim3(S), dim3(ngeo<RT>::NT), sizeof(SM), st, N, a.d_symmetric, al16, c, a, status);
  else
    hipLaunchKernelGGL(kg, dim3(S), dim3(ngeo<RT>::NT), sizeof(SM), st, N, 1, al16, c, a, status);
  VSM_LAUNCH_CHECK("k_ia_native");
  return VSM_OK;
}
#endif
}  // namespace vsm
#ifdef VSM_IA_PHASES
extern "C" int vsm_debug_ia_phases(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_ia_phase_cycles), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_ia_phase_cycles), z, sizeof(z));
  }
  return 0;
}
#endif

#else  // ---- dispatcher object: pre-pass, layout conversion, the run object and its C ABI -----------------------------------

VSM_NATIVE_DECL(1)
VSM_NATIVE_DECL(2)
VSM_NATIVE_DECL(3)
VSM_NATIVE_DECL(4)
VSM_NATIVE_DECL(5)
VSM_NATIVE_DECL(6)
VSM_NATIVE_DECL(7)
VSM_NATIVE_DECL(8)
VSM_NATIVE_DECL(9)
VSM_NATIVE_DECL(10)
VSM_NATIVE_DECL(11)
VSM_NATIVE_DECL(12)
VSM_NATIVE_DECL(13)
VSM_NATIVE_DECL(14)
VSM_NATIVE_DECL(15)
VSM_NATIVE_DECL(16)
VSM_NATIVE_DECL(17)
VSM_NATIVE_DECL(18)
VSM_NATIVE_DECL(19)
VSM_NATIVE_DECL(20)
VSM_NATIVE_DECL(21)
VSM_NATIVE_DECL(22)
VSM_NATIVE_DECL(23)
VSM_NATIVE_DECL(24)

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// Elemental pre-pass: elemental! incl. the SFI source (elemental.jl:289-392) of every (point, sub-problem) of a layer as two
// A-form images + vectors (ngeo::PRE_STRIDE).  Sub-row i of a group g (Stokes components g[0..gsz)) is row
// (i / gsz) n_stokes + g[i % gsz] of the full problem.  Thread = (row, column phase): 16 consecutive threads write 16
// consecutive rows of one column (128 contiguous bytes of an image block).
// ---------------------------------------------------------------------------------------------------------------------------
// ST = the storage type of the caller's arrays (double, or float: a Float32 run whose blocks fit these kernels -- inputs are
// converted on load, everything from here to the export of the composite is FP64).
template <typename ST>
struct nsub_pre {
  int m, gsz;
  int g[4];
  zsrc<ST> z;
};
template <typename ST>
struct npre_args {
  nsub_pre<ST> s[NSUB_MAX];
};
template <int RT, bool MIX, typename ST>
__global__ __launch_bounds__(64 * RT) void k_elemental_native(quad<ST> q, int n, int ndoubl, const ST* __restrict__ dtau,
                                                              const ST* __restrict__ varpi, const ST* __restrict__ tau_sum,
                                                              const ST* __restrict__ F0, npre_args<ST> a,
                                                              double* __restrict__ pre) {
  using G = ngeo<RT>;
  constexpr int NP = G::NP;
  __shared__ double mus[NP], xs[NP], es[NP], ems[NP], wts[NP];
  __shared__ int frow[NP];
  __shared__ int thick_flag;
  const int s = blockIdx.x, isub = blockIdx.y, tid = threadIdx.x;
  const nsub_pre<ST>& sp = a.s[isub];
  const int N = q.N, ns = q.n_stokes, m = sp.m, gsz = sp.gsz;
  const zsrc<ST> z = sp.z;
  const double d = (double)dtau[s], w = (double)varpi[s];
  const int ncomp = MIX ? z.ncomp : 0;
  const long long NNz = (long long)N * N;
  const ST* Zp = z.Zpp + (ncomp ? 0 : (long long)s * z.zs);
  const ST* Zm = z.Zmp + (ncomp ? 0 : (long long)s * z.zs);
  double fk[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < ncomp) fk[k] = (double)z.fcomp[(long long)s * ncomp + k];
  auto zget = [&](const ST* Z, long long zo) {
    if (ncomp == 0) return (double)Z[zo];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ncomp) acc += fk[k] * (double)Z[k * NNz + zo];
    return acc;
  };
  if (tid < NP) {
    const bool in = tid < n;
    const int ga = sp.g[tid % gsz];
    const int fr = in ? (tid / gsz) * ns + ga : 0;
    const double mu = in ? (double)q.mu[fr] : 1.0;
    const double x = d / mu;
    frow[tid] = fr;
    mus[tid] = mu;
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
    wts[tid] = in ? (double)q.wt[fr] : 0.0;
  }
  if (tid == 0) thick_flag = 0;
  __syncthreads();
  if (tid < n && xs[tid] >= 0.5) thick_flag = 1;   // (benign race: every writer stores 1)
  __syncthreads();
  const bool thick = thick_flag != 0;
  const int i = tid % NP, ph = tid / NP;           // 64 RT threads = NP rows x 4 column phases
  const int ic = min(i, n - 1);
  const double mi = mus[i], xi = xs[i], ai = ems[i], ei = es[i];
  const bool uvi = i < n && sp.g[i % gsz] >= 2;
  const double sg = (ndoubl >= 1 && uvi) ? -1.0 : 1.0;   // starred R* = D R (elemental.jl:403-422)
  double* out = pre + ((long long)isub * gridDim.x + s) * G::PRE_STRIDE;
  double* R = out;
  double* T = out + G::AF;
  const int Kend = ((n + 3) >> 2) << 2;
  const int c1 = Kend, c2 = Kend + 1;
  const bool riders_in = ndoubl > 0 && c2 < NP;   // (n = 61..64: no spare column; the layer kernel uses mat-vecs)
  const int fi = frow[ic];
  for (int j = ph; j < NP; j += 4) {
    if (riders_in && (j == c1 || j == c2)) continue;   // written below
    double rv = 0.0, tv = 0.0;
    if (j < n) {
      const double wt = wts[j];
      const double wct = (m == 0) ? wt / 2.0 : wt / 4.0;
      const long long zo = fi + (long long)N * frow[j];
      double rr, tt;
      elemental_pair(w, zget(Zp, zo), zget(Zm, zo), mi, xi, ai, ei, mus[j], xs[j], ems[j], es[j], wct, i == j, thick, rr, tt);
      const bool active = wct > (double)num<ST>::eps();   // eps(FT) of the MODEL's float type (elemental.jl:296)
      if (i < n) {
        rv = active ? rr * sg : 0.0;
        tv = active ? tt : ((i == j) ? ei : 0.0);
      }
    }
    R[naf_idx<RT>(i, j)] = rv;
    T[naf_idx<RT>(i, j)] = tv;
  }
  if (ph == 0) {   // SFI source of the solar beam: the same formulas with the solar column (see elemental_pair)
    const int i0 = ns * q.i_mu0;
    const double mu0n = (double)q.mu[i0];
    const double x0 = d / mu0n, e0 = exp(-x0), a0 = expm1(-x0);
    double zp = 0.0, zm = 0.0;
    for (int qq = 0; qq < ns; ++qq) {
      const long long zo = fi + (long long)N * (i0 + qq);
      const double f = (double)F0[qq + (long long)ns * s];
      zp += zget(Zp, zo) * f;
      zm += zget(Zm, zo) * f;
    }
    double rr, tt;
    // (the diagonal case mu_i == mu_0 of the source never takes the i == j form: elemental.jl:370-382)
    elemental_pair(w, zp, zm, mi, xi, ai, ei, mu0n, x0, a0, e0, (m == 0) ? 0.5 : 0.25, false, thick || x0 >= 0.5, rr, tt);
    const double att = exp(-(double)tau_sum[s] / mu0n);
    const double vp = (i < n) ? tt * att : 0.0;
    const double vm = (i < n) ? rr * att * sg : 0.0;
    const double expk0 = exp(-d / (double)q.mu0);
    out[2 * G::AF + i] = vp;
    out[2 * G::AF + NP + i] = vm;
    out[2 * G::AF + 2 * NP + i] = expk0;
    if (riders_in) {   // t[:, c1] = j0+, t[:, c2] = j1- = j0- expk ;  r[:, c1] = j0-, r[:, c2] = j0+
      T[naf_idx<RT>(i, c1)] = vp;
      T[naf_idx<RT>(i, c2)] = vm * expk0;
      R[naf_idx<RT>(i, c1)] = vm;
      R[naf_idx<RT>(i, c2)] = vp;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// native <-> reference layout ([N,N,S] column-major composite arrays), one workgroup per point
// ---------------------------------------------------------------------------------------------------------------------------
struct ngroup_map {
  int ngroups;
  int grp_of[4];    // group of Stokes component a
  int pos_in[4];    // its position inside the group
  int gsz[4];       // per group
  int rt[4];        // per group: row tiles
  int n[4];         // per group: rows
  double* base[4];  // per group: native composites [S] (stride COMP_STRIDE of its RT)
};
__device__ __forceinline__ int nnat_idx_rt(int rt, int i, int j) {   // nnat_idx<RT> with the row-tile count as an argument
  const int w = j >> 4, l15 = j & 15, ta = i >> 4, mm = i & 15, kq = mm & 3, r = mm >> 2;
  const int lane = (kq << 4) | l15, u = 2 * ta + (r >> 1);
  return ((w * 2 * rt + u) * 64 + lane) * 2 + (r & 1);
}
template <typename ST>
__global__ __launch_bounds__(256) void k_native_export(int N, int ns, ngroup_map gm, composite<ST> c) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  ST* out[4] = {c.R_mp + s * NN, c.R_pm + s * NN, c.T_pp + s * NN, c.T_mm + s * NN};
  for (int e = tid; e < N * N; e += 256) {
    const int i = e % N, j = e / N;
    const int ai = i % ns, aj = j % ns;
    const int gi = gm.grp_of[ai];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (gi == gm.grp_of[aj]) {
      const int rt = gm.rt[gi], np = 16 * rt, af = np * np;
      const double* cn = gm.base[gi] + (long long)s * (4 * af + 2 * np);
      const int is = (i / ns) * gm.gsz[gi] + gm.pos_in[ai], js = (j / ns) * gm.gsz[gi] + gm.pos_in[aj];
      const int ix = nnat_idx_rt(rt, is, js);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = cn[k * af + ix];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k][e] = (ST)v[k];
  }
  if (tid < N) {
    const int ai = tid % ns, gi = gm.grp_of[ai];
    const int rt = gm.rt[gi], np = 16 * rt, af = np * np;
    const double* cn = gm.base[gi] + (long long)s * (4 * af + 2 * np);
    const int is = (tid / ns) * gm.gsz[gi] + gm.pos_in[ai];
    c.J0_p[(long long)s * N + tid] = (ST)cn[4 * af + is];
    c.J0_m[(long long)s * N + tid] = (ST)cn[4 * af + np + is];
  }
}
// reference layout -> native (padding zero); elements that couple different groups are dropped (they are exact zeros for a
// composite that was built under the same coupling)
template <typename ST>
__global__ __launch_bounds__(256) void k_native_import(int N, int ns, ngroup_map gm, composite<ST> c) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  const ST* in[4] = {c.R_mp + s * NN, c.R_pm + s * NN, c.T_pp + s * NN, c.T_mm + s * NN};
  for (int g = 0; g < gm.ngroups; ++g) {
    const int rt = gm.rt[g], np = 16 * rt, af = np * np, n = gm.n[g], gsz = gm.gsz[g];
    double* cn = gm.base[g] + (long long)s * (4 * af + 2 * np);
    int comp_of[4] = {0, 0, 0, 0};   // Stokes component of position k of the group
    for (int a = 0; a < ns; ++a)
      if (gm.grp_of[a] == g) comp_of[gm.pos_in[a]] = a;
    for (int e = tid; e < af; e += 256) {
      const int is = e % np, js = e / np;
      double v[4] = {0.0, 0.0, 0.0, 0.0};
      if (is < n && js < n) {
        const int i = (is / gsz) * ns + comp_of[is % gsz], j = (js / gsz) * ns + comp_of[js % gsz];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (double)in[k][i + (long long)N * j];
      }
      const int ix = nnat_idx_rt(rt, is, js);
#pragma unroll
      for (int k = 0; k < 4; ++k) cn[k * af + ix] = v[k];
    }
    if (tid < np) {
      double vp = 0.0, vm = 0.0;
      if (tid < n) {
        const int i = (tid / gsz) * ns + comp_of[tid % gsz];
        vp = (double)c.J0_p[(long long)s * N + i];
        vm = (double)c.J0_m[(long long)s * N + i];
      }
      cn[4 * af + tid] = vp;
      cn[4 * af + np + tid] = vm;
    }
  }
}

// Which Stokes components a phase matrix couples: bit 4 a + b of mask[b] is set when some element (i, j) with i % ns == a,
// j % ns == b of block b of Zpp / Zmp [N,N,nblocks] is not exactly zero.
template <typename ST>
__global__ __launch_bounds__(256) void k_stokes_coupling(int N, int ns, const ST* __restrict__ Zpp, const ST* __restrict__ Zmp,
                                                         int* __restrict__ mask) {
  unsigned mm = 0;
  const long long NN = (long long)N * N;
  const ST* zp = Zpp + NN * blockIdx.y;
  const ST* zm = Zmp + NN * blockIdx.y;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < NN; e += 256ll * gridDim.x) {
    const int i = (int)(e % N), j = (int)(e / N);
    if (zp[e] != ST(0) || zm[e] != ST(0)) mm |= 1u << (4 * (i % ns) + (j % ns));
  }
  if (mm) atomicOr(mask + blockIdx.y, (int)mm);
}

// Guard of the contract "the coupling masks cover every Z handed to vsm_run_layer" (a caller outside this repository derives the
// masks itself): bit 4 a + b of allowed[moment] = an element (i, j) with i % ns == a, j % ns == b may be non-zero -- a and b in one
// block of the run AND that block not taken as a diagonal step in this layer.  Any other non-zero element of the layer's phase
// matrices raises VSM_DEVSTAT_MASK (vsm_device_status): the run drops that coupling silently otherwise.  One workgroup per matrix
// (blockIdx.x: the scatterer's block of a component stack, the spectral point of a per-point Z, or the one shared matrix).
template <typename ST>
struct nguard_args {
  const ST* zp[NSUB_MAX];
  const ST* zm[NSUB_MAX];
  unsigned allowed[NSUB_MAX];
};
template <typename ST>
__global__ __launch_bounds__(256) void k_coupling_guard(int N, int ns, long long mat_stride, nguard_args<ST> a, int* status) {
  const long long NN = (long long)N * N;
  const int im = blockIdx.y;
  const ST* zp = a.zp[im] + mat_stride * blockIdx.x;
  const ST* zm = a.zm[im] + mat_stride * blockIdx.x;
  unsigned mm = 0;
  for (int e = threadIdx.x; e < NN; e += 256) {
    const int i = e % N, j = e / N;
    if (zp[e] != ST(0) || zm[e] != ST(0)) mm |= 1u << (4 * (i % ns) + (j % ns));
  }
  if (mm & ~a.allowed[im]) atomicOr(&status[0], (int)VSM_DEVSTAT_MASK);
}

// A layer step of a sub-problem whose block of the phase matrix is exactly zero in this layer (a Stokes block that nothing
// scatters into -- U at m = 0 in a Rayleigh atmosphere --, a Rayleigh layer at m >= 3): elemental! gives r = 0, j = 0,
// t = diag(e^{-dtau / mu_i}), doubling! squares t (rt_helpers.jl:102-166 with r = 0: G = I, t' = t t), and
// interaction!(_11) with r-+ = r+- = 0, j0 = 0 (interaction.jl:207-266: T01 = T--, T21 = t++) scales the composite:
//   T++ <- t T++,  T-- <- T-- t,  R+- <- t R+- t,  J0+ <- t J0+;   R-+, J0- unchanged.
// t = e^{-dtau 2^nd / mu_i} here (the doubled layer in one exponential).  One workgroup per (point, sub-problem) over the native
// records: element (i, j) from its position.  TOA: the composite is the layer itself (R = 0, T = diag(t), J = 0).
struct ndiag_args {
  double* comp[NSUB_MAX];
  int gsz[NSUB_MAX];
  int g0[NSUB_MAX];    // the group's Stokes components, 4 bits each
};
// mode 0: the scaling above; 1: TOA; 2: the composite is still (R = 0, T = diag, J = 0) -- every layer above was such a step --:
// only the two diagonals change.
template <typename ST>
__global__ __launch_bounds__(256) void k_native_diag_layer(quad<ST> q, int rt, int n, double scale, int mode,
                                                           const ST* __restrict__ dtau, ndiag_args a) {
  __shared__ double tt[NATIVE_MAX_ROWS];
  const int s = blockIdx.x, isub = blockIdx.y, tid = threadIdx.x;
  const int np = 16 * rt, af = np * np, gsz = a.gsz[isub];
  double* comp = a.comp[isub] + (long long)s * (4 * af + 2 * np);
  if (tid < np) {
    double t = 0.0;
    if (tid < n) {
      const int fr = (tid / gsz) * q.n_stokes + ((a.g0[isub] >> (4 * (tid % gsz))) & 15);
      t = exp(-(double)dtau[s] * scale / (double)q.mu[fr]);
    }
    tt[tid] = t;
    if (mode == 2 && tid < n) {
      const int ix = nnat_idx_rt(rt, tid, tid);
      comp[NC_TPP * af + ix] *= t;
      comp[NC_TMM * af + ix] *= t;
    }
  }
  if (mode == 2) return;
  const int toa = mode == 1;
  __syncthreads();
  for (int e = tid; e < af; e += 256) {
    const int half = e & 1, lane = (e >> 1) & 63, unit = (e >> 7) % (2 * rt), w = (e >> 7) / (2 * rt);
    const int i = 16 * (unit >> 1) + (lane >> 4) + 4 * (2 * (unit & 1) + half), j = 16 * w + (lane & 15);
    const double ti = tt[i], tj = tt[j];
    if (toa) {
      const double d = (i == j) ? ti : 0.0;
      comp[NC_RMP * af + e] = 0.0;
      comp[NC_RPM * af + e] = 0.0;
      comp[NC_TPP * af + e] = d;
      comp[NC_TMM * af + e] = d;
    } else {
      comp[NC_RPM * af + e] *= ti * tj;
      comp[NC_TPP * af + e] *= ti;
      comp[NC_TMM * af + e] *= tj;
    }
  }
  if (tid < np) {
    if (toa) {
      comp[4 * af + tid] = 0.0;
      comp[4 * af + np + tid] = 0.0;
    } else {
      comp[4 * af + tid] *= tt[tid];
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// the run object (vsm_native_run.h)
// ---------------------------------------------------------------------------------------------------------------------------
}  // namespace vsm

namespace vsm {
namespace {

// groups of Stokes components = connected components of the (symmetrised) coupling mask; coupling < 0: one dense group
static int stokes_groups(int ns, int coupling, int grp_of[4], int groups[4][4], int gsz[4]) {
  bool adj[4][4] = {};
  for (int a = 0; a < ns; ++a)
    for (int b = 0; b < ns; ++b) {
      const bool on = coupling < 0 || ((coupling >> (4 * a + b)) & 1) || ((coupling >> (4 * b + a)) & 1) || a == b;
      adj[a][b] = on;
    }
  int ng = 0;
  for (int a = 0; a < 4; ++a) grp_of[a] = -1;
  for (int a = 0; a < ns; ++a) {
    if (grp_of[a] >= 0) continue;
    int stack[4], top = 0;
    stack[top++] = a;
    grp_of[a] = ng;
    while (top) {
      const int x = stack[--top];
      for (int b = 0; b < ns; ++b)
        if (adj[x][b] && grp_of[b] < 0) {
          grp_of[b] = ng;
          stack[top++] = b;
        }
    }
    ++ng;
  }
  for (int g = 0; g < ng; ++g) {
    gsz[g] = 0;
    for (int a = 0; a < ns; ++a)
      if (grp_of[a] == g) groups[g][gsz[g]++] = a;   // ascending component order inside a group
  }
  return ng;
}
static inline int rt_of(int n) {   // (as the launchers of the layer kernels choose it)
  const int ks = (n + 3) / 4;
  return (ks % 4 == 0 && ks >= 4) ? ks / 4 : (4 * ks + 2 + 15) / 16;
}
static inline size_t comp_stride_rt(int rt) { return (size_t)4 * (16 * rt) * (16 * rt) + 2 * 16 * rt; }
static inline size_t pre_stride_rt(int rt) { return (size_t)2 * (16 * rt) * (16 * rt) + 3 * 16 * rt; }

// f32: the FP32 family (vsm_native32.hip: blocks of <= 96 rows, records in floats); else the FP64 one (<= 64 rows, doubles)
static int plan_subs(bool f32, int N, int ns, int nm, const int* m, const int* coupling, std::vector<nat_sub>& subs, size_t S,
                     size_t& total) {
  if (N <= 0 || ns < 1 || ns > 4 || N % ns) {
    set_error("vsm_run: bad N = %d / n_stokes = %d", N, ns);
    return VSM_ERR_INVALID_ARG;
  }
  const int nq = N / ns;
  total = 0;
  subs.clear();
  for (int im = 0; im < nm; ++im) {
    int grp_of[4], groups[4][4], gsz[4];
    const int ng = stokes_groups(ns, coupling ? coupling[im] : -1, grp_of, groups, gsz);
    for (int g = 0; g < ng; ++g) {
      nat_sub sb;
      sb.im = im;
      sb.m = m ? m[im] : 0;
      sb.gsz = gsz[g];
      sb.uvmask = 0;
      for (int k = 0; k < 4; ++k) sb.g[k] = k < gsz[g] ? groups[g][k] : groups[g][0];
      for (int k = 0; k < gsz[g]; ++k)
        if (groups[g][k] >= 2) sb.uvmask |= 1u << k;
      sb.n = nq * gsz[g];
      const int max_rows = f32 ? NATIVE32_MAX_ROWS : NATIVE_MAX_ROWS;
      if (sb.n > max_rows) {
        set_error("vsm_run: a block of %d rows (N = %d, %d coupled Stokes components) is beyond the native kernels (%d)", sb.n, N,
                  gsz[g], max_rows);
        return VSM_ERR_UNSUPPORTED;
      }
      sb.ks = (sb.n + 3) / 4;
      sb.rt = f32 ? native32_rt_of(sb.n) : rt_of(sb.n);
      sb.comp_off = total;
      total += (f32 ? native32_comp_stride(sb.rt) : comp_stride_rt(sb.rt)) * S;
      subs.push_back(sb);
    }
  }
  return VSM_OK;
}

static int launch_layer_native(int ks, int S, int nsub, int n, unsigned uvmask, int gsz, int ndoubl, int toa, const double* pre,
                               const nlayer_comps& comps, int* status, hipStream_t st) {
#define VSM_NL(KS) \
  case KS: return VSM_NCAT(launch_layer_native_, KS)(S, nsub, n, uvmask, gsz, ndoubl, toa, pre, comps, status, st);
  switch (ks) {
    VSM_NL(1) VSM_NL(2) VSM_NL(3) VSM_NL(4) VSM_NL(5) VSM_NL(6) VSM_NL(7) VSM_NL(8) VSM_NL(9) VSM_NL(10) VSM_NL(11) VSM_NL(12)
    VSM_NL(13) VSM_NL(14) VSM_NL(15) VSM_NL(16) VSM_NL(17) VSM_NL(18) VSM_NL(19) VSM_NL(20) VSM_NL(21) VSM_NL(22) VSM_NL(23) VSM_NL(24)
    default: break;
  }
#undef VSM_NL
  set_error("launch_layer_native: no kernel for %d k-steps", ks);
  return VSM_ERR_UNSUPPORTED;
}

}  // namespace
// interaction!(::ScatteringInterface_11) on the reference-layout arrays, FP64, N <= 64 (VSM_ERR_UNSUPPORTED beyond)
int native_interaction11(int N, int S, const composite<double>& c, const added<double>& a, hipStream_t st) {
  if (N < 1 || N > 64) return VSM_ERR_UNSUPPORTED;   // (six live strips: beyond four row tiles the interaction alone stays on k_ia128)
  if (S <= 0) return VSM_OK;
  int* status = device_status();
  if (!status) return VSM_ERR_HIP;
#define VSM_NI(KS) \
  case KS: return VSM_NCAT(launch_ia_native_, KS)(N, S, c, a, status, st);
  switch ((N + 3) / 4) {
    VSM_NI(1) VSM_NI(2) VSM_NI(3) VSM_NI(4) VSM_NI(5) VSM_NI(6) VSM_NI(7) VSM_NI(8) VSM_NI(9) VSM_NI(10) VSM_NI(11) VSM_NI(12)
    VSM_NI(13) VSM_NI(14) VSM_NI(15) VSM_NI(16)
    default: break;
  }
#undef VSM_NI
  return VSM_ERR_UNSUPPORTED;
}
namespace {

template <int RT, typename ST>
static void launch_pre(bool mix, const quad<ST>& q, int S, int nsub, int n, int ndoubl, const ST* dtau, const ST* varpi,
                       const ST* tau_sum, const ST* F0, const npre_args<ST>& a, double* pre, hipStream_t st) {
  const dim3 grid(S, nsub), block(64 * RT);
  if (mix)
    hipLaunchKernelGGL((k_elemental_native<RT, true, ST>), grid, block, 0, st, q, n, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  else
    hipLaunchKernelGGL((k_elemental_native<RT, false, ST>), grid, block, 0, st, q, n, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
}

static ngroup_map group_map(const vsm_run* run, int im) {
  ngroup_map gm;
  gm.ngroups = 0;
  for (int a = 0; a < 4; ++a) {
    gm.grp_of[a] = gm.pos_in[a] = 0;
    gm.gsz[a] = gm.rt[a] = 1;
    gm.n[a] = 0;
    gm.base[a] = nullptr;
  }
  for (const nat_sub& sb : run->subs) {
    if (sb.im != im) continue;
    const int g = gm.ngroups++;
    gm.gsz[g] = sb.gsz;
    gm.rt[g] = sb.rt;
    gm.n[g] = sb.n;
    gm.base[g] = static_cast<double*>(run->ws) + sb.comp_off;
    for (int k = 0; k < sb.gsz; ++k) {
      gm.grp_of[sb.g[k]] = g;
      gm.pos_in[sb.g[k]] = k;
    }
  }
  return gm;
}

template <typename ST>
static quad<ST> run_quad(const vsm_run* run) {
  return quad<ST>{static_cast<const ST*>(run->mu), static_cast<const ST*>(run->wt), run->N, run->ns, run->i_mu0, (ST)run->mu0};
}

template <typename ST, typename Q>
static int run_create(const Q* q, int S, int nm, const int* m, const int* coupling, void* workspace, size_t workspace_bytes,
                      vsm_run** run_out) {
  VSM_REQUIRE(q && q->mu && q->wt && run_out && m, "vsm_run_create: null argument");
  VSM_REQUIRE(S >= 0 && nm >= 1 && nm <= NSUB_MAX, "vsm_run_create: bad S = %d / nm = %d (at most %d moments per run)", S, nm,
              NSUB_MAX);
  vsm_run* run = new vsm_run;
  run->mu = q->mu;
  run->wt = q->wt;
  run->N = q->N;
  run->ns = q->n_stokes;
  run->i_mu0 = q->i_mu0;
  run->mu0 = (double)q->mu0;
  run->elem_size = (int)sizeof(ST);
  run->S = S;
  run->nm = nm;
  run->m.assign(m, m + nm);
  size_t total = 0;
  const int rc = plan_subs(sizeof(ST) == 4, q->N, q->n_stokes, nm, m, coupling, run->subs, (size_t)S, total);
  if (rc) {
    delete run;
    return rc;
  }
  if (total * sizeof(ST) > workspace_bytes || (total && !workspace) || ((size_t)workspace & 15)) {
    set_error("vsm_run_create: workspace of %zu bytes (16-byte aligned) required, got %zu", total * sizeof(ST), workspace_bytes);
    delete run;
    return VSM_ERR_INVALID_ARG;
  }
  run->ws = workspace;
  run->ws_elems = total;
  run->pure_diag.assign(run->subs.size(), 0);
  // classes: sub-problems that one launch can take (equal n, group size and U/V pattern), at most NSUB_MAX each
  for (size_t i = 0; i < run->subs.size(); ++i) {
    const nat_sub& sb = run->subs[i];
    bool placed = false;
    for (auto& cl : run->classes) {
      const nat_sub& h = run->subs[cl[0]];
      if (h.n == sb.n && h.gsz == sb.gsz && h.uvmask == sb.uvmask && (int)cl.size() < NSUB_MAX) {
        cl.push_back((int)i);
        placed = true;
        break;
      }
    }
    if (!placed) run->classes.push_back(std::vector<int>{(int)i});
  }
  *run_out = run;
  return VSM_OK;
}

template <typename ST>
static int run_layer(vsm_run* run, int ndoubl, const ST* dtau, const ST* varpi, const ST* tau_sum, const ST* F0, int ncomp,
                     const ST* const* Zpp, const ST* const* Zmp, long long z_stride, const ST* fcomp, int toa,
                     const int* layer_coupling, void* stream) {
  VSM_REQUIRE(run && dtau && varpi && tau_sum && F0 && Zpp && Zmp, "vsm_run_layer: null argument");
  VSM_REQUIRE(run->elem_size == (int)sizeof(ST), "vsm_run_layer: the run was created for %d-byte arrays", run->elem_size);
  VSM_REQUIRE(ndoubl >= 0 && ndoubl < 60 && ncomp >= 0 && ncomp <= 4 && (ncomp == 0 || fcomp), "vsm_run_layer: bad ndoubl / component mix");
  if (run->S == 0) return VSM_OK;
  hipStream_t st = as_stream(stream);
  int* status = device_status();
  if (!status) return VSM_ERR_HIP;
  const quad<ST> q = run_quad<ST>(run);
  // a sub-problem whose Stokes block the layer's phase matrices leave exactly zero takes the diagonal step
  auto trivial = [&](const nat_sub& sb) {
    if (!layer_coupling || layer_coupling[sb.im] < 0) return false;
    for (int a = 0; a < sb.gsz; ++a)
      for (int b = 0; b < sb.gsz; ++b)
        if ((layer_coupling[sb.im] >> (4 * sb.g[a] + sb.g[b])) & 1) return false;
    return true;
  };
  {   // the masks against the layer's actual phase matrices (VSM_DEVSTAT_MASK)
    nguard_args<ST> ga;
    for (int k = 0; k < NSUB_MAX; ++k) {
      const int im = k < run->nm ? k : 0;
      VSM_REQUIRE(Zpp[im] && Zmp[im], "vsm_run_layer: null Z of moment %d", im);
      ga.zp[k] = Zpp[im];
      ga.zm[k] = Zmp[im];
      ga.allowed[k] = 0;
    }
    for (const nat_sub& sb : run->subs) {
      if (trivial(sb)) continue;
      for (int a = 0; a < sb.gsz; ++a)
        for (int b = 0; b < sb.gsz; ++b) ga.allowed[sb.im] |= 1u << (4 * sb.g[a] + sb.g[b]);
    }
    const int nmat = ncomp ? ncomp : (z_stride ? run->S : 1);
    hipLaunchKernelGGL((k_coupling_guard<ST>), dim3(nmat, run->nm), dim3(256), 0, st, run->N, run->ns,
                       ncomp ? (long long)run->N * run->N : z_stride, ga, status);
    VSM_LAUNCH_CHECK("k_coupling_guard");
  }
  if constexpr (sizeof(ST) == 4) {   // Float32 model: FP32 records and arithmetic (vsm_native32.hip)
    return native32_run_layer(run, ndoubl, dtau, varpi, tau_sum, F0, ncomp, Zpp, Zmp, z_stride, fcomp, toa, layer_coupling, status, st);
  } else {
  double* const ws = static_cast<double*>(run->ws);
  // the pre-pass images of the layer: one record per (sub-problem, point)
  size_t pre_total = 0;
  for (const nat_sub& sb : run->subs)
    if (!trivial(sb)) pre_total += pre_stride_rt(sb.rt) * (size_t)run->S;
  double* pre = nullptr;
  if (pre_total) {
    pre = static_cast<double*>(scratch(pre_total * sizeof(double), 3, st));
    if (!pre) return VSM_ERR_HIP;
  }
  size_t off = 0;
  for (const auto& cl : run->classes) {
    const nat_sub& h = run->subs[cl[0]];
    // diagonal steps: the full pass (mode 0 / 1 at TOA), or only the diagonals while the composite is still diagonal (mode 2)
    std::vector<int> act, triv[3];
    for (int i : cl) {
      if (!trivial(run->subs[i])) {
        act.push_back(i);
        run->pure_diag[i] = 0;
      } else if (toa) {
        triv[1].push_back(i);
        run->pure_diag[i] = 1;
      } else {
        triv[run->pure_diag[i] ? 2 : 0].push_back(i);
      }
    }
    for (int mode = 0; mode < 3; ++mode) {
      if (triv[mode].empty()) continue;
      ndiag_args da;
      const int nt = (int)triv[mode].size();
      for (int k = 0; k < NSUB_MAX; ++k) {
        const nat_sub& sb = run->subs[triv[mode][k < nt ? k : 0]];
        da.comp[k] = ws + sb.comp_off;
        da.gsz[k] = sb.gsz;
        da.g0[k] = sb.g[0] | (sb.g[1] << 4) | (sb.g[2] << 8) | (sb.g[3] << 12);
      }
      hipLaunchKernelGGL((k_native_diag_layer<ST>), dim3(run->S, nt), dim3(256), 0, st, q, h.rt, h.n, ldexp(1.0, ndoubl), mode, dtau,
                         da);
      VSM_LAUNCH_CHECK("k_native_diag_layer");
    }
    const int nsub = (int)act.size();
    if (!nsub) continue;
    npre_args<ST> pa;
    nlayer_comps lc;
    for (int k = 0; k < NSUB_MAX; ++k) {
      const nat_sub& sb = run->subs[act[k < nsub ? k : 0]];
      pa.s[k].m = sb.m;
      pa.s[k].gsz = sb.gsz;
      for (int a = 0; a < 4; ++a) pa.s[k].g[a] = sb.g[a];
      VSM_REQUIRE(Zpp[sb.im] && Zmp[sb.im], "vsm_run_layer: null Z of moment %d", sb.im);
      pa.s[k].z = zsrc<ST>{Zpp[sb.im], Zmp[sb.im], ncomp ? 0 : z_stride, ncomp, fcomp};
      lc.c[k] = ws + sb.comp_off;
    }
    double* pre_cl = pre + off;
    off += pre_stride_rt(h.rt) * (size_t)run->S * nsub;
    switch (h.rt) {
      case 1: launch_pre<1, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 2: launch_pre<2, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 3: launch_pre<3, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 4: launch_pre<4, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 5: launch_pre<5, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      default: launch_pre<6, ST>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
    }
    VSM_LAUNCH_CHECK("k_elemental_native");
    const int rc = launch_layer_native(h.ks, run->S, nsub, h.n, h.uvmask, h.gsz, ndoubl, toa, pre_cl, lc, status, st);
    if (rc) return rc;
  }
  return VSM_OK;
  }
}

template <typename ST, typename CS, bool IMPORT>
static int run_convert(vsm_run* run, const CS* comps, void* stream) {
  VSM_REQUIRE(run && comps, "vsm_run_export / import: null argument");
  VSM_REQUIRE(run->elem_size == (int)sizeof(ST), "vsm_run_export / import: the run was created for %d-byte arrays", run->elem_size);
  if (run->S == 0) return VSM_OK;
  if (IMPORT) run->pure_diag.assign(run->subs.size(), 0);
  for (int im = 0; im < run->nm; ++im) {
    const CS& c = comps[im];
    VSM_REQUIRE(c.R_mp && c.R_pm && c.T_pp && c.T_mm && c.J0_p && c.J0_m, "vsm_run_export / import: null composite array (moment %d)", im);
    const composite<ST> cc{c.R_mp, c.R_pm, c.T_pp, c.T_mm, c.J0_p, c.J0_m};
    if constexpr (sizeof(ST) == 4) {
      const int rc = native32_convert(run, im, cc, IMPORT, as_stream(stream));
      if (rc) return rc;
      continue;
    }
    const ngroup_map gm = group_map(run, im);
    if (IMPORT)
      hipLaunchKernelGGL((k_native_import<ST>), dim3(run->S), dim3(256), 0, as_stream(stream), run->N, run->ns, gm, cc);
    else
      hipLaunchKernelGGL((k_native_export<ST>), dim3(run->S), dim3(256), 0, as_stream(stream), run->N, run->ns, gm, cc);
    VSM_LAUNCH_CHECK("k_native_export / import");
  }
  return VSM_OK;
}

template <typename ST>
static int stokes_coupling(int N, int n_stokes, int nblocks, const ST* Zpp, const ST* Zmp, int* mask_d, void* stream) {
  VSM_REQUIRE(N > 0 && n_stokes >= 1 && n_stokes <= 4 && N % n_stokes == 0 && nblocks >= 1 && nblocks <= 65535 && Zpp && Zmp &&
                  mask_d, "vsm_stokes_coupling: bad argument");
  hipStream_t st = as_stream(stream);
  VSM_HIP(hipMemsetAsync(mask_d, 0, sizeof(int) * nblocks, st));
  const long long tot = (long long)N * N;
  const int blocks = (int)((tot + 255) / 256 < 16 ? (tot + 255) / 256 : 16);
  hipLaunchKernelGGL((k_stokes_coupling<ST>), dim3(blocks, nblocks), dim3(256), 0, st, N, n_stokes, Zpp, Zmp, mask_d);
  VSM_LAUNCH_CHECK("k_stokes_coupling");
  return VSM_OK;
}

}  // namespace
}  // namespace vsm

using namespace vsm;

extern "C" {

static int run_supported(int N, int n_stokes, int coupling, int max_rows) {
  if (N <= 0 || n_stokes < 1 || n_stokes > 4 || N % n_stokes) return 0;
  int grp_of[4], groups[4][4], gsz[4];
  const int ng = stokes_groups(n_stokes, coupling, grp_of, groups, gsz);
  for (int g = 0; g < ng; ++g)
    if ((N / n_stokes) * gsz[g] > max_rows) return 0;
  return 1;
}
int vsm_run_supported(int N, int n_stokes, int coupling) { return run_supported(N, n_stokes, coupling, NATIVE_MAX_ROWS); }
int vsm_run_supported_f32(int N, int n_stokes, int coupling) { return run_supported(N, n_stokes, coupling, NATIVE32_MAX_ROWS); }

size_t vsm_run_workspace_bytes(int N, int n_stokes, int S, int nm, const int* coupling) {
  std::vector<nat_sub> subs;
  size_t total = 0;
  if (S < 0 || nm < 0 || plan_subs(false, N, n_stokes, nm, nullptr, coupling, subs, (size_t)S, total)) return 0;
  return total * sizeof(double);
}
size_t vsm_run_workspace_bytes_f32(int N, int n_stokes, int S, int nm, const int* coupling) {
  std::vector<nat_sub> subs;
  size_t total = 0;
  if (S < 0 || nm < 0 || plan_subs(true, N, n_stokes, nm, nullptr, coupling, subs, (size_t)S, total)) return 0;
  return total * sizeof(float);
}

int vsm_run_create_f64(const vsm_quad_f64* q, int S, int nm, const int* m, const int* coupling, void* workspace,
                       size_t workspace_bytes, vsm_run** run_out) {
  return run_create<double>(q, S, nm, m, coupling, workspace, workspace_bytes, run_out);
}
int vsm_run_create_f32(const vsm_quad_f32* q, int S, int nm, const int* m, const int* coupling, void* workspace,
                       size_t workspace_bytes, vsm_run** run_out) {
  return run_create<float>(q, S, nm, m, coupling, workspace, workspace_bytes, run_out);
}

int vsm_run_destroy(vsm_run* run) {
  delete run;
  return VSM_OK;
}

int vsm_run_layer_f64(vsm_run* run, int ndoubl, const double* dtau, const double* varpi, const double* tau_sum, const double* F0,
                      int ncomp, const double* const* Zpp, const double* const* Zmp, long long z_stride, const double* fcomp,
                      int toa, const int* layer_coupling, void* stream) {
  return run_layer<double>(run, ndoubl, dtau, varpi, tau_sum, F0, ncomp, Zpp, Zmp, z_stride, fcomp, toa, layer_coupling, stream);
}
int vsm_run_layer_f32(vsm_run* run, int ndoubl, const float* dtau, const float* varpi, const float* tau_sum, const float* F0,
                      int ncomp, const float* const* Zpp, const float* const* Zmp, long long z_stride, const float* fcomp,
                      int toa, const int* layer_coupling, void* stream) {
  return run_layer<float>(run, ndoubl, dtau, varpi, tau_sum, F0, ncomp, Zpp, Zmp, z_stride, fcomp, toa, layer_coupling, stream);
}

int vsm_run_export_f64(vsm_run* run, const vsm_composite_f64* comps, void* stream) {
  return run_convert<double, vsm_composite_f64, false>(run, comps, stream);
}
int vsm_run_export_f32(vsm_run* run, const vsm_composite_f32* comps, void* stream) {
  return run_convert<float, vsm_composite_f32, false>(run, comps, stream);
}
int vsm_run_import_f64(vsm_run* run, const vsm_composite_f64* comps, void* stream) {
  return run_convert<double, vsm_composite_f64, true>(run, comps, stream);
}
int vsm_run_import_f32(vsm_run* run, const vsm_composite_f32* comps, void* stream) {
  return run_convert<float, vsm_composite_f32, true>(run, comps, stream);
}

int vsm_stokes_coupling_f64(int N, int n_stokes, int nblocks, const double* Zpp, const double* Zmp, int* mask_d, void* stream) {
  return stokes_coupling<double>(N, n_stokes, nblocks, Zpp, Zmp, mask_d, stream);
}
int vsm_stokes_coupling_f32(int N, int n_stokes, int nblocks, const float* Zpp, const float* Zmp, int* mask_d, void* stream) {
  return stokes_coupling<float>(N, n_stokes, nblocks, Zpp, Zmp, mask_d, stream);
}

}  // extern "C"

#endif  // VSM_NATIVE_KS
