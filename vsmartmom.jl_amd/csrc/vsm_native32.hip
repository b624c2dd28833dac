// FP32 native-layout layer kernels (Float32 models, sub-problems of n <= 96 rows): rt_kernel!'s scattering branch
// (rt_kernel.jl:175-250: elemental! -> doubling! -> interaction!(::ScatteringInterface_11) | TOA copy) in the model's own float
// type -- the reference runs doubling.jl:38-131 / interaction.jl:207-266 in FT -- on v_mfma_f32_16x16x4_f32, with the
// CompositeLayer kept in FP32 strip records between the layer steps of a run (vsm_run_*_f32).  The algebra, the phases and the
// barriers are those of vsm_native.hip (doubling step: rt_helpers.jl:102-166 as [E | W] = r [r | t]; interaction with ONE inverse
// and at most six live strips); what differs is in vsm_native32_dev.h: the permuted A-form that makes the FP32 accumulator layout
// coincide with the FP64 one, 16-byte units of four rows, two spectral points per workgroup from five row tiles on.
//   RT = 1..4 (n <= 64):  RT waves per workgroup, four waves per SIMD (a strip is 4 RT registers)
//   RT = 5, 6 (n <= 96):  two points per workgroup of 2 RT waves, three waves per SIMD (168 registers) -- C4's dense moments (N = 96)
//   RT = 7, 8 (n <= 128): one point per workgroup of 7 / 8 waves, two waves per SIMD (256 registers: six live strips of 32)
#include <stdlib.h>

#include <vector>

#include "vsm_native32_dev.h"
#include "vsm_native_run.h"

#ifndef VSM_N32_PARK_RT
#define VSM_N32_PARK_RT 5
#endif

namespace vsm {

struct n32layer_comps {
  float* c[NSUB_MAX];
};

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// elemental (from the pre-pass images) + ndoubl x doubling step + apply_D
// ---------------------------------------------------------------------------------------------------------------------------
// On return (all waves past a barrier): r_s = strip of the final r-+ (row signs of apply_D applied), t_s = strip of t++ (rider
// columns cleared), sm.vec[0] = j0+, sm.vec[1] = j0- (final sign), sm.usg filled.
template <int RT, int KS>
__device__ __forceinline__ void n32ed_body(n32smem<RT, (4 * KS + 2 > 16 * RT)>& sm, n32pos<RT>& p, int n, int gsz, unsigned uvmask,
                                           int ndoubl, const float* __restrict__ img, n32strip<RT>& r_s, n32strip<RT>& t_s,
                                           int* status) {
  using G = n32geo<RT>;
  constexpr unsigned dP = 0, dQ = G::AF * 4;
  float* jp = sm.vec[0];
  float* jm = sm.vec[1];
  float* rsg = sm.vec[3];   // row sign of apply_D
  const int tid = p.tid;
  constexpr int c1 = 4 * KS, c2 = 4 * KS + 1;   // spare columns (>= n, never read as k) carry j0+ / j1- through a step
  constexpr bool RID = c2 < G::NP;              // KS = 4 RT: no spare column, the source vectors by VALU mat-vecs
  static_assert(RID || KS == 4 * RT, "no spare columns for the source vectors");
  const bool own_wave = RID && p.wave == (c1 >> 4);
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2), laneAB = laneA || laneB;
  n32copy_image<RT>(sm.P, img, p);
  n32copy_image<RT>(sm.Q, img + G::AF, p);
  if (tid < G::NP) {
    jp[tid] = img[2 * G::AF + tid];
    jm[tid] = img[2 * G::AF + G::NP + tid];
    const bool uv = (uvmask >> (tid % gsz)) & 1u;
    sm.usg[tid] = uv ? -1.0f : 1.0f;
    rsg[tid] = (ndoubl >= 1 && uv) ? -1.0f : 1.0f;
  }
  const float expk0 = img[2 * G::AF + 2 * G::NP];
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA writes have landed in LDS
  __syncthreads();
  n32load(r_s, dP, p);
  n32load(t_s, dQ, p);
  // ---- doubling (rt_helpers.jl:102-166) -----------------------------------------------------------------------------------
  //   [E | W]   = r [r | t]                 A = P = [r]
  //   G         = (I - E)^-1                Horner series on [E] in P
  //   tt        = t G                       A = Q = [t]
  //   [r' | t'] = [r | 0] + tt [W | t]      A = P = [tt]
  // Sources (rt_helpers.jl:128-134: j0- += tt (j1- + r j0+), j0+ = j1+ + tt (j0+ + r j1-)) ride in the spare columns:
  //   t_s[c1] = j0+, t_s[c2] = j1- = j0- expk  ->  W[c1] = r j0+, W[c2] = r j1-
  //   r_s[c1] = j0-, r_s[c2] = j0+             ->  W[c1] += r_s[c1] expk, W[c2] += r_s[c2], r_s[c2] *= expk: lane-local;
  //                                                r'[c1] = j0- + tt (j1- + r j0+), r'[c2] = j1+ + tt (j0+ + r j1-)
  float expk = expk0;
  int slot = 0;
  const n32inv_ctx cx{dP, sm.P, sm.gjs(), status};
  for (int it = 0; it < ndoubl; ++it) {
    n32strip<RT> W, tt;
    const float fW = laneA ? expk : (laneB ? 1.0f : 0.0f), fR = laneB ? expk : 1.0f;
    if constexpr (!RID) {   // r j0+ , r j1-   (P = [r])
      n32mv_rows<RT, KS>(dP, jp, 1.0f, sm.mv[0], p);
      n32mv_rows<RT, KS>(dP, jm, expk, sm.mv[1], p);
    }
    {
      n32strip<RT> Gs;
      {
        n32strip<RT> E;
        n32mm2<RT, KS, true, true>(E, W, dP, r_s, t_s, p);
        if (own_wave) {
#pragma unroll
          for (int ta = 0; ta < RT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              W.v[ta][r] = fmaf(r_s.v[ta][r], fW, W.v[ta][r]);
              r_s.v[ta][r] *= fR;
            }
        }
        float nrm_o;
        const float nrm = n32norm(E, n, sm, slot, p, nrm_o);   // (its barrier: [r] is free)
        if constexpr (!RID) {   // u1 = j1- + r j0+ , u2 = j0+ + r j1-   (every wave is past the norm reduction's barrier)
          for (int i = tid; i < G::NP; i += G::NTP) {
            sm.vec[4][i] = jm[i] * expk + sm.mv[0][i];
            sm.vec[5][i] = jp[i] + sm.mv[1][i];
          }
        }
        n32invert_pair<RT, KS>(n32inv_order(nrm, status, p.live && tid == 0), n32series_order(nrm_o), E, Gs, n, cx, p);
      }
      n32mm<RT, KS, true>(tt, dQ, Gs, p);    // tt = t G
    }
    n32load(t_s, dQ, p);                     // t's strip (with its riders) is not kept in registers across the inverse
    __syncthreads();                         // P ([E]) and Q ([t]) no longer read
    n32store(dP, tt, p);
    __syncthreads();
    if constexpr (!RID) {   // tt u1 , tt u2
      n32mv_rows<RT, KS>(dP, sm.vec[4], 1.0f, sm.mv[0], p);
      n32mv_rows<RT, KS>(dP, sm.vec[5], 1.0f, sm.mv[1], p);
    }
    {
      n32strip<RT> tn;
      n32mm2<RT, KS, false, true>(r_s, tn, dP, W, t_s, p);   // r' = r + tt W (riders: the new j0-, j0+) ; t' = tt t
      t_s = tn;
    }
    const float expk_step = expk;
    expk = expk * expk;
    if (own_wave) {
      const float ft = laneB ? expk : 1.0f;   // t_s[c1] = j0+', t_s[c2] = j1-' = j0-' expk'
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float u = n32dpp_swap1(r_s.v[ta][r]) * ft;
          t_s.v[ta][r] = laneAB ? u : t_s.v[ta][r];
        }
    }
    if (it + 1 < ndoubl || !RID) __syncthreads();   // everybody finished reading P ([tt])
    if constexpr (!RID) {   // j0- += tt u1 ; j0+ = j0+ expk + tt u2   (visible after the barrier below / after the loop)
      for (int i = tid; i < G::NP; i += G::NTP) {
        jm[i] += sm.mv[0][i];
        jp[i] = jp[i] * expk_step + sm.mv[1][i];
      }
    }
    if (it + 1 < ndoubl) {
      n32store(dP, r_s, p);
      n32store(dQ, t_s, p);
      __syncthreads();
    }
  }
  if constexpr (RID) {
    if (ndoubl > 0 && own_wave) {   // the riders go back to the LDS vectors; the strips leave the loop with clean padding
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          float* dpv = laneA ? jp : sm.vec[7];   // (the other lanes write to a dummy vector)
          float* dmv = laneA ? jm : sm.vec[7];
          dpv[row] = t_s.v[ta][r];          // lane A: j0+
          dmv[row] = r_s.v[ta][r];          // lane A: j0-
          t_s.v[ta][r] = laneAB ? 0.0f : t_s.v[ta][r];
          r_s.v[ta][r] = laneAB ? 0.0f : r_s.v[ta][r];
        }
    }
  }
  __syncthreads();
  // ---- apply_D (doubling.jl:178-252): r-+ = D r*, j0- = D j0-* ------------------------------------------------------------
  if (ndoubl >= 1) {
#pragma unroll
    for (int ta = 0; ta < RT; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] *= rsg[p.row(ta, r)];
    for (int i = tid; i < G::NP; i += G::NTP) jm[i] *= rsg[i];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266) on the run's native records.  With
// G2 = (I - R+- r-+)^-1 and the push-through identities (I - r R)^-1 r = r G2, (I - r R)^-1 = I + r G2 R:
//   [E2 | Z] = R+- [r-+ | t--]          A = P = [R+-]   z  = J0+ + R+- j0-   (j0- rides in a spare column of r-+)
//   G2       = (I - E2)^-1              A = P = [E2]    BEFORE the products of [T--]: V is not live across the series
//   [S  | V] = T-- [r-+ | t--]          A = Q = [T--]   vs = T-- j0-
//   T21 = t++ G2 ;  Y = S G2            A = P = [t++], Q = [S]
//   [R+- | T++] = [r+- | 0] + T21 [Z | T++]      A = P = [T21]   J0+ = j0+ + T21 z   (z rides in a spare column of T++)
//   [R-+ | T--] = [R-+ | V] + Y [T++ | Z]        A = Q = [Y]     J0- = J0- + vs + Y z
// On entry: r_s / t_s = strips of the added layer's r-+ / t++ (columns >= n zero), sm.vec[0] / vec[1] = its j0+ / j0-, all waves
// past a barrier, P and Q free.  r+- = D r-+ D, t-- = D t++ D (added layers from doubling).
// ---------------------------------------------------------------------------------------------------------------------------
template <int RT, int KS>
__device__ __forceinline__ void n32ia_body(n32smem<RT, (4 * KS + 2 > 16 * RT)>& sm, n32pos<RT>& p, int n, float* __restrict__ comp,
                                           n32strip<RT>& r_s, n32strip<RT>& t_s, int* status) {
  using G = n32geo<RT>;
  constexpr unsigned dP = 0, dQ = G::AF * 4;
  float* vjp = sm.vec[0];
  float* vjm = sm.vec[1];
  float* vJp = sm.vec[2];
  float* vJm = sm.vec[3];
  float* vs = sm.vec[4];
  float* vz = sm.vec[5];
  const int tid = p.tid;
  float* R_mp = comp + N32_RMP * G::AF;
  float* R_pm = comp + N32_RPM * G::AF;
  float* T_pp = comp + N32_TPP * G::AF;
  float* T_mm = comp + N32_TMM * G::AF;
  float* J0_p = comp + 4 * G::AF;
  float* J0_m = J0_p + G::NP;
  constexpr int c1 = 4 * KS, c2 = 4 * KS + 1;
  constexpr bool RID = c2 < G::NP;
  // Five and six row tiles (168 registers, strips of 20 / 24): at most FIVE live strips.  Z = R+- t-- is not needed between its
  // product and the closing pairs: it waits in the R+- record of the composite (consumed into P at the entry, overwritten by the
  // new R+- at the end; the lane reads back exactly the addresses it wrote), and R-+ is requested after R+- / T++ have left.
  constexpr bool PARK = VSM_N32_PARK_RT > 0 && RT >= VSM_N32_PARK_RT;
  const bool own_wave = RID && p.wave == (c1 >> 4);
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2);
  int slot = 0;
  const n32dpar<RT> dp(sm.usg, p);
  const n32inv_ctx cx{dP, sm.P, sm.gjs(), status};
  // ---- stage: composite vectors, [R+-] -> P, [T--] -> Q ---------------------------------------------------------------------
  for (int i = tid; i < G::NP; i += G::NTP) {
    vJp[i] = J0_p[i];
    vJm[i] = J0_m[i];
  }
  n32strip<RT> Z, Gs;
  {
    n32strip<RT> A1, A2;
    n32ld_native(A1, R_pm, p);
    n32ld_native(A2, T_mm, p);
    n32store(dP, A1, p);
    n32store(dQ, A2, p);
  }
  if constexpr (RID) {
    if (own_wave) {  // j0- rides in the spare column c2 of r-+:  E2[:, c2] = R+- j0-, S[:, c2] = T-- j0-
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) r_s.v[ta][r] = laneB ? vjm[p.row(ta, r)] : r_s.v[ta][r];
    }
  }
  n32dsym(t_s, t_s, dp);   // t-- = D t++ D in place (undone below: D is an involution)
  __syncthreads();                                                                                       // (a)
  if constexpr (!RID) {   // R+- j0- , T-- j0-
    n32mv_rows<RT, KS>(dP, vjm, 1.0f, sm.mv[0], p);
    n32mv_rows<RT, KS>(dQ, vjm, 1.0f, sm.mv[1], p);
  }
  {
    n32strip<RT> E;
    n32mm2<RT, KS, true, true>(E, Z, dP, r_s, t_s, p);
    if constexpr (RID) {
      if (own_wave) {
        float* zd = laneB ? vz : sm.vec[7];   // (the other lanes write to a dummy vector)
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = p.row(ta, r);
            zd[row] = vJp[row] + E.v[ta][r];
          }
      }
    }
    if constexpr (PARK) n32st_native(R_pm, Z, p, p.live);   // (its old contents are in P; the new R+- overwrites it at the end)
    float nrm_o;
    const float nrm = n32norm(E, n, sm, slot, p, nrm_o);   // (b): every wave is done reading [R+-]
    if constexpr (!RID) {   // z = J0+ + R+- j0- ; vs = T-- j0-
      for (int i = tid; i < G::NP; i += G::NTP) {
        vz[i] = vJp[i] + sm.mv[0][i];
        vs[i] = sm.mv[1][i];
      }
    }
    n32invert_pair<RT, KS>(n32inv_order(nrm, status, p.live && tid == 0), n32series_order(nrm_o), E, Gs, n, cx, p);   // [E2] -> P, series
  }
  n32strip<RT> V;
  {
    n32strip<RT> S;
    n32mm2<RT, KS, true, true>(S, V, dQ, r_s, t_s, p);   // (Q = [T--] has not been touched since (a))
    if constexpr (RID) {
      if (own_wave) {
        float* sd = laneB ? vs : sm.vec[7];
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) sd[p.row(ta, r)] = S.v[ta][r];
      }
    }
    // r_s <- r+- (the accumulator of the R+- update; its rider column is never used), t_s <- t++
    n32dsym(r_s, r_s, dp);
    n32dsym(t_s, t_s, dp);
    __syncthreads();                      // (d): [E2] (series) and [T--] no longer read
    n32store(dQ, S, p);                   // [S]   -> Q
    n32store(dP, t_s, p);                 // [t++] -> P
  }
  __syncthreads();                        // (e)
  {
    n32strip<RT> X, Y;
    n32mm<RT, KS, true>(X, dP, Gs, p);    // T21 = t++ G2
    n32mm<RT, KS, true>(Y, dQ, Gs, p);    // Y = S G2 = T01 r-+
    __syncthreads();                      // (f): [t++], [S] no longer read
    n32store(dP, X, p);                   // [T21] -> P
    n32store(dQ, Y, p);                   // [Y]   -> Q
  }
  __builtin_amdgcn_sched_barrier(0);      // (the composite strips are requested once X and Y are dead, not above their stores)
  n32strip<RT> Tpp, Rmp;
  n32ld_native(Tpp, T_pp, p);
  if constexpr (PARK) n32ld_native(Z, R_pm, p); else n32ld_native(Rmp, R_mp, p);
  __syncthreads();                        // (g)
  if constexpr (!RID) {   // T21 z , Y z
    n32mv_rows<RT, KS>(dP, vz, 1.0f, sm.mv[0], p);
    n32mv_rows<RT, KS>(dQ, vz, 1.0f, sm.mv[1], p);
  }
  if constexpr (RID) {
    if (own_wave) {  // z rides in the spare column c1 of T++:  (T21 T++)[:, c1] = T21 z, (Y T++)[:, c1] = Y z
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tpp.v[ta][r] = laneA ? vz[p.row(ta, r)] : Tpp.v[ta][r];
    }
  }
  {
    n32strip<RT> acc;
    n32mm2<RT, KS, false, true>(r_s, acc, dP, Z, Tpp, p);   // R+- = r+- + T21 Z ; T++ = T21 T++
    n32st_native(R_pm, r_s, p, p.live);
    n32st_native(T_pp, acc, p, p.live);
    if (laneA && p.live) {
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
    n32ld_native(Rmp, R_mp, p);
  }
  if constexpr (RID) {
    if (own_wave) {   // (the native record keeps the rider column of the previous layer step)
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) Rmp.v[ta][r] = laneA ? 0.0f : Rmp.v[ta][r];
    }
  }
  n32mm2<RT, KS>(Rmp, V, dQ, Tpp, Z, p);       // R-+ = R-+ + Y T++ ; T-- = V + Y Z
  n32st_native(R_mp, Rmp, p, p.live);
  n32st_native(T_mm, V, p, p.live);
  if (laneA && p.live) {
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
    if (p.live) {
      for (int i = tid; i < G::NP; i += G::NTP) {
        J0_p[i] = vjp[i] + sm.mv[0][i];
        J0_m[i] = vJm[i] + vs[i] + sm.mv[1][i];
      }
    }
  }
}

template <int RT, int KS>
__global__ __launch_bounds__(n32geo<RT>::NT, n32geo<RT>::WPS) void k_layer_native32(int S, int n, int gsz, unsigned uvmask, int ndoubl,
                                                                                    int toa, const float* __restrict__ pre,
                                                                                    n32layer_comps a, int* status) {
  using G = n32geo<RT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  using SM = n32smem<RT, (4 * KS + 2 > 16 * RT)>;
  static_assert(sizeof(SM) % 16 == 0, "the second point's LDS block starts on a 16-byte boundary");
  const int pt = G::PP == 1 ? 0 : (int)(threadIdx.x >> 6) / RT;
  SM& sm = reinterpret_cast<SM*>(smem_raw)[pt];
  n32pos<RT> p(sm.P);
  int s = blockIdx.x * G::PP + pt;
  const int isub = blockIdx.y;
  p.live = s < S;
  s = min(s, S - 1);   // (the filler point of an odd batch walks the last point again, for the barriers; it stores nothing)
  const float* img = pre + ((long long)isub * S + s) * G::PRE_STRIDE;
  float* comp = a.c[isub] + (long long)s * G::COMP_STRIDE;
  n32strip<RT> r_s, t_s;
  n32ed_body<RT, KS>(sm, p, n, gsz, uvmask, ndoubl, img, r_s, t_s, status);
  if (toa) {   // copy_added_to_composite! (rt_helpers.jl:188-200)
    const n32dpar<RT> dp(sm.usg, p);
    n32st_native(comp + N32_RMP * G::AF, r_s, p, p.live);
    n32st_native(comp + N32_TPP * G::AF, t_s, p, p.live);
    n32strip<RT> d;
    n32dsym(d, r_s, dp);
    n32st_native(comp + N32_RPM * G::AF, d, p, p.live);
    n32dsym(d, t_s, dp);
    n32st_native(comp + N32_TMM * G::AF, d, p, p.live);
    if (p.live) {
      for (int i = p.tid; i < G::NP; i += G::NTP) {
        comp[4 * G::AF + i] = sm.vec[0][i];
        comp[4 * G::AF + G::NP + i] = sm.vec[1][i];
      }
    }
    return;
  }
  n32ia_body<RT, KS>(sm, p, n, comp, r_s, t_s, status);
}

}  // namespace

// one object per k-step count (parallel build): vsm_native32_<KS>.o is this file built with -DVSM_NATIVE32_KS=<KS>
#define VSM_N32CAT2(a, b) a##b
#define VSM_N32CAT(a, b) VSM_N32CAT2(a, b)
#define VSM_NATIVE32_DECL(KS) \
  int VSM_N32CAT(launch_layer_native32_, KS)(int, int, int, unsigned, int, int, int, const float*, const n32layer_comps&, int*, hipStream_t);

constexpr int native32_rt_of_ks(int ks) { return (ks % 4 == 0 && ks >= 4) ? ks / 4 : (4 * ks + 2 + 15) / 16; }

#ifdef VSM_NATIVE32_KS
VSM_NATIVE32_DECL(VSM_NATIVE32_KS)
int VSM_N32CAT(launch_layer_native32_, VSM_NATIVE32_KS)(int S, int nsub, int n, unsigned uvmask, int gsz, int ndoubl, int toa,
                                                        const float* pre, const n32layer_comps& comps, int* status, hipStream_t st) {
  constexpr int KS = VSM_NATIVE32_KS;
  // KS = 4, 8, ... 24: KS / 4 row tiles without spare columns (mat-vec source path) instead of one row tile more for two columns
  constexpr int RT = native32_rt_of_ks(KS);
  static_assert(RT >= 1 && RT <= 8, "n <= 128");
  using G = n32geo<RT>;
  using SM = n32smem<RT, (4 * KS + 2 > 16 * RT)>;
  auto kern = k_layer_native32<RT, KS>;
  const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kern), sizeof(SM) * G::PP, "hipFuncSetAttribute(k_layer_native32)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(kern, dim3((S + G::PP - 1) / G::PP, nsub), dim3(G::NT), sizeof(SM) * G::PP, st, S, n, gsz, uvmask, ndoubl, toa, pre,
                     comps, status);
  VSM_LAUNCH_CHECK("k_layer_native32");
  return VSM_OK;
}
}  // namespace vsm

#else  // ---- dispatcher object: pre-pass, layout conversion, the FP32 side of the run object ---------------------------------------

VSM_NATIVE32_DECL(1)
VSM_NATIVE32_DECL(2)
VSM_NATIVE32_DECL(3)
VSM_NATIVE32_DECL(4)
VSM_NATIVE32_DECL(5)
VSM_NATIVE32_DECL(6)
VSM_NATIVE32_DECL(7)
VSM_NATIVE32_DECL(8)
VSM_NATIVE32_DECL(9)
VSM_NATIVE32_DECL(10)
VSM_NATIVE32_DECL(11)
VSM_NATIVE32_DECL(12)
VSM_NATIVE32_DECL(13)
VSM_NATIVE32_DECL(14)
VSM_NATIVE32_DECL(15)
VSM_NATIVE32_DECL(16)
VSM_NATIVE32_DECL(17)
VSM_NATIVE32_DECL(18)
VSM_NATIVE32_DECL(19)
VSM_NATIVE32_DECL(20)
VSM_NATIVE32_DECL(21)
VSM_NATIVE32_DECL(22)
VSM_NATIVE32_DECL(23)
VSM_NATIVE32_DECL(24)
VSM_NATIVE32_DECL(25)
VSM_NATIVE32_DECL(26)
VSM_NATIVE32_DECL(27)
VSM_NATIVE32_DECL(28)
VSM_NATIVE32_DECL(29)
VSM_NATIVE32_DECL(30)
VSM_NATIVE32_DECL(31)
VSM_NATIVE32_DECL(32)

namespace {

__host__ __device__ __forceinline__ int n32af_idx_rt(int rt, int row, int k) {
  const int ks = k >> 2, kk = k & 3, rl = row & 15;
  return (ks * rt + (row >> 4)) * 64 + ((kk << 4) | ((rl & 3) << 2) | ((rl >> 2) ^ (ks & 3)));
}
__host__ __device__ __forceinline__ int n32nat_idx_rt(int rt, int i, int j) {
  const int w = j >> 4, l15 = j & 15, ta = i >> 4, mm = i & 15, kq = mm & 3, r = mm >> 2;
  return ((w * rt + ta) * 64 + ((kq << 4) | l15)) * 4 + r;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Elemental pre-pass: elemental! incl. the SFI source (elemental.jl:289-392) of every (point, sub-problem) of a layer as two
// A-form images + vectors (n32geo::PRE_STRIDE), in Float32 like the reference's Float32 run (vsm_elemental.h: the formulas of the
// FP32 strip kernels' pre-pass).  Sub-row i of a group g (Stokes components g[0..gsz)) is row (i / gsz) n_stokes + g[i % gsz] of
// the full problem.  Thread = (row tile t, kqr, k' | column phase): it owns the rows 16 t + 4 rr + kqr, rr = 0..3, of the columns
// 4 ks + k', ks = phase, phase + 4, ... -- the four words of an image block that are adjacent (n32af_idx): ONE 16-byte store per
// image and column, sixteen threads (kqr, k') write 256 contiguous bytes.
// ---------------------------------------------------------------------------------------------------------------------------
struct n32sub_pre {
  int m, gsz;
  int g[4];
  zsrc<float> z;
};
struct n32pre_args {
  n32sub_pre s[NSUB_MAX];
};
template <int RT, bool MIX>
__global__ __launch_bounds__(64 * RT) void k_elemental_native32(quad<float> q, int n, int ndoubl, const float* __restrict__ dtau,
                                                                const float* __restrict__ varpi, const float* __restrict__ tau_sum,
                                                                const float* __restrict__ F0, n32pre_args a, float* __restrict__ pre) {
  using G = n32geo<RT>;
  constexpr int NP = G::NP;
  __shared__ float mus[NP], xs[NP], es[NP], ems[NP], wts[NP];
  __shared__ int frow[NP];
  __shared__ int thick_flag;
  const int s = blockIdx.x, isub = blockIdx.y, tid = threadIdx.x;
  const n32sub_pre& sp = a.s[isub];
  const int N = q.N, ns = q.n_stokes, m = sp.m, gsz = sp.gsz;
  const zsrc<float> z = sp.z;
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
  if (tid < NP) {
    const bool in = tid < n;
    const int ga = sp.g[tid % gsz];
    const int fr = in ? (tid / gsz) * ns + ga : 0;
    const float mu = in ? q.mu[fr] : 1.0f;
    const float x = d / mu;
    frow[tid] = fr;
    mus[tid] = mu;
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
    wts[tid] = in ? q.wt[fr] : 0.0f;
  }
  if (tid == 0) thick_flag = 0;
  __syncthreads();
  if (tid < n && xs[tid] >= 0.5f) thick_flag = 1;   // (benign race: every writer stores 1)
  __syncthreads();
  const bool thick = thick_flag != 0;
  float* out = pre + ((long long)isub * gridDim.x + s) * G::PRE_STRIDE;
  float* R = out;
  float* T = out + G::AF;
  const int Kend = ((n + 3) >> 2) << 2;
  const int c1 = Kend, c2 = Kend + 1;
  const bool riders_in = ndoubl > 0 && c2 < NP;   // (no spare column: the layer kernel uses mat-vecs)
  // thread -> (kqr, k', row tile, column phase)
  const int kqr = tid & 3, kk = (tid >> 2) & 3, g16 = tid >> 4, t = g16 % RT, ph = g16 / RT;
  float mi[4], xi[4], ai[4], ei[4], sg[4];
  int fi[4];
  bool rin[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int i = 16 * t + 4 * rr + kqr;
    mi[rr] = mus[i];
    xi[rr] = xs[i];
    ai[rr] = ems[i];
    ei[rr] = es[i];
    rin[rr] = i < n;
    fi[rr] = frow[min(i, n - 1)];
    const bool uvi = i < n && sp.g[i % gsz] >= 2;
    sg[rr] = (ndoubl >= 1 && uvi) ? -1.0f : 1.0f;   // starred R* = D R (elemental.jl:403-422)
  }
  for (int ks = ph; ks < 4 * RT; ks += 4) {
    const int j = 4 * ks + kk;
    if (riders_in && (j == c1 || j == c2)) continue;   // written below
    f4_t rv = {0.f, 0.f, 0.f, 0.f}, tv = {0.f, 0.f, 0.f, 0.f};
    if (j < n) {
      const float wt = wts[j];
      const float wct = (m == 0) ? wt / 2.0f : wt / 4.0f;
      const bool active = wct > num<float>::eps();   // eps(FT) of the model's float type (elemental.jl:296)
      const float mj = mus[j], xj = xs[j], aj = ems[j], ej = es[j];
      const long long zc = (long long)N * frow[j];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = 16 * t + 4 * rr + kqr;
        float rres, tres;
        elemental_pair<float>(w, zget(Zp, fi[rr] + zc), zget(Zm, fi[rr] + zc), mi[rr], xi[rr], ai[rr], ei[rr], mj, xj, aj, ej, wct, i == j,
                              thick, rres, tres);
        const float rvv = rin[rr] ? (active ? rres * sg[rr] : 0.0f) : 0.0f;
        const float tvv = rin[rr] ? (active ? tres : ((i == j) ? ei[rr] : 0.0f)) : 0.0f;
        rv[rr ^ (ks & 3)] = rvv;
        tv[rr ^ (ks & 3)] = tvv;
      }
    }
    const int o = (ks * RT + t) * 64 + ((kk << 4) | (kqr << 2));
    *reinterpret_cast<f4_t*>(R + o) = rv;
    *reinterpret_cast<f4_t*>(T + o) = tv;
  }
  if (tid < NP) {   // SFI source of the solar beam: the same formulas with the solar column (see elemental_pair)
    const int i = tid, ic = min(i, n - 1);
    const int fri = frow[ic];
    const int i0 = ns * q.i_mu0;
    const float mu0n = q.mu[i0];
    const float x0 = d / mu0n, e0 = exp(-x0), a0 = expm1(-x0);
    float zp = 0.0f, zm = 0.0f;
    for (int qq = 0; qq < ns; ++qq) {
      const long long zo = fri + (long long)N * (i0 + qq);
      const float f = F0[qq + (long long)ns * s];
      zp += zget(Zp, zo) * f;
      zm += zget(Zm, zo) * f;
    }
    const bool uvi = i < n && sp.g[i % gsz] >= 2;
    const float sgi = (ndoubl >= 1 && uvi) ? -1.0f : 1.0f;
    float rr, tt;
    // (the diagonal case mu_i == mu_0 of the source never takes the i == j form: elemental.jl:370-382)
    elemental_pair<float>(w, zp, zm, mus[i], xs[i], ems[i], es[i], mu0n, x0, a0, e0, (m == 0) ? 0.5f : 0.25f, false, thick || x0 >= 0.5f,
                          rr, tt);
    const float att = exp(-tau_sum[s] / mu0n);
    const float vp = (i < n) ? tt * att : 0.0f;
    const float vm = (i < n) ? rr * att * sgi : 0.0f;
    const float expk0 = exp(-d / q.mu0);
    out[2 * G::AF + i] = vp;
    out[2 * G::AF + NP + i] = vm;
    out[2 * G::AF + 2 * NP + i] = expk0;
    if (riders_in) {   // t[:, c1] = j0+, t[:, c2] = j1- = j0- expk ;  r[:, c1] = j0-, r[:, c2] = j0+
      T[n32af_idx<RT>(i, c1)] = vp;
      T[n32af_idx<RT>(i, c2)] = vm * expk0;
      R[n32af_idx<RT>(i, c1)] = vm;
      R[n32af_idx<RT>(i, c2)] = vp;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// native (FP32 records) <-> reference layout ([N,N,S] column-major composite arrays), one workgroup per point
// ---------------------------------------------------------------------------------------------------------------------------
struct n32group_map {
  int ngroups;
  int grp_of[4];    // group of Stokes component a
  int pos_in[4];    // its position inside the group
  int gsz[4];       // per group
  int rt[4];        // per group: row tiles
  int n[4];         // per group: rows
  float* base[4];   // per group: native composites [S] (stride COMP_STRIDE of its RT)
};
__global__ __launch_bounds__(256) void k_native32_export(int N, int ns, n32group_map gm, composite<float> c) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  float* out[4] = {c.R_mp + s * NN, c.R_pm + s * NN, c.T_pp + s * NN, c.T_mm + s * NN};
  for (int e = tid; e < N * N; e += 256) {
    const int i = e % N, j = e / N;
    const int ai = i % ns, aj = j % ns;
    const int gi = gm.grp_of[ai];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (gi == gm.grp_of[aj]) {
      const int rt = gm.rt[gi], np = 16 * rt, af = np * np;
      const float* cn = gm.base[gi] + (long long)s * (4 * af + 2 * np);
      const int is = (i / ns) * gm.gsz[gi] + gm.pos_in[ai], js = (j / ns) * gm.gsz[gi] + gm.pos_in[aj];
      const int ix = n32nat_idx_rt(rt, is, js);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = cn[k * af + ix];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k][e] = v[k];
  }
  if (tid < N) {
    const int ai = tid % ns, gi = gm.grp_of[ai];
    const int rt = gm.rt[gi], np = 16 * rt, af = np * np;
    const float* cn = gm.base[gi] + (long long)s * (4 * af + 2 * np);
    const int is = (tid / ns) * gm.gsz[gi] + gm.pos_in[ai];
    c.J0_p[(long long)s * N + tid] = cn[4 * af + is];
    c.J0_m[(long long)s * N + tid] = cn[4 * af + np + is];
  }
}
// reference layout -> native (padding zero); elements that couple different groups are dropped (they are exact zeros for a
// composite that was built under the same coupling)
__global__ __launch_bounds__(256) void k_native32_import(int N, int ns, n32group_map gm, composite<float> c) {
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  const float* in[4] = {c.R_mp + s * NN, c.R_pm + s * NN, c.T_pp + s * NN, c.T_mm + s * NN};
  for (int g = 0; g < gm.ngroups; ++g) {
    const int rt = gm.rt[g], np = 16 * rt, af = np * np, n = gm.n[g], gsz = gm.gsz[g];
    float* cn = gm.base[g] + (long long)s * (4 * af + 2 * np);
    int comp_of[4] = {0, 0, 0, 0};   // Stokes component of position k of the group
    for (int a = 0; a < ns; ++a)
      if (gm.grp_of[a] == g) comp_of[gm.pos_in[a]] = a;
    for (int e = tid; e < af; e += 256) {
      const int is = e % np, js = e / np;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (is < n && js < n) {
        const int i = (is / gsz) * ns + comp_of[is % gsz], j = (js / gsz) * ns + comp_of[js % gsz];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[k][i + (long long)N * j];
      }
      const int ix = n32nat_idx_rt(rt, is, js);
#pragma unroll
      for (int k = 0; k < 4; ++k) cn[k * af + ix] = v[k];
    }
    if (tid < np) {
      float vp = 0.f, vm = 0.f;
      if (tid < n) {
        const int i = (tid / gsz) * ns + comp_of[tid % gsz];
        vp = c.J0_p[(long long)s * N + i];
        vm = c.J0_m[(long long)s * N + i];
      }
      cn[4 * af + tid] = vp;
      cn[4 * af + np + tid] = vm;
    }
  }
}

// A layer step of a sub-problem whose block of the phase matrix is exactly zero in this layer (vsm_native.hip:
// k_native_diag_layer): T++ <- t T++, T-- <- T-- t, R+- <- t R+- t, J0+ <- t J0+ with t = e^{-dtau 2^nd / mu_i}; R-+, J0- unchanged.
// mode 0: that scaling; 1: TOA (R = 0, T = diag(t), J = 0); 2: the composite is still diagonal: only the two diagonals change.
struct n32diag_args {
  float* comp[NSUB_MAX];
  int gsz[NSUB_MAX];
  int g0[NSUB_MAX];    // the group's Stokes components, 4 bits each
};
__global__ __launch_bounds__(256) void k_native32_diag_layer(quad<float> q, int rt, int n, float scale, int mode,
                                                             const float* __restrict__ dtau, n32diag_args a) {
  __shared__ float tt[NATIVE32_MAX_ROWS];
  const int s = blockIdx.x, isub = blockIdx.y, tid = threadIdx.x;
  const int np = 16 * rt, af = np * np, gsz = a.gsz[isub];
  float* comp = a.comp[isub] + (long long)s * (4 * af + 2 * np);
  if (tid < np) {
    float t = 0.f;
    if (tid < n) {
      const int fr = (tid / gsz) * q.n_stokes + ((a.g0[isub] >> (4 * (tid % gsz))) & 15);
      t = exp(-dtau[s] * scale / q.mu[fr]);
    }
    tt[tid] = t;
    if (mode == 2 && tid < n) {
      const int ix = n32nat_idx_rt(rt, tid, tid);
      comp[N32_TPP * af + ix] *= t;
      comp[N32_TMM * af + ix] *= t;
    }
  }
  if (mode == 2) return;
  const int toa = mode == 1;
  __syncthreads();
  for (int e = tid; e < af; e += 256) {
    const int r = e & 3, lane = (e >> 2) & 63, ta = (e >> 8) % rt, w = (e >> 8) / rt;
    const int i = 16 * ta + (lane >> 4) + 4 * r, j = 16 * w + (lane & 15);
    const float ti = tt[i], tj = tt[j];
    if (toa) {
      const float d = (i == j) ? ti : 0.f;
      comp[N32_RMP * af + e] = 0.f;
      comp[N32_RPM * af + e] = 0.f;
      comp[N32_TPP * af + e] = d;
      comp[N32_TMM * af + e] = d;
    } else {
      comp[N32_RPM * af + e] *= ti * tj;
      comp[N32_TPP * af + e] *= ti;
      comp[N32_TMM * af + e] *= tj;
    }
  }
  if (tid < np) {
    if (toa) {
      comp[4 * af + tid] = 0.f;
      comp[4 * af + np + tid] = 0.f;
    } else {
      comp[4 * af + tid] *= tt[tid];
    }
  }
}

static inline size_t pre32_stride_rt(int rt) { return (size_t)2 * (16 * rt) * (16 * rt) + 3 * 16 * rt; }

static int launch_layer_native32(int ks, int S, int nsub, int n, unsigned uvmask, int gsz, int ndoubl, int toa, const float* pre,
                                 const n32layer_comps& comps, int* status, hipStream_t st) {
#define VSM_N32L(KS) \
  case KS: return VSM_N32CAT(launch_layer_native32_, KS)(S, nsub, n, uvmask, gsz, ndoubl, toa, pre, comps, status, st);
  switch (ks) {
    VSM_N32L(1) VSM_N32L(2) VSM_N32L(3) VSM_N32L(4) VSM_N32L(5) VSM_N32L(6) VSM_N32L(7) VSM_N32L(8) VSM_N32L(9) VSM_N32L(10)
    VSM_N32L(11) VSM_N32L(12) VSM_N32L(13) VSM_N32L(14) VSM_N32L(15) VSM_N32L(16) VSM_N32L(17) VSM_N32L(18) VSM_N32L(19)
    VSM_N32L(20) VSM_N32L(21) VSM_N32L(22) VSM_N32L(23) VSM_N32L(24) VSM_N32L(25) VSM_N32L(26) VSM_N32L(27) VSM_N32L(28)
    VSM_N32L(29) VSM_N32L(30) VSM_N32L(31) VSM_N32L(32)
    default: break;
  }
#undef VSM_N32L
  set_error("launch_layer_native32: no kernel for %d k-steps", ks);
  return VSM_ERR_UNSUPPORTED;
}

template <int RT>
static void launch_pre32(bool mix, const quad<float>& q, int S, int nsub, int n, int ndoubl, const float* dtau, const float* varpi,
                         const float* tau_sum, const float* F0, const n32pre_args& a, float* pre, hipStream_t st) {
  const dim3 grid(S, nsub), block(64 * RT);
  if (mix)
    hipLaunchKernelGGL((k_elemental_native32<RT, true>), grid, block, 0, st, q, n, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  else
    hipLaunchKernelGGL((k_elemental_native32<RT, false>), grid, block, 0, st, q, n, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
}

}  // namespace

int native32_rt_of(int n) { return native32_rt_of_ks((n + 3) / 4); }
size_t native32_comp_stride(int rt) { return (size_t)4 * (16 * rt) * (16 * rt) + 2 * 16 * rt; }

// rt_kernel!(::noRS) of one scattering layer for all moments of a Float32 run (the body of vsm_run_layer_f32 behind the coupling guard)
int native32_run_layer(vsm_run* run, int ndoubl, const float* dtau, const float* varpi, const float* tau_sum, const float* F0, int ncomp,
                       const float* const* Zpp, const float* const* Zmp, long long z_stride, const float* fcomp, int toa,
                       const int* layer_coupling, int* status, hipStream_t st) {
  const quad<float> q{static_cast<const float*>(run->mu), static_cast<const float*>(run->wt), run->N, run->ns, run->i_mu0, (float)run->mu0};
  float* ws = static_cast<float*>(run->ws);
  // a sub-problem whose Stokes block the layer's phase matrices leave exactly zero takes the diagonal step
  auto trivial = [&](const nat_sub& sb) {
    if (!layer_coupling || layer_coupling[sb.im] < 0) return false;
    for (int a = 0; a < sb.gsz; ++a)
      for (int b = 0; b < sb.gsz; ++b)
        if ((layer_coupling[sb.im] >> (4 * sb.g[a] + sb.g[b])) & 1) return false;
    return true;
  };
  // the pre-pass images of the layer: one record per (sub-problem, point)
  size_t pre_total = 0;
  for (const nat_sub& sb : run->subs)
    if (!trivial(sb)) pre_total += pre32_stride_rt(sb.rt) * (size_t)run->S;
  float* pre = nullptr;
  if (pre_total) {
    pre = static_cast<float*>(scratch(pre_total * sizeof(float), 3, st));
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
      n32diag_args da;
      const int nt = (int)triv[mode].size();
      for (int k = 0; k < NSUB_MAX; ++k) {
        const nat_sub& sb = run->subs[triv[mode][k < nt ? k : 0]];
        da.comp[k] = ws + sb.comp_off;
        da.gsz[k] = sb.gsz;
        da.g0[k] = sb.g[0] | (sb.g[1] << 4) | (sb.g[2] << 8) | (sb.g[3] << 12);
      }
      hipLaunchKernelGGL(k_native32_diag_layer, dim3(run->S, nt), dim3(256), 0, st, q, h.rt, h.n, (float)ldexp(1.0, ndoubl), mode, dtau,
                         da);
      VSM_LAUNCH_CHECK("k_native32_diag_layer");
    }
    const int nsub = (int)act.size();
    if (!nsub) continue;
    n32pre_args pa;
    n32layer_comps lc;
    for (int k = 0; k < NSUB_MAX; ++k) {
      const nat_sub& sb = run->subs[act[k < nsub ? k : 0]];
      pa.s[k].m = sb.m;
      pa.s[k].gsz = sb.gsz;
      for (int a = 0; a < 4; ++a) pa.s[k].g[a] = sb.g[a];
      VSM_REQUIRE(Zpp[sb.im] && Zmp[sb.im], "vsm_run_layer: null Z of moment %d", sb.im);
      pa.s[k].z = zsrc<float>{Zpp[sb.im], Zmp[sb.im], ncomp ? 0 : z_stride, ncomp, fcomp};
      lc.c[k] = ws + sb.comp_off;
    }
    float* pre_cl = pre + off;
    off += pre32_stride_rt(h.rt) * (size_t)run->S * nsub;
    switch (h.rt) {
      case 1: launch_pre32<1>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 2: launch_pre32<2>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 3: launch_pre32<3>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 4: launch_pre32<4>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 5: launch_pre32<5>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 6: launch_pre32<6>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      case 7: launch_pre32<7>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
      default: launch_pre32<8>(ncomp > 0, q, run->S, nsub, h.n, ndoubl, dtau, varpi, tau_sum, F0, pa, pre_cl, st); break;
    }
    VSM_LAUNCH_CHECK("k_elemental_native32");
    const int rc = launch_layer_native32(h.ks, run->S, nsub, h.n, h.uvmask, h.gsz, ndoubl, toa, pre_cl, lc, status, st);
    if (rc) return rc;
  }
  return VSM_OK;
}

int native32_convert(vsm_run* run, int im, const composite<float>& c, bool import, hipStream_t st) {
  n32group_map gm;
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
    gm.base[g] = static_cast<float*>(run->ws) + sb.comp_off;
    for (int k = 0; k < sb.gsz; ++k) {
      gm.grp_of[sb.g[k]] = g;
      gm.pos_in[sb.g[k]] = k;
    }
  }
  if (import)
    hipLaunchKernelGGL(k_native32_import, dim3(run->S), dim3(256), 0, st, run->N, run->ns, gm, c);
  else
    hipLaunchKernelGGL(k_native32_export, dim3(run->S), dim3(256), 0, st, run->N, run->ns, gm, c);
  VSM_LAUNCH_CHECK("k_native32_export / import");
  return VSM_OK;
}

}  // namespace vsm

#endif  // VSM_NATIVE32_KS
