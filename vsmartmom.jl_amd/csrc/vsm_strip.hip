// Column-strip kernels (FP64, 32 < N <= 60): elemental! + ndoubl x doubling step + apply_D! with HALF the LDS
// footprint of vsm_fused.hip, so that TWO workgroups share a CU and one workgroup's barriers, LDS stores and
// pipeline fill hide behind the other's MFMAs.
//
// Layout.  A workgroup (4 waves) owns one spectral point.  Wave w owns the 16-column strip [16w, 16w+16) of
// EVERY matrix and keeps it in registers in the v_mfma_f64_16x16x4 accumulator layout: 4 row tiles, element r
// of tile ta = (row 16 ta + (lane>>4) + 4 r, column 16 w + (lane&15)).  For the f64 MFMA that layout IS the
// B-operand layout of four consecutive k-steps (k = 4 r + (lane>>4)), so a product  C_s = A * B_s  takes its B
// operand straight from the registers that hold B's strip -- results of one product feed the next without
// touching LDS.  Only A operands live in LDS ("A-form": column-major 64 x 64, the swizzle of vsm_lds.h), and
// at most two of them are alive at a time:  2 x 32 KB + vectors < 80 KB per workgroup.
//
//   per doubling step (rt_helpers.jl:102-166):      A operand (LDS)      B operand (registers)
//     E   = r r                                      P = r                r_s
//     G   = (I - E)^-1   series by squaring          P = E^(2^p)          powers / partial sums
//     tt  = t G                                      Q = t                G_s
//     tmp = tt r ;  t' = tt t   (share A)            P = tt               r_s ; t_s (+ j0+, j1- in spare columns)
//     r'  = r + tmp t                                Q = tmp              t_s
#include <stdlib.h>

#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_lds.h"

namespace vsm {

#ifdef VSM_PHASE_TIMING
__device__ unsigned long long vsm_phase_cycles_strip[32];
// Stamps accumulate in (scalar) registers and are flushed once per workgroup with atomics: a global read-modify-write per stamp
// would sit in every phase it precedes.  EVERY workgroup contributes (slot 28 / 29 count the flushes): workgroup 0 alone shows
// the first round of a launch, where all workgroups of a CU start in phase.
#define VSM_STAMP_DECL                                                    \
  unsigned long long _is[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};            \
  unsigned long long _it = __builtin_readcyclecounter()
#define VSM_STAMP(i)                                                     \
  do {                                                                   \
    const unsigned long long _t = __builtin_readcyclecounter();          \
    _is[(i) - 8] += _t - _it;                                            \
    _it = _t;                                                            \
  } while (0)
#define VSM_STAMP_FLUSH()                                                                         \
  do {                                                                                            \
    if (threadIdx.x == 0) {                                                                       \
      for (int _i = 0; _i < 10; ++_i) atomicAdd(&vsm_phase_cycles_strip[8 + _i], _is[_i]);        \
      atomicAdd(&vsm_phase_cycles_strip[29], 1ull);                                               \
    }                                                                                             \
  } while (0)
#define VSM_LIFE_DECL const unsigned long long _lt0 = __builtin_readcyclecounter()
#define VSM_LIFE_FLUSH()                                                                                    \
  do {                                                                                                      \
    if (threadIdx.x == 0) {                                                                                 \
      atomicAdd(&vsm_phase_cycles_strip[27], (unsigned long long)(__builtin_readcyclecounter() - _lt0));    \
      atomicAdd(&vsm_phase_cycles_strip[26], 1ull);                                                         \
    }                                                                                                       \
  } while (0)
#define VSM_RSTAMP_DECL                                  \
  unsigned long long _rs[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
  const unsigned long long _rt0 = __builtin_readcyclecounter(), _rr0 = __builtin_amdgcn_s_memrealtime(); \
  unsigned long long _rt = _rt0
#define VSM_RSTAMP(i)                                                \
  do {                                                               \
    const unsigned long long _t = __builtin_readcyclecounter();      \
    _rs[i] += _t - _rt;                                              \
    _rt = _t;                                                        \
  } while (0)
#define VSM_RSTAMP_FLUSH()                                                                               \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      for (int _i = 0; _i < 8; ++_i) atomicAdd(&vsm_phase_cycles_strip[_i], _rs[_i]);                    \
      for (int _i = 8; _i < 12; ++_i) atomicAdd(&vsm_phase_cycles_strip[12 + _i], _rs[_i]);              \
      atomicAdd(&vsm_phase_cycles_strip[28], 1ull);                                                      \
      atomicAdd(&vsm_phase_cycles_strip[30], (unsigned long long)(__builtin_readcyclecounter() - _rt0)); \
      atomicAdd(&vsm_phase_cycles_strip[31], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - _rr0)); \
    }                                                                                                    \
  } while (0)
#else
#define VSM_STAMP_DECL
#define VSM_STAMP(i)
#define VSM_STAMP_FLUSH()
#define VSM_LIFE_DECL
#define VSM_LIFE_FLUSH()
#define VSM_RSTAMP_DECL
#define VSM_RSTAMP(i)
#define VSM_RSTAMP_FLUSH()
#endif

}  // namespace vsm
#include "vsm_strip_dev.h"
namespace vsm {
namespace {


// ---------------------------------------------------------------------------
// elemental! + doubling! + apply_D!   (strip form)
// ---------------------------------------------------------------------------
// Body shared by k_ed_strip and k_layer_strip.  On return (all waves past a barrier): r_s = strip of the final
// r-+ (row signs of apply_D applied), t_s = strip of t++, sm.vec[0] = j0+, sm.vec[1] = j0- (final sign).
// THERMAL: the `:thermal` per-source slot of the layer (rt_kernel.jl:205-232; contribute!(::PreparedThermalEmission),
// Sources/thermal_emission.jl:241-301) instead of the solar beam: j0+- = 2 pi (1 - varpi) B (1 - e^{-dtau/mu_i}) on the I rows,
// the slot's expk = 1 (doubling.jl:62-81); `F0` then points at B[S] and tau_sum is not read.
// PRE: the elemental layer comes from the pre-pass (k_elemental_img) as two A-form images + vectors at `img`.
template <int KS, bool MIX, bool THERMAL = false, bool PRE = false>   // KS k-steps per product: 4 KS >= N (columns >= N of the A-forms are zero); MIX: Z = sum_k f_k Z_k
__device__ __forceinline__ void ed_body(ssmem& sm, spos& p, const quad<double>& q, int m, int ndoubl,
                                        const double* __restrict__ dtau, const double* __restrict__ varpi,
                                        const double* __restrict__ tau_sum, const double* __restrict__ F0,
                                        const zsrc<double>& z, sstrip& r_s, sstrip& t_s, const double* __restrict__ img = nullptr) {
  VSM_RSTAMP_DECL;
  double* P = sm.P;
  double* Q = sm.Q;
  double* jp = sm.vec[0];
  double* jm = sm.vec[1];
  double* mus = sm.vec[2];
  double* rsg = sm.vec[3];   // row sign of apply_D
  double* xs = sm.vec[4];
  double* es = sm.vec[5];
  double* ems = sm.vec[6];
  const int s = blockIdx.x;
  const int N = q.N, ns = q.n_stokes;
  const int tid = threadIdx.x;
  const int Kend = ((N + 3) >> 2) << 2;
  // spare columns Kend, Kend+1 (>= N, never read as k) carry j0+ and j1- through the products of a step
  const int c1 = Kend, c2 = Kend + 1;
  constexpr bool RID = 4 * KS + 2 <= SNP;   // spare columns for the source vectors (N <= 60); else VALU mat-vecs (N = 61..64)
  static_assert(RID || PRE, "N > 60: the elemental step comes from the pre-pass");
  const bool own_wave = RID && (p.wave == (c1 >> 4));
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2), laneAB = laneA || laneB;
  double expk0;
  if constexpr (PRE) {
    // the elemental layer of the pre-pass: images -> P, Q (LDS DMA), vectors, sign tables
    copy_image_to_lds(P, img, p);
    copy_image_to_lds(Q, img + PRE_IMG, p);
    if (tid < SNP) {
      jp[tid] = img[2 * PRE_IMG + tid];
      jm[tid] = img[2 * PRE_IMG + SNP + tid];
      const bool uv = is_uv_row(tid, ns);
      sm.usg[tid] = uv ? -1.0 : 1.0;
      rsg[tid] = (ndoubl >= 1 && uv) ? -1.0 : 1.0;
    }
    expk0 = img[2 * PRE_IMG + 2 * SNP];
    VSM_RSTAMP(8);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA writes have landed in LDS
    VSM_RSTAMP(9);
  } else {
    const double d = dtau[s], w = varpi[s];
    // Z of this point: one block (z.ncomp == 0), or the mix  sum_k f_k(s) Z_k  of up to 4 scattering components
    // (types.jl:1262-1292 `+` of CoreScatteringOpticalProperties, evaluated where Z is consumed)
    const int ncomp = MIX ? z.ncomp : 0;   // (a compile-time switch: the run-time one cost 4.5 % on the plain path)
    const long long NNz = (long long)q.N * q.N;
    const double* Zp = z.Zpp + (ncomp ? 0 : (long long)s * z.zs);
    const double* Zm = z.Zmp + (ncomp ? 0 : (long long)s * z.zs);
    double fk[4] = {0.0, 0.0, 0.0, 0.0};
  #pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ncomp) fk[k] = z.fcomp[(long long)s * ncomp + k];
    auto zget = [&](const double* Z, long long zo) {
      if (ncomp == 0) return Z[zo];
      double acc = 0.0;
  #pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < ncomp) acc += fk[k] * Z[k * NNz + zo];
      return acc;
    };

    // per-row tables; `thick` = some dtau / mu_i >= 1/2 (then the differences of exponentials go through expm1 per element)
    if (tid < SNP) {
      const bool in = tid < N;
      const double mu = in ? q.mu[tid] : 1.0;
      mus[tid] = mu;
      const double x = d / mu;
      xs[tid] = x;
      es[tid] = exp(-x);
      ems[tid] = expm1(-x);
      const bool uv = is_uv_row(tid, ns);
      sm.usg[tid] = uv ? -1.0 : 1.0;
      rsg[tid] = (ndoubl >= 1 && uv) ? -1.0 : 1.0;   // starred R* = D R (elemental.jl:403-422)
      const unsigned long long any_thick = __ballot(in && x >= 0.5);   // (tid < 64: exactly wave 0)
      if (tid == 0) sm.flags[0] = any_thick != 0ull;
    }
    VSM_RSTAMP(8);
    __syncthreads();
    VSM_RSTAMP(9);
    const bool thick = __builtin_amdgcn_readfirstlane(sm.flags[0]) != 0;

    // ---- elemental (elemental.jl:289-334) and its SFI source (elemental.jl:348-392) -------------------------------------------
    //   r-+_ij = varpi Z-+_ij  mu_j / (mu_i + mu_j) w_j (1 - e^{-x_i} e^{-x_j}),   1 - e^{-x_i} e^{-x_j} = -(a_i + a_j + a_i a_j), a = expm1(-x)
    //   t++_ij = varpi Z++_ij  mu_j / (mu_i - mu_j) w_j (e^{-x_i} - e^{-x_j})      (mu_i != mu_j)
    //          = delta_ij e^{-x_i} + e^{-x_j} varpi Z++_ij x_i w_j                 (mu_i == mu_j)
    // A lane computes the 16 elements of its column of the strip; the results go straight into the A-forms [r] -> P, [t] -> Q (the
    // first doubling step needs them there) and the strips are read back, so that the loop over the row tiles stays ROLLED: a
    // quarter of the code (the unrolled form was 45 KB of straight-line code executed once per workgroup behind a cold
    // instruction cache: 1.1 10^5 cycles per workgroup, a sixth of its lifetime) and a few live registers.
    // The source vectors have the same form with the solar column in place of column j (mu_j -> mu_0, x_j -> dtau / mu_0,
    // w_j -> (1 + delta_m0) / 4, Z_ij -> sum_q Z_{i, i0 + q} F0_q):  j0+ is the "t" formula, j0- the "r" formula, times the beam
    // attenuation exp(-tau_sum / mu_0).  The lanes that own the spare columns c1, c2 (never read as a contraction index) evaluate
    // them in place of their (zero) matrix elements and leave them where the doubling loop wants them (see below):
    //   t[:, c1] = j0+, t[:, c2] = j1- = j0- expk ;  r[:, c1] = j0-, r[:, c2] = j0+     (ndoubl > 0)
    expk0 = THERMAL ? 1.0 : exp(-d / q.mu0);
    {
      const int j = p.col;
      const int jc = min(j, N - 1);
      const int i0 = ns * q.i_mu0;
      const int jt = laneAB ? i0 : jc;               // table / Z column
      double fz[4] = {1.0, 0.0, 0.0, 0.0};
      double att = 1.0;
      if (laneAB && !THERMAL) {
  #pragma unroll
        for (int qq = 0; qq < 4; ++qq) fz[qq] = (qq < ns) ? F0[qq + (long long)ns * s] : 0.0;
        att = exp(-tau_sum[s] / mus[i0]);
      }
      const int nz = (own_wave && !THERMAL) ? ns : 1;   // (wave-uniform)
      const double wt = (j < N) ? q.wt[jc] : 0.0;
      const double wct = laneAB ? ((m == 0) ? 0.5 : 0.25) : ((m == 0) ? wt / 2.0 : wt / 4.0);
      const bool active = laneAB || wct > num<double>::eps();
      const double mj = mus[jt], xj = xs[jt], emj = ems[jt], ej = es[jt];
      const double thB = THERMAL ? 6.283185307179586476925286766559 * (1.0 - w) * F0[s] : 0.0;
      const bool riders_in = ndoubl > 0;
  #pragma unroll 1
      for (int ta = 0; ta < 4; ++ta) {
        double zp[4] = {0.0, 0.0, 0.0, 0.0}, zm[4] = {0.0, 0.0, 0.0, 0.0};
        for (int qq = 0; qq < nz; ++qq) {
          const double f = (qq == 0) ? fz[0] : ((qq == 1) ? fz[1] : ((qq == 2) ? fz[2] : fz[3]));
          const int jz = laneAB ? jt + qq : jt;
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const long long zo = min(p.row(ta, r), N - 1) + (long long)N * jz;
            zp[r] += zget(Zp, zo) * f;
            zm[r] += zget(Zm, zo) * f;
          }
        }
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = p.row(ta, r);
          const double mi = mus[i], xi = xs[i], emi = ems[i], ei = es[i], sg = rsg[i];
          double rr, tt;
          elemental_pair(w, zp[r], zm[r], mi, xi, emi, ei, mj, xj, emj, ej, wct, i == j, thick, rr, tt);
          const bool in = i < N && j < N;
          double rv = (in && active) ? rr * sg : 0.0;
          double tv = in ? (active ? tt : ((i == j) ? ei : 0.0)) : 0.0;
          if (own_wave) {
            double vp, vm;
            if (THERMAL) {
              vp = vm = (i < N && i % ns == 0 && mi > num<double>::eps()) ? thB * (-emi) : 0.0;
            } else {
              vp = (i < N) ? tt * att : 0.0;
              vm = (i < N) ? rr * att * sg : 0.0;
            }
            if (laneAB) {
              tv = riders_in ? (laneA ? vp : vm * expk0) : 0.0;
              rv = riders_in ? (laneA ? vm : vp) : 0.0;
            }
            double* dp = laneA ? jp : sm.vec[7];   // (the other lanes write to a dummy vector)
            double* dm = laneA ? jm : sm.vec[7];
            dp[i] = vp;
            dm[i] = vm;
          }
          *p.sptr(p.delta(P), ta, r) = rv;
          *p.sptr(p.delta(Q), ta, r) = tv;
        }
      }
    }
    VSM_RSTAMP(10);
  }
  __syncthreads();
  VSM_RSTAMP(11);
  load_strip(r_s, P, p);
  load_strip(t_s, Q, p);
  VSM_RSTAMP(7);   // elemental
  // ---- doubling (rt_helpers.jl:102-166) -----------------------------------------------------------------------------------
  //   [E | W]   = r [r | t]                 A = P = [r]   (one pass over the fragments of [r])
  //   G         = (I - E)^-1                Horner series on [E] in P (Gauss-Jordan / long series: out of line)
  //   tt        = t G                       A = Q = [t]
  //   [r' | t'] = [r | 0] + tt [W | t]      A = P = [tt]  (one pass over the fragments of [tt])
  // i.e. r' = r + tt (r t) instead of (tt r) t: the intermediate tmp = tt r, its A-form store, its barrier and its product
  // phase are gone (four A-form stores and six barriers per step instead of five and seven at series order 2; the same six
  // products).  The padding rows / columns (>= N) of every strip are zero by construction (elemental writes zeros, a product
  // inherits zero rows from its A operand and zero columns from its B operand), so the A-form stores need no mask.
  // Sources (rt_helpers.jl:128-134: j0- += tt (j1- + r j0+), j0+ = j1+ + tt (j0+ + r j1-)) ride in the spare columns
  // c1, c2 (>= 4 KS: never read as a contraction index) of the strips that are live anyway, for the WHOLE loop:
  //   t_s[c1] = j0+, t_s[c2] = j1- = j0- expk  ->  W[c1] = r j0+, W[c2] = r j1-
  //   r_s[c1] = j0-, r_s[c2] = j0+             ->  W[c1] += r_s[c1] expk (= j1-), W[c2] += r_s[c2] (= j0+), r_s[c2] *= expk (= j1+):
  //                                                lane-local;  r'[c1] = j0- + tt (j1- + r j0+), r'[c2] = j1+ + tt (j0+ + r j1-)
  // -- the reference's statements term for term.  The new j0-', j0+' are already where r_s wants them; the t_s riders of the
  // next step come from the neighbour lane with one DPP quad swap per element: no LDS traffic, no extra live registers.
  double expk = expk0;
  int slot = 0;
  auto asis = [](double a, int, int) { return a; };
  VSM_RSTAMP(0);
  for (int n = 0; n < ndoubl; ++n) {
    // on entry: P = [r], Q = [t] (A-forms incl. the rider columns), r_s / t_s in registers, all waves past a barrier
    sstrip W;
    sstrip tt;
    const double fW = laneA ? expk : (laneB ? 1.0 : 0.0), fR = laneB ? expk : 1.0;
    double* const mvA[4] = {sm.xw[0], sm.xw[1], sm.xw[2], sm.xw[3]};             // (the transposer tiles are idle in the loop)
    double* const mvB[4] = {sm.xw[0] + SNP, sm.xw[1] + SNP, sm.xw[2] + SNP, sm.xw[3] + SNP};
    if constexpr (!RID) {   // r j0+ , r j1-
      mv_part(P, jp, 1.0, mvA[p.wave], p);
      mv_part(P, jm, expk, mvB[p.wave], p);
    }
    {
      sstrip G;
      {
        sstrip E;
        E.zero();
        W.zero();
        mm_ab2<KS>(E, W, P, r_s, t_s, p);
        if (own_wave) {   // W[c1] += j1- = j0- expk (lane c1 holds j0- in r_s), W[c2] += j0+ (lane c2 holds it in r_s); then
                          // r_s[c2] -> j1+ = j0+ expk, the addend of the second product: all lane-local
#pragma unroll
          for (int ta = 0; ta < 4; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              W.v[ta][r] = fma(r_s.v[ta][r], fW, W.v[ta][r]);
              r_s.v[ta][r] *= fR;
            }
        }
        VSM_RSTAMP(1);
        invert_strip_horner<KS>(E, G, P, N, sm, slot, p);   // (its first barrier: [r] is free)
        VSM_RSTAMP(2);
      }
      if constexpr (!RID) {   // u1 = j1- + r j0+ , u2 = j0+ + r j1-   (every wave is past the norm reduction's barrier)
        if (tid < SNP) {
          sm.vec[4][tid] = jm[tid] * expk + mv_sum(mvA, tid);
          sm.vec[5][tid] = jp[tid] + mv_sum(mvB, tid);
        }
      }
      tt.zero();
      mm_ab<KS>(tt, Q, G, p);              // tt = t G
    }
    load_strip(t_s, Q, p);                 // t's strip (with its riders) is not kept in registers across the inverse (kept: 75 more spilled VGPRs, no gain)
    __syncthreads();                       // P ([E]) and Q ([t]) no longer read
    store_strip(P, tt, p, asis);
    __syncthreads();
    VSM_RSTAMP(3);
    if constexpr (!RID) {   // tt u1 , tt u2   (the partial sums of r j were consumed two barriers ago)
      mv_part(P, sm.vec[4], 1.0, mvA[p.wave], p);
      mv_part(P, sm.vec[5], 1.0, mvB[p.wave], p);
    }
    {
      sstrip tn;
      tn.zero();
      mm_ab2<KS>(r_s, tn, P, W, t_s, p);   // r' = r + tt W (riders: the new j0-, j0+) ; t' = tt t
      t_s = tn;
    }
    VSM_RSTAMP(4);
    const double expk_step = expk;
    expk = expk * expk;
    if (own_wave) {
      const double ft = laneB ? expk : 1.0;   // t_s[c1] = j0+', t_s[c2] = j1-' = j0-' expk';  r_s[c1] = j0-', r_s[c2] = j0+' stay
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double u = dpp_swap1(r_s.v[ta][r]) * ft;   // lane A: j0+'   lane B: j0-' expk'
          t_s.v[ta][r] = laneAB ? u : t_s.v[ta][r];
        }
    }
    VSM_RSTAMP(5);
    if (n + 1 < ndoubl || !RID) __syncthreads();  // everybody finished reading P ([tt])
    if constexpr (!RID) {   // j0- += tt u1 ; j0+ = j0+ expk + tt u2   (visible after the barrier below / after the loop)
      if (tid < SNP) {
        jm[tid] += mv_sum(mvA, tid);
        jp[tid] = jp[tid] * expk_step + mv_sum(mvB, tid);
      }
    }
    if (n + 1 < ndoubl) {
      store_strip(P, r_s, p, asis);
      store_strip(Q, t_s, p, asis);
      __syncthreads();
    }
    VSM_RSTAMP(6);
  }
  if (ndoubl > 0 && own_wave) {   // the riders go back to the LDS vectors; the strips leave the loop with clean padding
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        double* dp = laneA ? jp : sm.vec[7];   // (the other lanes write to a dummy vector)
        double* dm = laneA ? jm : sm.vec[7];
        dp[row] = t_s.v[ta][r];           // lane A: j0+
        dm[row] = r_s.v[ta][r];           // lane A: j0-
        t_s.v[ta][r] = laneAB ? 0.0 : t_s.v[ta][r];
        r_s.v[ta][r] = laneAB ? 0.0 : r_s.v[ta][r];
      }
  }
  __syncthreads();

  VSM_RSTAMP_FLUSH();
  // ---- apply_D (doubling.jl:178-252): r-+ = D r*, j0- = D j0-* ------------------------------------------------
  if (ndoubl >= 1) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] *= rsg[p.row(ta, r)];
    if (tid < SNP) jm[tid] *= rsg[tid];
  }
  __syncthreads();
}

template <int KS>
__global__ __launch_bounds__(SNT, 2) void k_ed_strip(quad<double> q, int m, int ndoubl, const double* __restrict__ dtau,
                                                     const double* __restrict__ varpi, const double* __restrict__ tau_sum,
                                                     const double* __restrict__ F0, zsrc<double> z, added<double> out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ssmem& sm = *reinterpret_cast<ssmem*>(smem_raw);
  spos p;
  p.bind(sm.P);
  sstrip r_s, t_s;
  ed_body<KS, false>(sm, p, q, m, ndoubl, dtau, varpi, tau_sum, F0, z, r_s, t_s);
  const int s = blockIdx.x, tid = threadIdx.x, N = q.N, ns = q.n_stokes;
  double* P = sm.P;
  double* Q = sm.Q;
  const double* jp = sm.vec[0];
  const double* jm = sm.vec[1];
  auto keepN = [N](double a, int r, int c) { return (r < N && c < N) ? a : 0.0; };
  // ---- write the added layer through LDS (coalesced stores) -----------------------------------------------------
  store_strip(P, r_s, p, keepN);
  store_strip(Q, t_s, p, keepN);
  __syncthreads();
  double* g_rmp = out.r_mp + (long long)s * out.mat_stride;
  double* g_tpp = out.t_pp + (long long)s * out.mat_stride;
  double* g_rpm = out.r_pm + (long long)s * out.mat_stride;
  double* g_tmm = out.t_mm + (long long)s * out.mat_stride;
  for (int e = tid; e < N * N; e += SNT) {
    const int i = e % N, j = e / N;
    const int ix = lidx<SNP>(i, j);
    const double r = P[ix], t = Q[ix];
    g_rmp[e] = r;
    g_tpp[e] = t;
    if (!out.d_symmetric) {
      const bool same = is_uv_row(i, ns) == is_uv_row(j, ns);
      g_rpm[e] = same ? r : -r;
      g_tmm[e] = same ? t : -t;
    }
  }
  if (tid < N) {
    out.j0_p[(long long)s * N + tid] = jp[tid];
    out.j0_m[(long long)s * N + tid] = jm[tid];
  }
}


// ---------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11)  (interaction.jl:207-266), strip form
// ---------------------------------------------------------------------------

// Body shared by k_ia_strip and k_layer_strip.  On entry: r_s / t_s = strips of the added layer's r-+ / t++ (columns >= N zero),
// sm.vec[0] / vec[1] = its j0+ / j0-, all waves past a barrier, P and Q free.  DSYM: r+- = D r-+ D, t-- = D t++ D (added layers
// from doubling, ns = n_stokes); otherwise they are read from r_pm / t_mm (surface layers).
//
// The reference's statements (interaction.jl:207-266)
//   T01 = T-- (I - r-+ R+-)^-1 ;  R-+ += T01 r-+ T++ ;  J0- += T01 (j0- + r-+ J0+) ;  T-- = T01 t--
//   T21 = t++ (I - R+- r-+)^-1 ;  J0+ = j0+ + T21 (J0+ + R+- j0-) ;  T++ = T21 T++ ;  R+- = r+- + T21 R+- t--
// are evaluated with ONE inverse G2 = (I - R+- r-+)^-1 and the push-through identities (I - r R)^-1 r = r G2,
// (I - r R)^-1 = I + r G2 R  (exact; valid whenever the inverses exist):
//   [E2 | Z] = R+- [r-+ | t--]          A = P = [R+-]   z  = J0+ + R+- j0-   (j0- rides in a spare column of r-+)
//   [S  | V] = T-- [r-+ | t--]          A = Q = [T--]   vs = T-- j0-         (the same rider)
//   G2       = (I - E2)^-1              A = P = [E2]    Horner series (Gauss-Jordan / long series out of line)
//   T21 = t++ G2 ;  Y = S G2            A = P = [t++], Q = [S]               (Y = T01 r-+)
//   [R+- | T++] = [r+- | 0] + T21 [Z | T++]      A = P = [T21]   J0+ = j0+ + T21 z   (z rides in a spare column of T++)
//   [R-+ | T--] = [R-+ | V] + Y [T++ | Z]        A = Q = [Y]     J0- = J0- + vs + Y z  (the same rider)
// since T01 t-- = T-- t-- + (T-- r G2)(R+- t--) = V + Y Z and T01 (j0- + r J0+) = T-- j0- + Y (R+- j0- + J0+).  Ten products
// and the series (the as-written order: twelve and two inverses), seven barriers, three product phases of four, two and four
// products, six live strips at most; every composite matrix crosses the memory system once in each direction ([R+-], [T--]
// coalesced into their A-forms, T++ and R-+ as strips).
template <int KS, bool DSYM>
__device__ __forceinline__ void ia_body(ssmem& sm, spos& p, int N, int ns, const composite<double>& c, sstrip& r_s,
                                        sstrip& t_s, const double* __restrict__ r_pm, const double* __restrict__ t_mm) {
  double* P = sm.P;
  double* Q = sm.Q;
  double* vjp = sm.vec[0];
  double* vjm = sm.vec[1];
  double* vJp = sm.vec[2];
  double* vJm = sm.vec[3];
  double* vs = sm.vec[4];
  double* vz = sm.vec[5];
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  double* R_mp = c.R_mp + s * NN;
  double* R_pm = c.R_pm + s * NN;
  double* T_pp = c.T_pp + s * NN;
  double* T_mm = c.T_mm + s * NN;
  double* J0_p = c.J0_p + (long long)s * N;
  double* J0_m = c.J0_m + (long long)s * N;
  const int Kend = ((N + 3) >> 2) << 2;
  const int c1 = Kend, c2 = Kend + 1;
  constexpr bool RID = 4 * KS + 2 <= SNP;   // spare columns for the vectors (N <= 60); else VALU mat-vecs (N = 61..64)
  const bool own_wave = RID && (p.wave == (c1 >> 4));
  const bool laneA = own_wave && (p.col == c1), laneB = own_wave && (p.col == c2);
  // slots of the mat-vec partial sums (!RID): the Gauss-Jordan scratch (idle outside the inverse) and two spare vectors
  double* const gjv = &sm.gj.col[0][0];
  double* const mvA[4] = {gjv, gjv + SNP, gjv + 2 * SNP, gjv + 3 * SNP};
  double* const mvB[4] = {gjv + 4 * SNP, gjv + 5 * SNP, sm.vec[6], sm.vec[7]};
  auto keepN = [N](double x, int r, int cc) { return (r < N && cc < N) ? x : 0.0; };
  int slot = 0;
  double* xw = sm.xw[p.wave];
  const dpar dp(sm.usg, p);   // (filled by the caller; all +1 for a surface layer)

  VSM_STAMP_DECL;
  // ---- stage: composite vectors, [R+-] -> P, [T--] -> Q (all 32 column loads of a lane in flight together) ---------------
  if (tid < SNP) {
    const bool in = tid < N;
    vJp[tid] = in ? J0_p[tid] : 0.0;
    vJm[tid] = in ? J0_m[tid] : 0.0;
  }
  sstrip Z, V, G;                          // (Z doubles as the t-- strip of a surface layer until the first products)
  if (!DSYM) load_strip_global_c8_issue(G, t_mm, N, p);
  stage_aform_full2(P, R_pm, Q, T_mm, N, p);
  if (!DSYM) load_strip_global_c8_finish(G, p, xw);
  __syncthreads();                                                                                       // (a)
  VSM_STAMP(8);
  if constexpr (!RID) {   // R+- j0- , T-- j0-
    mv_part(P, vjm, 1.0, mvA[p.wave], p);
    mv_part(Q, vjm, 1.0, mvB[p.wave], p);
  }
  if (own_wave) {  // j0- rides in the spare column c2 of r-+:  E2[:, c2] = R+- j0-, S[:, c2] = T-- j0-
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) r_s.v[ta][r] = laneB ? vjm[p.row(ta, r)] : r_s.v[ta][r];
  }
  if (DSYM) dsym_strip(t_s, t_s, dp);   // t-- = D t++ D in place (undone below: D is an involution)
  {
    sstrip E, S;
    {
      const sstrip& tb = DSYM ? t_s : G;
      E.zero();
      Z.zero();
      mm_ab2<KS>(E, Z, P, r_s, tb, p);
      S.zero();
      V.zero();
      mm_ab2<KS>(S, V, Q, r_s, tb, p);
    }
    if (own_wave) {
      double* zd = laneB ? vz : sm.vec[7];   // (the other lanes write to a dummy vector)
      double* sd = laneB ? vs : sm.vec[7];
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          zd[row] = vJp[row] + E.v[ta][r];
          sd[row] = S.v[ta][r];
        }
    }
    // r_s <- r+- (the accumulator of the R+- update; its rider column is never stored), t_s <- t++
    if (DSYM) {
      dsym_strip(r_s, r_s, dp);
      dsym_strip(t_s, t_s, dp);
    } else {
      load_strip_global_c8(r_s, r_pm, N, p, xw);
    }
    VSM_STAMP(9);
    const double nrm = strip_norm_bound_clean(E, N, sm, slot, p);   // (b): every wave is done reading [R+-] and [T--]
    if constexpr (!RID) {   // z = J0+ + R+- j0- ; vs = T-- j0-   (before the inverse may use the Gauss-Jordan scratch)
      if (tid < SNP) {
        vz[tid] = vJp[tid] + mv_sum(mvA, tid);
        vs[tid] = mv_sum(mvB, tid);
      }
    }
    store_strip(Q, S, p, keepN);                                    // [S] -> Q  (read after barrier (e))
    invert_strip_horner_k<KS>(series_order(nrm), E, G, P, N, sm, p);   // [E2] -> P, barrier (c), series
  }
  VSM_STAMP(10);
  __syncthreads();                        // (d): [E2] (series) no longer read
  store_strip(P, t_s, p, keepN);          // [t++] -> P
  __syncthreads();                        // (e)
  VSM_STAMP(11);
  {
    sstrip X, Y;
    X.zero();
    mm_ab<KS>(X, P, G, p);                // T21 = t++ G2
    Y.zero();
    mm_ab<KS>(Y, Q, G, p);                // Y = S G2 = T01 r-+
    VSM_STAMP(12);
    __syncthreads();                      // (f): [t++], [S] no longer read
    store_strip(P, X, p, keepN);          // [T21] -> P
    store_strip(Q, Y, p, keepN);          // [Y]   -> Q
  }
  sstrip Tpp, Rmp;
  load_strip_global_c8_issue(Tpp, T_pp, N, p);
  load_strip_global_c8_issue(Rmp, R_mp, N, p);
  __syncthreads();                        // (g)
  VSM_STAMP(13);
  if constexpr (!RID) {   // T21 z , Y z
    mv_part(P, vz, 1.0, mvA[p.wave], p);
    mv_part(Q, vz, 1.0, mvB[p.wave], p);
  }
  load_strip_global_c8_finish(Tpp, p, xw);
  if (own_wave) {  // z rides in the spare column c1 of T++:  (T21 T++)[:, c1] = T21 z, (Y T++)[:, c1] = Y z
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) Tpp.v[ta][r] = laneA ? vz[p.row(ta, r)] : Tpp.v[ta][r];
  }
  {
    sstrip acc;
    acc.zero();
    mm_ab2<KS>(r_s, acc, P, Z, Tpp, p);   // R+- = r+- + T21 Z ; T++ = T21 T++
    VSM_STAMP(14);
    load_strip_global_c8_finish(Rmp, p, xw);
    store_strip_global_c8(R_pm, r_s, N, p, xw);
    store_strip_global_c8(T_pp, acc, N, p, xw);
    if (laneA) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          if (row < N) J0_p[row] = vjp[row] + acc.v[ta][r];
        }
    }
  }
  VSM_STAMP(15);
  mm_ab2<KS>(Rmp, V, Q, Tpp, Z, p);       // R-+ = R-+ + Y T++ ; T-- = V + Y Z
  VSM_STAMP(16);
  store_strip_global_c8(R_mp, Rmp, N, p, xw);
  store_strip_global_c8(T_mm, V, N, p, xw);
  if (laneA) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        if (row < N) J0_m[row] = vJm[row] + vs[row] + Rmp.v[ta][r];
      }
  }
  if constexpr (!RID) {   // J0+ = j0+ + T21 z ; J0- = J0- + T-- j0- + Y z
    __syncthreads();
    if (tid < N) {
      J0_p[tid] = vjp[tid] + mv_sum(mvA, tid);
      J0_m[tid] = vJm[tid] + vs[tid] + mv_sum(mvB, tid);
    }
  }
  VSM_STAMP(17);
  VSM_STAMP_FLUSH();
}

template <int KS, bool DSYM>
__global__ __launch_bounds__(SNT, 2) void k_ia_strip(int N, composite<double> c, added<double> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ssmem& sm = *reinterpret_cast<ssmem*>(smem_raw);
  spos p;
  p.bind(sm.P);
  const int s = blockIdx.x, tid = threadIdx.x;
  if (tid < SNP) {
    const bool in = tid < N;
    sm.vec[0][tid] = in ? a.j0_p[(long long)s * N + tid] : 0.0;
    sm.vec[1][tid] = in ? a.j0_m[(long long)s * N + tid] : 0.0;
    sm.usg[tid] = (DSYM && is_uv_row(tid, a.d_symmetric)) ? -1.0 : 1.0;
  }
  sstrip r_s, t_s;
  load_strip_global_c8(r_s, a.r_mp + s * a.mat_stride, N, p, sm.xw[p.wave]);
  load_strip_global_c8(t_s, a.t_pp + s * a.mat_stride, N, p, sm.xw[p.wave]);
  __syncthreads();
  ia_body<KS, DSYM>(sm, p, N, a.d_symmetric, c, r_s, t_s, DSYM ? nullptr : a.r_pm + s * a.mat_stride,
                    DSYM ? nullptr : a.t_mm + s * a.mat_stride);
}

// rt_kernel!(::noRS) for a scattering layer (rt_kernel.jl:175-250) in ONE launch: elemental! + doubling! and then
// either the TOA copy (iz == 1: copy_added_to_composite!, rt_helpers.jl:188-200) or interaction!(::_11).  The added
// layer never leaves the chip.
template <int KS, bool MIX, bool THERMAL, bool PRE = false>
__device__ __forceinline__ void layer_body(const quad<double>& q, int m, int ndoubl, const double* __restrict__ dtau,
                                           const double* __restrict__ varpi, const double* __restrict__ tau_sum,
                                           const double* __restrict__ F0, const zsrc<double>& z, int toa,
                                           const composite<double>& c, const double* __restrict__ img = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  ssmem& sm = *reinterpret_cast<ssmem*>(smem_raw);
  VSM_LIFE_DECL;
  spos p;
  p.bind(sm.P);
  sstrip r_s, t_s;
  ed_body<KS, MIX, THERMAL, PRE>(sm, p, q, m, ndoubl, dtau, varpi, tau_sum, F0, z, r_s, t_s, img);
  const int N = q.N, ns = q.n_stokes;
  if (toa) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long NN = (long long)N * N;
    double* xw = sm.xw[p.wave];
    store_strip_global_c8(c.R_mp + s * NN, r_s, N, p, xw);
    store_strip_global_c8(c.T_pp + s * NN, t_s, N, p, xw);
    sstrip d;
    const dpar dp(sm.usg, p);
    dsym_strip(d, r_s, dp);
    store_strip_global_c8(c.R_pm + s * NN, d, N, p, xw);
    dsym_strip(d, t_s, dp);
    store_strip_global_c8(c.T_mm + s * NN, d, N, p, xw);
    if (tid < N) {
      c.J0_p[(long long)s * N + tid] = sm.vec[0][tid];
      c.J0_m[(long long)s * N + tid] = sm.vec[1][tid];
    }
    return;
  }
  ia_body<KS, true>(sm, p, N, ns, c, r_s, t_s, nullptr, nullptr);
  VSM_LIFE_FLUSH();
}
// The same for several Fourier moments at once: blockIdx.y picks the moment's composite.  The elemental layers come from the
// pre-pass k_elemental_img (below) as A-form images: inside this kernel their ~2 10^3 VALU / LDS instructions per wave would
// each queue behind an FP64 MFMA of the workgroup that shares the SIMDs (on gfx950 the FP64 matrix rate equals the FP64
// vector rate: an MFMA occupies the lanes for its 64 cycles) -- measured 1.3 10^5 cycles per workgroup, a fifth of its
// lifetime, against ~2 10^4 for the image copy.  The pre-pass runs at full occupancy with nothing to wait for.
struct layer_mm_comps {
  composite<double> c[VSM_MM_MAX];
};
template <int KS>
__global__ __launch_bounds__(SNT, 2) void k_layer_strip_mm(quad<double> q, int ndoubl, layer_mm_comps a, int toa,
                                                           const double* __restrict__ pre) {
  const int im = blockIdx.y;
  const double* img = pre + ((long long)im * gridDim.x + blockIdx.x) * PRE_STRIDE;
  layer_body<KS, false, false, true>(q, 0, ndoubl, nullptr, nullptr, nullptr, nullptr, zsrc<double>{}, toa, a.c[im], img);
}


// ---------------------------------------------------------------------------
// Operator-level batched product in strip form:  C = alpha A*B + beta D + gamma I   (FP64, M, Nc, K <= 64)
// One workgroup per (spectral point, parameter): A is staged ONCE into LDS (coalesced), every wave takes its 16-column
// strip of B straight from global memory into the MFMA B-operand registers and writes its strip of C -- each operand
// crosses the memory system once, where k_gemm's tile-waves re-read A rows / B columns 16/3 times at N = 60.
// 40 KB of LDS per workgroup: four workgroups per CU.
// ---------------------------------------------------------------------------
struct gsmem {
  double A[SNP * SNP];
};
template <int KS>
__global__ __launch_bounds__(SNT, 4) void k_gemm_strip(int M, int Nc, int K, const double* __restrict__ A, long long sa,
                                                       long long pa, const double* __restrict__ B, long long sb, long long pb,
                                                       double* C, long long sc, long long pc, double alpha, const double* D,
                                                       long long sd, long long pd, double beta, double gamma) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  gsmem& sm = *reinterpret_cast<gsmem*>(smem_raw);
  spos p;
  p.bind(sm.A);
  const long long s = blockIdx.x, pp = blockIdx.y;
  const double* As = A + s * sa + pp * pa;
  const double* Bs = B + s * sb + pp * pb;
  double* Cs = C + s * sc + pp * pc;
  const double* Ds = D ? D + s * sd + pp * pd : nullptr;
#pragma unroll 4
  for (int j = p.wave; j < 4 * KS; j += 4)
    sm.A[lidx<SNP>(p.lane, j)] = (p.lane < M && j < K) ? As[p.lane + (long long)M * j] : 0.0;
  sstrip b, acc;
  const int cc = min(p.col, Nc - 1);
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const double v = Bs[min(row, K - 1) + (long long)K * cc];
      b.v[ta][r] = (row < K && p.col < Nc) ? v : 0.0;
    }
  __syncthreads();
  acc.zero();
  mm_ab<KS>(acc, sm.A, b, p);
  if (p.col < Nc) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = p.row(ta, r);
        if (row < M) {
          double v = alpha * acc.v[ta][r];
          if (Ds) v += beta * Ds[row + (long long)M * p.col];
          if (row == p.col) v += gamma;
          Cs[row + (long long)M * p.col] = v;
        }
      }
  }
}

}  // namespace

// The layer kernels are instantiated per KS in separately compiled objects (make: vsm_strip_<KS>.o is this file built
// with -DVSM_STRIP_KS=<KS>) so that the seven instantiations build in parallel; the object built WITHOUT the macro holds
// the dispatchers and the strip GEMM.
#define VSM_CAT2(a, b) a##b
#define VSM_CAT(a, b) VSM_CAT2(a, b)
#define VSM_STRIP_DECL(KS)                                                                                                  \
  int VSM_CAT(launch_ed_strip_, KS)(const quad<double>&, int, int, int, const double*, const double*, const double*,       \
                                    const double*, const zsrc<double>&, const added<double>&, hipStream_t);                 \
  int VSM_CAT(launch_ia_strip_, KS)(int, int, const composite<double>&, const added<double>&, hipStream_t);                \
  int VSM_CAT(launch_layer_strip_mm_, KS)(const quad<double>&, int, int, int, const layer_mm_args<double>&, int,           \
                                          const double*, hipStream_t);

#ifdef VSM_STRIP_KS
VSM_STRIP_DECL(VSM_STRIP_KS)
template <typename K>
static int strip_enable_lds(K kern, const char* what) {
  return ensure_dyn_lds(reinterpret_cast<const void*>(kern), sizeof(ssmem), what);   // once per (device, kernel)
}
int VSM_CAT(launch_ed_strip_, VSM_STRIP_KS)(const quad<double>& q, int S, int m, int ndoubl, const double* dtau,
                                            const double* varpi, const double* tau_sum, const double* F0,
                                            const zsrc<double>& z, const added<double>& a, hipStream_t st) {
#if VSM_STRIP_KS < 16
  const int prepared = strip_enable_lds(k_ed_strip<VSM_STRIP_KS>, "hipFuncSetAttribute(k_ed_strip)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(k_ed_strip<VSM_STRIP_KS>, dim3(S), dim3(SNT), sizeof(ssmem), st, q, m, ndoubl, dtau, varpi, tau_sum, F0, z,
                     a);
  VSM_LAUNCH_CHECK("k_ed_strip");
  return VSM_OK;
#else   // N = 61..64: elemental! + doubling! alone stay on the LDS-resident kernels (the strip form needs the pre-pass)
  return VSM_ERR_UNSUPPORTED;
#endif
}
int VSM_CAT(launch_ia_strip_, VSM_STRIP_KS)(int N, int S, const composite<double>& c, const added<double>& a, hipStream_t st) {
  const int prepared = strip_enable_lds(k_ia_strip<VSM_STRIP_KS, true>, "hipFuncSetAttribute(k_ia_strip)");
  const int prepared_g = strip_enable_lds(k_ia_strip<VSM_STRIP_KS, false>, "hipFuncSetAttribute(k_ia_strip general)");
  if (prepared) return prepared;
  if (prepared_g) return prepared_g;
  if (a.d_symmetric)   // (d_symmetric carries n_stokes)
    hipLaunchKernelGGL((k_ia_strip<VSM_STRIP_KS, true>), dim3(S), dim3(SNT), sizeof(ssmem), st, N, c, a);
  else
    hipLaunchKernelGGL((k_ia_strip<VSM_STRIP_KS, false>), dim3(S), dim3(SNT), sizeof(ssmem), st, N, c, a);
  VSM_LAUNCH_CHECK("k_ia_strip");
  return VSM_OK;
}
int VSM_CAT(launch_layer_strip_mm_, VSM_STRIP_KS)(const quad<double>& q, int S, int nm, int ndoubl,
                                                  const layer_mm_args<double>& a, int toa, const double* pre, hipStream_t st) {
  const int prepared = strip_enable_lds(k_layer_strip_mm<VSM_STRIP_KS>, "hipFuncSetAttribute(k_layer_strip_mm)");
  if (prepared) return prepared;
  layer_mm_comps cc;
  for (int i = 0; i < VSM_MM_MAX; ++i) cc.c[i] = a.c[i];
  hipLaunchKernelGGL((k_layer_strip_mm<VSM_STRIP_KS>), dim3(S, nm), dim3(SNT), sizeof(ssmem), st, q, ndoubl, cc, toa, pre);
  VSM_LAUNCH_CHECK("k_layer_strip_mm");
  return VSM_OK;
}

#else  // ---- dispatcher object -----------------------------------------------------------------------------------------

VSM_STRIP_DECL(9)
VSM_STRIP_DECL(10)
VSM_STRIP_DECL(11)
VSM_STRIP_DECL(12)
VSM_STRIP_DECL(13)
VSM_STRIP_DECL(14)
VSM_STRIP_DECL(15)
VSM_STRIP_DECL(16)

bool strip_supported(int N) {
  const int Kend = ((N + 3) >> 2) << 2;
  return N > 32 && Kend + 2 <= SNP;
}
// the layer step (pre-pass + layer kernel) and interaction!(_11) also take N = 61..64 (KS = 16: no spare columns, the source
// vectors by mat-vecs over the A-forms)
bool strip_layer_supported(int N) { return N > 32 && N <= SNP; }

template <int KS>
static void launch_gemm_strip(int M, int Nc, int K, int S, int P, const double* A, long long sa, long long pa, const double* B,
                              long long sb, long long pb, double* C, long long sc, long long pc, double alpha, const double* D,
                              long long sd, long long pd, double beta, double gamma, hipStream_t st) {
  hipLaunchKernelGGL(k_gemm_strip<KS>, dim3(S, P), dim3(SNT), sizeof(gsmem), st, M, Nc, K, A, sa, pa, B, sb, pb, C, sc, pc, alpha,
                     D, sd, pd, beta, gamma);
}
// returns VSM_ERR_UNSUPPORTED when the shape is left to k_gemm
int strip_gemm(int M, int Nc, int K, int S, int P, const double* A, long long sa, long long pa, const double* B, long long sb,
               long long pb, double* C, long long sc, long long pc, double alpha, const double* D, long long sd, long long pd,
               double beta, double gamma, hipStream_t st) {
  if (M > SNP || Nc > SNP || K > SNP || M <= 16 || Nc <= 8 || K <= 8 || P > 65535) return VSM_ERR_UNSUPPORTED;
  static const bool off = ab_switch("VSM_NO_STRIP") || ab_switch("VSM_NO_STRIP_GEMM");
  if (off) return VSM_ERR_UNSUPPORTED;
  const int ks = (K + 3) >> 2;
#define VSM_G(KS) launch_gemm_strip<KS>(M, Nc, K, S, P, A, sa, pa, B, sb, pb, C, sc, pc, alpha, D, sd, pd, beta, gamma, st)
  if (ks <= 4) VSM_G(4);
  else if (ks <= 8) VSM_G(8);
  else if (ks <= 12) VSM_G(12);
  else if (ks <= 15) VSM_G(15);
  else VSM_G(16);
#undef VSM_G
  VSM_LAUNCH_CHECK("k_gemm_strip");
  return VSM_OK;
}

#define VSM_STRIP_SWITCH16(N_, CALL)      \
  if ((((N_) + 3) >> 2) == 16) return CALL(16);
#define VSM_STRIP_SWITCH(N_, CALL)        \
  switch (((N_) + 3) >> 2) {               \
    case 9: return CALL(9);                \
    case 10: return CALL(10);              \
    case 11: return CALL(11);              \
    case 12: return CALL(12);              \
    case 13: return CALL(13);              \
    case 14: return CALL(14);              \
    case 15: return CALL(15);              \
    default: break;                        \
  }

// ---------------------------------------------------------------------------
// Elemental pre-pass of the multi-moment layer kernel: elemental! (elemental.jl:289-392) for every (point, moment) of a layer
// into A-form images (layout: vsm_strip_dev.h, PRE_STRIDE).  One workgroup per (point, moment); lane = row, a wave walks the
// columns j = wave, wave + 4, ... (uniform per wave), so the image is written in full 512-byte columns.
// ---------------------------------------------------------------------------
template <bool MIX, bool THERMAL = false>   // THERMAL: the `:thermal` source slot (F0 = B[S], expk = 1; see ed_body)
__global__ __launch_bounds__(SNT) void k_elemental_img(quad<double> q, int ndoubl, const double* __restrict__ dtau,
                                                       const double* __restrict__ varpi, const double* __restrict__ tau_sum,
                                                       const double* __restrict__ F0, layer_mm_args<double> a,
                                                       double* __restrict__ pre) {
  __shared__ double mus[SNP], xs[SNP], es[SNP], ems[SNP];
  __shared__ int thick_flag;
  const int s = blockIdx.x, im = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int N = q.N, ns = q.n_stokes, m = a.m[im];
  const zsrc<double> z = a.z[im];
  const double d = dtau[s], w = varpi[s];
  const int ncomp = MIX ? z.ncomp : 0;
  const long long NNz = (long long)N * N;
  const double* Zp = z.Zpp + (ncomp ? 0 : (long long)s * z.zs);
  const double* Zm = z.Zmp + (ncomp ? 0 : (long long)s * z.zs);
  double fk[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (k < ncomp) fk[k] = z.fcomp[(long long)s * ncomp + k];
  auto zget = [&](const double* Z, long long zo) {
    if (ncomp == 0) return Z[zo];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < ncomp) acc += fk[k] * Z[k * NNz + zo];
    return acc;
  };
  if (tid < SNP) {
    const bool in = tid < N;
    const double mu = in ? q.mu[tid] : 1.0;
    const double x = d / mu;
    mus[tid] = mu;
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
    const unsigned long long any_thick = __ballot(in && x >= 0.5);
    if (tid == 0) thick_flag = any_thick != 0ull;
  }
  __syncthreads();
  const bool thick = thick_flag != 0;
  const int i = lane, ic = min(i, N - 1);
  const double mi = mus[i], xi = xs[i], ai = ems[i], ei = es[i];
  const double sg = (ndoubl >= 1 && is_uv_row(i, ns)) ? -1.0 : 1.0;   // starred R* = D R (elemental.jl:403-422)
  double* out = pre + ((long long)im * gridDim.x + s) * PRE_STRIDE;
  double* R = out;
  double* T = out + PRE_IMG;
  const int Kend = ((N + 3) >> 2) << 2;
  const int c1 = Kend, c2 = Kend + 1;   // the spare columns that carry the source vectors through the doubling loop
  const bool riders_in = ndoubl > 0 && Kend + 2 <= SNP;   // (N = 61..64: no spare column; the layer kernel uses mat-vecs)
#pragma unroll 1
  for (int j = wave; j < SNP; j += 4) {
    if (riders_in && (j == c1 || j == c2)) continue;   // written below
    double rv = 0.0, tv = 0.0;
    if (j < N) {                                        // (wave-uniform)
      const double wt = q.wt[j];
      const double wct = (m == 0) ? wt / 2.0 : wt / 4.0;
      const long long zo = ic + (long long)N * j;
      double rr, tt;
      elemental_pair(w, zget(Zp, zo), zget(Zm, zo), mi, xi, ai, ei, mus[j], xs[j], ems[j], es[j], wct, i == j, thick, rr, tt);
      const bool active = wct > num<double>::eps();
      if (i < N) {
        rv = active ? rr * sg : 0.0;
        tv = active ? tt : ((i == j) ? ei : 0.0);
      }
    }
    R[lidx<SNP>(i, j)] = rv;
    T[lidx<SNP>(i, j)] = tv;
  }
  if (wave == 0) {   // SFI source of the solar beam: the same formulas with the solar column (see elemental_pair)
    double vp, vm, expk0;
    if (THERMAL) {   // j0+- = 2 pi (1 - varpi) B (1 - e^{-dtau / mu_i}) on the I rows (Sources/thermal_emission.jl:241-301)
      vp = vm = (i < N && i % ns == 0 && mi > num<double>::eps()) ? 6.283185307179586476925286766559 * (1.0 - w) * F0[s] * (-ai) : 0.0;
      expk0 = 1.0;
    } else {
      const int i0 = ns * q.i_mu0;
      double zp = 0.0, zm = 0.0;
      for (int qq = 0; qq < ns; ++qq) {
        const long long zo = ic + (long long)N * (i0 + qq);
        const double f = F0[qq + (long long)ns * s];
        zp += zget(Zp, zo) * f;
        zm += zget(Zm, zo) * f;
      }
      double rr, tt;
      elemental_pair(w, zp, zm, mi, xi, ai, ei, mus[i0], xs[i0], ems[i0], es[i0], (m == 0) ? 0.5 : 0.25, false, thick, rr, tt);
      const double att = exp(-tau_sum[s] / mus[i0]);
      vp = (i < N) ? tt * att : 0.0;
      vm = (i < N) ? rr * att * sg : 0.0;
      expk0 = exp(-d / q.mu0);
    }
    out[2 * PRE_IMG + i] = vp;
    out[2 * PRE_IMG + SNP + i] = vm;
    out[2 * PRE_IMG + 2 * SNP + i] = expk0;
    if (riders_in) {   // t[:, c1] = j0+, t[:, c2] = j1- = j0- expk ;  r[:, c1] = j0-, r[:, c2] = j0+
      T[lidx<SNP>(i, c1)] = vp;
      T[lidx<SNP>(i, c2)] = vm * expk0;
      R[lidx<SNP>(i, c1)] = vm;
      R[lidx<SNP>(i, c2)] = vp;
    }
  }
}

static int strip_layer_forward_pre(const quad<double>& q, int S, int nm, int ndoubl, const double* dtau, const double* varpi,
                                   const double* tau_sum, const double* F0, const layer_mm_args<double>& a, int toa,
                                   hipStream_t st, int thermal) {
  if (S <= 0 || nm <= 0) return VSM_OK;
  if (!strip_layer_supported(q.N)) {
    set_error("strip_layer_forward_mm: N=%d outside (32, 64]", q.N);
    return VSM_ERR_UNSUPPORTED;
  }
  double* pre = static_cast<double*>(scratch((size_t)nm * S * PRE_STRIDE * sizeof(double), 3, st));
  if (!pre) return VSM_ERR_HIP;
  const dim3 grid(S, nm), block(SNT);
  if (thermal) {   // F0 = B[S]; tau_sum is not read
    if (a.z[0].ncomp > 0)
      hipLaunchKernelGGL((k_elemental_img<true, true>), grid, block, 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
    else
      hipLaunchKernelGGL((k_elemental_img<false, true>), grid, block, 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  } else if (a.z[0].ncomp > 0) {
    hipLaunchKernelGGL((k_elemental_img<true>), grid, block, 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  } else {
    hipLaunchKernelGGL((k_elemental_img<false>), grid, block, 0, st, q, ndoubl, dtau, varpi, tau_sum, F0, a, pre);
  }
  VSM_LAUNCH_CHECK("k_elemental_img");
#define VSM_CALL(KS) VSM_CAT(launch_layer_strip_mm_, KS)(q, S, nm, ndoubl, a, toa, pre, st)
  VSM_STRIP_SWITCH16(q.N, VSM_CALL)
  VSM_STRIP_SWITCH(q.N, VSM_CALL)
#undef VSM_CALL
  set_error("strip_layer_forward_mm: N=%d outside (32, 60]", q.N);
  return VSM_ERR_UNSUPPORTED;
}
int strip_layer_forward_mm(const quad<double>& q, int S, int nm, int ndoubl, const double* dtau, const double* varpi,
                           const double* tau_sum, const double* F0, const layer_mm_args<double>& a, int toa, hipStream_t st) {
  return strip_layer_forward_pre(q, S, nm, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st, 0);
}
// one moment (solar, or the `:thermal` slot: F0 = B[S], m = 0): the same pair with nm = 1
int strip_layer_forward(const quad<double>& q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                        const double* tau_sum, const double* F0, const zsrc<double>& z, int toa,
                        const composite<double>& c, hipStream_t st, int thermal) {
  if (S <= 0) return VSM_OK;
  layer_mm_args<double> a;
  for (int i = 0; i < VSM_MM_MAX; ++i) {
    a.m[i] = m;
    a.z[i] = z;
    a.c[i] = c;
  }
  return strip_layer_forward_pre(q, S, 1, ndoubl, dtau, varpi, tau_sum, F0, a, toa, st, thermal);
}

int strip_interaction11(int N, int S, const composite<double>& c, const added<double>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
#define VSM_CALL(KS) VSM_CAT(launch_ia_strip_, KS)(N, S, c, a, st)
  VSM_STRIP_SWITCH16(N, VSM_CALL)
  VSM_STRIP_SWITCH(N, VSM_CALL)
#undef VSM_CALL
  set_error("strip_interaction11: N=%d outside (32, 60]", N);
  return VSM_ERR_UNSUPPORTED;
}

int strip_elemental_doubling(const quad<double>& q, int S, int m, int ndoubl, const double* dtau, const double* varpi,
                             const double* tau_sum, const double* F0, const zsrc<double>& z, const added<double>& a,
                             hipStream_t st) {
  if (S <= 0) return VSM_OK;
#define VSM_CALL(KS) VSM_CAT(launch_ed_strip_, KS)(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, a, st)
  VSM_STRIP_SWITCH(q.N, VSM_CALL)
#undef VSM_CALL
  set_error("strip_elemental_doubling: N=%d outside (32, 60]", q.N);
  return VSM_ERR_UNSUPPORTED;
}
#endif  // VSM_STRIP_KS

}  // namespace vsm

#if defined(VSM_PHASE_TIMING) && defined(VSM_STRIP_KS) && VSM_STRIP_KS == 15   // stamps of the C2 instantiation
extern "C" int vsm_debug_phase_cycles_strip(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_phase_cycles_strip), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_phase_cycles_strip), z, sizeof(z));
  }
  return 0;
}
#endif
