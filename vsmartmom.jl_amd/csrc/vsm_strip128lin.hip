// Linearized (Jacobian) doubling and interaction for FP64, 60 < N <= 128, in the column-strip scheme of vsm_strip128.hip:
//
//   k_dbl128_lin<RT>   doubling_allparams_helper! (src/CoreRT/CoreKernel/doubling_lin.jl:216-339) + apply_D! with derivative
//                      slots (:374-421): ALL doubling steps of one spectral point, forward + every active parameter, in one
//                      persistent workgroup, in place on the AddedLayer / AddedLayerLin
//   k_ia128_lin<RT>    one half of interaction_helper!(::ScatteringInterface_11) with derivatives (interaction_lin.jl:217-331);
//                      the interaction is two launches with different operand bindings (the second
//                      half reads only arrays the first leaves untouched)
//
// Both are the same recurrence (vsm_striplin.hip): with a left operand LA, right operands ER, S2, S3, a transmission LT,
//     E = LA ER ;  G = (I - E)^-1 ;  tt = LT G ;  rt = LA S2 ;        out0 = ACC0 + tt rt ;  out1 = tt S3
//     per parameter:  X1 = PA ER + LA D1 ;  X2 = PA S2 + LA D2 ;  Y = YI + tt X1 ;  ttdot = Y G      (Gdot is never formed)
//                     outp0 = ACCP + tt X2 + ttdot rt ;  outp1 = tt D3 + ttdot S3
// (doubling: LA = ER = ACC0 = r, LT = S2 = S3 = t, PA = D1 = ACCP = rdot, D2 = D3 = YI = tdot).  The shapes of 61 <= N <= 128 leave
// no room for the four A-forms and ten live strips the kernels of N <= 60 keep on chip, so, as in vsm_strip128.hip:
//   * ONE A-form in LDS; the products run ordered by left operand: [LA] -> [E] -> [LT] -> [tt], then per parameter
//     [LA] -> [PA] -> [tt] -> [Y] -> [ttdot] (parameters one after the other: the scratch does not grow with their number and the
//     source-vector tables stay two columns wide);
//   * every strip that is not an operand of the running product waits in the workgroup's global scratch as lane-linear 32-byte
//     records (2 KB per instruction, L2 / MALL resident); the doubling state (r, t, rdot_p, tdot_p) stays there between steps and
//     meets the AddedLayer's column-major arrays only on the way in and out;
//   * source vectors: one more 16 x 16 MFMA tile per wave behind the products that carry them (rider_tile: row tile w of
//     [A] (x_0 | x_1), x from an LDS table), the vector algebra by N threads between the phases.  nw = RT waves.
// Workgroups are persistent (grid = CUs; two per CU at RT = 4) and walk the spectral axis.
#include "vsm_strip128_dev.h"

namespace vsm {
namespace {

// ---- shared pieces -------------------------------------------------------------------------------------------------------------
template <int RT>
struct lin128 {
  static constexpr int NP = 16 * RT;
  // LDS (doubles): A-form | reduction slots (32) | pivot / permutation of the Gauss-Jordan path (128) | NVEC vector tables of NP
  // | one transposer tile per wave
  static constexpr int NVEC = 20;
  static constexpr int XW = 16 * XS8B;   // per-wave transposer tile of load_c8 / store_c8
  static constexpr size_t lds_bytes() { return (size_t)(NP * NP + 32 + 128 + NVEC * NP + RT * XW) * sizeof(double); }
};

// y_0 = [A] x_0, y_1 = [A] x_1 for the two vectors of the LDS table at xt (x_0 = xt[0..NP), x_1 = xt[NP..2 NP)): wave w forms row
// tile w (rider_tile), the lanes of columns 0 / 1 leave the rows in y0 / y1.  Needs nw = RT.
template <int RT>
__device__ __forceinline__ void rider2(double* y0, double* y1, const double* xt, bpos<RT>& p) {
  constexpr int NP = 16 * RT;
  const unsigned xb = lds_addr128(xt) + 8u * (unsigned)((p.l15 & 1) * NP + p.kq);
  d4_t y = acc_zero<double>();
  rider_tile(y, xb, p);
  if (p.l15 < 2) {
    double* d = p.l15 ? y1 : y0;
    const int rrow = 16 * p.wave + p.kq;
#pragma unroll
    for (int r = 0; r < 4; ++r) d[rrow + 4 * r] = y[r];
  }
}

// The wave's record of scratch slot `slot`: wbase (slot 0) + slot * one slot's stride.  The base goes through an empty asm so that
// the sum is formed where it is used (two VALU instructions) instead of twenty hoisted -- and spilled -- slot addresses.
template <int RT>
__device__ __forceinline__ d4_t* slot_base(d4_t* scr, int nslots, const bpos<RT>& p) {
  return scr + ((long long)blockIdx.x * nslots * B_MAXW + p.wave) * (RT * 64) + p.lane;
}
template <int RT>
__device__ __forceinline__ d4_t* slot_ptr(d4_t* wbase, int slot) {
  asm volatile("" : "+v"(wbase));
  return wbase + (long long)slot * (B_MAXW * RT * 64);
}

// Residency policy.  At four row tiles (N <= 64) a strip is 32 registers and a workgroup's wave can hold six beside the
// fragments, so the strips every parameter needs again stay in registers (K = true): the state r, t across ALL steps, rdot_p and
// the accumulators X1, Q2 from the [r] phase to the [tt] phase -- a doubling step with one parameter then parks / fetches 14
// strips instead of 40.  From five row tiles on (40 .. 64 registers per strip) everything that is not an operand waits in the
// scratch, as in vsm_strip128.hip.
template <int RT>
struct keep128 {
  static constexpr bool value = RT <= 4;
};
// Persistent workgroups per CU (a workgroup is RT waves; the register budget is 256 up to four row tiles, i.e. two waves per SIMD):
// eight waves per CU up to four row tiles, one workgroup beyond.
template <int RT>
struct wg_per_cu {
  static constexpr int value = RT == 1 ? 8 : (RT == 2 ? 4 : (RT <= 4 ? 2 : 1));
};
// acc += [A] * (the strip `held` when K, else the parked strip at `slot`)
template <bool K, int RT>
__device__ __forceinline__ void mm_held(bstrip<RT>& acc, const bstrip<RT>& held, const d4_t* slot, bpos<RT>& p) {
  if constexpr (K) {
    mm128(acc, held, p);
  } else {
    bstrip<RT> B;
    fill(B, slot, p);
    mm128(acc, B, p);
  }
}

// ---- doubling ------------------------------------------------------------------------------------------------------------------
// scratch slots: 0 r, 1 t, 2 W = r t, 3 G, 4 tt, then (rdot_p, tdot_p) for p = 0 .. P-1   (X1, Q2, Y never leave the registers:
// three live strips is what a wave holds at eight row tiles, and the phases between the A-form switches need no more)
constexpr int DL_FIXED = 5;
// ST: the element type of the caller's arrays (double, or float: Float32 runs -- storage in single, arithmetic in double)
template <int RT, typename ST>
__global__ __launch_bounds__(64 * RT, RT <= 4 ? 2 : 1) void k_dbl128_lin(int N, int ns, int S, int P, int nd,
                                                                         ST* __restrict__ expk_g, ST* __restrict__ ekl_g,
                                                                         added<ST> a, added_lin<ST> al,
                                                                         d4_t* __restrict__ scr, int* __restrict__ status) {
  constexpr int NP = 16 * RT;
  constexpr bool K = keep128<RT>::value;
  extern __shared__ __attribute__((aligned(16))) double lds128[];
  double* AF = lds128;
  float* red = reinterpret_cast<float*>(lds128 + NP * NP);
  double* gjs = lds128 + NP * NP + 32;
  double* vt = gjs + 128;
  double *jp = vt, *jm = vt + NP, *J1p = vt + 2 * NP, *J1m = vt + 3 * NP, *Av = vt + 4 * NP, *Bv = vt + 5 * NP;
  double *xt = vt + 6 * NP;                       // x_0 | x_1 of the running rider pass (2 NP)
  double *y0 = vt + 8 * NP, *y1 = vt + 9 * NP;    // its result
  double *aJp = vt + 12 * NP, *aJm = vt + 13 * NP, *aJ1p = vt + 14 * NP, *aJ1m = vt + 15 * NP;
  double *ra = vt + 16 * NP, *rb = vt + 17 * NP;  // r aJ+ , r aJ1-  ->  v, u
  double *aJpn = vt + 18 * NP, *aJmn = vt + 19 * NP;
  const inv128_ctx icx{AF, gjs, status};
  bpos<RT> p(lds_addr128(AF), N);
  double* xw = vt + lin128<RT>::NVEC * NP + lin128<RT>::XW * p.wave;
  const int nw = blockDim.x >> 6, tid = threadIdx.x;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  const int nslots = DL_FIXED + 2 * P;
  int slot = 0;
  d4_t* const wbase = slot_base<RT>(scr, nslots, p);
  auto sl = [&](int i) { return slot_ptr<RT>(wbase, i); };

  for (int s = blockIdx.x; s < S; s += gridDim.x) {
    double k = expk_g[s];
    bstrip<RT> r_s, t_s;                           // (K: the state; else scratch copies that die at once)
    {
      ldg(r_s, a.r_mp + NN * s, N, p, xw);
      if constexpr (!K) spill(sl(0), r_s, p);
      ldg(t_s, a.t_pp + NN * s, N, p, xw);
      if constexpr (!K) spill(sl(1), t_s, p);
      bstrip<RT> x;
      for (int pp = 0; pp < P; ++pp) {
        ldg(x, al.ap_r_mp + pp * MS + NN * s, N, p, xw);
        spill(sl(DL_FIXED + 2 * pp), x, p);
        ldg(x, al.ap_t_pp + pp * MS + NN * s, N, p, xw);
        spill(sl(DL_FIXED + 2 * pp + 1), x, p);
      }
    }
    for (int i = tid; i < NP; i += blockDim.x) {
      jp[i] = i < N ? a.j0_p[(long long)N * s + i] : 0.0;
      jm[i] = i < N ? a.j0_m[(long long)N * s + i] : 0.0;
    }
    __syncthreads();

    for (int n = 0; n < nd; ++n) {
      // ================= forward: E, W, G, tt =================  (rt_helpers.jl:102-166; sources :128-134)
      {
        if constexpr (!K) fill(r_s, sl(0), p);
        store_af(r_s, N, p);
        for (int i = tid; i < NP; i += blockDim.x) {
          J1p[i] = jp[i] * k;
          J1m[i] = jm[i] * k;
          xt[i] = jp[i];
          xt[NP + i] = jm[i] * k;
        }
        __syncthreads();
        bstrip<RT> E, G;
        E.zero();
        mm128(E, r_s, p);                      // E = r r
        {
          bstrip<RT> W;
          W.zero();
          mm_held<K, RT>(W, t_s, sl(1), p);    // W = r t
          spill(sl(2), W, p);
        }
        rider2(y0, y1, xt, p);                 // r j0+ | r j1-
        const double nrm = norm128(E, N, nw, red, slot, p);   // (barrier: [r] is free, y0 / y1 complete)
        for (int i = tid; i < NP; i += blockDim.x) {
          Av[i] = J1m[i] + y0[i];              // A = j1- + r j0+
          Bv[i] = jp[i] + y1[i];               // B = j0+ + r j1-
        }
        invert128(inv_order128(nrm, status), E, G, N, icx, p);
        spill(sl(3), G, p);
        if constexpr (!K) fill(t_s, sl(1), p);
        __syncthreads();                       // [E] no longer read
        store_af(t_s, N, p);
        __syncthreads();
        bstrip<RT> tt;
        tt.zero();
        mm128(tt, G, p);                       // tt = t G
        spill(sl(4), tt, p);
      }
      __syncthreads();                         // [t] no longer read

      // ================= parameters ================= (doubling_lin.jl:216-339; Gdot eliminated: t Gdot = tt X1 G)
      for (int pp = 0; pp < P; ++pp) {
        const double kl = ekl_g[s + (long long)S * pp];
        const int iRd = DL_FIXED + 2 * pp, iTd = iRd + 1;
        ST* g_aJp = al.ap_J0_p + pp * VS + (long long)N * s;
        ST* g_aJm = al.ap_J0_m + pp * VS + (long long)N * s;
        bstrip<RT> rd, X1, Q2;                 // X1, Q2 live from here to the [tt] phase at every RT; K: rd too
        // ---- [r]: X1 = r rdot ; Q2 = r tdot ; riders r aJ+ | r aJ1-
        if constexpr (!K) fill(r_s, sl(0), p);
        store_af(r_s, N, p);                   // (the A-form is free: barrier above / at the end of the previous parameter)
        for (int i = tid; i < NP; i += blockDim.x) {
          const double vp = i < N ? g_aJp[i] : 0.0, vm = i < N ? g_aJm[i] : 0.0;
          aJp[i] = vp;
          aJm[i] = vm;
          aJ1p[i] = vp * k + jp[i] * kl;       // aJ1+- = aJ+- expk + j0+- ekl_p
          const double v1m = vm * k + jm[i] * kl;
          aJ1m[i] = v1m;
          xt[i] = vp;
          xt[NP + i] = v1m;
        }
        __syncthreads();
        {
          fill(rd, sl(iRd), p);
          X1.zero();
          mm128(X1, rd, p);                    // X1 = r rdot
          bstrip<RT> B;
          fill(B, sl(iTd), p);
          Q2.zero();
          mm128(Q2, B, p);                     // Q2 = r tdot
          rider2(ra, rb, xt, p);
        }
        // ---- [rdot]: X1 += rdot r ; Q2 += rdot t ; riders rdot j0+ | rdot j1-
        if constexpr (!K) fill(rd, sl(iRd), p);
        __syncthreads();                       // [r] no longer read
        store_af(rd, N, p);
        for (int i = tid; i < NP; i += blockDim.x) {
          xt[i] = jp[i];
          xt[NP + i] = J1m[i];
        }
        __syncthreads();
        {
          mm_held<K, RT>(X1, r_s, sl(0), p);   // X1 = r rdot + rdot r
          mm_held<K, RT>(Q2, t_s, sl(1), p);   // Q2 = r tdot + rdot t
          rider2(y0, y1, xt, p);
        }
        // ---- [tt]: Y = tdot + tt X1 ; rdot' = rdot + tt Q2 ; tdot' = tt tdot ; riders tt v | tt u
        {
          bstrip<RT> tt;
          fill(tt, sl(4), p);
          __syncthreads();                     // [rdot] no longer read, y complete
          store_af(tt, N, p);
        }
        for (int i = tid; i < NP; i += blockDim.x) {
          const double v = aJ1m[i] + y0[i] + ra[i];   // v = aJ1- + rdot j0+ + r aJ+
          const double u = aJp[i] + y1[i] + rb[i];    // u = aJ+ + rdot j1- + r aJ1-
          xt[i] = v;
          xt[NP + i] = u;
        }
        __syncthreads();
        bstrip<RT> Y, tdn;                     // (rdot' accumulates in rd)
        {
          fill(Y, sl(iTd), p);
          mm128(Y, X1, p);                     // Y = tdot + tt X1   (stays in registers until it becomes the A-form [Y])
          if constexpr (!K) fill(rd, sl(iRd), p);
          mm128(rd, Q2, p);                    // rdot' = rdot + tt Q2 (+ ttdot W below)
          if constexpr (!K) spill(sl(iRd), rd, p);
          bstrip<RT> B;
          fill(B, sl(iTd), p);
          tdn.zero();
          mm128(tdn, B, p);                    // tdot' = tt tdot (+ ttdot t below)
          if constexpr (!K) spill(sl(iTd), tdn, p);
          rider2(y0, y1, xt, p);
        }
        // ---- [Y]: ttdot = Y G
        __syncthreads();                       // [tt] no longer read, y complete
        store_af(Y, N, p);
        for (int i = tid; i < NP; i += blockDim.x) {
          aJmn[i] = aJm[i] + y0[i];            // aJ-' = aJ- + tt v (+ ttdot A below)
          aJpn[i] = aJ1p[i] + y1[i];           // aJ+' = aJ1+ + tt u (+ ttdot B below)
          xt[i] = Av[i];
          xt[NP + i] = Bv[i];
        }
        __syncthreads();
        bstrip<RT> ttl;
        {
          bstrip<RT> Gs;
          fill(Gs, sl(3), p);
          ttl.zero();
          mm128(ttl, Gs, p);
        }
        // ---- [ttdot]: rdot' += ttdot W ; tdot' += ttdot t ; riders ttdot A | ttdot B
        __syncthreads();                       // [Y] no longer read
        store_af(ttl, N, p);
        __syncthreads();
        {
          bstrip<RT> B;
          if constexpr (!K) fill(rd, sl(iRd), p);
          fill(B, sl(2), p);
          mm128(rd, B, p);
          spill(sl(iRd), rd, p);               // rdot' = rdot + tt Q2 + ttdot rt
          if constexpr (!K) fill(tdn, sl(iTd), p);
          mm_held<K, RT>(tdn, t_s, sl(1), p);
          spill(sl(iTd), tdn, p);              // tdot' = tt tdot + ttdot t
          rider2(y0, y1, xt, p);
        }
        __syncthreads();                       // [ttdot] free, y complete
        if (tid < N) {
          g_aJm[tid] = (ST)(aJmn[tid] + y0[tid]);
          g_aJp[tid] = (ST)(aJpn[tid] + y1[tid]);
        }
        if (tid == 0) ekl_g[s + (long long)S * pp] = (ST)(2.0 * k * kl);      // (k_ekl_step, before expk is squared)
      }

      // ================= forward: r' = r + tt W ; t' = tt t ; sources (after the parameters: they read the old r, t) ==========
      {
        bstrip<RT> tt;
        fill(tt, sl(4), p);
        store_af(tt, N, p);                    // (the A-form is free: barrier above)
      }
      for (int i = tid; i < NP; i += blockDim.x) {
        xt[i] = Av[i];
        xt[NP + i] = Bv[i];
      }
      __syncthreads();
      {
        bstrip<RT> B, tn;
        if constexpr (!K) fill(r_s, sl(0), p);
        fill(B, sl(2), p);
        mm128(r_s, B, p);                      // r' = r + tt W
        if constexpr (!K) spill(sl(0), r_s, p);
        tn.zero();
        mm_held<K, RT>(tn, t_s, sl(1), p);     // t' = tt t
        if constexpr (K) t_s = tn; else spill(sl(1), tn, p);
        rider2(y0, y1, xt, p);                 // tt A | tt B
      }
      __syncthreads();                         // [tt] free, y complete
      for (int i = tid; i < NP; i += blockDim.x) {
        jm[i] = jm[i] + y0[i];                 // j0-' = j0- + tt A
        jp[i] = J1p[i] + y1[i];                // j0+' = j1+ + tt B
      }
      k = k * k;
      __syncthreads();
    }

    // ---- out: apply_D! with derivative slots (doubling_lin.jl:374-421) when ns > 0, else the plain state
    {
      const bool uj = ns > 0 && is_uv_row(min(p.col, N - 1), ns);
      auto put = [&](const bstrip<RT>& x, ST* g_f, ST* g_b, bool flip) {
        // g_f = x with its U / V rows negated (flip), g_b = D g_f D ... as k_apply_D_lin
        bstrip<RT> f, bb;
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool ui = ns > 0 && is_uv_row(p.row(ta, r), ns);
            const double v = (flip && ui) ? -x.v[ta][r] : x.v[ta][r];
            f.v[ta][r] = v;
            bb.v[ta][r] = (ui == uj) ? v : -v;
          }
        stg(g_f + NN * s, f, N, p, xw);
        if (ns > 0) stg(g_b + NN * s, bb, N, p, xw);
      };
      if constexpr (!K) fill(r_s, sl(0), p);
      put(r_s, a.r_mp, a.r_pm, true);
      if constexpr (!K) fill(t_s, sl(1), p);
      put(t_s, a.t_pp, a.t_mm, false);
      bstrip<RT> x;
      for (int pp = 0; pp < P; ++pp) {
        fill(x, sl(DL_FIXED + 2 * pp), p);
        put(x, al.ap_r_mp + pp * MS, al.ap_r_pm + pp * MS, true);
        fill(x, sl(DL_FIXED + 2 * pp + 1), p);
        put(x, al.ap_t_pp + pp * MS, al.ap_t_mm + pp * MS, false);
      }
      if (ns > 0) {                            // the slots of the inactive parameters are zero (as k_dbl_lin_multi)
        x.zero();
        for (int pp = P; pp < al.P; ++pp) {
          stg(al.ap_r_pm + pp * MS + NN * s, x, N, p, xw);
          stg(al.ap_t_mm + pp * MS + NN * s, x, N, p, xw);
        }
      }
      if (tid < N) {
        const double sg = (ns > 0 && is_uv_row(tid, ns)) ? -1.0 : 1.0;
        a.j0_p[(long long)N * s + tid] = (ST)jp[tid];
        a.j0_m[(long long)N * s + tid] = (ST)(sg * jm[tid]);
        if (ns > 0 && sg < 0.0)
          for (int pp = 0; pp < P; ++pp) {
            ST* g = al.ap_J0_m + pp * VS + (long long)N * s + tid;
            *g = -*g;                          // (written by this very thread in the last step)
          }
      }
      if (tid == 0) expk_g[s] = (ST)k;
    }
    __syncthreads();   // the next point overwrites the tables and the A-form
  }
}

// ---- interaction half ---------------------------------------------------------------------------------------------------------
// operand bindings of one half
template <typename ST>
struct ia128_half {
  const ST *LA, *ER, *LT, *S2, *S3, *ACC0;
  long long sLA, sER, sLT, sS2, sS3, sACC0;
  ST *OUT0, *OUT1;
  const ST *VR, *VADD, *VACC;
  ST* VOUT;
  const ST *PA, *D1, *D2, *YI, *ACCP, *D3;
  long long sPA, pPA, sD1, pD1, sD2, pD2, sYI, pYI, sACCP, pACCP, sD3, pD3;
  ST *OUTP0, *OUTP1;
  const ST *VDR, *VDADD, *VDACC;
  ST* VDOUT;
};
// scratch slots: 0 ER, 1 S2, 2 S3, 3 rt, 4 G, 5 tt, 6 outp0, 7 outp1   (X1, X2, Y never leave the registers; K, four row tiles:
// neither do ER, S2 -- for the whole point -- and the two outputs of a parameter)
constexpr int IL_SLOTS = 8;
template <int RT, typename ST>
__global__ __launch_bounds__(64 * RT, RT <= 4 ? 2 : 1) void k_ia128_lin(int N, int S, int P, ia128_half<ST> h, d4_t* __restrict__ scr,
                                                                        int* __restrict__ status) {
  constexpr int NP = 16 * RT;
  constexpr bool K = keep128<RT>::value;
  extern __shared__ __attribute__((aligned(16))) double lds128[];
  double* AF = lds128;
  float* red = reinterpret_cast<float*>(lds128 + NP * NP);
  double* gjs = lds128 + NP * NP + 32;
  double* vt = gjs + 128;
  double *vr = vt, *vadd = vt + NP, *vacc = vt + 2 * NP, *rtv = vt + 3 * NP;
  double *xt = vt + 4 * NP;                       // x_0 | x_1 (2 NP)
  double *y0 = vt + 6 * NP, *y1 = vt + 7 * NP;
  double *vdr = vt + 8 * NP, *vdadd = vt + 9 * NP, *vdacc = vt + 10 * NP, *pv = vt + 11 * NP, *vd = vt + 13 * NP;
  const inv128_ctx icx{AF, gjs, status};
  bpos<RT> p(lds_addr128(AF), N);
  double* xw = vt + lin128<RT>::NVEC * NP + lin128<RT>::XW * p.wave;
  const int nw = blockDim.x >> 6, tid = threadIdx.x;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  int slot = 0;
  d4_t* const wbase = slot_base<RT>(scr, IL_SLOTS, p);
  auto sl = [&](int i) { return slot_ptr<RT>(wbase, i); };

  for (int s = blockIdx.x; s < S; s += gridDim.x) {
    // ================= forward =================
    bstrip<RT> er, s2;                             // (K: live for the whole point)
    {
      ldg(er, h.ER + s * h.sER, N, p, xw);
      ldg(s2, h.S2 + s * h.sS2, N, p, xw);
      stage_af(AF, h.LA + s * h.sLA, N, nw, p);
      for (int i = tid; i < NP; i += blockDim.x) {
        const long long o = (long long)N * s + i;
        const double v = i < N ? h.VR[o] : 0.0;
        vr[i] = v;
        vadd[i] = i < N ? h.VADD[o] : 0.0;
        vacc[i] = i < N ? h.VACC[o] : 0.0;
        xt[i] = v;
        xt[NP + i] = v;
      }
      if constexpr (!K) {
        spill(sl(0), er, p);
        spill(sl(1), s2, p);
      }
      __syncthreads();
      bstrip<RT> E, G;
      E.zero();
      mm128(E, er, p);                         // E = LA ER
      {
        bstrip<RT> rt;
        rt.zero();
        mm128(rt, s2, p);                      // rt = LA S2
        spill(sl(3), rt, p);
      }
      rider2(y0, y1, xt, p);                   // LA VR
      {
        bstrip<RT> s3;                         // (parked now: the second half's OUT1 overwrites S3's array)
        ldg(s3, h.S3 + s * h.sS3, N, p, xw);
        spill(sl(2), s3, p);
      }
      const double nrm = norm128(E, N, nw, red, slot, p);     // (barrier: [LA] free, y complete)
      for (int i = tid; i < NP; i += blockDim.x) rtv[i] = vadd[i] + y0[i];
      invert128(inv_order128(nrm, status), E, G, N, icx, p);
      spill(sl(4), G, p);
      __syncthreads();                         // [E] no longer read
      stage_af(AF, h.LT + s * h.sLT, N, nw, p);
      __syncthreads();
      bstrip<RT> tt;
      tt.zero();
      mm128(tt, G, p);                         // tt = LT G
      spill(sl(5), tt, p);
    }
    __syncthreads();                           // [LT] no longer read

    // ================= parameters =================
    for (int pp = 0; pp < P; ++pp) {
      bstrip<RT> X1, X2;                       // live to the [tt] phase
      // ---- [PA]: X1 = PA ER ; X2 = PA S2 ; rider PA VR
      stage_af(AF, h.PA + s * h.sPA + pp * h.pPA, N, nw, p);
      for (int i = tid; i < NP; i += blockDim.x) {
        const long long o = (long long)pp * VS + (long long)N * s + i;
        vdr[i] = i < N ? h.VDR[o] : 0.0;
        vdadd[i] = i < N ? h.VDADD[o] : 0.0;
        vdacc[i] = i < N ? h.VDACC[o] : 0.0;
        xt[i] = vr[i];
        xt[NP + i] = vr[i];
      }
      __syncthreads();
      {
        X1.zero();
        mm_held<K, RT>(X1, er, sl(0), p);      // X1 = PA ER
        X2.zero();
        mm_held<K, RT>(X2, s2, sl(1), p);      // X2 = PA S2
        rider2(pv, y1, xt, p);
      }
      // ---- [LA]: X1 += LA D1 ; X2 += LA D2 ; rider LA VDR
      __syncthreads();                         // [PA] no longer read
      stage_af(AF, h.LA + s * h.sLA, N, nw, p);
      for (int i = tid; i < NP; i += blockDim.x) {
        xt[i] = vdr[i];
        xt[NP + i] = vdr[i];
      }
      __syncthreads();
      {
        bstrip<RT> B;
        ldg(B, h.D1 + s * h.sD1 + pp * h.pD1, N, p, xw);
        mm128(X1, B, p);                       // X1 = PA ER + LA D1
        ldg(B, h.D2 + s * h.sD2 + pp * h.pD2, N, p, xw);
        mm128(X2, B, p);                       // X2 = PA S2 + LA D2
        rider2(y0, y1, xt, p);
      }
      // ---- [tt]: Y = YI + tt X1 ; outp0 = ACCP + tt X2 ; outp1 = tt D3 ; rider tt x2v
      {
        bstrip<RT> tt;
        fill(tt, sl(5), p);
        __syncthreads();                       // [LA] no longer read, y complete
        store_af(tt, N, p);
      }
      for (int i = tid; i < NP; i += blockDim.x) {
        const double v = vdadd[i] + pv[i] + y0[i];   // X2[:, c] = PA VR + LA VDR + VDADD
        xt[i] = v;
        xt[NP + i] = v;
      }
      __syncthreads();
      bstrip<RT> Y, o0, o1;
      {
        ldg(Y, h.YI + s * h.sYI + pp * h.pYI, N, p, xw);
        mm128(Y, X1, p);                       // Y = YI + tt X1   (stays in registers until it becomes the A-form [Y])
        ldg(o0, h.ACCP + s * h.sACCP + pp * h.pACCP, N, p, xw);
        mm128(o0, X2, p);                      // outp0 = ACCP + tt X2 (+ ttdot rt below)
        if constexpr (!K) spill(sl(6), o0, p);
        bstrip<RT> B;
        ldg(B, h.D3 + s * h.sD3 + pp * h.pD3, N, p, xw);
        o1.zero();
        mm128(o1, B, p);                       // outp1 = tt D3 (+ ttdot S3 below)
        if constexpr (!K) spill(sl(7), o1, p);
        rider2(y0, y1, xt, p);
      }
      // ---- [Y]: ttdot = Y G
      __syncthreads();                         // [tt] no longer read, y complete
      store_af(Y, N, p);
      for (int i = tid; i < NP; i += blockDim.x) {
        vd[i] = vdacc[i] + y0[i];              // VDACC + tt x2v (+ ttdot rtv below)
        xt[i] = rtv[i];
        xt[NP + i] = rtv[i];
      }
      __syncthreads();
      bstrip<RT> ttl;
      {
        bstrip<RT> Gs;
        fill(Gs, sl(4), p);
        ttl.zero();
        mm128(ttl, Gs, p);
      }
      // ---- [ttdot]: outp0 += ttdot rt ; outp1 += ttdot S3 ; rider ttdot rtv
      __syncthreads();                         // [Y] no longer read
      store_af(ttl, N, p);
      __syncthreads();
      {
        bstrip<RT> B;
        if constexpr (!K) fill(o0, sl(6), p);
        fill(B, sl(3), p);
        mm128(o0, B, p);
        stg(h.OUTP0 + (long long)pp * MS + NN * s, o0, N, p, xw);
        if constexpr (!K) fill(o1, sl(7), p);
        fill(B, sl(2), p);
        mm128(o1, B, p);
        stg(h.OUTP1 + (long long)pp * MS + NN * s, o1, N, p, xw);
        rider2(y0, y1, xt, p);
      }
      __syncthreads();                         // [ttdot] free, y complete
      if (tid < N) h.VDOUT[(long long)pp * VS + (long long)N * s + tid] = (ST)(vd[tid] + y0[tid]);
    }

    // ================= forward outputs (last: OUT0 / OUT1 alias operands of the parameter loop) =================
    {
      bstrip<RT> tt;
      fill(tt, sl(5), p);
      store_af(tt, N, p);                      // (barrier above / after the [LT] product: the A-form is free)
    }
    for (int i = tid; i < NP; i += blockDim.x) {
      xt[i] = rtv[i];
      xt[NP + i] = rtv[i];
    }
    __syncthreads();
    {
      bstrip<RT> acc, B;
      ldg(acc, h.ACC0 + s * h.sACC0, N, p, xw);
      fill(B, sl(3), p);
      mm128(acc, B, p);                        // out0 = ACC0 + tt rt
      stg(h.OUT0 + NN * s, acc, N, p, xw);
      fill(B, sl(2), p);
      acc.zero();
      mm128(acc, B, p);                        // out1 = tt S3
      stg(h.OUT1 + NN * s, acc, N, p, xw);
      rider2(y0, y1, xt, p);
    }
    __syncthreads();
    if (tid < N) h.VOUT[(long long)N * s + tid] = (ST)(vacc[tid] + y0[tid]);
    __syncthreads();   // the next point overwrites the tables and the A-form
  }
}

template <int RT, typename ST>
int launch_dbl128_lin(int N, int ns, int S, int P, int nd, ST* expk, ST* ekl, const added<ST>& a,
                      const added_lin<ST>& al, hipStream_t st) {
  constexpr size_t lds = lin128<RT>::lds_bytes();
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl128_lin<RT, ST>), lds, "hipFuncSetAttribute(k_dbl128_lin)"))
    return prepared;
  const int per_cu = wg_per_cu<RT>::value;
  const int grid = S < per_cu * cu_count() ? S : per_cu * cu_count();
  const int nslots = DL_FIXED + 2 * P;
  d4_t* scr = static_cast<d4_t*>(scratch((size_t)grid * nslots * B_MAXW * RT * 64 * sizeof(d4_t), 3, st));
  int* status = device_status();
  if (!scr || !status) return VSM_ERR_HIP;
  hipLaunchKernelGGL((k_dbl128_lin<RT, ST>), dim3(grid), dim3(64 * RT), lds, st, N, ns, S, P, nd, expk, ekl, a, al, scr, status);
  VSM_LAUNCH_CHECK("k_dbl128_lin");
  return VSM_OK;
}
template <int RT, typename ST>
int launch_ia128_lin(int N, int S, int P, const ia128_half<ST>& h, hipStream_t st) {
  constexpr size_t lds = lin128<RT>::lds_bytes();
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_ia128_lin<RT, ST>), lds, "hipFuncSetAttribute(k_ia128_lin)"))
    return prepared;
  const int per_cu = wg_per_cu<RT>::value;
  const int grid = S < per_cu * cu_count() ? S : per_cu * cu_count();
  d4_t* scr = static_cast<d4_t*>(scratch((size_t)grid * IL_SLOTS * B_MAXW * RT * 64 * sizeof(d4_t), 3, st));
  int* status = device_status();
  if (!scr || !status) return VSM_ERR_HIP;
  hipLaunchKernelGGL((k_ia128_lin<RT, ST>), dim3(grid), dim3(64 * RT), lds, st, N, S, P, h, scr, status);
  VSM_LAUNCH_CHECK("k_ia128_lin");
  return VSM_OK;
}

}  // namespace

// FP64, 60 < N <= 128 (N = 61 .. 64 on four row tiles: the strips of vsm_striplin.hip have no spare column left there)
// Which shapes land here.  The interaction: every N <= 128 (vsm_striplin.hip's k_ia_lin_half, 0.39 of peak for two rounds, lost to
// k_ia128_lin<4> on the shapes both took -- 0.867 -> 0.726 ms per half at 2048 points of the C2 shape -- and is gone).  The doubling:
// beyond the reach of vsm_striplin.hip's k_dbl_lin_multi (49 <= N <= 60 stays there: 12.99 vs 12.61 10^3 points/s on the C2 shape)
// and N <= 48; the thresholds are build parameters so that tools/variants_lin128.sh can re-measure.
#ifndef VSM_LIN128_DBL_MIN
#define VSM_LIN128_DBL_MIN 60
#endif
// ... and the shapes of one to three row tiles (N <= 48), where vsm_striplin.hip pads to 64 rows (N = 30: 27.7 -> 84.3 10^3 points/s,
// N = 48: 20.0 -> 23.4, profiles/r04/shape_sweep_f64.txt)
#ifndef VSM_LIN128_SMALL_MAX
#define VSM_LIN128_SMALL_MAX 48
#endif
static bool lin128_small(int N) { return N >= 1 && N <= VSM_LIN128_SMALL_MAX; }
bool strip128_lin_dbl_supported(int N) { return (N > VSM_LIN128_DBL_MIN && N <= 128) || lin128_small(N); }


// All ndoubl doubling steps (forward + P active parameters) in one launch, apply_D! included when ns (n_stokes) > 0
template <typename ST>
int strip128_doubling_lin(int N, int S, int P, int nd, int ns, ST* expk, ST* ekl, const added<ST>& a,
                          const added_lin<ST>& al, hipStream_t st) {
  // (Float32 arrays: every N <= 128 -- there is no other fused linearized kernel for them)
  const bool shape_ok = sizeof(ST) == 4 ? (N >= 1 && N <= 128) : strip128_lin_dbl_supported(N);
  if (!shape_ok || P < 0 || nd < 1 || a.mat_stride != (long long)N * N || al.mat_stride != (long long)N * N)
    return VSM_ERR_UNSUPPORTED;
  if (S <= 0) return VSM_OK;
  switch ((N + 15) / 16) {
    case 1: return launch_dbl128_lin<1, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 2: return launch_dbl128_lin<2, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 3: return launch_dbl128_lin<3, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 4: return launch_dbl128_lin<4, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 5: return launch_dbl128_lin<5, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 6: return launch_dbl128_lin<6, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 7: return launch_dbl128_lin<7, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
    case 8: return launch_dbl128_lin<8, ST>(N, ns, S, P, nd, expk, ekl, a, al, st);
  }
  return VSM_ERR_UNSUPPORTED;
}

// ScatteringInterface_11 interaction with derivatives: two launches (first half, second half), bindings as
// vsm_striplin.hip's strip_interaction11_lin
template <typename ST>
int strip128_interaction11_lin(int N, int S, const composite<ST>& c, const composite_lin<ST>& cl, const added<ST>& a,
                               const added_lin<ST>& al, hipStream_t st) {
  if (N < 1 || N > 128) return VSM_ERR_UNSUPPORTED;
  if (S <= 0) return VSM_OK;
  const int P = cl.P;
  const long long NN = (long long)N * N, MS = NN * S;
  const long long as = a.mat_stride, als = al.mat_stride, alp = (als == 0) ? NN : MS;
  ia128_half<ST> h1{};
  h1.LA = a.r_mp;   h1.sLA = as;
  h1.ER = c.R_pm;   h1.sER = NN;
  h1.LT = c.T_mm;   h1.sLT = NN;
  h1.S2 = c.T_pp;   h1.sS2 = NN;
  h1.S3 = a.t_mm;   h1.sS3 = as;
  h1.ACC0 = c.R_mp; h1.sACC0 = NN;
  h1.OUT0 = c.R_mp; h1.OUT1 = c.T_mm;
  h1.VR = c.J0_p; h1.VADD = a.j0_m; h1.VACC = c.J0_m; h1.VOUT = c.J0_m;
  h1.PA = al.ap_r_mp; h1.sPA = als; h1.pPA = alp;
  h1.D1 = cl.R_pm;    h1.sD1 = NN;  h1.pD1 = MS;
  h1.D2 = cl.T_pp;    h1.sD2 = NN;  h1.pD2 = MS;
  h1.YI = cl.T_mm;    h1.sYI = NN;  h1.pYI = MS;
  h1.ACCP = cl.R_mp;  h1.sACCP = NN; h1.pACCP = MS;
  h1.D3 = al.ap_t_mm; h1.sD3 = als; h1.pD3 = alp;
  h1.OUTP0 = cl.R_mp; h1.OUTP1 = cl.T_mm;
  h1.VDR = cl.J0_p; h1.VDADD = al.ap_J0_m; h1.VDACC = cl.J0_m; h1.VDOUT = cl.J0_m;
  ia128_half<ST> h2{};
  h2.LA = c.R_pm;   h2.sLA = NN;
  h2.ER = a.r_mp;   h2.sER = as;
  h2.LT = a.t_pp;   h2.sLT = as;
  h2.S2 = a.t_mm;   h2.sS2 = as;
  h2.S3 = c.T_pp;   h2.sS3 = NN;
  h2.ACC0 = a.r_pm; h2.sACC0 = as;
  h2.OUT0 = c.R_pm; h2.OUT1 = c.T_pp;
  h2.VR = a.j0_m; h2.VADD = c.J0_p; h2.VACC = a.j0_p; h2.VOUT = c.J0_p;
  h2.PA = cl.R_pm;    h2.sPA = NN;  h2.pPA = MS;
  h2.D1 = al.ap_r_mp; h2.sD1 = als; h2.pD1 = alp;
  h2.D2 = al.ap_t_mm; h2.sD2 = als; h2.pD2 = alp;
  h2.YI = al.ap_t_pp; h2.sYI = als; h2.pYI = alp;
  h2.ACCP = al.ap_r_pm; h2.sACCP = als; h2.pACCP = alp;
  h2.D3 = cl.T_pp;    h2.sD3 = NN;  h2.pD3 = MS;
  h2.OUTP0 = cl.R_pm; h2.OUTP1 = cl.T_pp;
  h2.VDR = al.ap_J0_m; h2.VDADD = cl.J0_p; h2.VDACC = al.ap_J0_p; h2.VDOUT = cl.J0_p;
  int rc;
  switch ((N + 15) / 16) {
#define VSM_CASE(RT)                                                  \
  case RT:                                                            \
    if ((rc = launch_ia128_lin<RT, ST>(N, S, P, h1, st))) return rc;  \
    return launch_ia128_lin<RT, ST>(N, S, P, h2, st);
    VSM_CASE(1)
    VSM_CASE(2)
    VSM_CASE(3)
    VSM_CASE(4)
    VSM_CASE(5)
    VSM_CASE(6)
    VSM_CASE(7)
    VSM_CASE(8)
#undef VSM_CASE
  }
  return VSM_ERR_UNSUPPORTED;
}

template int strip128_doubling_lin<double>(int, int, int, int, int, double*, double*, const added<double>&, const added_lin<double>&,
                                           hipStream_t);
template int strip128_doubling_lin<float>(int, int, int, int, int, float*, float*, const added<float>&, const added_lin<float>&,
                                          hipStream_t);
template int strip128_interaction11_lin<double>(int, int, const composite<double>&, const composite_lin<double>&, const added<double>&,
                                                const added_lin<double>&, hipStream_t);
template int strip128_interaction11_lin<float>(int, int, const composite<float>&, const composite_lin<float>&, const added<float>&,
                                               const added_lin<float>&, hipStream_t);

}  // namespace vsm
