// FP64 column-strip kernels for 64 < N <= 128 (the reference's own baseline shapes live here: Natraj N = 108,
// test/test_CoreRT.jl:110-157; VLIDORT case A N = 112, test/vlidort_baseline/cases/case_A_siewert2000.jl:29-50).
//
//   k_dbl128<RT>   doubling! (src/CoreRT/CoreKernel/doubling.jl:38-99, rt_helpers.jl:102-166, apply_D doubling.jl:178-252):
//                  the whole doubling loop of one spectral point in one workgroup, in place on the AddedLayer
//   k_ia128<RT>    interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266), one inverse, in place on the composite
//   k_inv1m128<RT> (I - A B)^-1, the inverse of the operator chains (linearized / Raman runs) at these shapes
//
// Scheme (vsm_strip.hip's, re-dimensioned): a matrix is padded to NP = 16 RT rows (RT = 5..8 row tiles); wave w owns the
// 16-column strip w of every operator as RT accumulator tiles of v_mfma_f64_16x16x4 -- the accumulator layout IS the B-operand
// layout, so a strip is the right operand of the next product without any movement.  The left operand is an "A-form" of the
// matrix in LDS that all waves read.  One A-form of 128 x 128 doubles is 128 KB: there is room for ONE (vsm_strip.hip keeps two
// of 32 KB), and a wave's 256 registers hold THREE strips of 64, not five.  So
//   * the products of a step are ordered by their left operand, [r] -> [E] -> [t] -> [tt], each A-form written once from the
//     strips (two barriers), every product of that operand running off it:
//         W = r t ; E = r r ;  G = (I - E)^-1 (Neumann series off [E]) ;  tt = t G ;  r' = r + tt W ;  t' = tt t
//   * the strips that are not an operand of the running phase (r, t, W) wait in a per-workgroup global scratch (lane-linear,
//     2 KB per instruction, L2 / MALL resident: 4 stores + 5 loads of 16 KB per wave and step against >= 6 x 32 RT MFMAs).
// Workgroups are persistent (one per CU, 1 + RT waves at most: two waves per SIMD) and walk the spectral axis.
// Source vectors: N <= 80 -- two spare columns cb, cb + 1 of the strips (cb = N rounded up to even), exactly as in vsm_strip.hip;
// N > 80 -- each wave forms one more 16 x 16 tile per product that carries vectors: row tile w of [A] (x_0 | x_1), the vectors read
// from an LDS table (mm128r).  That is the only way at N = 127, 128 (eight full strips) and measured 1 ... 5 % faster from N = 96 on
// (no vector-only wave at N = 96 / 112, no rider lane code in the strips' registers); 3 % slower at five row tiles.
//
// A-form layout (verified conflict-free for both directions): block (ks, t) = the 16 x 4 fragment of row tile t and k-step ks,
// 64 doubles; inside a block, element (m, k') of k-step ks sits at word
//     k' << 4 | (m ^ (k' | j << 2)),     j = ks & 3
// so that a fragment read (lanes = (k', m); ds_read_b64 in 32-lane groups, or ds_read2st64_b64 for two row tiles in 16-lane
// groups) covers its part of the 512-byte block linearly, and a strip store (ds_write_b64, 16-lane groups = 16 columns = four
// k-steps x four k': k' | j << 2 is the lane's l15) hits 16 distinct 8-byte slots of a 128-byte bank row.
#include "vsm_strip128_dev.h"

namespace vsm {
#ifdef VSM_PHASE_TIMING   // diagnostic build (make timing; tools/phase_timing128.py): cycles per phase of k_ia128, every workgroup
__device__ unsigned long long vsm_phase_cycles_128[64];   // k_ia128: 0..17, points [31] ; k_dbl128: 32..43, steps [62], points [63]
#define B128_STAMP_DECL                  \
  unsigned long long _bs[20] = {};       \
  unsigned long long _bt = __builtin_readcyclecounter()
#define B128_STAMP(i)                                             \
  do {                                                            \
    const unsigned long long _t = __builtin_readcyclecounter();   \
    _bs[i] += _t - _bt;                                           \
    _bt = _t;                                                     \
  } while (0)
#define B128_STAMP_FLUSH_AT(base, cnt, npoints)                                                   \
  do {                                                                                            \
    if (threadIdx.x == 0) {                                                                       \
      for (int _i = 0; _i < 20; ++_i) atomicAdd(&vsm_phase_cycles_128[(base) + _i], _bs[_i]);     \
      atomicAdd(&vsm_phase_cycles_128[cnt], (unsigned long long)(npoints));                       \
    }                                                                                             \
  } while (0)
#define B128_STAMP_FLUSH(npoints) B128_STAMP_FLUSH_AT(0, 31, npoints)
#else
#define B128_STAMP_DECL
#define B128_STAMP(i)
#define B128_STAMP_FLUSH(n) (void)(n)
#define B128_STAMP_FLUSH_AT(b, c, n) (void)(n)
#endif
namespace {

// ---- doubling --------------------------------------------------------------------------------------------------------------
// MR: the source vectors as an extra MFMA tile per wave (mm128r) instead of rider columns -- N = 127, 128
// ST: the element type of the caller's arrays (double; float = the reference's Float32 runs of 96 < N <= 128: storage in single,
// arithmetic in double)
template <int RT, bool MR, typename ST>
__global__ __launch_bounds__(64 * B_MAXW) void k_dbl128(int N, int ns, int S, int ndoubl, ST* __restrict__ expk_g,
                                                         added<ST> a, d4_t* __restrict__ scr, int* __restrict__ status) {
  constexpr int NP = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) double lds128[];
  double* AF = lds128;
  float* red = reinterpret_cast<float*>(lds128 + NP * NP);
  float* dsg = red + 32;   // D of apply_D: -1 on the U / V rows, +1 elsewhere
  double* xt = lds128 + NP * NP + 128;   // (MR) x_0 = j0+, x_1 = j1- of the [r] phase ; u1, u2 of the [tt] phase ; j0+, j0-
  double* xu = xt + 2 * NP;
  double* vjp = xt + 4 * NP;
  double* vjm = xt + 5 * NP;
  const inv128_ctx icx{AF, lds128 + NP * NP + 128 + (MR ? 6 * NP : 0), status};
  bpos<RT> p(lds_addr128(AF), N);
  const unsigned xb = lds_addr128(xt) + 8u * (unsigned)((p.l15 & 1) * NP + p.kq);
  const int rrow = 16 * p.wave + p.kq;   // (MR) the lane's rows of the rider tile: rrow + 4 r
  for (int i = threadIdx.x; i < NP; i += blockDim.x) dsg[i] = is_uv_row(i, ns) ? -1.f : 1.f;
  const int nw = blockDim.x >> 6;
  const int cb = (N + 1) & ~1;   // rider columns cb, cb + 1: one strip, an even / odd lane pair
  const bool own_wave = p.wave == (cb >> 4);
  const bool laneA = own_wave && p.col == cb, laneB = own_wave && p.col == cb + 1, laneAB = laneA || laneB;
  const long long NN = (long long)N * N;
  d4_t* const sR = scr + ((long long)(blockIdx.x * 3 + 0) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  d4_t* const sT = scr + ((long long)(blockIdx.x * 3 + 1) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  d4_t* const sW = scr + ((long long)(blockIdx.x * 3 + 2) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  int slot = 0;
  B128_STAMP_DECL;
  int npts = 0;

  for (int s = blockIdx.x; s < S; s += gridDim.x) {
    ++npts;
    ST* const g_r = a.r_mp + NN * s;
    ST* const g_t = a.t_pp + NN * s;
    ST* const g_jp = a.j0_p + (long long)N * s;
    ST* const g_jm = a.j0_m + (long long)N * s;
    double expk = expk_g[s];
    bstrip<RT> r_s, t_s;
    load_global128(r_s, g_r, N, p);
    load_global128(t_s, g_t, N, p);
    // riders (rt_helpers.jl:128-134 term for term, see vsm_strip.hip):
    //   t_s[cb] = j0+, t_s[cb+1] = j1- = j0- expk ;  r_s[cb] = j0-, r_s[cb+1] = j0+
    if (!MR && own_wave) {
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r), rc = min(row, N - 1);
          const double vp = row < N ? g_jp[rc] : 0.0, vm = row < N ? g_jm[rc] : 0.0;
          t_s.v[ta][r] = laneAB ? (laneA ? vp : vm * expk) : t_s.v[ta][r];
          r_s.v[ta][r] = laneAB ? (laneA ? vm : vp) : r_s.v[ta][r];
        }
    }
    if constexpr (MR) {   // x_0 = j0+, x_1 = j1- = j0- expk
      for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        const double vp = i < N ? g_jp[i] : 0.0, vm = i < N ? g_jm[i] : 0.0;
        vjp[i] = vp;
        vjm[i] = vm;
        xt[i] = vp;
        xt[NP + i] = vm * expk;
      }
    }
    spill(sT, t_s, p);
    if constexpr (RT >= 8) spill(sR, r_s, p);   // (RT < 8: r stays in registers from one step to the next)
    store_af(r_s, N, p);
    __syncthreads();
    B128_STAMP(0);

    // Register plan: a strip is 8 RT registers, the fragments RT more: three strips live at most (the series; t' = tt t with
    // r' waiting for its A-form store), two in the products that carry the source vectors -- r and W wait in the scratch while they are not an operand.
    for (int n = 0; n < ndoubl; ++n) {
      // on entry: [A] = [r]; t_s in registers; sR = r, sT = t (strips incl. their riders)
      constexpr bool KEEPW = RT <= 6;        // (five row tiles: a fourth strip fits beside the series' three -- W is not parked)
      bstrip<RT> G, Wk;
      {
        {
          bstrip<RT> W;
          W.zero();
          constexpr bool EARLY = RT < 8;     // request parked strips a phase ahead where a fourth strip's registers exist
          if constexpr (MR) {
            d4_t u = acc_zero<double>();
            mm128r(W, t_s, u, xb, p);        // W = r t ; tile: r j0+ | r j1-
            if (p.l15 < 2) {                 // u1 = j1- + r j0+ | u2 = j0+ + r j1-   (read in the [tt] phase, barriers away)
#pragma unroll
              for (int r = 0; r < 4; ++r) xu[p.l15 * NP + rrow + 4 * r] = u[r] + xt[(1 - p.l15) * NP + rrow + 4 * r];
            }
          } else {
            mm128(W, t_s, p);                // W = r t            (riders: r j0+, r j1-)
          }
          if constexpr (!EARLY) fill(r_s, sR, p);
          if (!MR && own_wave) {             // W[cb] += j1- = j0- expk, W[cb+1] += j0+ ; r_s[cb+1] -> j1+ = j0+ expk
            const double fW = laneA ? expk : (laneB ? 1.0 : 0.0), fR = laneB ? expk : 1.0;
#pragma unroll
            for (int ta = 0; ta < RT; ++ta)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                W.v[ta][r] = fma(r_s.v[ta][r], fW, W.v[ta][r]);
                r_s.v[ta][r] *= fR;
              }
          }
          if constexpr (KEEPW) Wk = W; else spill(sW, W, p);
        }
        B128_STAMP(1);
        bstrip<RT> E;
        E.zero();
        mm128(E, r_s, p);                    // E = r r
        spill(sR, r_s, p);                   // (the addend of r' = r + tt W, its rider column scaled)
        B128_STAMP(2);
        const double nrm = norm128(E, N, nw, red, slot, p);   // (its barrier: [r] is free)
        invert128(inv_order128(nrm, status), E, G, N, icx, p);
      }
      B128_STAMP(3);
      constexpr bool EARLY = RT < 8;
      bstrip<RT> W, t2;
      fill(t2, sT, p);
      __syncthreads();                       // [E] no longer read
      store_af(t2, N, p);
      __syncthreads();
      B128_STAMP(4);
      {
        bstrip<RT> tt;
        tt.zero();
        mm128(tt, G, p);                     // tt = t G
        B128_STAMP(5);
        if constexpr (EARLY) {               // (the operands of the [tt] phase: requested across the barriers)
          fill(r_s, sR, p);
          if constexpr (KEEPW) W = Wk; else fill(W, sW, p);
        }
        __syncthreads();                     // [t] no longer read
        store_af(tt, N, p);
      }
      __syncthreads();
      B128_STAMP(6);
      if constexpr (EARLY) {
        fill(t2, sT, p);
      } else {
        fill(r_s, sR, p);
        if constexpr (KEEPW) W = Wk; else fill(W, sW, p);
      }
      {
        if constexpr (MR) {
          d4_t z = acc_zero<double>();
          mm128r(r_s, W, z, xb + 16u * NP, p);   // r' = r + tt W ; tile: tt u1 | tt u2
          if (p.l15 < 2) {                   // j0- += tt u1 | j0+ = j0+ expk + tt u2     (rt_helpers.jl:128-134)
            double* vj = p.l15 ? vjp : vjm;
            const double f = p.l15 ? expk : 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) vj[rrow + 4 * r] = fma(vj[rrow + 4 * r], f, z[r]);
          }
        } else {
          mm128(r_s, W, p);                  // r' = r + tt W      (riders: the new j0-, j0+)
        }
      }
      B128_STAMP(7);
      if constexpr (!EARLY) fill(t2, sT, p);
      t_s.zero();
      mm128(t_s, t2, p);                     // t' = tt t
      B128_STAMP(8);
      expk = expk * expk;
      if (n + 1 < ndoubl) {
        if (!MR && own_wave) {   // t_s[cb] = j0+', t_s[cb+1] = j1-' = j0-' expk'   (from the neighbour lane of r_s)
          const double ft = laneB ? expk : 1.0;
#pragma unroll
          for (int ta = 0; ta < RT; ++ta)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double u = dpp_swap1_128(r_s.v[ta][r]) * ft;
              t_s.v[ta][r] = laneAB ? u : t_s.v[ta][r];
            }
        }
        spill(sT, t_s, p);
        if constexpr (RT >= 8) spill(sR, r_s, p);
        __syncthreads();                     // [tt] no longer read
        store_af(r_s, N, p);
        if constexpr (MR) {
          for (int i = threadIdx.x; i < NP; i += blockDim.x) {
            xt[i] = vjp[i];
            xt[NP + i] = vjm[i] * expk;
          }
        }
        __syncthreads();
        B128_STAMP(9);
      }
    }

    // ---- out: apply_D (doubling.jl:178-252) ----
    if (ndoubl > 0) {
      // r-+ = D r* ; r+- = D r-+ D = r* D ; t-- = D t++ D ; j0- = D j0-*      (products by +-1: exact)
      const bool cok = p.col < N;
      const double sc = (double)dsg[min(p.col, NP - 1)];
      long long o = NN * s + (long long)N * p.col + p.kq;
      asm volatile("" : "+v"(o));
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          const double sr = (double)dsg[row];
          const double rs = r_s.v[ta][r], tv = t_s.v[ta][r], rv = rs * sr;
          const bool rok = ta < RT - 1 || row < N;
          if (rok && cok) {
            const long long e = o + 16 * ta + 4 * r;
            a.r_mp[e] = (ST)rv;
            a.t_pp[e] = (ST)tv;
            a.r_pm[e] = (ST)(rs * sc);
            a.t_mm[e] = (ST)(tv * (sr * sc));
          }
          if (!MR && rok && laneA) g_jm[row] = (ST)rv;   // j0- (sign of apply_D_SFI)
          if (!MR && rok && laneB) g_jp[row] = (ST)rs;   // j0+
        }
      if constexpr (MR) {
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
          g_jp[i] = (ST)vjp[i];
          g_jm[i] = (ST)(vjm[i] * (double)dsg[i]);
        }
      }
      if (threadIdx.x == 0) expk_g[s] = (ST)expk;
    }
    __syncthreads();   // the next point overwrites the A-form and the reduction slots
    B128_STAMP(10);
  }
  if (threadIdx.x == 0) atomicAdd(&status[2], npts);   // (vsm_device_status: which kernel family ran)
  B128_STAMP_FLUSH_AT(32, 63, npts);
}

template <int RT, bool MR, typename ST>
int launch_dbl128(int N, int ns, int S, int ndoubl, ST* expk, const added<ST>& a, int grid, int nw, d4_t* scr,
                  int* status, hipStream_t st) {
  constexpr size_t lds = (size_t)(16 * RT) * (16 * RT) * sizeof(double) + 1024 + (MR ? 6 * 16 * RT * sizeof(double) : 0) + GJS_BYTES;
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl128<RT, MR, ST>), lds, "hipFuncSetAttribute(k_dbl128)")) return prepared;
  hipLaunchKernelGGL((k_dbl128<RT, MR, ST>), dim3(grid), dim3(64 * nw), lds, st, N, ns, S, ndoubl, expk, a, scr, status);
  VSM_LAUNCH_CHECK("k_dbl128");
  return VSM_OK;
}

// ---- interaction (ScatteringInterface_11) ------------------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266) with ONE inverse G2 = (I - R+- r-+)^-1 and the
// push-through identities of vsm_strip.hip's ia_body, ordered by left operand for the single A-form:
//   [R+-]: E2 = R+- r-+ , Z = R+- t--      (rider: R+- j0-  ->  z = J0+ + R+- j0-)
//   [T--]: V = T-- t-- , S = T-- r-+       (rider: vs = T-- j0-)
//   [E2] : G2 (series)
//   [t++]: T21 = t++ G2          [S]: Y = S G2
//   [T21]: R+- = r+- + T21 Z , T++ = T21 T++     (rider z in a spare column of T++: J0+ = j0+ + T21 z)
//   [Y]  : R-+ = R-+ + Y T++ , T-- = V + Y Z     (the same rider: J0- = J0- + vs + Y z)
// Three strips live at most; E2, Z, V, S wait in the workgroup's scratch.  The composite's [R+-], [T--] and the layer's [t++]
// are staged from global memory with whole-column requests; every other operand is a strip of its owner wave.
template <int RT, bool MR, typename ST>
__global__ __launch_bounds__(64 * B_MAXW) void k_ia128(int N, int S, composite<ST> c, added<ST> a, d4_t* __restrict__ scr,
                                                        int* __restrict__ status) {
  constexpr int NP = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) double lds128[];
  double* AF = lds128;
  double* vec = lds128 + NP * NP;   // j0+, j0-, J0+, J0-, z, vs, (MR) the rider table x_0 = x_1
  double* vjp = vec, *vjm = vec + NP, *vJp = vec + 2 * NP, *vJm = vec + 3 * NP, *vz = vec + 4 * NP, *vs = vec + 5 * NP;
  double* xt = vec + 6 * NP;
  float* red = reinterpret_cast<float*>(vec + 8 * NP);
  const inv128_ctx icx{AF, vec + 8 * NP + 32, status};
  double* xw = vec + 8 * NP + 32 + 128 + 16 * XS8B * (threadIdx.x >> 6);   // the wave's transposer tile (load_c8 / store_c8)
  bpos<RT> p(lds_addr128(AF), N);
  const unsigned xb = lds_addr128(xt) + 8u * (unsigned)((p.l15 & 1) * NP + p.kq);
  const int rrow = 16 * p.wave + p.kq;
  const int nw = blockDim.x >> 6, tid = threadIdx.x;
  const int cr = (N + 1) & ~1;   // the rider column (the doubling kernel's cb: the same strip count)
  const bool laneR = !MR && p.col == cr;
  const long long NN = (long long)N * N;
  d4_t* const sE = scr + ((long long)(blockIdx.x * 4 + 0) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  d4_t* const sZ = scr + ((long long)(blockIdx.x * 4 + 1) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  d4_t* const sV = scr + ((long long)(blockIdx.x * 4 + 2) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  d4_t* const sS = scr + ((long long)(blockIdx.x * 4 + 3) * B_MAXW + p.wave) * (RT * 64) + p.lane;
  int slot = 0;
  B128_STAMP_DECL;
  int npts = 0;

  for (int s = blockIdx.x; s < S; s += gridDim.x) {
    ++npts;
    ST* const R_mp = c.R_mp + NN * s;
    ST* const R_pm = c.R_pm + NN * s;
    ST* const T_pp = c.T_pp + NN * s;
    ST* const T_mm = c.T_mm + NN * s;
    ST* const J0_p = c.J0_p + (long long)N * s;
    ST* const J0_m = c.J0_m + (long long)N * s;
    const ST* const a_r_mp = a.r_mp + a.mat_stride * s;
    const ST* const a_r_pm = a.r_pm + a.mat_stride * s;
    const ST* const a_t_pp = a.t_pp + a.mat_stride * s;
    const ST* const a_t_mm = a.t_mm + a.mat_stride * s;
    for (int i = tid; i < NP; i += blockDim.x) {
      const bool in = i < N;
      vjp[i] = in ? a.j0_p[(long long)N * s + i] : 0.0;
      vjm[i] = in ? a.j0_m[(long long)N * s + i] : 0.0;
      vJp[i] = in ? J0_p[i] : 0.0;
      vJm[i] = in ? J0_m[i] : 0.0;
      if constexpr (MR) xt[i] = xt[NP + i] = vjm[i];
    }
    bstrip<RT> r_s;
    ldg_issue(r_s, a_r_mp, N, p);                   // (in flight with the staging)
    stage_af(AF, R_pm, N, nw, p);
    ldg_finish<ST>(r_s, p, xw);
    __syncthreads();                                    // (a)
    B128_STAMP(0);
    B128_STAMP(1);
    if (laneR) {                                        // j0- rides in the spare column of r-+
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) r_s.v[ta][r] = vjm[p.row(ta, r)];
    }
    bstrip<RT> tm;
    ldg_issue(tm, a_t_mm, N, p);                    // (requested a product ahead; permuted where it is first used)
    {
      bstrip<RT> E;
      E.zero();
      if constexpr (MR) {
        d4_t y = acc_zero<double>();
        mm128r(E, r_s, y, xb, p);                       // E2 = R+- r-+ ; tile: R+- j0-
        if (p.l15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) vz[rrow + 4 * r] = vJp[rrow + 4 * r] + y[r];
        }
      } else {
        mm128(E, r_s, p);                               // E2 = R+- r-+
      }
      if (laneR) {
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) vz[p.row(ta, r)] = vJp[p.row(ta, r)] + E.v[ta][r];
      }
      spill(sE, E, p);
    }
    B128_STAMP(2);
    ldg_finish<ST>(tm, p, xw);
    {
      {
        bstrip<RT> Z;
        Z.zero();
        mm128(Z, tm, p);                                // Z = R+- t--
        spill(sZ, Z, p);
      }
      B128_STAMP(3);
      __syncthreads();                                  // (b) [R+-] no longer read
      stage_af(AF, T_mm, N, nw, p);
      __syncthreads();                                  // (c)
      B128_STAMP(4);
      bstrip<RT> V;
      V.zero();
      mm128(V, tm, p);                                  // V = T-- t--
      spill(sV, V, p);
    }
    B128_STAMP(5);
    bstrip<RT> E;
    fill(E, sE, p);                                     // (requested a product ahead)
    {
      bstrip<RT> Sx;
      Sx.zero();
      if constexpr (MR) {
        d4_t y = acc_zero<double>();
        mm128r(Sx, r_s, y, xb, p);                      // S = T-- r-+ ; tile: T-- j0-
        if (p.l15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) vs[rrow + 4 * r] = y[r];
        }
      } else {
        mm128(Sx, r_s, p);                              // S = T-- r-+
      }
      if (laneR) {
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) vs[p.row(ta, r)] = Sx.v[ta][r];
      }
      spill(sS, Sx, p);
    }
    B128_STAMP(6);
    bstrip<RT> G;
    {
      const double nrm = norm128(E, N, nw, red, slot, p);   // (d) [T--] no longer read
      if constexpr (MR) {                               // (nobody reads the table any more) x = z for [T21], [Y]
        for (int i = tid; i < NP; i += blockDim.x) xt[i] = xt[NP + i] = vz[i];
      }
      invert128(inv_order128(nrm, status), E, G, N, icx, p);
    }
    B128_STAMP(7);
    __syncthreads();                                    // (e) [E2] no longer read
    stage_af(AF, a_t_pp, N, nw, p);
    __syncthreads();                                    // (f)
    B128_STAMP(8);
    bstrip<RT> X, Sx;
    fill(Sx, sS, p);                                    // (requested a product ahead)
    X.zero();
    mm128(X, G, p);                                     // T21 = t++ G2
    B128_STAMP(9);
    __syncthreads();                                    // (g) [t++] no longer read
    store_af(Sx, N, p);
    __syncthreads();                                    // (h)
    B128_STAMP(10);
    bstrip<RT> Y;
    Y.zero();
    mm128(Y, G, p);                                     // Y = S G2
    B128_STAMP(11);
    __syncthreads();                                    // (i) [S] no longer read
    store_af(X, N, p);
    bstrip<RT> Tpp;
    {
      bstrip<RT> acc, Z;
      ldg_issue(acc, a_r_pm, N, p);                 // (requested across the barrier)
      fill(Z, sZ, p);
      __syncthreads();                                  // (j)
      B128_STAMP(12);
      ldg_finish<ST>(acc, p, xw);
      mm128(acc, Z, p);                                 // R+- = r+- + T21 Z
      ldg_issue(Tpp, T_pp, N, p);
      stg(R_pm, acc, N, p, xw);
      ldg_finish<ST>(Tpp, p, xw);
    }
    B128_STAMP(13);
    if (laneR) {                                        // z rides in the spare column of T++
#pragma unroll
      for (int ta = 0; ta < RT; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) Tpp.v[ta][r] = vz[p.row(ta, r)];
    }
    {
      bstrip<RT> acc;
      acc.zero();
      if constexpr (MR) {
        d4_t y = acc_zero<double>();
        mm128r(acc, Tpp, y, xb, p);                     // T++ = T21 T++ ; tile: T21 z
        if (p.l15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (rrow + 4 * r < N) J0_p[rrow + 4 * r] = (ST)(vjp[rrow + 4 * r] + y[r]);
        }
      } else {
        mm128(acc, Tpp, p);                             // T++ = T21 T++ ; rider: T21 z
      }
      stg(T_pp, acc, N, p, xw);
      if (laneR) {
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = p.row(ta, r);
            if (row < N) J0_p[row] = (ST)(vjp[row] + acc.v[ta][r]);
          }
      }
    }
    __syncthreads();                                    // (k) [T21] no longer read
    B128_STAMP(14);
    store_af(Y, N, p);
    __syncthreads();                                    // (l)
    B128_STAMP(15);
    bstrip<RT> V2, Z2;
    {
      bstrip<RT> acc;
      ldg(acc, R_mp, N, p, xw);
      if constexpr (MR) {
        d4_t y = acc_zero<double>();
        mm128r(acc, Tpp, y, xb, p);                     // R-+ = R-+ + Y T++ ; tile: Y z
        if (p.l15 == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (rrow + 4 * r < N) J0_m[rrow + 4 * r] = (ST)(vJm[rrow + 4 * r] + vs[rrow + 4 * r] + y[r]);
        }
      } else {
        mm128(acc, Tpp, p);                             // R-+ = R-+ + Y T++ ; rider: Y z
      }
      fill(V2, sV, p);                                  // (the last product's operands, requested ahead of the stores)
      fill(Z2, sZ, p);
      stg(R_mp, acc, N, p, xw);
      if (laneR) {
#pragma unroll
        for (int ta = 0; ta < RT; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = p.row(ta, r);
            if (row < N) J0_m[row] = (ST)(vJm[row] + vs[row] + acc.v[ta][r]);
          }
      }
    }
    {
      bstrip<RT>& acc = V2;
      bstrip<RT>& Z = Z2;
      mm128(acc, Z, p);                                 // T-- = V + Y Z
      B128_STAMP(16);
      stg(T_mm, acc, N, p, xw);
    }
    __syncthreads();                                    // (m) the next point restages the A-form and the vectors
    B128_STAMP(17);
  }
  if (threadIdx.x == 0) atomicAdd(&status[3], npts);   // (vsm_device_status: which kernel family ran)
  B128_STAMP_FLUSH(npts);
}

template <int RT, bool MR, typename ST>
int launch_ia128(int N, int S, const composite<ST>& c, const added<ST>& a, int grid, int nw, d4_t* scr, int* status,
                 hipStream_t st) {
  constexpr size_t lds = (size_t)(16 * RT) * (16 * RT) * sizeof(double) + 8 * 16 * RT * sizeof(double) + 256 + GJS_BYTES +
                         B_MAXW * 16 * XS8B * sizeof(double);
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_ia128<RT, MR, ST>), lds, "hipFuncSetAttribute(k_ia128)")) return prepared;
  hipLaunchKernelGGL((k_ia128<RT, MR, ST>), dim3(grid), dim3(64 * nw), lds, st, N, S, c, a, scr, status);
  VSM_LAUNCH_CHECK("k_ia128");
  return VSM_OK;
}

// ---- (I - A B)^-1 (the operator chains of the linearized and Raman runs at these shapes: one launch instead of a product and
// a pivoted global-memory inverse) -------------------------------------------------------------------------------------------
template <int RT>
__global__ __launch_bounds__(64 * B_MAXW) void k_inv1m128(int N, int S, const double* __restrict__ A, long long sa,
                                                           const double* __restrict__ B, long long sb, double* __restrict__ X,
                                                           int* __restrict__ status) {
  constexpr int NP = 16 * RT;
  extern __shared__ __attribute__((aligned(16))) double lds128[];
  double* AF = lds128;
  float* red = reinterpret_cast<float*>(lds128 + NP * NP);
  const inv128_ctx icx{AF, lds128 + NP * NP + 32, status};
  bpos<RT> p(lds_addr128(AF), N);
  const int nw = blockDim.x >> 6;
  int slot = 0;
  for (int s = blockIdx.x; s < S; s += gridDim.x) {
    bstrip<RT> b;
    load_global128(b, B + sb * s, N, p);
    stage_af(AF, A + sa * s, N, nw, p);
    __syncthreads();
    bstrip<RT> E, G;
    E.zero();
    mm128(E, b, p);
    const double nrm = norm128(E, N, nw, red, slot, p);
    invert128(inv_order128(nrm, status), E, G, N, icx, p);
    store_global128(X + (long long)N * N * s, G, N, p);
    __syncthreads();   // the next point restages the A-form
  }
}
template <int RT>
int launch_inv1m128(int N, int S, const double* A, long long sa, const double* B, long long sb, double* X, int grid, int* status,
                    hipStream_t st) {
  constexpr size_t lds = (size_t)(16 * RT) * (16 * RT) * sizeof(double) + 256 + GJS_BYTES;
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_inv1m128<RT>), lds, "hipFuncSetAttribute(k_inv1m128)")) return prepared;
  hipLaunchKernelGGL(k_inv1m128<RT>, dim3(grid), dim3(64 * RT), lds, st, N, S, A, sa, B, sb, X, status);
  VSM_LAUNCH_CHECK("k_inv1m128");
  return VSM_OK;
}


}  // namespace

bool strip128_supported(int N) { return N > 64 && N <= 128; }
// Source vectors: rider columns of the strips for N <= 80 (five row tiles: +0 MFMAs, measured 3 % faster there), an extra MFMA
// tile per wave beyond (no rider-only wave at N = 96 / 112, no rider lane code: +1 ... 5 % at N = 96 ... 126; the only way at 127, 128)
static bool mr_policy(int N) { return N > 80; }

template <typename ST>
int strip128_doubling(int N, int n_stokes, int S, int ndoubl, ST* expk, const added<ST>& a, hipStream_t st) {
  if (ndoubl == 0 || S <= 0) return VSM_OK;   // doubling.jl:50
  const bool mr = mr_policy(N);
  const int RT = (N + 15) / 16, nw = mr ? RT : (((N + 1) & ~1) >> 4) + 1;
  const int grid = S < cu_count() ? S : cu_count();
  d4_t* scr = static_cast<d4_t*>(scratch((size_t)grid * 3 * B_MAXW * RT * 64 * sizeof(d4_t), 3, st));
  if (!scr) return VSM_ERR_HIP;
  int* status = device_status();
  if (!status) return VSM_ERR_HIP;
  switch (RT) {
    case 5: return launch_dbl128<5, false, ST>(N, n_stokes, S, ndoubl, expk, a, grid, nw, scr, status, st);
    case 6: return launch_dbl128<6, true, ST>(N, n_stokes, S, ndoubl, expk, a, grid, nw, scr, status, st);
    case 7: return launch_dbl128<7, true, ST>(N, n_stokes, S, ndoubl, expk, a, grid, nw, scr, status, st);
    case 8: return launch_dbl128<8, true, ST>(N, n_stokes, S, ndoubl, expk, a, grid, nw, scr, status, st);
  }
  set_error("strip128_doubling: N=%d outside 65..128", N);
  return VSM_ERR_UNSUPPORTED;
}

template <typename ST>
int strip128_interaction11(int N, int S, const composite<ST>& c, const added<ST>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const bool mr = mr_policy(N);
  const int RT = (N + 15) / 16, nw = mr ? RT : (((N + 1) & ~1) >> 4) + 1;
  const int grid = S < cu_count() ? S : cu_count();
  d4_t* scr = static_cast<d4_t*>(scratch((size_t)grid * 4 * B_MAXW * RT * 64 * sizeof(d4_t), 3, st));
  if (!scr) return VSM_ERR_HIP;
  int* status = device_status();
  if (!status) return VSM_ERR_HIP;
  switch (RT) {
    case 5: return launch_ia128<5, false, ST>(N, S, c, a, grid, nw, scr, status, st);
    case 6: return launch_ia128<6, true, ST>(N, S, c, a, grid, nw, scr, status, st);
    case 7: return launch_ia128<7, true, ST>(N, S, c, a, grid, nw, scr, status, st);
    case 8: return launch_ia128<8, true, ST>(N, S, c, a, grid, nw, scr, status, st);
  }
  set_error("strip128_interaction11: N=%d outside 65..128", N);
  return VSM_ERR_UNSUPPORTED;
}

int strip128_inv_one_minus(int N, int S, const double* A, long long sa, const double* B, long long sb, double* X, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int RT = (N + 15) / 16;
  const int grid = S < cu_count() ? S : cu_count();
  int* status = device_status();
  if (!status) return VSM_ERR_HIP;
  switch (RT) {
    case 5: return launch_inv1m128<5>(N, S, A, sa, B, sb, X, grid, status, st);
    case 6: return launch_inv1m128<6>(N, S, A, sa, B, sb, X, grid, status, st);
    case 7: return launch_inv1m128<7>(N, S, A, sa, B, sb, X, grid, status, st);
    case 8: return launch_inv1m128<8>(N, S, A, sa, B, sb, X, grid, status, st);
  }
  set_error("strip128_inv_one_minus: N=%d outside 65..128", N);
  return VSM_ERR_UNSUPPORTED;
}

template int strip128_doubling<double>(int, int, int, int, double*, const added<double>&, hipStream_t);
template int strip128_doubling<float>(int, int, int, int, float*, const added<float>&, hipStream_t);
template int strip128_interaction11<double>(int, int, const composite<double>&, const added<double>&, hipStream_t);
template int strip128_interaction11<float>(int, int, const composite<float>&, const added<float>&, hipStream_t);

}  // namespace vsm

#ifdef VSM_PHASE_TIMING
extern "C" int vsm_debug_phase_cycles_128(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_phase_cycles_128), sizeof(unsigned long long) * 64);
  if (reset) {
    unsigned long long z[64] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_phase_cycles_128), z, sizeof(z));
  }
  return 0;
}
#endif
