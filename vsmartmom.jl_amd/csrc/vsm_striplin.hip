// Linearized doubling step, column-strip form (FP64, 32 < N <= 60): ONE launch per doubling step does the forward
// recurrences AND the derivative recurrences of every active parameter of doubling_allparams_helper!
// (doubling_lin.jl:216-339) for one spectral point per workgroup, instead of ~40 batched-operator launches that
// each stream three [N,N,S] arrays through HBM.
//
// Same layout as vsm_strip.hip (strips in MFMA accumulator registers = B operands, A operands in LDS A-form), with
// FOUR A-form buffers (128 KB: one workgroup per CU, up to 512 VGPRs per lane):
//
//     BR = r           BT = t -> tt = t G           BX = rdot_p -> ttdot_p           BY = series scratch, tdot_p -> Y_p
//
//   forward:   E = r r ; G = (I - E)^-1 ; tt = t G ; rt = r t                      (A: BR, BT;  B: r_s, G_s, t_s)
//   per parameter p (rdot = ap_r-+, tdot = ap_t++), with Gdot = G X1 G eliminated (t Gdot = tt X1 G):
//     X1 = rdot r + r rdot ;  Q2 = rdot t + r tdot                                 (A: BX, BR shared by both products)
//     Y  = tdot + tt X1 ;  ttdot = Y G                                             (A: BT ; BY)
//     rdot' = rdot + ttdot rt + tt Q2 ;  tdot' = ttdot t + tt tdot                 (A: BX, BT shared by both products)
//   forward:   r' = r + tt rt ;  t' = tt t                                         (A: BT shared)
//   = 5 products + inverse, + 10 products per parameter (the reference: 12 per parameter).
//
// All 4 + 8 P matrix-vector products of the source recurrences ride in the two spare columns (Kend, Kend+1) of the B
// strips of these products:
//     rt[:,c1] = r j0+ (+ j1-) = A      rt[:,c2] = r j1- (+ j0+) = B           j1+- = j0+- expk
//     X1[:,c1] = rdot j0+ + r aJ+       X1[:,c2] = rdot j1- + r aJ1-           aJ1+- = aJ+- expk + j0+- ekl_p
//     rdot'[:,c1] = aJ- + ttdot A + tt v = aJ-'      rdot'[:,c2] = aJ1+ + ttdot B + tt u = aJ+'
//     r'[:,c1] = j0- + tt A = j0-'                   r'[:,c2] = j1+ + tt B = j0+'
// State (r, t, j, rdot, tdot, aJ, expk, ekl) lives in HBM between steps: 115 KB (1 + P) per point and step against
// (12 + 20 P) N^3 flop -- far above the ridge.
#include <stdlib.h>

#include "vsm_internal.h"
#include "vsm_strip_dev.h"


namespace vsm {

namespace {

struct lsmem {
  double BR[SNP * SNP];
  double BT[SNP * SNP];
  double BX[SNP * SNP];
  double BY[SNP * SNP];
  double vec[8][SNP];   // doubling: j0+, j0-, (aJ+_p, aJ-_p) x up to 3 ; interaction halves: VR, VADD, VACC, VDR, VDADD, VDACC
  float red[2][4];
  gj_scratch<double, SNP> gj;
  double xw[4][16 * 34];   // per-wave transposer of load/store_strip_global_c
};

// Strip <-> global through a per-wave LDS transposer.  load_strip_global / store_strip_global touch 16 columns x 32 B
// per instruction (sixteen half-used cache lines, re-fetched from L2 by the next instruction because the four waves'
// strips overflow the vector L1): measured at half the time of the interaction kernel of rounds 1-3.  Here each lane moves 64 contiguous
// bytes of one column (full lines, 16-byte accesses) and the (row, lane) permutation to the MFMA layout happens in a
// wave-private 16 x 32 tile of LDS -- wave-private, so no barrier (LDS operations of one wave complete in order).
constexpr int XS = 34;   // column stride of the tile in doubles: 2 l15 + kq is conflict-free over a 32-lane pass
__device__ __forceinline__ void load_strip_global_c(sstrip& x, const double* __restrict__ g, int N, const spos& p,
                                                    double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  const double* src = g + (long long)N * min(col, N - 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row0 = 32 * h + 8 * q;
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int r = row0 + i;
      if (col < N && r + 1 < N) {
        const d2_t t = *reinterpret_cast<const d2_t*>(src + r);
        v[i] = t.a;
        v[i + 1] = t.b;
      } else {
        v[i] = (col < N && r < N) ? src[r] : 0.0;
        v[i + 1] = 0.0;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) xw[c * XS + 8 * q + i] = v[i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) x.v[2 * h + t][r] = xw[p.l15 * XS + 16 * t + p.kq + 4 * r];
  }
}
// The same load in two halves: `issue` only requests the 16 doubles of the lane (they stay in flight while the wave goes on with
// its MFMAs), `finish` runs them through the transposer.  With one wave per SIMD nothing else hides a global round trip.
struct raw_strip {
  double v[2][8];
};
__device__ __forceinline__ void issue_strip_load(raw_strip& w, const double* __restrict__ g, int N, const spos& p) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  const double* src = g + (long long)N * min(col, N - 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int row0 = 32 * h + 8 * q;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int r = row0 + i;
      if (col < N && r + 1 < N) {
        const d2_t t = *reinterpret_cast<const d2_t*>(src + r);
        w.v[h][i] = t.a;
        w.v[h][i + 1] = t.b;
      } else {
        w.v[h][i] = (col < N && r < N) ? src[r] : 0.0;
        w.v[h][i + 1] = 0.0;
      }
    }
  }
}
__device__ __forceinline__ void finish_strip_load(sstrip& x, const raw_strip& w, const spos& p, double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) xw[c * XS + 8 * q + i] = w.v[h][i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) x.v[2 * h + t][r] = xw[p.l15 * XS + 16 * t + p.kq + 4 * r];
  }
}
// A-form staging in two halves (the lane's 16 column values: wave w takes the columns w, w + 4, ...)
struct raw_aform {
  double v[16];
};
__device__ __forceinline__ void issue_aform_load(raw_aform& w, const double* __restrict__ g, int N, const spos& p) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int j = p.wave + 4 * i;
    w.v[i] = (p.lane < N && j < N) ? g[p.lane + (long long)N * j] : 0.0;
  }
}
__device__ __forceinline__ void finish_aform_load(double* L, const raw_aform& w, const spos& p) {
#pragma unroll
  for (int i = 0; i < 16; ++i) L[lidx<SNP>(p.lane, p.wave + 4 * i)] = w.v[i];
}

__device__ __forceinline__ void store_strip_global_c(double* __restrict__ g, const sstrip& x, int N, const spos& p,
                                                     double* __restrict__ xw) {
  const int c = p.lane >> 2, q = p.lane & 3;
  const int col = 16 * p.wave + c;
  double* dst = g + (long long)N * min(col, N - 1);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[p.l15 * XS + 16 * t + p.kq + 4 * r] = x.v[2 * h + t][r];
    __builtin_amdgcn_wave_barrier();
    const int row0 = 32 * h + 8 * q;
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
      const int r = row0 + i;
      d2_t t;
      t.a = xw[c * XS + 8 * q + i];
      t.b = xw[c * XS + 8 * q + i + 1];
      if (col < N && r + 1 < N)
        *reinterpret_cast<d2_t*>(dst + r) = t;
      else if (col < N && r < N)
        dst[r] = t.a;
    }
  }
}

// First of the two spare columns that carry the riders: N rounded up to a k-step -- and never below the 4 KS contraction range
// of the instantiation (8 <= N <= 32 runs on KS = 9: riders at columns 36, 37; inside the range they would be contracted against
// the zero padding rows of every right operand, harmless only as long as every source vector is finite).
template <int KS>
__device__ __forceinline__ int rider_base(int N) {
  const int kend = ((N + 3) >> 2) << 2;
  return (kend < 4 * KS && N <= 32) ? 4 * KS : kend;
}
// spare-column access: lanes of the owning wave with col == c1 (A) / c1 + 1 (B)
struct spare {
  bool own, A, B, AB;
  __device__ __forceinline__ spare(const spos& p, int c1) {
    own = (p.wave == (c1 >> 4));
    A = own && (p.col == c1);
    B = own && (p.col == c1 + 1);
    AB = A || B;
  }
  // x[:, c1] = fa(row, old), x[:, c1+1] = fb(row, old)
  template <typename FA, typename FB>
  __device__ __forceinline__ void put(sstrip& x, const spos& p, FA fa, FB fb) const {
    if (own) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          const double old = x.v[ta][r];
          const double va = fa(row, old), vb = fb(row, old);
          x.v[ta][r] = A ? va : (B ? vb : old);
        }
    }
  }
  // dstA[row] = x[:, c1], dstB[row] = x[:, c1+1]   (global vectors of length N)
  __device__ __forceinline__ void get(const sstrip& x, const spos& p, int N, double* dstA, double* dstB) const {
    if (AB && (A || dstB != nullptr)) {
      double* d = A ? dstA : dstB;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          if (row < N) d[row] = x.v[ta][r];
        }
    }
  }
};

template <int KS>
__global__ __launch_bounds__(SNT, 1) void k_dbl_lin_step(int N, int S, int P, double* __restrict__ expk,
                                                         double* __restrict__ ekl, added<double> a, added_lin<double> al) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lsmem& sm = *reinterpret_cast<lsmem*>(smem_raw);
  double* BR = sm.BR;
  double* BT = sm.BT;
  double* BX = sm.BX;
  double* BY = sm.BY;
  double* jp = sm.vec[0];
  double* jm = sm.vec[1];
  double* ajp = sm.vec[2];
  double* ajm = sm.vec[3];
  spos p;
  p.bind(sm.BR);
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  const int Kend = rider_base<KS>(N);
  const spare sp(p, Kend);
  double* xw = sm.xw[p.wave];
  const double k = expk[s];
  double* g_r = a.r_mp + (long long)s * NN;
  double* g_t = a.t_pp + (long long)s * NN;
  // A-form stores carry no mask: the padding rows of every strip are zero by construction (zero-padded loads, products inherit
  // zero rows from their A operand) and the spare columns with their riders sit at or beyond 4 KS (rider_base): never read as k
  auto asis = [](double x, int, int) { return x; };

  stage_aform_full2(BR, g_r, BT, g_t, N, p);
  if (tid < SNP) {
    jp[tid] = (tid < N) ? a.j0_p[(long long)s * N + tid] : 0.0;
    jm[tid] = (tid < N) ? a.j0_m[(long long)s * N + tid] : 0.0;
  }
  __syncthreads();
  // r's strip is re-read from BR where it is needed (register budget); t's is kept (BT is overwritten by tt)
  auto load_r_with_riders = [&](sstrip& x) {
    load_strip(x, BR, p);
    sp.put(x, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
  };
  sstrip t_s, G, rt;
  load_strip(t_s, BT, p);
  int slot = 0;
  {
    sstrip E, r_s;
    load_strip(r_s, BR, p);
    E.zero();
    mm_ab<KS>(E, BR, r_s, p);
    invert_strip_horner<KS>(E, G, BY, N, sm, slot, p);   // (E: clean padding -- the rows of a product come from its zero-padded A-form)
  }
  sp.put(t_s, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
  {
    sstrip tt;
    tt.zero();
    mm_ab<KS>(tt, BT, G, p);
    rt.zero();
    mm_ab<KS>(rt, BR, t_s, p);   // r t  (+ r j0+, r j1-)
    __syncthreads();             // BT (t) and BY (series powers) no longer read
    store_strip(BT, tt, p, asis);
  }
  sp.put(rt, p, [&](int row, double o) { return jm[row] * k + o; }, [&](int row, double o) { return jp[row] + o; });
  __syncthreads();   // tt complete in BT

  for (int pp = 0; pp < P; ++pp) {
    const double kl = ekl[s + (long long)S * pp];
    double* g_ar = al.ap_r_mp + (long long)pp * MS + (long long)s * NN;
    double* g_at = al.ap_t_pp + (long long)pp * MS + (long long)s * NN;
    double* g_ajp = al.ap_J0_p + (long long)pp * VS + (long long)s * N;
    double* g_ajm = al.ap_J0_m + (long long)pp * VS + (long long)s * N;
    stage_aform_full2(BX, g_ar, BY, g_at, N, p);
    if (tid < SNP) {
      ajp[tid] = (tid < N) ? g_ajp[tid] : 0.0;
      ajm[tid] = (tid < N) ? g_ajm[tid] : 0.0;
    }
    __syncthreads();
    sstrip X1, Q2;
    X1.zero();
    Q2.zero();
    {
      sstrip r_s;
      load_r_with_riders(r_s);
      mm_ab2<KS>(X1, Q2, BX, r_s, t_s, p);   // rdot r (+ rdot j0+, rdot j1-) ; rdot t
    }
    {
      sstrip rd, td;   // (scoped: the strips of rdot / tdot are re-read where they are needed again -- register budget)
      load_strip(rd, BX, p);
      load_strip(td, BY, p);
      sp.put(rd, p, [&](int row, double) { return ajp[row]; }, [&](int row, double) { return ajm[row] * k + jm[row] * kl; });
      mm_ab2<KS>(X1, Q2, BR, rd, td, p);     // + r rdot (+ r aJ+, r aJ1-)   ; + r tdot
    }
    // v = aJ1- + rdot j0+ + r aJ+ ; u = aJ+ + rdot j1- + r aJ1-   -> spare columns of Q2
    if (sp.own) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          const double x = X1.v[ta][r];
          const double v = ajm[row] * k + jm[row] * kl + x, u = ajp[row] + x;
          Q2.v[ta][r] = sp.A ? v : (sp.B ? u : Q2.v[ta][r]);
        }
    }
    {
      sstrip Y;
      load_strip(Y, BY, p);      // tdot
      mm_ab<KS>(Y, BT, X1, p);   // Y = tdot + tt X1
      __syncthreads();           // every wave has read tdot (BY) and is done with rdot's A-form (BX)
      store_strip(BY, Y, p, asis);
    }
    __syncthreads();   // Y complete in BY
    {
      sstrip ttl;
      ttl.zero();
      mm_ab<KS>(ttl, BY, G, p);   // ttdot = Y G
      store_strip(BX, ttl, p, asis);
    }
    sstrip rd, tdn;
    load_strip_global_c(rd, g_ar, N, p, xw);
    sp.put(rd, p, [&](int row, double) { return ajm[row]; }, [&](int row, double) { return ajp[row] * k + jp[row] * kl; });
    tdn.zero();
    __syncthreads();   // ttdot complete in BX
    mm_ab2<KS>(rd, tdn, BX, rt, t_s, p);   // rdot += ttdot rt (+ ttdot A, ttdot B) ; tdot' = ttdot t
    {
      sstrip td;
      load_strip_global_c(td, g_at, N, p, xw);
      mm_ab2<KS>(rd, tdn, BT, Q2, td, p);    // rdot += tt Q2 (+ tt v, tt u)          ; tdot' += tt tdot
    }
    store_strip_global_c(g_ar, rd, N, p, xw);
    store_strip_global_c(g_at, tdn, N, p, xw);
    sp.get(rd, p, N, g_ajm, g_ajp);
    if (tid == 0) ekl[s + (long long)S * pp] = 2.0 * k * kl;
    __syncthreads();   // BX, BY, aJ+- free for the next parameter
  }

  // forward update: r' = r + tt rt (+ tt A, tt B on top of j0-, j1+) ; t' = tt t
  sstrip r_s;
  load_strip(r_s, BR, p);
  sp.put(r_s, p, [&](int row, double) { return jm[row]; }, [&](int row, double) { return jp[row] * k; });
  sstrip tn;
  tn.zero();
  mm_ab2<KS>(r_s, tn, BT, rt, t_s, p);
  store_strip_global_c(g_r, r_s, N, p, xw);
  store_strip_global_c(g_t, tn, N, p, xw);
  sp.get(r_s, p, N, a.j0_m + (long long)s * N, a.j0_p + (long long)s * N);
  if (tid == 0) expk[s] = k * k;
}

// ---------------------------------------------------------------------------
// ALL ndoubl doubling steps of a layer in ONE launch, for PA = 1..3 active parameters (gas columns; the surface slot is not
// doubled).  Same arithmetic and statement order as k_dbl_lin_step, but the state stays on the chip between the
// steps: [r] lives in BR (rewritten at the end of a step), t, rdot_p and tdot_p live as strips in registers (their A-form
// buffers are overwritten within a step: BT by tt, BX by ttdot, BY by the series scratch / Y), the source vectors in LDS, expk / ekl
// in registers.  Per step that removes the staging of four matrices, two strip loads and four strip stores through global
// memory (the per-step kernel spends ~30 % of its time there) and seven of eight launches.
// ---------------------------------------------------------------------------
template <int KS, int PA>
__global__ __launch_bounds__(SNT, 1) void k_dbl_lin_multi(int N, int S, int nd, int ns, double* __restrict__ expk,
                                                          double* __restrict__ ekl, added<double> a, added_lin<double> al) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lsmem& sm = *reinterpret_cast<lsmem*>(smem_raw);
  double* BR = sm.BR;
  double* BT = sm.BT;
  double* BX = sm.BX;
  double* BY = sm.BY;
  double* jp = sm.vec[0];
  double* jm = sm.vec[1];
  spos p;
  p.bind(sm.BR);
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N, MS = NN * S, VS = (long long)N * S;
  const int Kend = rider_base<KS>(N);
  const spare sp(p, Kend);
  double* xw = sm.xw[p.wave];
  double k = expk[s];
  double kl[PA];
  double* g_r = a.r_mp + (long long)s * NN;
  double* g_t = a.t_pp + (long long)s * NN;
  // A-form stores carry no mask: the padding rows of every strip are zero by construction (zero-padded loads, products inherit
  // zero rows from their A operand) and the spare columns with their riders sit at or beyond 4 KS (rider_base): never read as k
  auto asis = [](double x, int, int) { return x; };
  // a strip with its two spare columns cleared (they carry riders during a product)
  auto clean = [&](sstrip& x) { sp.put(x, p, [](int, double) { return 0.0; }, [](int, double) { return 0.0; }); };
  // the riders of a result strip go back to the LDS vectors (only the wave that owns the spare columns touches them)
  auto take = [&](const sstrip& x, double* dA, double* dB) {
    if (sp.AB) {
      double* d = sp.A ? dA : dB;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = p.row(ta, r);
          if (row < N) d[row] = x.v[ta][r];
        }
    }
  };

  stage_aform_full(BR, g_r, N, p);
  if (tid < SNP) {
    jp[tid] = (tid < N) ? a.j0_p[(long long)s * N + tid] : 0.0;
    jm[tid] = (tid < N) ? a.j0_m[(long long)s * N + tid] : 0.0;
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
      sm.vec[2 + 2 * pp][tid] = (tid < N) ? al.ap_J0_p[pp * VS + (long long)s * N + tid] : 0.0;
      sm.vec[3 + 2 * pp][tid] = (tid < N) ? al.ap_J0_m[pp * VS + (long long)s * N + tid] : 0.0;
    }
  }
  sstrip t_s, r_new;
  sstrip rd_s[PA], td_s[PA];          // rdot_p, tdot_p: register strips between the steps
  load_strip_global_c(t_s, g_t, N, p, xw);
#pragma unroll
  for (int pp = 0; pp < PA; ++pp) {
    kl[pp] = ekl[s + (long long)S * pp];
    load_strip_global_c(rd_s[pp], al.ap_r_mp + pp * MS + (long long)s * NN, N, p, xw);
    load_strip_global_c(td_s[pp], al.ap_t_pp + pp * MS + (long long)s * NN, N, p, xw);
  }
  int slot = 0;
  for (int n = 0; n < nd; ++n) {
    // on entry: BR = [r] stored (visible after the barrier below), t_s / rd_s / td_s clean strips, vectors in LDS
    store_strip(BT, t_s, p, asis);
    __syncthreads();
    auto load_r_with_riders = [&](sstrip& x) {
      load_strip(x, BR, p);
      sp.put(x, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
    };
    sstrip G, rt;
    {
      sstrip E, r_s;
      load_strip(r_s, BR, p);
      E.zero();
      mm_ab<KS>(E, BR, r_s, p);
      invert_strip_horner<KS>(E, G, BY, N, sm, slot, p);   // (E: clean padding -- the rows of a product come from its zero-padded A-form)
    }
    sp.put(t_s, p, [&](int row, double) { return jp[row]; }, [&](int row, double) { return jm[row] * k; });
    {
      sstrip tt;
      tt.zero();
      mm_ab<KS>(tt, BT, G, p);
      rt.zero();
      mm_ab<KS>(rt, BR, t_s, p);   // r t  (+ r j0+, r j1-)
      __syncthreads();             // BT (t) and BY (series powers) no longer read
      store_strip(BT, tt, p, asis);
    }
    sp.put(rt, p, [&](int row, double o) { return jm[row] * k + o; }, [&](int row, double o) { return jp[row] + o; });
    // ---- the parameters ----------------------------------------------------------------------------------------------------
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
      double* ajp = sm.vec[2 + 2 * pp];
      double* ajm = sm.vec[3 + 2 * pp];
      sstrip& rd = rd_s[pp];
      sstrip& td = td_s[pp];
      store_strip(BX, rd, p, asis);   // [rdot_p] -> BX, [tdot_p] -> BY  (both free: barrier above / end of the previous parameter)
      store_strip(BY, td, p, asis);
      __syncthreads();   // tt complete in BT (first parameter), rdot_p / tdot_p complete
      sstrip X1, Q2;
      X1.zero();
      Q2.zero();
      {
        sstrip r_s;
        load_r_with_riders(r_s);
        mm_ab2<KS>(X1, Q2, BX, r_s, t_s, p);   // rdot r (+ rdot j0+, rdot j1-) ; rdot t
      }
      {
        sstrip rdr = rd;
        sp.put(rdr, p, [&](int row, double) { return ajp[row]; }, [&](int row, double) { return ajm[row] * k + jm[row] * kl[pp]; });
        mm_ab2<KS>(X1, Q2, BR, rdr, td, p);    // + r rdot (+ r aJ+, r aJ1-)   ; + r tdot
      }
      if (sp.own) {   // v = aJ1- + rdot j0+ + r aJ+ ; u = aJ+ + rdot j1- + r aJ1-   -> spare columns of Q2
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = p.row(ta, r);
            const double x = X1.v[ta][r];
            const double v = ajm[row] * k + jm[row] * kl[pp] + x, u = ajp[row] + x;
            Q2.v[ta][r] = sp.A ? v : (sp.B ? u : Q2.v[ta][r]);
          }
      }
      {
        sstrip Y = td;
        mm_ab<KS>(Y, BT, X1, p);   // Y = tdot + tt X1
        __syncthreads();           // every wave has read tdot (BY) and is done with rdot's A-form (BX)
        store_strip(BY, Y, p, asis);
      }
      __syncthreads();   // Y complete in BY
      {
        sstrip ttl;
        ttl.zero();
        mm_ab<KS>(ttl, BY, G, p);   // ttdot = Y G
        store_strip(BX, ttl, p, asis);
      }
      sp.put(rd, p, [&](int row, double) { return ajm[row]; }, [&](int row, double) { return ajp[row] * k + jp[row] * kl[pp]; });
      sstrip tdn;
      tdn.zero();
      __syncthreads();   // ttdot complete in BX
      mm_ab2<KS>(rd, tdn, BX, rt, t_s, p);      // rdot += ttdot rt (+ ttdot A, ttdot B) ; tdot' = ttdot t
      mm_ab2<KS>(rd, tdn, BT, Q2, td, p);       // rdot += tt Q2 (+ tt v, tt u)          ; tdot' += tt tdot
      take(rd, ajm, ajp);                       // (this parameter's vectors are not read again in this step)
      clean(rd);
      clean(tdn);
      td = tdn;
      kl[pp] = 2.0 * k * kl[pp];
      if (pp + 1 < PA) __syncthreads();   // BX, BY free for the next parameter
    }
    // forward update: r' = r + tt rt (+ tt A, tt B on top of j0-, j1+) ; t' = tt t
    sstrip r_s;
    load_strip(r_s, BR, p);
    sp.put(r_s, p, [&](int row, double) { return jm[row]; }, [&](int row, double) { return jp[row] * k; });
    sstrip tn;
    tn.zero();
    mm_ab2<KS>(r_s, tn, BT, rt, t_s, p);
    take(r_s, jm, jp);
    clean(r_s);
    clean(tn);
    t_s = tn;
    r_new = r_s;
    k = k * k;
    __syncthreads();   // everybody is done reading BR, BT, BX, BY of this step
    if (n + 1 < nd) store_strip(BR, r_new, p, asis);
  }
  // apply_D! with derivative slots (doubling_lin.jl:374-421) on the way out when ns > 0: r-+ gets its U,V rows negated,
  // r+- = D r-+ D and t-- = D t++ D are derived (also for the derivatives; the slots of the inactive parameters are zero)
  const bool uj = ns > 0 && is_uv_row(p.col, ns);
  auto dsign = [&](const sstrip& x, sstrip& rowflip, sstrip& both, bool flip) {
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ui = is_uv_row(p.row(ta, r), ns);
        const double v = (flip && ui) ? -x.v[ta][r] : x.v[ta][r];
        rowflip.v[ta][r] = v;
        both.v[ta][r] = (ui == uj) ? v : -v;
      }
  };
  if (ns > 0) {
    sstrip f, b;
    dsign(r_new, f, b, true);
    store_strip_global_c(g_r, f, N, p, xw);
    store_strip_global_c(a.r_pm + (long long)s * NN, b, N, p, xw);
    dsign(t_s, f, b, false);
    store_strip_global_c(g_t, f, N, p, xw);
    store_strip_global_c(a.t_mm + (long long)s * NN, b, N, p, xw);
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
      dsign(rd_s[pp], f, b, true);
      store_strip_global_c(al.ap_r_mp + pp * MS + (long long)s * NN, f, N, p, xw);
      store_strip_global_c(al.ap_r_pm + pp * MS + (long long)s * NN, b, N, p, xw);
      dsign(td_s[pp], f, b, false);
      store_strip_global_c(al.ap_t_pp + pp * MS + (long long)s * NN, f, N, p, xw);
      store_strip_global_c(al.ap_t_mm + pp * MS + (long long)s * NN, b, N, p, xw);
    }
    f.zero();
    for (int pp = PA; pp < al.P; ++pp) {
      store_strip_global_c(al.ap_r_pm + pp * MS + (long long)s * NN, f, N, p, xw);
      store_strip_global_c(al.ap_t_mm + pp * MS + (long long)s * NN, f, N, p, xw);
    }
  } else {
    store_strip_global_c(g_r, r_new, N, p, xw);
    store_strip_global_c(g_t, t_s, N, p, xw);
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
      store_strip_global_c(al.ap_r_mp + pp * MS + (long long)s * NN, rd_s[pp], N, p, xw);
      store_strip_global_c(al.ap_t_pp + pp * MS + (long long)s * NN, td_s[pp], N, p, xw);
    }
  }
  if (tid < N) {   // (the vectors were last written by the owning wave before the final barrier of the loop)
    const double sg = (ns > 0 && is_uv_row(tid, ns)) ? -1.0 : 1.0;
    a.j0_p[(long long)s * N + tid] = jp[tid];
    a.j0_m[(long long)s * N + tid] = sg * jm[tid];
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) {
      al.ap_J0_p[pp * VS + (long long)s * N + tid] = sm.vec[2 + 2 * pp][tid];
      al.ap_J0_m[pp * VS + (long long)s * N + tid] = sg * sm.vec[3 + 2 * pp][tid];
    }
  }
  if (tid == 0) {
    expk[s] = k;
#pragma unroll
    for (int pp = 0; pp < PA; ++pp) ekl[s + (long long)S * pp] = kl[pp];
  }
}

}  // namespace

#define VSM_CAT2(a, b) a##b
#define VSM_CAT(a, b) VSM_CAT2(a, b)
#define VSM_STRIPLIN_DECL(KS)                                                                                                      \
  int VSM_CAT(launch_dbl_lin_step_, KS)(int, int, int, double*, double*, const added<double>&, const added_lin<double>&, hipStream_t); \
  int VSM_CAT(launch_dbl_lin_multi_, KS)(int, int, int, int, int, double*, double*, const added<double>&, const added_lin<double>&, hipStream_t);

#ifdef VSM_STRIP_KS
VSM_STRIPLIN_DECL(VSM_STRIP_KS)
int VSM_CAT(launch_dbl_lin_step_, VSM_STRIP_KS)(int N, int S, int P, double* expk, double* ekl, const added<double>& a,
                                                const added_lin<double>& al, hipStream_t st) {
  const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl_lin_step<VSM_STRIP_KS>), sizeof(lsmem), "hipFuncSetAttribute(k_dbl_lin_step)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(k_dbl_lin_step<VSM_STRIP_KS>, dim3(S), dim3(SNT), sizeof(lsmem), st, N, S, P, expk, ekl, a, al);
  VSM_LAUNCH_CHECK("k_dbl_lin_step");
  return VSM_OK;
}

int VSM_CAT(launch_dbl_lin_multi_, VSM_STRIP_KS)(int N, int S, int PA, int nd, int ns, double* expk, double* ekl, const added<double>& a,
                                                 const added_lin<double>& al, hipStream_t st) {
  int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl_lin_multi<VSM_STRIP_KS, 1>), sizeof(lsmem), "hipFuncSetAttribute(k_dbl_lin_multi)");
  if (!prepared) prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl_lin_multi<VSM_STRIP_KS, 2>), sizeof(lsmem), "hipFuncSetAttribute(k_dbl_lin_multi)");
  if (!prepared) prepared = ensure_dyn_lds(reinterpret_cast<const void*>(k_dbl_lin_multi<VSM_STRIP_KS, 3>), sizeof(lsmem), "hipFuncSetAttribute(k_dbl_lin_multi)");
  if (prepared) return prepared;
  if (PA == 1)
    hipLaunchKernelGGL((k_dbl_lin_multi<VSM_STRIP_KS, 1>), dim3(S), dim3(SNT), sizeof(lsmem), st, N, S, nd, ns, expk, ekl, a, al);
  else if (PA == 2)
    hipLaunchKernelGGL((k_dbl_lin_multi<VSM_STRIP_KS, 2>), dim3(S), dim3(SNT), sizeof(lsmem), st, N, S, nd, ns, expk, ekl, a, al);
  else
    hipLaunchKernelGGL((k_dbl_lin_multi<VSM_STRIP_KS, 3>), dim3(S), dim3(SNT), sizeof(lsmem), st, N, S, nd, ns, expk, ekl, a, al);
  VSM_LAUNCH_CHECK("k_dbl_lin_multi");
  return VSM_OK;
}

#else  // ---- dispatcher object -----------------------------------------------------------------------------------------

VSM_STRIPLIN_DECL(9)
VSM_STRIPLIN_DECL(10)
VSM_STRIPLIN_DECL(11)
VSM_STRIPLIN_DECL(12)
VSM_STRIPLIN_DECL(13)
VSM_STRIPLIN_DECL(14)
VSM_STRIPLIN_DECL(15)

// The linearized strip kernels take 8 <= N <= 60; shapes below 33 run on the KS = 9 instantiation (k padded to 36: the strips'
// padding rows are zero, so the extra k-steps add nothing) -- a quarter of the MFMA work is useful at N = 30, but the operator
// chain these shapes would take instead is launch-bound (wall ratio 20 against a flop ratio of 3.4 at N = 30).
static bool strip_lin_supported(int N) { return N >= 8 && (strip_supported(N) || N <= 32); }
static int strip_lin_ks(int N) { return N <= 32 ? 9 : (N + 3) / 4; }
// One fused doubling step (forward + P parameters); VSM_ERR_UNSUPPORTED outside 32 < N <= 60 or when the added layer is
// not the plain [N,N,S] layout.
int strip_doubling_lin_step(int N, int S, int P, double* expk, double* ekl, const added<double>& a,
                            const added_lin<double>& al, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_STRIP_LIN");
  if (off || !strip_lin_supported(N) || a.mat_stride != (long long)N * N || al.mat_stride != (long long)N * N)
    return VSM_ERR_UNSUPPORTED;
  switch (strip_lin_ks(N)) {
#define VSM_CASE(KS) \
  case KS:           \
    return VSM_CAT(launch_dbl_lin_step_, KS)(N, S, P, expk, ekl, a, al, st);
    VSM_CASE(9)
    VSM_CASE(10)
    VSM_CASE(11)
    VSM_CASE(12)
    VSM_CASE(13)
    VSM_CASE(14)
    VSM_CASE(15)
#undef VSM_CASE
    default:
      return VSM_ERR_UNSUPPORTED;
  }
}

// All ndoubl doubling steps in one launch (one to three active parameters), apply_D! included when ns (n_stokes) > 0;
// VSM_ERR_UNSUPPORTED otherwise.
int strip_doubling_lin_multi(int N, int S, int P, int nd, int ns, double* expk, double* ekl, const added<double>& a,
                             const added_lin<double>& al, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_STRIP_LIN") || ab_switch("VSM_NO_LIN_MULTI");
  if (off || P < 1 || P > 3 || nd < 1 || !strip_lin_supported(N) || a.mat_stride != (long long)N * N || al.mat_stride != (long long)N * N)
    return VSM_ERR_UNSUPPORTED;
  switch (strip_lin_ks(N)) {
#define VSM_CASE(KS) \
  case KS:           \
    return VSM_CAT(launch_dbl_lin_multi_, KS)(N, S, P, nd, ns, expk, ekl, a, al, st);
    VSM_CASE(9)
    VSM_CASE(10)
    VSM_CASE(11)
    VSM_CASE(12)
    VSM_CASE(13)
    VSM_CASE(14)
    VSM_CASE(15)
#undef VSM_CASE
    default:
      return VSM_ERR_UNSUPPORTED;
  }
}

#endif


}  // namespace vsm
