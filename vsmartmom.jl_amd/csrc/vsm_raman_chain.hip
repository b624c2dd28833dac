// Rotational-Raman inelastic doubling -- ALL STEPS OF A LAYER IN ONE LAUNCH, the state of a line on chip (FP64, 20 <= N <= 22; round 4).
//
// vsm_raman_quad.hip walks one doubling step per launch and is bound by the bytes of a step: the inelastic state of a line
// (ier, iet, ieJ0+-) is read two to three times and written once per step, nine steps per layer on the C5 shape.  The elastic side
// of a step (r, t, ttg, gt, gr, grt and four vectors per point) does not depend on the inelastic state, so the host runs the
// elastic chain of all steps first and keeps every step's operands (`rc_steps`: step-major arrays in library scratch); this kernel
// then walks the steps with the state of its lines in LDS: ier, iet as flat images (both lane maps of the 4 x 4 x 4 MFMA read the
// same image; the results of a step are written back into them), the source vectors in a small table.  The state crosses the memory
// system once per layer.
//
// With the LDS holding state there is no room to stage donor operands, so they stream from L2 straight into the fragment registers
// (32-byte pieces per four lanes), several k-steps ahead.  One wave per SIMD cannot hide that stream (measured with four lines per
// wave: 1.50 ms per step against the step kernel's 1.30); here a line's matrix is split over TWO lane groups (a column half each:
// lane = 16 q + 4 (2 line + half) + l), a wave walks two lines with 36 registers per matrix, fits 256 registers and 19 KB of LDS, and
// two waves share a SIMD: one wave's loads are in flight under the other's products.
//
// The two lines of a wave share their DONOR (workgroup = donor point x pair of line offsets; recipients np - shift[d]): both pairs of
// lane groups stream the same donor blocks, so a step's donor operands are fetched once per wave (recipient-major pairs: once per
// line; +4 % on the C5 shape).
//
// Measured (C5 shape, N = 21, K = 40, 4000 points): 9.65 ms per layer-moment of nine steps against 11.9 ms of the
// step kernels, 13 + 3 GB through the memory system against 48: 7.7 k points/s against 6.7 k (K = 100: 3.38 against 2.91 k); MFMA
// pipe 0.52 busy, SQ_WAIT_ANY 0.35 at 1.8 waves per SIMD (256 registers hold a prefetch distance of one k-step, deeper ones spill).
// Where the two column halves of a line are poorly filled it loses (N = 15: 8.1 k against 15.1 k points/s with the first version),
// so it takes N = 20 ... 22 only; the other sizes stay on the step kernels.
//
// Layout (see vsm_raman_quad.hip for the instruction's lane map):  B / D operand register (I, j) = element [4 I + q][4 (JH half + j) + l]
// of the lane's line;  A operand register (I, K) = element [4 I + l][4 K + q] (both lane groups of a line hold the same values).
#include <type_traits>

#include "vsm_common.h"
#include "vsm_internal.h"

namespace vsm {
namespace {

template <int N>
struct ccfg {
  static constexpr int RB = (N + 3) / 4;                 // row / contraction blocks
  static constexpr int JH = ((N + 2 + 3) / 4 + 1) / 2;   // column blocks per lane group (both groups: incl. the rider columns N, N + 1)
  static constexpr int cA = N, cB = N + 1;
  static constexpr int REM = N - 4 * (RB - 1);           // valid rows (columns) of the last block of real rows (columns)
  static constexpr int hE = (RB - 1) / JH, jE = (RB - 1) % JH;   // lane group / register of that block as a column block
  static constexpr int FL = ((N * N + 31) / 32) * 32 + 4;        // line stride of the state images ((FL mod 32) = 4)
};
template <int R, int C>
struct cmat {
  double v[R][C];
};
template <int R>
struct camat {
  double v[R][R];
};
template <int R>
struct cvecr {
  double x[R];   // row 4 I + q
};
struct cpos {
  int lane, q, l, line, half;
  int tr4[2];   // ds_bpermute address of the transposed lane (l, (line, h), q) per source column half h
  __device__ __forceinline__ void init(int lane_) {
    lane = lane_;
    q = lane >> 4;
    l = lane & 3;
    line = (lane >> 3) & 1;
    half = (lane >> 2) & 1;
    tr4[0] = 4 * (16 * l + 4 * (2 * line + 0) + q);
    tr4[1] = 4 * (16 * l + 4 * (2 * line + 1) + q);
  }
};
template <int R, int C>
__device__ __forceinline__ void c_zero(cmat<R, C>& m) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) m.v[i][j] = 0.0;
}
__device__ __forceinline__ double c_bperm(double x, int addr4) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
// accumulator layout -> left-operand layout: column block k of the matrix lives in lane group k / C, register k % C
template <int R, int C>
__device__ __forceinline__ void c_transpose(camat<R>& a, const cmat<R, C>& m, const cpos& p) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int k = 0; k < R; ++k) a.v[i][k] = c_bperm(m.v[i][k % C], p.tr4[k / C]);
}
template <int Ccol, int C>
__device__ __forceinline__ bool c_owns(const cpos& p) {
  return p.half == (Ccol >> 2) / C && p.l == (Ccol & 3);
}
template <int Ccol, int R, int C>
__device__ __forceinline__ cvecr<R> c_col(const cmat<R, C>& m) {   // valid on the lanes that own the column
  cvecr<R> x;
#pragma unroll
  for (int i = 0; i < R; ++i) x.x[i] = m.v[i][(Ccol >> 2) % C];
  return x;
}
template <int Ccol, int R, int C>
__device__ __forceinline__ void c_set_col(cmat<R, C>& m, const cvecr<R>& x, const cpos& p) {
  const bool mine = c_owns<Ccol, C>(p);
#pragma unroll
  for (int i = 0; i < R; ++i) m.v[i][(Ccol >> 2) % C] = mine ? x.x[i] : m.v[i][(Ccol >> 2) % C];
}
// the values held by the lanes that own column Ccol -> every lane of the same (q, line)
template <int Ccol, int C, int R>
__device__ __forceinline__ cvecr<R> c_bcast(const cvecr<R>& x, const cpos& p) {
  cvecr<R> y;
  const int src4 = 4 * ((p.lane & 48) | ((2 * p.line + (Ccol >> 2) / C) << 2) | (Ccol & 3));
#pragma unroll
  for (int i = 0; i < R; ++i) y.x[i] = c_bperm(x.x[i], src4);
  return y;
}
template <int N>
__device__ __forceinline__ void c_load_v(cvecr<ccfg<N>::RB>& x, const double* __restrict__ g, const cpos& p) {
  constexpr int RB = ccfg<N>::RB, REM = ccfg<N>::REM;
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const bool e = (i == RB - 1) && REM < 4;
    const double v = g[4 * i + (e ? min(p.q, REM - 1) : p.q)];
    x.x[i] = (!e || p.q < REM) ? v : 0.0;
  }
}

// ---- operand fetchers ----------------------------------------------------------------------------------------------------------
template <int R>
struct ca_reg {
  const camat<R>& A;
  __device__ __forceinline__ double operator()(int i, int k) const { return A.v[i][k]; }
};
template <int R, int C>
struct cb_reg {
  const cmat<R, C>& B;
  __device__ __forceinline__ double operator()(int k, int j) const { return B.v[k][j]; }
};
// Per-lane element offsets into a flat column-major N x N block for the two lane maps (interior / last row block / last column
// block / corner; the lanes past the last real row or column are clamped onto it and their values masked).  One table per wave:
// a fetcher is then a base pointer + these 32-bit offsets (six 64-bit pointers per operand were what the allocator spilled).
template <int N>
struct coff {
  int a00, a10, a01, a11;             // left operand:  element [4 i + l][4 k + q]
  int b00, b10, b01, b11, bd0, bd1;   // right operand: element [4 k + q][4 (JH half + j) + l]; bd: the second group's blocks past the real columns
  bool arok, acok, brok, cedge, second, mA, mB;
  __device__ __forceinline__ void init(const cpos& p) {
    constexpr int REM = ccfg<N>::REM, JH = ccfg<N>::JH, jE = ccfg<N>::jE;
    const int lc = min(p.l, REM - 1), qc = min(p.q, REM - 1);
    a00 = p.l + N * p.q;
    a10 = lc + N * p.q;
    a01 = p.l + N * qc;
    a11 = lc + N * qc;
    const int c0 = 4 * JH * p.half + p.l;                  // column of register j = 0
    const int cl = min(c0 + 4 * jE, N - 1) - 4 * jE;       // the same, clamped for the register of the edge block
    const int cd = min(c0, N - 1);
    b00 = p.q + N * c0;
    b10 = qc + N * c0;
    b01 = p.q + N * cl;
    b11 = qc + N * cl;
    bd0 = p.q + N * cd;
    bd1 = qc + N * cd;
    arok = p.l < REM;
    acok = p.q < REM;
    brok = p.q < REM;
    cedge = p.half == ccfg<N>::hE && p.l >= REM;
    second = p.half == 1;
    mA = c_owns<ccfg<N>::cA, JH>(p);
    mB = c_owns<ccfg<N>::cB, JH>(p);
  }
};
// left operand from a flat block (an LDS state image or global memory)
template <int N>
struct ca_mem {
  const double* g;
  const coff<N>& o;
  __device__ __forceinline__ double operator()(int i, int k) const {
    constexpr int RB = ccfg<N>::RB, REM = ccfg<N>::REM;
    const int off = 4 * i + N * 4 * k;
    const bool ie = (i == RB - 1) && REM < 4, ke = (k == RB - 1) && REM < 4;
    const double x = g[(ie ? (ke ? o.a11 : o.a10) : (ke ? o.a01 : o.a00)) + off];
    return ((!ie || o.arok) && (!ke || o.acok)) ? x : 0.0;
  }
};
// left operand = the transpose of a matrix held in the accumulator layout, fragment by fragment (ds_bpermute: no 72-register copy)
template <int R, int C>
struct ca_tr {
  const cmat<R, C>& M;
  const cpos& p;
  __device__ __forceinline__ double operator()(int i, int k) const { return c_bperm(M.v[i][k % C], p.tr4[k / C]); }
};
// right operand from a flat block; NR rider columns (N: r1, N + 1: r2) from registers
template <int N, int NR>
struct cb_mem {
  const double* g;
  const coff<N>& o;
  const cvecr<ccfg<N>::RB>* r1;
  const cvecr<ccfg<N>::RB>* r2;
  __device__ __forceinline__ double operator()(int k, int j) const {
    constexpr int RB = ccfg<N>::RB, JH = ccfg<N>::JH, REM = ccfg<N>::REM, cA = ccfg<N>::cA, cB = ccfg<N>::cB, jE = ccfg<N>::jE;
    const bool re = (k == RB - 1) && REM < 4, ce = (j == jE) && REM < 4;
    const int off = 4 * k + N * 4 * j;
    const int ob = re ? (ce ? o.b11 : o.b10) : (ce ? o.b01 : o.b00);
    double x;
    if (JH + j < RB) {   // a block of real columns in both lane groups
      const double y = g[ob + off];
      x = ((!re || o.brok) && (!ce || !o.cedge)) ? y : 0.0;
    } else {             // real in the first group only: the second group's lanes read a valid address and are masked
      const double y = g[o.second ? (re ? o.bd1 : o.bd0) + 4 * k : ob + off];
      x = (!o.second && (!re || o.brok) && (!ce || !o.cedge)) ? y : 0.0;
    }
    if (NR >= 1 && j == ((cA >> 2) % JH)) x = o.mA ? r1->x[k] : x;
    if (NR >= 2 && j == ((cB >> 2) % JH)) x = o.mB ? r2->x[k] : x;
    return x;
  }
};
// acc += A B with the left / right fragments requested DA / DB k-steps ahead.  (Measured on the C5 shape: distances 1 ... 3 for the
// operands that stream from memory are within 2 % of each other -- 7.06 / 6.94 / 7.07 / 6.89 k points/s for (DB, DA) = (1, 1), (2, 1),
// (2, 2), (3, 2) -- the second wave of the SIMD, not the distance, covers the stream; 1 / 1 has the fewest scratch reloads.)
template <int DA, int DB, int R, int C, typename FA, typename FB>
__device__ __forceinline__ void c_mm(cmat<R, C>& acc, const FA& fa, const FB& fb) {
  static_assert(DA >= 1 && DB >= 1, "depth");
  double a[DA + 1][R], bb[DB + 1][C];
#pragma unroll
  for (int d = 0; d < DA; ++d)
    if (d < R) {
#pragma unroll
      for (int i = 0; i < R; ++i) a[d][i] = fa(i, d);
    }
#pragma unroll
  for (int d = 0; d < DB; ++d)
    if (d < R) {
#pragma unroll
      for (int j = 0; j < C; ++j) bb[d][j] = fb(d, j);
    }
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (k + DB < R) {
#pragma unroll
      for (int j = 0; j < C; ++j) bb[(k + DB) % (DB + 1)][j] = fb(k + DB, j);
    }
    if (k + DA < R) {
#pragma unroll
      for (int i = 0; i < R; ++i) a[(k + DA) % (DA + 1)][i] = fa(i, k + DA);
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < C; ++j)
        acc.v[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k % (DA + 1)][i], bb[k % (DB + 1)][j], acc.v[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);   // (keep the written order: fragments of the steps ahead requested, then the MFMAs of step k --
                                         //  left alone the scheduler hoists every load of a product to its top and the wave spills)
  }
}
template <int N, int NR>
__device__ __forceinline__ void c_read_b(cmat<ccfg<N>::RB, ccfg<N>::JH>& m, const cb_mem<N, NR>& fb) {
#pragma unroll
  for (int k = 0; k < ccfg<N>::RB; ++k)
#pragma unroll
    for (int j = 0; j < ccfg<N>::JH; ++j) m.v[k][j] = fb(k, j);
}
// the N x N part of a matrix -> the lane's line of a flat state image
template <int N>
__device__ __forceinline__ void c_write_img(double* img, const cmat<ccfg<N>::RB, ccfg<N>::JH>& m, const cpos& p) {
  constexpr int RB = ccfg<N>::RB, JH = ccfg<N>::JH;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < JH; ++j) {
      const int row = 4 * i + p.q, col = 4 * (JH * p.half + j) + p.l;
      if (row < N && col < N) img[row + N * col] = m.v[i][j];
    }
}

struct rc_steps {
  const double *r, *t, *ttg, *gt, *gr, *grt;   // [nd][S][N N]
  const double *jp, *j1m, *tmp1, *tmp2;         // [nd][S][N]
  const double* expk;                           // [nd][S]
  long long sm, sv, ss;                         // step strides (elements) of the matrix, vector and scalar arrays
};

#ifndef RC_DEPTH
#define RC_DEPTH 1
#endif
#ifndef RC_DEPTH_A
#define RC_DEPTH_A 1
#endif

template <int N>
__global__ __launch_bounds__(64, 2) void k_raman_doubling_chain(int S, int K, int nd, const int* __restrict__ shift, const rc_steps e,
                                                                double* ier, double* iet, double* ieJp, double* ieJm, int ns,
                                                                double* ier_pm, double* iet_mm) {
  using Q = ccfg<N>;
  constexpr int RB = Q::RB, JH = Q::JH, NN = N * N, cA = Q::cA, cB = Q::cB, FL = Q::FL;
  constexpr int DG = RC_DEPTH, DGA = RC_DEPTH_A;   // prefetch distance (k-steps) of right / left operands that stream from memory
  __shared__ double IER[2 * FL];
  __shared__ double IET[2 * FL];
  __shared__ double VJ[2][2][4 * RB];   // iej0+, iej0- of the two lines
  cpos p;
  p.init(threadIdx.x);
  coff<N> o;
  o.init(p);
  // workgroup -> (DONOR point, pair rank).  The two lines of a wave share their donor n0 = np (recipients np - shift[d]): both
  // lane-group pairs stream the same donor blocks -- identical addresses within an instruction -- so a step's donor operands are
  // fetched once per wave instead of once per line.  XCD-aware and DONOR-major (the pair rank runs fastest): XCD x owns a contiguous
  // eighth of the points, and the ~ 256 waves in flight on it are ALL pair ranks of a dozen neighbouring donors walking the steps
  // together -- a donor's operands of a step come into that L2 once and are hit by the other waves of the donor (pair rank by pair rank
  // every wave fetched them itself: 12 of the 21 GB a layer-moment moved).
  const int NP = (K + 1) >> 1;
  const int per = (S + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int pi = idx % NP, np = xcd * per + idx / NP;
  if (np >= S) return;
  int dsel[2] = {-1, -1};
  int cnt = 0;
  {
    int base = 0;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int i = 64 * hh + p.lane;
      const int sh = (i < K) ? shift[i] : 0;
      const int nr = np - sh;   // the recipient of line i whose donor is np
      const bool inb = i < K && nr >= 0 && nr < S;
      const unsigned long long bal = __ballot(inb);
      const int rank = base + __popcll(bal & ((1ull << p.lane) - 1ull));
      base += __popcll(bal);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        const unsigned long long mb = __ballot(inb && rank == 2 * pi + bb);
        if (mb) dsel[bb] = 64 * hh + __ffsll((long long)mb) - 1;
      }
    }
    cnt = base;
  }
  if (pi == 0) {   // out-of-band lines keep zero D-mirrors (the operator-level apply_D! writes every block)
    for (int dz = 0; dz < K; ++dz) {
      const int n0 = np + shift[dz];   // (this chore is per RECIPIENT: the wave of pair rank 0 of point np does it for recipient np)
      if (n0 >= 0 && n0 < S) continue;
      const long long oz = ((long long)np + (long long)S * dz) * NN;
      for (int el = p.lane; el < NN; el += 64) {
        ier_pm[oz + el] = 0.0;
        iet_mm[oz + el] = 0.0;
      }
    }
  }
  if (2 * pi >= cnt) return;
  const bool valid = 2 * pi + p.line < cnt;
  const int dd = (valid && p.line == 1) ? dsel[1] : dsel[0];   // (a one-line last pair repeats its line; nothing of the copy is stored)
  const int n0 = np, n1 = np - shift[dd];   // the lane's line: donor np, recipient n1
  const long long o4 = ((long long)n1 + (long long)S * dd) * NN, o4v = ((long long)n1 + (long long)S * dd) * N;
  const long long e4_ = (long long)n0 * NN, e1_ = (long long)n0 * N, s4_ = (long long)n1 * NN;
  double* ierL = IER + p.line * FL;
  double* ietL = IET + p.line * FL;
  double* vjp = &VJ[0][p.line][0];
  double* vjm = &VJ[1][p.line][0];
  const bool rid = c_owns<cA, JH>(p);

  // ---- the state of the two lines goes on chip
  {
    cmat<RB, JH> m;
    c_read_b<N, 0>(m, cb_mem<N, 0>{ier + o4, o, nullptr, nullptr});
    c_write_img<N>(ierL, m, p);
    c_read_b<N, 0>(m, cb_mem<N, 0>{iet + o4, o, nullptr, nullptr});
    c_write_img<N>(ietL, m, p);
    cvecr<RB> v, w;
    c_load_v<N>(v, ieJp + o4v, p);
    c_load_v<N>(w, ieJm + o4v, p);
    if (rid) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        vjp[4 * i + p.q] = v.x[i];
        vjm[4 * i + p.q] = w.x[i];
      }
    }
  }
  cmat<RB, JH> O1, O2;
  cvecr<RB> nJp, nJm;
  for (int s = 0; s < nd; ++s) {
    // (hide the loop invariance of the per-lane block offsets: hoisted out of the step loop, the derived 64-bit addresses of the
    // donor arrays are long-lived values the register allocator spills, and a scratch reload waits for every load in flight)
    long long e4 = e4_, e1 = e1_, s4 = s4_;
    asm volatile("" : "+v"(e4), "+v"(e1), "+v"(s4));
    const double* r_s = e.r + s * e.sm;
    const double* ttg_s = e.ttg + s * e.sm;
    const double e0 = e.expk[s * e.ss + n0];
    // ---- X = ier r0 + r1 ier      (rider columns of r0: j1-[n0], j0+[n0]  ->  X[:, cA] = ier j1-, X[:, cB] = ier j0+)
    cmat<RB, JH> X, WA, W1;
    {
      cvecr<RB> v1, v2;
      c_load_v<N>(v1, e.j1m + s * e.sv + e1, p);
      c_load_v<N>(v2, e.jp + s * e.sv + e1, p);
      c_zero(X);
      c_mm<1, DG>(X, ca_mem<N>{ierL, o}, cb_mem<N, 2>{r_s + e4, o, &v1, &v2});
    }
    c_read_b<N, 0>(WA, cb_mem<N, 0>{ierL, o, nullptr, nullptr});   // WA <- ier (rider columns zero): the accumulator of WA = ier + X gr0
    c_mm<DGA, 1>(X, ca_mem<N>{r_s + s4, o}, cb_reg<RB, JH>{WA});
    c_read_b<N, 0>(W1, cb_mem<N, 0>{ietL, o, nullptr, nullptr});   // W1 <- iet: the accumulator of W1 = iet + X gt0
    auto vj_read = [&](cvecr<RB>& a, const double* tab) {   // (the source vectors are re-read from their table where they are used)
#pragma unroll
      for (int i = 0; i < RB; ++i) a.x[i] = tab[4 * i + p.q];
    };
    {   // a3, a4 accumulate in the rider column cA of W1, WA (see k_raman_doubling_quad)
      cvecr<RB> cJp, cJm;
      vj_read(cJp, vjp);
      vj_read(cJm, vjm);
      const cvecr<RB> xA = c_col<cA>(X), xB = c_bcast<cB, JH>(c_col<cB>(X), p);
      cvecr<RB> u3, u4;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        u3.x[i] = cJp.x[i] + xA.x[i];
        u4.x[i] = cJm.x[i] * e0 + xB.x[i];
      }
      c_set_col<cA>(W1, u3, p);
      c_set_col<cA>(WA, u4, p);
    }
    {
      cvecr<RB> vt;
      c_load_v<N>(vt, e.tmp2 + s * e.sv + e1, p);
      c_mm<1, DG>(WA, ca_tr<RB, JH>{X, p}, cb_mem<N, 1>{e.gr + s * e.sm + e4, o, &vt, nullptr});   // WA = ier + X gr0 ; [:, cA] += X tmp2
      c_load_v<N>(vt, e.tmp1 + s * e.sv + e1, p);
      c_mm<1, DG>(W1, ca_tr<RB, JH>{X, p}, cb_mem<N, 1>{e.gt + s * e.sm + e4, o, &vt, nullptr});   // W1 = iet + X gt0 ; [:, cA] += X tmp1
    }
    // ---- R1IET = r1 [iet | iej1- | iej0+]
    cmat<RB, JH> W3;
    {
      cvecr<RB> j1, cJp;
      vj_read(cJp, vjp);
      vj_read(j1, vjm);
#pragma unroll
      for (int i = 0; i < RB; ++i) j1.x[i] *= e0;
      c_zero(W3);
      c_mm<DGA, 1>(W3, ca_mem<N>{r_s + s4, o}, cb_mem<N, 2>{ietL, o, &j1, &cJp});
    }
    cvecr<RB> a4;
    {
      const cvecr<RB> q1 = c_col<cA>(W3), q3 = c_col<cA>(W1);
      cvecr<RB> a3;
#pragma unroll
      for (int i = 0; i < RB; ++i) a3.x[i] = q3.x[i] + q1.x[i];
      c_set_col<cA>(W1, a3, p);
      const cvecr<RB> sv = c_bcast<cB, JH>(c_col<cB>(W3), p), q4 = c_col<cA>(WA);
#pragma unroll
      for (int i = 0; i < RB; ++i) a4.x[i] = q4.x[i] + sv.x[i];
    }
    // ---- W3 = WA t0 + r1 iet, column cA = a4
    {
      c_mm<1, DG>(W3, ca_tr<RB, JH>{WA, p}, cb_mem<N, 0>{e.t + s * e.sm + e4, o, nullptr, nullptr});
      c_set_col<cA>(W3, a4, p);
    }
    // ---- iet' = ttg1 W1 + iet gt0 ;  ier' = ier + iet grt0 + ttg1 W3   (columns cA: ttg1 a3 + iet tmp1, ttg1 a4 + iet tmp2)
    c_zero(O1);
    c_mm<DGA, 1>(O1, ca_mem<N>{ttg_s + s4, o}, cb_reg<RB, JH>{W1});
    c_read_b<N, 0>(O2, cb_mem<N, 0>{ierL, o, nullptr, nullptr});
    c_mm<DGA, 1>(O2, ca_mem<N>{ttg_s + s4, o}, cb_reg<RB, JH>{W3});
    {
      cvecr<RB> vt;
      c_load_v<N>(vt, e.tmp1 + s * e.sv + e1, p);
      c_mm<1, DG>(O1, ca_mem<N>{ietL, o}, cb_mem<N, 1>{e.gt + s * e.sm + e4, o, &vt, nullptr});
      c_load_v<N>(vt, e.tmp2 + s * e.sv + e1, p);
      c_mm<1, DG>(O2, ca_mem<N>{ietL, o}, cb_mem<N, 1>{e.grt + s * e.sm + e4, o, &vt, nullptr});
    }
    // ---- the new state: ieJ0+' = iej0+ expk0 + O1[:, cA] ;  ieJ0-' = iej0- + O2[:, cA] ;  iet' = O1, ier' = O2
    {
      cvecr<RB> cJp, cJm;
      vj_read(cJp, vjp);
      vj_read(cJm, vjm);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        nJp.x[i] = O1.v[i][(cA >> 2) % JH] + cJp.x[i] * e0;
        nJm.x[i] = O2.v[i][(cA >> 2) % JH] + cJm.x[i];
      }
    }
    if (s + 1 < nd) {
      c_write_img<N>(ietL, O1, p);
      c_write_img<N>(ierL, O2, p);
      if (rid) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
          vjp[4 * i + p.q] = nJp.x[i];
          vjm[4 * i + p.q] = nJm.x[i];
        }
      }
    }
  }
  // ---- out, with apply_D! (doubling_inelastic.jl:166-195: ier' rows flipped for U / V, D-mirrors ier+-, iet--)
  if (valid) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = 4 * i + p.q;
      const bool uvr = is_uv_row(row, ns);
      if (rid && row < N) {
        ieJp[o4v + row] = nJp.x[i];
        ieJm[o4v + row] = uvr ? -nJm.x[i] : nJm.x[i];
      }
#pragma unroll
      for (int j = 0; j < JH; ++j) {
        const int col = 4 * (JH * p.half + j) + p.l;
        if (row < N && col < N) {
          const long long o = o4 + row + N * col;
          const double a1 = O1.v[i][j], a2 = uvr ? -O2.v[i][j] : O2.v[i][j];
          const bool df = uvr != is_uv_row(col, ns);
          iet[o] = a1;
          ier[o] = a2;
          iet_mm[o] = df ? -a1 : a1;
          ier_pm[o] = df ? -a2 : a2;
        }
      }
    }
  }
}

#ifndef RC_N_LO
#define RC_N_LO 20
#endif
#ifndef RC_N_HI
#define RC_N_HI 22
#endif
// second window (round 5): N = 13, 14 -- N + 2 = 15, 16 columns fill the two column halves of a line exactly (two blocks of four
// each), unlike N = 15 ... 19; N = 14 is the Stokes_IQ scene of the moment m = 0 of a 7-stream Stokes_IQU run (C5).
#ifndef RC_N2_LO
#define RC_N2_LO 13
#endif
#ifndef RC_N2_HI
#define RC_N2_HI 14
#endif
template <int N, int HI, typename F>
int dispatch_rc(int n, F f) {
  if constexpr (N > HI) {
    return VSM_ERR_UNSUPPORTED;
  } else {
    if (n == N) return f(std::integral_constant<int, N>{});
    return dispatch_rc<N + 1, HI>(n, f);
  }
}

}  // namespace

// `stash`: the elastic operands of every step, step-major -- per array nd x S blocks: r, t, ttg, gt, gr, grt (N^2 each) | jp, j1m,
// tmp1, tmp2 (N each) | expk (1)
size_t raman_chain_stash_elems(int N, int S, int nd) {
  return (size_t)nd * ((size_t)6 * N * N * S + (size_t)4 * N * S + (size_t)S);
}
double* raman_chain_stash_ptr(double* stash, int N, int S, int nd, int step, int which) {
  const size_t per = (size_t)N * N * S, pv = (size_t)N * S;
  if (which < 6) return stash + (size_t)which * nd * per + (size_t)step * per;
  if (which < 10) return stash + (size_t)6 * nd * per + (size_t)(which - 6) * nd * pv + (size_t)step * pv;
  return stash + (size_t)6 * nd * per + (size_t)4 * nd * pv + (size_t)step * S;
}
bool raman_chain_supported(int N, int K) {
  static const bool off = ab_switch("VSM_NO_RAMAN_CHAIN");
  return !off && ((N >= RC_N_LO && N <= RC_N_HI) || (N >= RC_N2_LO && N <= RC_N2_HI)) && K <= 128 && K > 0;
}
// All nd doubling steps of the inelastic recurrences in one launch (FP64, 20 <= N <= 22, K <= 128; VSM_ERR_UNSUPPORTED otherwise).
// apply_D! of the inelastic operators happens on the way out (ns = n_stokes).  A wave reads and writes the blocks of its own lines only.
int raman_doubling_chain(int N, int S, int K, int nd, const int* shift, double* stash, double* ier, double* iet, double* ieJp,
                         double* ieJm, int ns, double* ier_pm, double* iet_mm, hipStream_t st) {
  if (!raman_chain_supported(N, K) || nd < 1 || ns < 1) return VSM_ERR_UNSUPPORTED;
  if (S <= 0) return VSM_OK;
  const long long blocks = 8LL * ((S + 7) / 8) * ((K + 1) / 2);
  if (blocks > 0x7fffffffLL) return VSM_ERR_UNSUPPORTED;
  rc_steps e;
  e.r = raman_chain_stash_ptr(stash, N, S, nd, 0, 0);
  e.t = raman_chain_stash_ptr(stash, N, S, nd, 0, 1);
  e.ttg = raman_chain_stash_ptr(stash, N, S, nd, 0, 2);
  e.gt = raman_chain_stash_ptr(stash, N, S, nd, 0, 3);
  e.gr = raman_chain_stash_ptr(stash, N, S, nd, 0, 4);
  e.grt = raman_chain_stash_ptr(stash, N, S, nd, 0, 5);
  e.jp = raman_chain_stash_ptr(stash, N, S, nd, 0, 6);
  e.j1m = raman_chain_stash_ptr(stash, N, S, nd, 0, 7);
  e.tmp1 = raman_chain_stash_ptr(stash, N, S, nd, 0, 8);
  e.tmp2 = raman_chain_stash_ptr(stash, N, S, nd, 0, 9);
  e.expk = raman_chain_stash_ptr(stash, N, S, nd, 0, 10);
  e.sm = (long long)N * N * S;
  e.sv = (long long)N * S;
  e.ss = S;
  auto launch = [&](auto tag) {
    hipLaunchKernelGGL((k_raman_doubling_chain<decltype(tag)::value>), dim3((unsigned)blocks), dim3(64), 0, st, S, K, nd, shift, e, ier,
                       iet, ieJp, ieJm, ns, ier_pm, iet_mm);
    VSM_LAUNCH_CHECK("k_raman_doubling_chain");
    return (int)VSM_OK;
  };
  if (N <= RC_N2_HI) return dispatch_rc<RC_N2_LO, RC_N2_HI>(N, launch);
  return dispatch_rc<RC_N_LO, RC_N_HI>(N, launch);
}

}  // namespace vsm
