// Rotational-Raman inelastic doubling step and interaction pass -- ONE WAVE PER RAMAN LINE (FP64, N <= 30).
//
// doubling_inelastic.jl:62-123 (the two `for dn` loops of doubling_helper!(::RRS, ...)).  Per (recipient point n1,
// line dn; donor n0 = n1 + shift[dn]) the step is ten 32 x 32 x Kend products plus eight mat-vecs:
//     X    = ier r0 + r1 ier              W1 = iet + X gt0          WA = ier + X gr0        W3 = WA t0 + r1 iet
//     iet' = ttg1 W1 + iet gt0            ier' = ier + iet grt0 + ttg1 W3          (+ the source recurrences, riding in
//                                                                                     the spare columns N, N+1)
// k_raman_doubling_lines (vsm_fused.hip) gives the four 16 x 16 output tiles of each product to four waves and needs
// ten workgroup barriers per line around 6-MFMA bursts.  Here a wave owns a whole line: right operands and every
// intermediate live in registers in the accumulator layout (the FP64 MFMA's B operand layout IS its accumulator layout,
// as in the strip kernels), left operands are wave-private k-major LDS images, and no barrier is needed after the two
// shared images (r1, ttg1) are staged.  Four waves (one per SIMD) walk the in-band lines of one recipient point
// round-robin.
//
// Global <-> register traffic is flat: a block is read as a contiguous array (lane + 64 j: coalesced, NF registers),
// written into a k-major image and read back in the accumulator layout by ds_read_b64 (gathering the accumulator layout
// straight from global memory costs 16 instructions per block, each touching 16 cache lines).  N is a template parameter:
// the chunk count, the image positions and the rider columns are compile-time, so the inner loop has no masks and no
// uniform branches.  The images are zero outside N x N except for the rider columns N, N+1; stale riders are harmless
// wherever they can appear (as columns k >= N of a left operand they meet zero rows of the right operand; as columns
// N, N+1 of an output they are never stored), so nothing is masked.
//
// LDS: 2 shared + 4 x 4 private images of 32 x 34 doubles = 153 KB.
#include "vsm_common.h"
#include "vsm_internal.h"
#include <type_traits>

namespace vsm {
namespace {

// diagnostic per-phase cycle stamps of workgroup 0 / wave 0 (build with -DRW_PHASE_TIMING; tools/raman_phase_timing.py)
#ifdef RW_PHASE_TIMING
__device__ unsigned long long rw_phase_cycles[16];
#define RW_STAMP_DECL                 \
  unsigned long long _t_acc[16] = {}; \
  unsigned long long _t_prev = __builtin_readcyclecounter()
#define RW_STAMP(i)                                             \
  do {                                                          \
    const unsigned long long _t = __builtin_readcyclecounter(); \
    _t_acc[i] += _t - _t_prev;                                  \
    _t_prev = _t;                                               \
  } while (0)
#define RW_STAMP_FLUSH()                                                           \
  do {                                                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0)                                       \
      for (int _i = 0; _i < 16; ++_i) atomicAdd(&rw_phase_cycles[_i], _t_acc[_i]); \
  } while (0)
#else
#define RW_STAMP_DECL
#define RW_STAMP(i)
#define RW_STAMP_FLUSH()
#endif

constexpr int WLD = 34;            // k-major row pitch: conflict-free ds_read_b64 for the (l15, kq) fragment pattern
constexpr int WIMG = 32 * WLD;     // doubles per image
constexpr int RW_WAVES = 4;
constexpr int RW_PRIV = 4;         // private images per wave
constexpr int RW_MAXN = 30;      // (rider columns N, N+1 must exist; rows 30, 31 of the pad columns take the stray lanes)
constexpr int RW_MAXN_SP = 24;   // software-pipelined doubling body: beyond, its blocks in flight spill

struct wmat {
  d4_t v[2][2];  // [row tile][column tile]; element r of a tile: row 16 a + kq + 4 r, column 16 b + l15
};
struct cvec {
  double x[2][4];  // rows 16 a + kq + 4 r (every lane of a kq group holds the same rows)
};
struct wpos {
  int l15, kq, lane;
};

__device__ __forceinline__ void w_zero(wmat& m) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) m.v[a][b] = acc_zero<double>();
}
__device__ __forceinline__ void w_add(wmat& m, const wmat& o) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) m.v[a][b] += o.v[a][b];
}
// accumulator layout -> k-major image (all 32 x 32)
__device__ __forceinline__ void w_to_img(double* L, const wmat& m, const wpos& p) {
  double* b0 = L + p.l15 + WLD * p.kq;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) b0[16 * b + WLD * (16 * a + 4 * r)] = m.v[a][b][r];
}
__device__ __forceinline__ void w_from_img(wmat& m, const double* img, const wpos& p) {
  const double* b0 = img + p.l15 + WLD * p.kq;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) m.v[a][b][r] = b0[16 * b + WLD * (16 * a + 4 * r)];
}
// acc += A B : A from a k-major image, B from registers; k runs in the order the accumulator layout stores it
template <int KS>
__device__ __forceinline__ void w_mm(wmat& acc, const double* A, const wmat& B, const wpos& p) {
  const double* a0 = A + p.kq + WLD * p.l15;
#pragma unroll
  for (int i = 0; i < KS; ++i) {
    const int a = i >> 2, r = i & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double af = a0[16 * a + 4 * r + WLD * 16 * t];
#pragma unroll
      for (int b = 0; b < 2; ++b) acc.v[t][b] = mfma<double>::mma(af, B.v[a][b][r], acc.v[t][b]);
    }
  }
}
// column C of m, valid on the lanes that hold it (l15 == C & 15)
template <int C>
__device__ __forceinline__ cvec w_col(const wmat& m) {
  cvec x;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) x.x[a][r] = m.v[a][C >> 4][r];
  return x;
}
template <int C>
__device__ __forceinline__ void w_set_col(wmat& m, const cvec& x, const wpos& p) {
  const bool mine = p.l15 == (C & 15);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) m.v[a][C >> 4][r] = mine ? x.x[a][r] : m.v[a][C >> 4][r];
}
// copy the values held by the lanes of column C to every lane of the same kq group
template <int C>
__device__ __forceinline__ cvec v_bcast(const cvec& x, const wpos& p) {
  cvec y;
  const int src = (p.lane & 48) | (C & 15);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) y.x[a][r] = __shfl(x.x[a][r], src, 64);
  return y;
}
// column `col` of an image (the pad columns 32, 33 hold vectors), rows in the cvec order
__device__ __forceinline__ void v_from_img(cvec& x, const double* img, int col, const wpos& p) {
  const double* b0 = img + col + WLD * p.kq;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) x.x[a][r] = b0[WLD * (16 * a + 4 * r)];
}
// the values the lanes of column C hold -> column `col` of an image
template <int C>
__device__ __forceinline__ void v_to_img(double* img, int col, const cvec& x, const wpos& p) {
  if (p.l15 == (C & 15)) {
    double* b0 = img + col + WLD * p.kq;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) b0[WLD * (16 * a + 4 * r)] = x.x[a][r];
  }
}

// compile-time loop: g(integral_constant<int, i>) for i in [LO, HI)
template <int LO, int HI, typename G>
__device__ __forceinline__ void static_for(G&& g) {
  if constexpr (LO < HI) {
    g(std::integral_constant<int, LO>{});
    static_for<LO + 1, HI>(g);
  }
}
// ---- flat (coalesced) block <-> image copies; chunk j covers the elements lane + 64 j ------------------------------------------
template <int N>
struct flat {
  static constexpr int NN = N * N, NF = (NN + 63) / 64, TAIL = NN - 64 * (NF - 1);   // TAIL lanes in the last chunk
  double f[NF];
};
template <int N>
struct flat_idx {
  int aidx[flat<N>::NF];   // image position of element lane + 64 j (the lanes past the block: an unused pad word)
  int etail;               // min(lane, TAIL - 1) + 64 (NF - 1)
};
template <int N>
__device__ __forceinline__ void f_index(flat_idx<N>& ix, int lane) {
  using F = flat<N>;
#pragma unroll
  for (int j = 0; j < F::NF; ++j) {
    const int e = lane + 64 * j, ec = min(e, F::NN - 1), col = ec / N;
    ix.aidx[j] = (e < F::NN) ? col + WLD * (ec - col * N) : 32 + (lane & 1) + WLD * (30 + ((lane >> 1) & 1));
  }
  ix.etail = min(lane, F::TAIL - 1) + 64 * (F::NF - 1);
}
template <int N>
__device__ __forceinline__ void f_load(flat<N>& x, const double* __restrict__ g, const flat_idx<N>& ix, int lane) {
  using F = flat<N>;
  const double* gl = g + lane;
#pragma unroll
  for (int j = 0; j < F::NF - 1; ++j) x.f[j] = gl[64 * j];
  x.f[F::NF - 1] = g[ix.etail];
}
template <int N>
__device__ __forceinline__ void f_to_img(double* img, const flat<N>& x, const flat_idx<N>& ix) {
#pragma unroll
  for (int j = 0; j < flat<N>::NF; ++j) img[ix.aidx[j]] = x.f[j];
}
template <int N>
__device__ __forceinline__ void img_to_global(double* __restrict__ g, const double* img, const flat_idx<N>& ix, int lane) {
  using F = flat<N>;
  double* gl = g + lane;
#pragma unroll
  for (int j = 0; j < F::NF - 1; ++j) gl[64 * j] = img[ix.aidx[j]];
  if (lane < F::TAIL) gl[64 * (F::NF - 1)] = img[ix.aidx[F::NF - 1]];
}

// apply_D! on the way out of the LAST doubling step (doubling_inelastic.jl:166-195): per flat chunk k, bit k of `flip`
// says the element sits in a U/V row, bit k of `diff` that row and column differ in that property.
struct dsign_masks {
  unsigned flip, diff;
  int ns;
};
template <int N>
__device__ __forceinline__ dsign_masks d_masks(int ns, int lane) {
  using F = flat<N>;
  dsign_masks m{0u, 0u, ns};
  if (ns > 0) {
#pragma unroll
    for (int j = 0; j < F::NF; ++j) {
      const int e = min(lane + 64 * j, F::NN - 1), col = e / N, row = e - col * N;
      const bool ui = is_uv_row(row, ns), uj = is_uv_row(col, ns);
      m.flip |= (ui ? 1u : 0u) << j;
      m.diff |= ((ui != uj) ? 1u : 0u) << j;
    }
  }
  return m;
}
// chunk k of an output block: g[e] = (FLIP and U/V row) ? -v : v ; when this is the last step also the D-mirror gm[e]
template <int N, int k, bool FLIP>
__device__ __forceinline__ void out_chunk(double* __restrict__ g, double* __restrict__ gm, double v, const dsign_masks& m,
                                          int lane) {
  using F = flat<N>;
  if (k < F::NF - 1 || lane < F::TAIL) {
    const double a = (FLIP && ((m.flip >> k) & 1u)) ? -v : v;
    g[lane + 64 * k] = a;
    if (m.ns > 0) gm[lane + 64 * k] = ((m.diff >> k) & 1u) ? -a : a;
  }
}
// out-of-band lines keep zero D-mirrors (the operator-level apply_D! writes every block)
template <int N>
__device__ __forceinline__ void zero_block(double* __restrict__ g, int lane) {
  using F = flat<N>;
#pragma unroll
  for (int j = 0; j < F::NF; ++j)
    if (j < F::NF - 1 || lane < F::TAIL) g[lane + 64 * j] = 0.0;
}
// the in-band lines of one wave (every RW_WAVES-th in-band line of the recipient point) as a 128-bit mask
struct line_list {
  unsigned long long mine[2];
  int sh[2];   // lane i: shift[i], shift[64 + i]
  int K;
  __device__ __forceinline__ void build(const int* __restrict__ shift, int K_, int S, int n1, int lane, int wave) {
    K = K_;
    int base = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 64 * h + lane;
      sh[h] = (i < K) ? shift[i] : 0;
      const int n0 = n1 + sh[h];
      const bool inb = i < K && n0 >= 0 && n0 < S;
      const unsigned long long bal = __ballot(inb);
      const int before = base + __popcll(bal & ((1ull << lane) - 1ull));
      base += __popcll(bal);
      mine[h] = __ballot(inb && (before & (RW_WAVES - 1)) == wave);
    }
  }
  __device__ __forceinline__ int next(int d) const {   // next line >= d of this wave, K if none
    if (d < 64) {
      const unsigned long long mm = mine[0] & (~0ull << d);
      if (mm) return __ffsll((long long)mm) - 1;
      d = 64;
    }
    if (d < 128) {
      const unsigned long long mm = mine[1] & (~0ull << (d - 64));
      if (mm) return 64 + __ffsll((long long)mm) - 1;
    }
    return K;
  }
  __device__ __forceinline__ int shift_of(int d) const { return __builtin_amdgcn_readlane(d < 64 ? sh[0] : sh[1], d & 63); }
};

// stage two shared N x N blocks as k-major images, zero the private images
template <int N>
__device__ __forceinline__ void stage_shared(double* A1, const double* __restrict__ g1, double* A2,
                                             const double* __restrict__ g2, double* priv, int tid) {
  for (int e = tid; e < 1024; e += 64 * RW_WAVES) {
    const int row = e & 31, col = e >> 5;
    const bool in = row < N && col < N;
    A1[col + WLD * row] = in ? g1[row + N * col] : 0.0;
    A2[col + WLD * row] = in ? g2[row + N * col] : 0.0;
  }
  for (int e = tid & 63; e < RW_PRIV * WIMG; e += 64) priv[e] = 0.0;
}

template <int N, bool LAST>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void k_raman_doubling_wave(
    int S, int K, const int* __restrict__ shift, const double* __restrict__ r, const double* __restrict__ t,
    const double* __restrict__ ttg, const double* __restrict__ gt, const double* __restrict__ gr,
    const double* __restrict__ grt, const double* __restrict__ jp, const double* __restrict__ j1m,
    const double* __restrict__ tmp1, const double* __restrict__ tmp2, const double* __restrict__ expk, double* ier,
    double* iet, double* ieJp, double* ieJm, int ns_arg, double* ier_pm, double* iet_mm) {
  const int ns = LAST ? ns_arg : 0;   // (compile-time zero in the steps before the last one: no D-mirror code at all)
  constexpr int KS = (N + 3) / 4, NN = N * N, cA = N, cB = N + 1;
  extern __shared__ __attribute__((aligned(16))) double rw_smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  wpos p;
  p.lane = tid & 63;
  p.l15 = p.lane & 15;
  p.kq = p.lane >> 4;
  const int lane = p.lane;
  double* R1a = rw_smem;
  double* TTGa = rw_smem + WIMG;
  double* IERa = rw_smem + (2 + RW_PRIV * wave) * WIMG;   // ier ; pad columns: iej0+, iej0-
  double* IETa = IERa + WIMG;                             // iet (+ columns N, N+1: iej1-, iej0+)
  double* Xa = IETa + WIMG;                               // X, later WA
  double* STa = Xa + WIMG;                                // staging of the right operands and of the outputs
  const int n1 = blockIdx.x;
  flat_idx<N> ix;
  f_index<N>(ix, lane);
  stage_shared<N>(R1a, r + (long long)n1 * NN, TTGa, ttg + (long long)n1 * NN, IERa, tid);
  __syncthreads();
  line_list ll;
  ll.build(shift, K, S, n1, lane, wave);
  const dsign_masks dm = d_masks<N>(ns, lane);
  if (ns > 0) {
    for (int dz = wave; dz < K; dz += RW_WAVES) {
      const int n0 = n1 + shift[dz];
      if (n0 >= 0 && n0 < S) continue;
      const long long oz = ((long long)n1 + (long long)S * dz) * (N * N);
      zero_block<N>(ier_pm + oz, lane);
      zero_block<N>(iet_mm + oz, lane);
    }
  }
  // the operands of a line as flat blocks (NF registers each) and one element per lane of the vectors;
  // rolling prefetch, at most five blocks in flight: a block is requested two to four product groups before its use
  flat<N> fIER, fIET, fR0, fGT, fGR, fGRT, fT0;
  double vJp = 0.0, vJm = 0.0, vj1m = 0.0, vjp0 = 0.0, vt1 = 0.0, vt2 = 0.0, e0 = 0.0;
  const bool vin = lane < N;
  const int vl = vin ? lane : 0;
  // per-lane image positions of the vector / rider stores; the lanes >= N write to an unused pad word, so the loop body
  // has no divergent branches and stays one scheduling region
  const int vdummy = 32 + (lane & 1) + WLD * (30 + ((lane >> 1) & 1));
  const int vp32 = vin ? 32 + WLD * lane : vdummy, vp33 = vin ? 33 + WLD * lane : vdummy;
  const int vpA = vin ? cA + WLD * lane : vdummy, vpB = vin ? cB + WLD * lane : vdummy;
  auto issue_ier = [&](int dd) {
    const long long o4 = ((long long)n1 + (long long)S * dd) * NN, o4v = ((long long)n1 + (long long)S * dd) * N;
    f_load<N>(fIER, ier + o4, ix, lane);
    vJp = ieJp[o4v + vl];
    vJm = ieJm[o4v + vl];
    e0 = expk[n1 + ll.shift_of(dd)];
  };
  auto issue_iet = [&](int dd) { f_load<N>(fIET, iet + ((long long)n1 + (long long)S * dd) * NN, ix, lane); };
  auto issue_r0_gt_gr = [&](int dd) {
    const int n0 = n1 + ll.shift_of(dd);
    const long long e4 = (long long)n0 * NN, e1 = (long long)n0 * N;
    f_load<N>(fR0, r + e4, ix, lane);
    f_load<N>(fGT, gt + e4, ix, lane);
    f_load<N>(fGR, gr + e4, ix, lane);
    vj1m = j1m[e1 + vl];
    vjp0 = jp[e1 + vl];
    vt1 = tmp1[e1 + vl];
    vt2 = tmp2[e1 + vl];
  };
  RW_STAMP_DECL;
  int d = ll.next(0);
  if (d < K) {
    issue_ier(d);
    issue_iet(d);
    issue_r0_gt_gr(d);
  }
  while (d < K) {
    const long long o4 = ((long long)n1 + (long long)S * d) * NN, o4v = ((long long)n1 + (long long)S * d) * N;
    const long long e4 = (long long)(n1 + ll.shift_of(d)) * NN;
    const int dnext = ll.next(d + 1);
    const int dpre = dnext < K ? dnext : d;   // (past the last line: re-request this line, unused)
    const double e0c = e0;
    RW_STAMP(15);
    // ---- images of ier, iet (+ riders iej1-, iej0+), r0 (+ riders j1-[n0], j0+[n0]); vectors into the pad columns
    f_to_img<N>(IERa, fIER, ix);
    f_to_img<N>(IETa, fIET, ix);
    f_to_img<N>(STa, fR0, ix);
    IERa[vp32] = vJp;
    IERa[vp33] = vJm;
    IETa[vpA] = vJm * e0c;
    IETa[vpB] = vJp;
    STa[vpA] = vj1m;
    STa[vpB] = vjp0;
    f_load<N>(fGRT, grt + e4, ix, lane);
    RW_STAMP(0);
    wmat X, R1IET, O2;
    {
      wmat Bw;
      w_from_img(Bw, STa, p);
      // X = ier r0 + r1 ier (columns N, N+1: ier j1-, ier j0+) ;  r1 iet (columns N, N+1: r1 iej1-, r1 iej0+)
      w_zero(X);
      w_mm<KS>(X, IERa, Bw, p);
      w_from_img(O2, IERa, p);
      w_mm<KS>(X, R1a, O2, p);
      w_from_img(Bw, IETa, p);
      w_zero(R1IET);
      w_mm<KS>(R1IET, R1a, Bw, p);
    }
    RW_STAMP(1);
    // ---- gt0 (+ rider tmp1) ;  X gt0 (column N: X tmp1), iet gt0 (column N: iet tmp1)
    f_to_img<N>(STa, fGT, ix);
    STa[vpA] = vt1;
    STa[vpB] = 0.0;
    w_to_img(Xa, X, p);
    f_load<N>(fT0, t + e4, ix, lane);
    RW_STAMP(2);
    wmat W1, O1;
    {
      wmat Bw;
      w_from_img(Bw, STa, p);
      w_zero(W1);
      w_mm<KS>(W1, Xa, Bw, p);
      w_zero(O1);
      w_mm<KS>(O1, IETa, Bw, p);
    }
    RW_STAMP(3);
    // ---- gr0 (+ rider tmp2) ;  X gr0 (column N: X tmp2)
    f_to_img<N>(STa, fGR, ix);
    STa[vpA] = vt2;
    issue_ier(dpre);
    RW_STAMP(4);
    wmat WA;
    {
      wmat Bw;
      w_from_img(Bw, STa, p);
      w_zero(WA);
      w_mm<KS>(WA, Xa, Bw, p);
    }
    RW_STAMP(5);
    // ---- grt0 (+ rider tmp2) ;  ier + iet grt0 (column N: iet tmp2)
    f_to_img<N>(STa, fGRT, ix);   // (the rider column still holds tmp2)
    issue_iet(dpre);
    RW_STAMP(6);
    {
      wmat Bw;
      w_from_img(Bw, STa, p);
      w_mm<KS>(O2, IETa, Bw, p);
    }
    RW_STAMP(7);
    // ---- W1 = iet + X gt0, column N = a3 = iej0+ + r1 iej1- + ier j1- + X tmp1
    cvec cJp, cJm;
    v_from_img(cJp, IERa, 32, p);
    v_from_img(cJm, IERa, 33, p);
    {
      const cvec q1 = w_col<cA>(R1IET), q2 = w_col<cA>(X), q3 = w_col<cA>(W1);
      cvec a3;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) a3.x[a][q] = cJp.x[a][q] + q1.x[a][q] + q2.x[a][q] + q3.x[a][q];
      wmat Bw;
      w_from_img(Bw, IETa, p);
      w_add(W1, Bw);
      w_set_col<cA>(W1, a3, p);
    }
    // ---- WA = ier + X gr0 ;  a4 = iej1- + ier j0+ + r1 iej0+ + X tmp2
    cvec a4;
    {
      const cvec q1 = w_col<cB>(X), q2 = w_col<cB>(R1IET), q3 = w_col<cA>(WA);
      cvec sv;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) sv.x[a][q] = q1.x[a][q] + q2.x[a][q];
      sv = v_bcast<cB>(sv, p);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) a4.x[a][q] = cJm.x[a][q] * e0c + sv.x[a][q] + q3.x[a][q];
      wmat Bw;
      w_from_img(Bw, IERa, p);
      w_add(WA, Bw);
    }
    w_to_img(Xa, WA, p);   // X is dead (the wave's own earlier reads retire in order)
    RW_STAMP(8);
    // ---- t0 ;  W3 = WA t0 + r1 iet, column N = a4
    f_to_img<N>(STa, fT0, ix);
    STa[vpA] = 0.0;
    issue_r0_gt_gr(dpre);
    RW_STAMP(9);
    {
      wmat Bw;
      w_from_img(Bw, STa, p);
      w_mm<KS>(R1IET, Xa, Bw, p);
      w_set_col<cA>(R1IET, a4, p);
    }
    // iet' = ttg1 W1 + iet gt0 ;  ieJ0+' = iej0+ expk0 + ttg1 a3 + iet tmp1  (column N of the same accumulator)
    w_mm<KS>(O1, TTGa, W1, p);
    // ier' = ier + iet grt0 + ttg1 W3 ;  ieJ0-' = iej0- + ttg1 a4 + iet tmp2
    w_mm<KS>(O2, TTGa, R1IET, p);
    RW_STAMP(10);
    // ---- outputs through the staging image: coalesced stores
    {
      cvec q1 = w_col<cA>(O1), q2 = w_col<cA>(O2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          q1.x[a][q] += cJp.x[a][q] * e0c;
          q2.x[a][q] += cJm.x[a][q];
        }
      v_to_img<cA>(STa, 32, q1, p);
      v_to_img<cA>(STa, 33, q2, p);
    }
    w_to_img(STa, O1, p);
    static_for<0, flat<N>::NF>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      out_chunk<N, k, false>(iet + o4, iet_mm + o4, STa[ix.aidx[k]], dm, lane);
    });
    w_to_img(STa, O2, p);
    static_for<0, flat<N>::NF>([&](auto kk) {
      constexpr int k = decltype(kk)::value;
      out_chunk<N, k, true>(ier + o4, ier_pm + o4, STa[ix.aidx[k]], dm, lane);
    });
    if (vin) {
      ieJp[o4v + lane] = STa[32 + WLD * lane];
      const double x = STa[33 + WLD * lane];
      ieJm[o4v + lane] = (ns > 0 && is_uv_row(lane, ns)) ? -x : x;
    }
    RW_STAMP(11);
    d = dnext;
  }
  RW_STAMP_FLUSH();
}

// ---- software-pipelined variant --------------------------------------------------------------------------------------------
// One wave per SIMD: nothing but the wave's own instruction order can overlap the LDS / VALU / global work with the MFMAs.
// Every product is issued as 4 KS "slots" of one MFMA; between the slots run the units of the side work that belongs to
// LATER products (staging the next right operand, reading it back, writing an intermediate's image, the rider algebra,
// the output copies), pinned in place by scheduling barriers.  The first phase of the next line (images of ier / iet / r0)
// rides in the last product of the current one.
#define RW_PIN() __builtin_amdgcn_sched_barrier(0)

// U units spread over the slots [S0, S1): run those of slot s
template <int U, int S0, int S1, int s, typename G>
__device__ __forceinline__ void units(G&& g) {
  if constexpr (s >= S0 && s < S1) {
    constexpr int n = S1 - S0, k = s - S0;
    static_for<k * U / n, (k + 1) * U / n>(g);
  }
}
// the 2 KS left-operand fragments of a product; requested during the previous product (units of its side work)
template <int KS>
struct afrag {
  double v[KS][2];
};
template <int KS, int u>
__device__ __forceinline__ void af_rd(afrag<KS>& f, const double* a0) {   // a0 = image + kq + WLD l15
  constexpr int i = u >> 1, t = u & 1;
  f.v[i][t] = a0[16 * (i >> 2) + 4 * (i & 3) + WLD * 16 * t];
}
template <int KS, typename F>
__device__ __forceinline__ void w_mm_s(wmat& acc, const afrag<KS>& af, const wmat& B, F&& side) {
  static_for<0, 4 * KS>([&](auto slot) {
    constexpr int s = decltype(slot)::value, i = s >> 2, t = (s >> 1) & 1, b = s & 1, a = i >> 2, r = i & 3;
    acc.v[t][b] = mfma<double>::mma(af.v[i][t], B.v[a][b][r], acc.v[t][b]);
    RW_PIN();
    side(slot);
    RW_PIN();
  });
}
// unit u of the 16 of an image <-> accumulator-layout copy
template <int u>
__device__ __forceinline__ void w_rd(wmat& m, const double* b0) {
  constexpr int a = u >> 3, b = (u >> 2) & 1, r = u & 3;
  m.v[a][b][r] = b0[16 * b + WLD * (16 * a + 4 * r)];
}
template <int u>
__device__ __forceinline__ void w_wr(double* b0, const wmat& m) {
  constexpr int a = u >> 3, b = (u >> 2) & 1, r = u & 3;
  b0[16 * b + WLD * (16 * a + 4 * r)] = m.v[a][b][r];
}
template <int u>
__device__ __forceinline__ void v_rd(cvec& x, const double* b0) {   // b0 = img + col + WLD kq
  x.x[u >> 2][u & 3] = b0[WLD * (16 * (u >> 2) + 4 * (u & 3))];
}

template <int N, bool LAST>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void k_raman_doubling_wave_sp(
    int S, int K, const int* __restrict__ shift, const double* __restrict__ r, const double* __restrict__ t,
    const double* __restrict__ ttg, const double* __restrict__ gt, const double* __restrict__ gr,
    const double* __restrict__ grt, const double* __restrict__ jp, const double* __restrict__ j1m,
    const double* __restrict__ tmp1, const double* __restrict__ tmp2, const double* __restrict__ expk, double* ier,
    double* iet, double* ieJp, double* ieJm, int ns_arg, double* ier_pm, double* iet_mm) {
  const int ns = LAST ? ns_arg : 0;   // (compile-time zero in the steps before the last one: no D-mirror code at all)
  constexpr int KS = (N + 3) / 4, SL = 4 * KS, H = 2 * KS, NN = N * N, cA = N, cB = N + 1;   // SL slots of ONE MFMA per product
  using F = flat<N>;
  constexpr int NF = F::NF;
  extern __shared__ __attribute__((aligned(16))) double rw_smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  wpos p;
  p.lane = tid & 63;
  p.l15 = p.lane & 15;
  p.kq = p.lane >> 4;
  const int lane = p.lane;
  double* R1a = rw_smem;
  double* TTGa = rw_smem + WIMG;
  double* IERa = rw_smem + (2 + RW_PRIV * wave) * WIMG;   // ier ; pad columns: iej0+, iej0-
  double* IETa = IERa + WIMG;                             // iet (+ columns N, N+1: iej1-, iej0+) ; pad column: iej0+ expk0
  double* Xa = IETa + WIMG;                               // X, later WA, later the image of ier'
  double* STa = Xa + WIMG;                                // staging of the right operands and of iet'
  const int n1 = blockIdx.x;
  flat_idx<N> ix;
  f_index<N>(ix, lane);
  stage_shared<N>(R1a, r + (long long)n1 * NN, TTGa, ttg + (long long)n1 * NN, IERa, tid);
  __syncthreads();
  line_list ll;
  ll.build(shift, K, S, n1, lane, wave);
  const dsign_masks dm = d_masks<N>(ns, lane);
  if (ns > 0) {
    for (int dz = wave; dz < K; dz += RW_WAVES) {
      const int n0 = n1 + shift[dz];
      if (n0 >= 0 && n0 < S) continue;
      const long long oz = ((long long)n1 + (long long)S * dz) * (N * N);
      zero_block<N>(ier_pm + oz, lane);
      zero_block<N>(iet_mm + oz, lane);
    }
  }
  flat<N> fIER, fIET, fR0, fGT, fGR, fGRT, fT0;
  double vJp = 0.0, vJm = 0.0, vj1m = 0.0, vjp0 = 0.0, vt1 = 0.0, vt2 = 0.0, e0 = 0.0;
  const bool vin = lane < N;
  const int vl = vin ? lane : 0;
  const int vdummy = 32 + (lane & 1) + WLD * (30 + ((lane >> 1) & 1));
  const int vp32 = vin ? 32 + WLD * lane : vdummy, vp33 = vin ? 33 + WLD * lane : vdummy;
  const int vpA = vin ? cA + WLD * lane : vdummy, vpB = vin ? cB + WLD * lane : vdummy;
  // accumulator-layout base offsets of the images
  const int wofs = p.l15 + WLD * p.kq, vofs = WLD * p.kq, aofs = p.kq + WLD * p.l15;
  afrag<KS> afA, afB;   // fragments of the current / next product's left operand
  auto af_units = [&](afrag<KS>& f, const double* img) {
    return [&f, img, aofs](auto u) { af_rd<KS, decltype(u)::value>(f, img + aofs); };
  };
  auto o4_of = [&](int dd) { return ((long long)n1 + (long long)S * dd) * NN; };
  auto o4v_of = [&](int dd) { return ((long long)n1 + (long long)S * dd) * N; };
  auto issue_ier = [&](int dd) {
    f_load<N>(fIER, ier + o4_of(dd), ix, lane);
    vJp = ieJp[o4v_of(dd) + vl];
    vJm = ieJm[o4v_of(dd) + vl];
    e0 = expk[n1 + ll.shift_of(dd)];
  };
  auto issue_iet = [&](int dd) { f_load<N>(fIET, iet + o4_of(dd), ix, lane); };
  auto issue_r0 = [&](int dd) {
    const int n0 = n1 + ll.shift_of(dd);
    f_load<N>(fR0, r + (long long)n0 * NN, ix, lane);
    vj1m = j1m[(long long)n0 * N + vl];
    vjp0 = jp[(long long)n0 * N + vl];
  };
  auto issue_gt_gr = [&](int dd) {
    const int n0 = n1 + ll.shift_of(dd);
    f_load<N>(fGT, gt + (long long)n0 * NN, ix, lane);
    f_load<N>(fGR, gr + (long long)n0 * NN, ix, lane);
    vt1 = tmp1[(long long)n0 * N + vl];
    vt2 = tmp2[(long long)n0 * N + vl];
  };
  auto issue_grt = [&](int dd) { f_load<N>(fGRT, grt + (long long)(n1 + ll.shift_of(dd)) * NN, ix, lane); };
  auto issue_t0 = [&](int dd) { f_load<N>(fT0, t + (long long)(n1 + ll.shift_of(dd)) * NN, ix, lane); };
  // unit lists of the first phase of a line (images of ier, iet, r0 with riders and pad vectors; then r0 read back)
  wmat Br0;
  auto head_units_a = [&](auto u) {   // NF + NF + 5 units
    constexpr int k = decltype(u)::value;
    if constexpr (k < NF) IERa[ix.aidx[k]] = fIER.f[k];
    else if constexpr (k < 2 * NF) IETa[ix.aidx[k - NF]] = fIET.f[k - NF];
    else if constexpr (k == 2 * NF) IERa[vp32] = vJp;
    else if constexpr (k == 2 * NF + 1) IERa[vp33] = vJm;
    else if constexpr (k == 2 * NF + 2) IETa[vpA] = vJm * e0;
    else if constexpr (k == 2 * NF + 3) IETa[vpB] = vJp;
    else IETa[vp32] = vJp * e0;
  };
  auto head_units_b = [&](auto u) {   // NF + 2 + 16 units
    constexpr int k = decltype(u)::value;
    if constexpr (k < NF) STa[ix.aidx[k]] = fR0.f[k];
    else if constexpr (k == NF) STa[vpA] = vj1m;
    else if constexpr (k == NF + 1) STa[vpB] = vjp0;
    else w_rd<k - NF - 2>(Br0, STa + wofs);
  };
  int d = ll.next(0);
  if (d < K) {
    issue_ier(d);
    issue_iet(d);
    issue_r0(d);
    issue_gt_gr(d);
    issue_grt(d);
    issue_t0(d);
    static_for<0, 2 * NF + 5>(head_units_a);
    static_for<0, NF + 18>(head_units_b);
    static_for<0, 2 * KS>(af_units(afA, IERa));
  }
  while (d < K) {
    const long long o4 = o4_of(d), o4v = o4v_of(d);
    const int dnext = ll.next(d + 1);
    const int dpre = dnext < K ? dnext : d;   // (past the last line: re-request this line, unused)
    wmat X, R1IET, O1, O2, W1, WA, Biet, Bgt, Bgr, Bgrt, Bt0;
    cvec cJp, cJm, cJ1m, cJpe, a3, a4;
    // P1: X = ier r0                                   | read ier (seed of ier', right operand of r1 ier) ; read iet + riders
    w_zero(X);
    w_mm_s<KS>(X, afA, Br0, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, H, SL, s>(af_units(afB, R1a));
      units<16, 0, H, s>([&](auto u) { w_rd<decltype(u)::value>(O2, IERa + wofs); });
      units<16, H, SL, s>([&](auto u) { w_rd<decltype(u)::value>(Biet, IETa + wofs); });
    });
    // P2: X += r1 ier                                  | stage gt0 (+ riders tmp1, 0) ; read it back
    w_mm_s<KS>(X, afB, O2, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, H, SL, s>(af_units(afA, IETa));
      units<NF + 2, 0, H, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < NF) STa[ix.aidx[k]] = fGT.f[k];
        else if constexpr (k == NF) STa[vpA] = vt1;
        else STa[vpB] = 0.0;
      });
      units<16, H, SL, s>([&](auto u) { w_rd<decltype(u)::value>(Bgt, STa + wofs); });
    });
    // P3: r1 iet (+ riders r1 iej1-, r1 iej0+)        | image of X ; stage gr0 (+ rider tmp2)
    w_zero(R1IET);
    w_mm_s<KS>(R1IET, afB, Biet, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<16, 0, H, s>([&](auto u) { w_wr<decltype(u)::value>(Xa + wofs, X); });
      units<NF + 1, H, SL, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < NF) STa[ix.aidx[k]] = fGR.f[k];
        else STa[vpA] = vt2;
      });
    });
    // P4: iet gt0 (column N: iet tmp1)                 | read gr0 ; stage grt0 (rider column still tmp2) ; next gt0, gr0
    w_zero(O1);
    w_mm_s<KS>(O1, afA, Bgt, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, H, SL, s>(af_units(afB, Xa));
      units<16, 0, H, s>([&](auto u) { w_rd<decltype(u)::value>(Bgr, STa + wofs); });
      units<NF, H, SL, s>([&](auto u) { STa[ix.aidx[decltype(u)::value]] = fGRT.f[decltype(u)::value]; });
      if constexpr (s == SL - 1) issue_gt_gr(dpre);
    });
    // P5: X gt0 (column N: X tmp1)                     | read grt0 ; next grt0
    w_zero(W1);
    w_mm_s<KS>(W1, afB, Bgt, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<16, 0, H, s>([&](auto u) { w_rd<decltype(u)::value>(Bgrt, STa + wofs); });
      if constexpr (s == SL - 1) issue_grt(dpre);
    });
    // P6: X gr0 (column N: X tmp2)                     | read iet, ier again (addends), pad vectors ; W1 += iet ; stage t0
    w_zero(WA);
    wmat Tiet, Tier;
    cvec w1A;
    w_mm_s<KS>(WA, afB, Bgr, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      if constexpr (s == 0) w1A = w_col<cA>(W1);   // X tmp1, before the riders of iet are added
      units<64, 0, H, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < 16) w_rd<k>(Tiet, IETa + wofs);
        else if constexpr (k < 32) w_rd<k - 16>(Tier, IERa + wofs);
        else if constexpr (k < 40) v_rd<k - 32>(cJp, IERa + 32 + vofs);
        else if constexpr (k < 48) v_rd<k - 40>(cJm, IERa + 33 + vofs);
        else if constexpr (k < 56) v_rd<k - 48>(cJ1m, IETa + cA + vofs);
        else v_rd<k - 56>(cJpe, IETa + 32 + vofs);
      });
      units<NF + 1 + 4, H, SL, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < NF) STa[ix.aidx[k]] = fT0.f[k];
        else if constexpr (k == NF) STa[vpA] = 0.0;
        else W1.v[(k - NF - 1) >> 1][(k - NF - 1) & 1] += Tiet.v[(k - NF - 1) >> 1][(k - NF - 1) & 1];
      });
      if constexpr (s == SL - 1) issue_ier(dpre);
    });
    // P7: ier + iet grt0 (column N: iet tmp2)          | a3, W1 column N ; WA += ier, a4 ; image of WA ; read t0 ; next t0, iet
    w_mm_s<KS>(O2, afA, Bgrt, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, 0, H, s>(af_units(afB, TTGa));
      if constexpr (s == 0) {
        // W1 column N = a3 = iej0+ + r1 iej1- + ier j1- + X tmp1
        const cvec q1 = w_col<cA>(R1IET), q2 = w_col<cA>(X);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) a3.x[a][q] = cJp.x[a][q] + q1.x[a][q] + q2.x[a][q] + w1A.x[a][q];
        w_set_col<cA>(W1, a3, p);
      }
      if constexpr (s == 2) {
        // a4 = iej1- + ier j0+ + r1 iej0+ + X tmp2
        const cvec q1 = w_col<cB>(X), q2 = w_col<cB>(R1IET), q3 = w_col<cA>(WA);
        cvec sv;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) sv.x[a][q] = q1.x[a][q] + q2.x[a][q];
        sv = v_bcast<cB>(sv, p);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int q = 0; q < 4; ++q) a4.x[a][q] = cJ1m.x[a][q] + sv.x[a][q] + q3.x[a][q];
      }
      // three 16-unit lists in the slots after the algebra (in thirds when there are at least three slots left)
      constexpr int B0 = SL > 4 ? 4 : 3, R = SL - B0, T = R >= 3 ? R / 3 : 1;
      constexpr int A_lo = B0, A_hi = B0 + T;
      constexpr int B_lo = R >= 2 ? B0 + T : B0, B_hi = R >= 3 ? B0 + 2 * T : SL;
      constexpr int C_lo = R >= 3 ? B0 + 2 * T : B_lo, C_hi = SL;
      units<4, A_lo, A_hi, s>([&](auto u) {   // WA += ier
        constexpr int k = decltype(u)::value;
        WA.v[k >> 1][k & 1] += Tier.v[k >> 1][k & 1];
      });
      units<16, B_lo, B_hi, s>([&](auto u) { w_wr<decltype(u)::value>(Xa + wofs, WA); });
      units<16, C_lo, C_hi, s>([&](auto u) { w_rd<decltype(u)::value>(Bt0, STa + wofs); });
      if constexpr (s == SL - 1) {
        issue_t0(dpre);
        issue_iet(dpre);
      }
    });
    // P8: iet' = iet gt0 + ttg1 W1 (column N: ttg1 a3 + iet tmp1)     | next r0
    w_mm_s<KS>(O1, afB, W1, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, H, SL, s>(af_units(afA, Xa));
      if constexpr (s == SL - 1) issue_r0(dpre);
    });
    // P9: W3 = r1 iet + WA t0                           | image of iet' (+ ieJ0+' into a pad column), its coalesced store
    flat<N> fOut;
    double vOut = 0.0;
    w_mm_s<KS>(R1IET, afA, Bt0, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<17, 0, H, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < 16) {
          w_wr<k>(STa + wofs, O1);
        } else {
          cvec q1 = w_col<cA>(O1);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) q1.x[a][q] += cJpe.x[a][q];
          v_to_img<cA>(STa, 32, q1, p);
        }
      });
      constexpr int M0 = H + (SL - H) / 2, R_hi = (M0 > H) ? M0 : H + 1, W_lo = (M0 > H) ? M0 : H;
      units<NF + 1, H, R_hi, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < NF) fOut.f[k] = STa[ix.aidx[k]];
        else vOut = STa[vin ? 32 + WLD * lane : vdummy];
      });
      units<NF + 1, W_lo, SL, s>([&](auto u) {
        constexpr int k = decltype(u)::value;
        if constexpr (k < NF) out_chunk<N, k, false>(iet + o4, iet_mm + o4, fOut.f[k], dm, lane);
        else {
          if (vin) ieJp[o4v + lane] = vOut;
        }
      });
    });
    w_set_col<cA>(R1IET, a4, p);   // W3 column N = a4
    // P10: ier' = ier + iet grt0 + ttg1 W3 (column N: ttg1 a4 + iet tmp2)    | first phase of the next line
    w_mm_s<KS>(O2, afB, R1IET, [&](auto sl) {
      constexpr int s = decltype(sl)::value;
      units<2 * KS, H, SL, s>(af_units(afA, IERa));
      units<2 * NF + 5, 0, H, s>(head_units_a);
      units<NF + 18, H, SL, s>(head_units_b);
    });
    // image of ier' (+ ieJ0-'), coalesced store
    {
      cvec q2 = w_col<cA>(O2);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) q2.x[a][q] += cJm.x[a][q];
      v_to_img<cA>(Xa, 33, q2, p);
      w_to_img(Xa, O2, p);
      static_for<0, NF>([&](auto kk) {
        constexpr int k = decltype(kk)::value;
        out_chunk<N, k, true>(ier + o4, ier_pm + o4, Xa[ix.aidx[k]], dm, lane);
      });
      if (vin) {
        const double x = Xa[33 + WLD * lane];
        ieJm[o4v + lane] = (ns > 0 && is_uv_row(lane, ns)) ? -x : x;
      }
    }
    d = dnext;
  }
}

// One pass of interaction_helper!(::RRS, ::ScatteringInterface_11) (interaction_inelastic.jl:319-521), one wave per line
// (operand roles: rs_ia_pass in vsm_internal.h):
//   W1 = L1 E0[n0] + L2 I1 ;  Y = YA + TI W1 ;  W3 = L1 E3[n0] + L2 I3
//   OUTA = ACCA + TI W3 + Y GX[n0] ;  OUTB = TI I4 + Y GY[n0]
//   V1 = L1 VE0[n0] + L2 VI1 + VADD ;  VOUT = VACC + TI V1 + Y VV[n0]          (riding in spare column N)
// Shared images: L2, TI; private: L1, Y, two staging images (right operands alternate between them).
template <int N>
__global__ __launch_bounds__(64 * RW_WAVES, 1) void k_raman_interaction_wave(int S, int K, const int* __restrict__ shift,
                                                                             rs_ia_pass<double> h) {
  constexpr int KS = (N + 3) / 4, NN = N * N, cA = N;
  extern __shared__ __attribute__((aligned(16))) double rw_smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  wpos p;
  p.lane = tid & 63;
  p.l15 = p.lane & 15;
  p.kq = p.lane >> 4;
  const int lane = p.lane;
  double* L2a = rw_smem;
  double* TIa = rw_smem + WIMG;
  double* L1a = rw_smem + (2 + RW_PRIV * wave) * WIMG;   // pad columns: VADD, VACC
  double* Ya = L1a + WIMG;
  double* SAa = Ya + WIMG;
  double* SBa = SAa + WIMG;
  const int n1 = blockIdx.x;
  flat_idx<N> ix;
  f_index<N>(ix, lane);
  stage_shared<N>(L2a, h.L2 + (long long)n1 * h.sL2, TIa, h.TI + (long long)n1 * NN, L1a, tid);
  __syncthreads();
  line_list ll;
  ll.build(shift, K, S, n1, lane, wave);
  // blocks in use order: L1, E0, I1 | E3, I3 | YA, ACCA | I4 | GX, GY
  flat<N> fL1, fE0, fI1, fE3, fI3, fYA, fAC, fI4, fGX, fGY;
  double vE0 = 0.0, vI1 = 0.0, vVV = 0.0, vAD = 0.0, vAC = 0.0;
  const bool vin = lane < N;
  const int vl = vin ? lane : 0;
  auto issue_a = [&](int dd) {   // L1, E0, I1 (+ VE0, VI1, VADD, VACC)
    const int n0 = n1 + ll.shift_of(dd);
    const long long o4 = ((long long)n1 + (long long)S * dd) * NN, o4v = ((long long)n1 + (long long)S * dd) * N;
    f_load<N>(fL1, h.L1 + o4, ix, lane);
    f_load<N>(fE0, h.E0 + n0 * h.sE0, ix, lane);
    f_load<N>(fI1, h.I1 + o4, ix, lane);
    vE0 = h.VE0[(long long)n0 * N + vl];
    vI1 = h.VI1[o4v + vl];
    vAD = h.VADD[o4v + vl];
    vAC = h.VACC[o4v + vl];
  };
  auto issue_b = [&](int dd) {   // E3, I3
    const int n0 = n1 + ll.shift_of(dd);
    const long long o4 = ((long long)n1 + (long long)S * dd) * NN;
    f_load<N>(fE3, h.E3 + n0 * h.sE3, ix, lane);
    f_load<N>(fI3, h.I3 + o4, ix, lane);
  };
  int d = ll.next(0);
  if (d < K) {
    issue_a(d);
    issue_b(d);
  }
  while (d < K) {
    const int n0 = n1 + ll.shift_of(d);
    const long long o4 = ((long long)n1 + (long long)S * d) * NN, o4v = ((long long)n1 + (long long)S * d) * N;
    const int dnext = ll.next(d + 1);
    // ---- L1 image; E0 (+ rider VE0[n0]) and I1 (+ rider VI1) ;  W1 = L1 E0 + L2 I1
    f_to_img<N>(L1a, fL1, ix);
    f_to_img<N>(SAa, fE0, ix);
    f_to_img<N>(SBa, fI1, ix);
    if (vin) {
      L1a[32 + WLD * lane] = vAD;
      L1a[33 + WLD * lane] = vAC;
      SAa[cA + WLD * lane] = vE0;
      SBa[cA + WLD * lane] = vI1;
    }
    f_load<N>(fYA, h.YA + o4, ix, lane);
    f_load<N>(fAC, h.ACCA + o4, ix, lane);
    wmat W1;
    {
      wmat Bw;
      w_from_img(Bw, SAa, p);
      w_zero(W1);
      w_mm<KS>(W1, L1a, Bw, p);
      w_from_img(Bw, SBa, p);
      w_mm<KS>(W1, L2a, Bw, p);
    }
    // ---- E3, I3 ;  W3 = L1 E3 + L2 I3
    f_to_img<N>(SAa, fE3, ix);
    f_to_img<N>(SBa, fI3, ix);
    if (vin) {
      SAa[cA + WLD * lane] = 0.0;
      SBa[cA + WLD * lane] = 0.0;
    }
    f_load<N>(fI4, h.I4 + o4, ix, lane);
    f_load<N>(fGX, h.GX + (long long)n0 * NN, ix, lane);
    vVV = h.VV[(long long)n0 * N + vl];
    wmat W3;
    {
      wmat Bw;
      w_from_img(Bw, SAa, p);
      w_zero(W3);
      w_mm<KS>(W3, L1a, Bw, p);
      w_from_img(Bw, SBa, p);
      w_mm<KS>(W3, L2a, Bw, p);
    }
    // column N of W1 becomes V1 = L1 VE0 + L2 VI1 + VADD
    cvec cAD, cAC;
    v_from_img(cAD, L1a, 32, p);
    v_from_img(cAC, L1a, 33, p);
    {
      cvec q = w_col<cA>(W1);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) q.x[a][i] += cAD.x[a][i];
      w_set_col<cA>(W1, q, p);
    }
    // ---- YA, ACCA as accumulator seeds ;  Y = YA + TI W1 (column N: TI V1) ;  A = ACCA + TI W3
    f_to_img<N>(SAa, fYA, ix);
    f_to_img<N>(SBa, fAC, ix);
    f_load<N>(fGY, h.GY + (long long)n0 * NN, ix, lane);
    wmat Y, A, B;
    w_from_img(Y, SAa, p);
    w_mm<KS>(Y, TIa, W1, p);
    w_from_img(A, SBa, p);
    w_mm<KS>(A, TIa, W3, p);
    // ---- I4 ;  B = TI I4
    f_to_img<N>(SAa, fI4, ix);
    if (dnext < K) issue_a(dnext);
    {
      wmat Bw;
      w_from_img(Bw, SAa, p);
      w_zero(B);
      w_mm<KS>(B, TIa, Bw, p);
    }
    const cvec vt = w_col<cA>(Y);
    w_to_img(Ya, Y, p);   // (its column N = TI V1 meets the zero row N of GX, GY)
    // ---- GX (+ rider VV[n0]), GY ;  A += Y GX (column N: Y VV) ;  B += Y GY
    f_to_img<N>(SBa, fGX, ix);
    if (vin) SBa[cA + WLD * lane] = vVV;
    f_to_img<N>(SAa, fGY, ix);
    if (dnext < K) issue_b(dnext);
    {
      wmat Bw;
      w_from_img(Bw, SBa, p);
      w_mm<KS>(A, Ya, Bw, p);
      w_from_img(Bw, SAa, p);
      w_mm<KS>(B, Ya, Bw, p);
    }
    // ---- outputs through the staging images
    {
      cvec q = w_col<cA>(A);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) q.x[a][i] += cAC.x[a][i] + vt.x[a][i];
      v_to_img<cA>(SAa, 32, q, p);
    }
    w_to_img(SBa, A, p);
    img_to_global<N>(h.OUTA + o4, SBa, ix, lane);
    w_to_img(SAa, B, p);
    img_to_global<N>(h.OUTB + o4, SAa, ix, lane);
    if (vin) h.VOUT[o4v + lane] = SAa[32 + WLD * lane];
    d = dnext;
  }
}

constexpr size_t RW_LDS_BYTES = (size_t)(2 + RW_PRIV * RW_WAVES) * WIMG * sizeof(double);

template <int N>
int launch_rw(int S, int K, const int* shift, const double* r, const double* t, const double* ttg, const double* gt,
              const double* gr, const double* grt, const double* jp, const double* j1m, const double* tmp1,
              const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp, double* ieJm, int ns,
              double* ier_pm, double* iet_mm, hipStream_t st) {
  static const bool plain = ab_switch("VSM_RAMAN_WAVE_PLAIN");   // the unpipelined body (A/B)
  using kern_t = void (*)(int, int, const int*, const double*, const double*, const double*, const double*, const double*,
                          const double*, const double*, const double*, const double*, const double*, const double*, double*,
                          double*, double*, double*, int, double*, double*);
  const bool last = ns > 0;
  kern_t kern = last ? (kern_t)k_raman_doubling_wave<N, true> : (kern_t)k_raman_doubling_wave<N, false>;
  if constexpr (N <= RW_MAXN_SP) {
    if (!plain) kern = last ? (kern_t)k_raman_doubling_wave_sp<N, true> : (kern_t)k_raman_doubling_wave_sp<N, false>;
  }
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kern), RW_LDS_BYTES, "hipFuncSetAttribute(k_raman_doubling_wave)"))
    return prepared;   // once per (device, kernel)
  hipLaunchKernelGGL(kern, dim3(S), dim3(64 * RW_WAVES), RW_LDS_BYTES, st, S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1,
                     tmp2, expk, ier, iet, ieJp, ieJm, ns, ier_pm, iet_mm);
  VSM_LAUNCH_CHECK("k_raman_doubling_wave");
  return VSM_OK;
}
template <int N>
int launch_rw_ia(int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st) {
  auto kern = k_raman_interaction_wave<N>;
  if (const int prepared = ensure_dyn_lds(reinterpret_cast<const void*>(kern), RW_LDS_BYTES, "hipFuncSetAttribute(k_raman_interaction_wave)"))
    return prepared;
  hipLaunchKernelGGL(kern, dim3(S), dim3(64 * RW_WAVES), RW_LDS_BYTES, st, S, K, shift, h);
  VSM_LAUNCH_CHECK("k_raman_interaction_wave");
  return VSM_OK;
}

// N -> instantiation.  The file is built in RW_PARTS objects (parallel build: each covers a range of N; -DRW_PART=k) or,
// without RW_PART, as one object with every N (tools/variantsrw.sh).
#ifdef RW_PART
#if RW_PART == 0
#define RW_N_LO 1
#define RW_N_HI 15
#elif RW_PART == 1
#define RW_N_LO 16
#define RW_N_HI 21
#else
#define RW_N_LO 22
#define RW_N_HI 30
#endif
#else
#define RW_N_LO 1
#define RW_N_HI 30
#endif
static_assert(RW_N_HI <= RW_MAXN, "range");
template <int N, typename F>
int dispatch_n(int n, F f) {
  if constexpr (N > RW_N_HI) {
    return VSM_ERR_UNSUPPORTED;
  } else {
    if (n == N) return f(std::integral_constant<int, N>{});
    return dispatch_n<N + 1>(n, f);
  }
}

}  // namespace

#define RW_DBL_ARGS                                                                                                        \
  int N, int S, int K, const int *shift, const double *r, const double *t, const double *ttg, const double *gt,            \
      const double *gr, const double *grt, const double *jp, const double *j1m, const double *tmp1, const double *tmp2,    \
      const double *expk, double *ier, double *iet, double *ieJp, double *ieJm, int ns, double *ier_pm, double *iet_mm,    \
      hipStream_t st
#define RW_DBL_PASS N, S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm, ns, ier_pm, iet_mm, st
#define RW_CAT_(a, b) a##b
#define RW_CAT(a, b) RW_CAT_(a, b)
#ifdef RW_PART
#define RW_PART_FN(name) RW_CAT(name, RW_PART)
#else
#define RW_PART_FN(name) RW_CAT(name, all)
#endif
int raman_doubling_wave_part_0(RW_DBL_ARGS);
int raman_doubling_wave_part_1(RW_DBL_ARGS);
int raman_doubling_wave_part_2(RW_DBL_ARGS);
int raman_interaction_wave_part_0(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st);
int raman_interaction_wave_part_1(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st);
int raman_interaction_wave_part_2(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st);

int RW_PART_FN(raman_doubling_wave_part_)(RW_DBL_ARGS) {
  if (N < RW_N_LO || N > RW_N_HI) return VSM_ERR_UNSUPPORTED;
  return dispatch_n<RW_N_LO>(N, [&](auto tag) {
    return launch_rw<decltype(tag)::value>(S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm,
                                           ns, ier_pm, iet_mm, st);
  });
}
int RW_PART_FN(raman_interaction_wave_part_)(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h,
                                             hipStream_t st) {
  if (N < RW_N_LO || N > RW_N_HI) return VSM_ERR_UNSUPPORTED;
  return dispatch_n<RW_N_LO>(N, [&](auto tag) { return launch_rw_ia<decltype(tag)::value>(S, K, shift, h, st); });
}

#if !defined(RW_PART) || RW_PART == 0
// FP64, N <= 30, K <= 128; VSM_ERR_UNSUPPORTED otherwise (the caller falls back to k_raman_doubling_lines / the operator
// chain).  K > 128: the line list is a 128-bit mask.  N 25..30 run the unpipelined body.  ns > 0 (n_stokes) marks the
// LAST doubling step of a layer: apply_D! of the inelastic operators happens on the way out (ier_pm, iet_mm are written).
int raman_doubling_wave(RW_DBL_ARGS) {
  static const bool off = ab_switch("VSM_NO_RAMAN_WAVE");
  if (off || N > RW_MAXN || N < 1 || K > 128) return VSM_ERR_UNSUPPORTED;
  if (S <= 0 || K <= 0) return VSM_OK;
#ifdef RW_PART
  int rc = raman_doubling_wave_part_0(RW_DBL_PASS);
  if (rc == VSM_ERR_UNSUPPORTED) rc = raman_doubling_wave_part_1(RW_DBL_PASS);
  if (rc == VSM_ERR_UNSUPPORTED) rc = raman_doubling_wave_part_2(RW_DBL_PASS);
  return rc;
#else
  return raman_doubling_wave_part_all(RW_DBL_PASS);
#endif
}

int raman_interaction_wave(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_RAMAN_WAVE") || ab_switch("VSM_NO_RAMAN_IA_WAVE");
  if (off || N > RW_MAXN || N < 1 || K > 128) return VSM_ERR_UNSUPPORTED;
  if (S <= 0 || K <= 0) return VSM_OK;
#ifdef RW_PART
  int rc = raman_interaction_wave_part_0(N, S, K, shift, h, st);
  if (rc == VSM_ERR_UNSUPPORTED) rc = raman_interaction_wave_part_1(N, S, K, shift, h, st);
  if (rc == VSM_ERR_UNSUPPORTED) rc = raman_interaction_wave_part_2(N, S, K, shift, h, st);
  return rc;
#else
  return raman_interaction_wave_part_all(N, S, K, shift, h, st);
#endif
}
#endif

#ifdef RW_PHASE_TIMING
extern "C" int vsm_debug_rw_phase(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(rw_phase_cycles), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(rw_phase_cycles), z, sizeof(z));
  }
  return 0;
}
#endif
}  // namespace vsm
