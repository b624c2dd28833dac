// Column-strip kernels, FP32 edition (64 < N <= 96): the fused layer step of vsm_strip.hip for the hyperspectral
// configuration (N = 96, Float32).  Same scheme -- every matrix lives as 16-column strips in the accumulator registers of
// the wave that owns the strip, B operands come straight from those registers, A operands from one of TWO LDS buffers,
// two workgroups per CU -- with the differences the f32 MFMA imposes:
//   * v_mfma_f32_16x16x4_f32 accumulator element r of a tile is row 4 (lane>>4) + r (f64: (lane>>4) + 4 r), so the tile
//     held by a lane, used as B operand of "k-step r", covers k = 16 tb + 4 (lane>>4) + r: the products walk k in that
//     permuted order (any order is fine as long as the A fragment is fetched for the same k);
//   * 6 waves (6 strips of 16 columns, 6 row tiles of 4 f32 = 24 VGPRs per strip), 384 threads;
//   * N = 96 leaves no spare column for the source vectors: their products  tt j, tmp j, r J0+, T01 u, R+- j0-, T21 z
//     are VALU mat-vecs over the A-form in LDS (4 lanes per row, two shuffles).
#include <stdlib.h>

#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_lds.h"

namespace vsm {

#ifdef VSM_PHASE_TIMING
__device__ unsigned long long vsm_phase_cycles_strip32v1[32];
#define VSM_STAMP_DECL unsigned long long _t_prev = __builtin_readcyclecounter()
#define VSM_STAMP(i)                                                     \
  do {                                                                   \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                           \
      const unsigned long long _t = __builtin_readcyclecounter();        \
      vsm_phase_cycles_strip32v1[i] += _t - _t_prev;                       \
      _t_prev = _t;                                                      \
    }                                                                    \
  } while (0)
#else
#define VSM_STAMP_DECL
#define VSM_STAMP(i)
#endif

namespace {

constexpr int FNP = 96;    // padded matrix size
constexpr int FNW = 6;     // waves = column strips
constexpr int FNT = 64 * FNW;
constexpr int FTL = FNP / 16;  // row tiles per strip

struct fstrip {
  f4_t v[FTL];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < FTL; ++a) v[a] = acc_zero<float>();
  }
};

struct fsmem32 {
  float P[FNP * FNP];
  float Q[FNP * FNP];
  float vec[10][FNP];
  float red[2][8];
  gj_scratch<float, FNP> gj;
};

// A-form swizzle of the FP32 strips.  The generic lidx() of vsm_lds.h separates the four k of an A fragment by the LOW
// bits of k; here a lane group (lane>>4 = kq) covers k = 16 tb + 4 kq + r, so the row XOR must depend on bits 2..3 of k:
//     lidx32(row, k) = (row ^ s32(k)) + 96 k ,   s32(k) = (k & 3) | (k & 8) | ((k & 4) << 2)
// s32 maps the low four bits of k onto span{1, 2, 8, 16} (4 is left out): both the A-fragment reads (16 rows x two kq
// per 32-lane group) and the accumulator-layout stores (rows 4 kq + r x 16 columns) hit 32 distinct banks
// (checked exhaustively; the generic swizzle gave 2-way conflicts on every fragment read and products at 1/3 of the
// MFMA rate).
__device__ __forceinline__ int s32(int b) { return (b & 3) | (b & 8) | ((b & 4) << 2); }
__device__ __forceinline__ int lidx32(int a, int b) { return (a ^ s32(b)) + FNP * b; }

// Per-lane addressing as "two base registers + immediate":
//   A fragment (row 16 t + l15, column k = 16 tb + 4 kq + r, ks = 4 tb + r):  lidx32 = ar[r] + bt[t] + 96 (16 tb + r)
//   strip element (row 16 ta + 4 kq + r, column col):                        lidx32 = sr[r] + ((16 ta) ^ p16)
struct fpos {
  int lane, wave, l15, kq, col;
  int ar[4], bt[FTL];
  int sr[4], p16;
  __device__ __forceinline__ fpos() {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    l15 = lane & 15;
    kq = lane >> 4;
    col = 16 * wave + l15;
    const int L = l15 ^ ((kq >> 1) << 3), pq = kq & 1;
#pragma unroll
    for (int r = 0; r < 4; ++r) ar[r] = (L ^ r) + 4 * FNP * kq;
#pragma unroll
    for (int t = 0; t < FTL; ++t) bt[t] = 16 * (t ^ pq);
    const int m = s32(col);
    p16 = m & 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) sr[r] = (r ^ (m & 3)) + ((4 * kq) ^ (m & 8)) + FNP * col;
  }
  __device__ __forceinline__ int row(int ta, int r) const { return 16 * ta + 4 * kq + r; }
  __device__ __forceinline__ int aidx(int t, int ks) const {
    return ar[ks & 3] + bt[t] + FNP * (16 * (ks >> 2) + (ks & 3));
  }
  __device__ __forceinline__ int sidx(int ta, int r) const { return sr[r] + ((16 * ta) ^ p16); }
};

// acc += A * B   (A: A-form in LDS, B: strip in registers); KS = 4 ceil(N / 16) MFMA steps
// Fragments are requested PF k-steps ahead: one f32 k-step is only 6 x 32 = 192 MFMA cycles, less than the LDS latency
// under load, so the single-step lookahead of the FP64 kernels (4 x 64 cycles per step) leaves the pipe at 40 %.
constexpr int PF = 1;
template <int KS>
__device__ __forceinline__ void mm_ab(fstrip& acc, const float* A, const fstrip& B, fpos& p) {
  float a[PF + 1][FTL];
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0)
#pragma unroll
    for (int t = 0; t < FTL; ++t) a[s0][t] = A[p.aidx(t, s0)];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + PF < KS) {
#pragma unroll
      for (int t = 0; t < FTL; ++t) a[(ks + PF) % (PF + 1)][t] = A[p.aidx(t, ks + PF)];
    }
    const float b = B.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < FTL; ++t) acc.v[t] = mfma<float>::mma(a[ks % (PF + 1)][t], b, acc.v[t]);
  }
}
template <int KS>
__device__ __forceinline__ void mm_ab2(fstrip& acc1, fstrip& acc2, const float* A, const fstrip& B1, const fstrip& B2,
                                       fpos& p) {
  float a[PF + 1][FTL];
#pragma unroll
  for (int s0 = 0; s0 < PF; ++s0)
#pragma unroll
    for (int t = 0; t < FTL; ++t) a[s0][t] = A[p.aidx(t, s0)];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    if (ks + PF < KS) {
#pragma unroll
      for (int t = 0; t < FTL; ++t) a[(ks + PF) % (PF + 1)][t] = A[p.aidx(t, ks + PF)];
    }
    const float b1 = B1.v[ks >> 2][ks & 3], b2 = B2.v[ks >> 2][ks & 3];
#pragma unroll
    for (int t = 0; t < FTL; ++t) {
      acc1.v[t] = mfma<float>::mma(a[ks % (PF + 1)][t], b1, acc1.v[t]);
      acc2.v[t] = mfma<float>::mma(a[ks % (PF + 1)][t], b2, acc2.v[t]);
    }
  }
}

template <typename F>
__device__ __forceinline__ void store_strip(float* dst, const fstrip& s, const fpos& p, F f) {
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      dst[p.sidx(ta, r)] = f(s.v[ta][r], row, p.col);
    }
}
__device__ __forceinline__ void load_strip(fstrip& s, const float* src, const fpos& p) {
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) s.v[ta][r] = src[p.sidx(ta, r)];
}
__device__ __forceinline__ void load_strip_global(fstrip& s, const float* __restrict__ g, int N, const fpos& p) {
  const int cc = min(p.col, N - 1);
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const float v = g[min(row, N - 1) + (long long)N * cc];
      s.v[ta][r] = (row < N && p.col < N) ? v : 0.0f;
    }
}
__device__ __forceinline__ void store_strip_global(float* __restrict__ g, const fstrip& s, int N, const fpos& p) {
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      if (row < N && p.col < N) g[row + (long long)N * p.col] = s.v[ta][r];
    }
}
__device__ __forceinline__ void dsym_strip(fstrip& d, const fstrip& x, int ns, const fpos& p) {
  const bool uc = is_uv_row(p.col, ns);
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) d.v[ta][r] = (is_uv_row(p.row(ta, r), ns) == uc) ? x.v[ta][r] : -x.v[ta][r];
}
// global column-major N x N -> A-form in LDS (zero padded); 96 of the 384 threads' lanes... one column per wave and
// pass: lanes 0..63 take rows 0..63, then rows 64..95
__device__ __forceinline__ void stage_aform(float* L, const float* __restrict__ g, int N, const fpos& p) {
  // all column loads in flight before the first LDS write (one workgroup per CU: a round trip per column is exposed)
  constexpr int NJ = FNP / FNW;
  float v0[NJ], v1[NJ];
#pragma unroll
  for (int c = 0; c < NJ; ++c) {
    const int j = p.wave + FNW * c;
    v0[c] = (p.lane < N && j < N) ? g[p.lane + (long long)N * j] : 0.0f;
    const int i = 64 + (p.lane & 31);
    v1[c] = (p.lane < 32 && i < N && j < N) ? g[i + (long long)N * j] : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < NJ; ++c) {
    const int j = p.wave + FNW * c;
    L[lidx32(p.lane, j)] = v0[c];
    if (p.lane < 32) L[lidx32(64 + p.lane, j)] = v1[c];
  }
}

// y1 = A x1, y2 = A x2 (x2 scaled) over the A-form in LDS.  Lane (kq, l15) of wave w walks row 16 w + l15 over the
// columns k = 16 tb + 4 kq + r -- exactly the A-fragment pattern of tile w, hence conflict-free -- and the four kq
// groups are summed with two shuffles.  Lanes with kq == 0 receive the sums.  x entries beyond N must be zero.
__device__ __forceinline__ void matvec2(const float* A, const float* x1, const float* x2, float scale2, float& y1, float& y2,
                                        const fpos& p) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ks = 0; ks < 4 * FTL; ++ks) {
    const int k = 16 * (ks >> 2) + 4 * p.kq + (ks & 3);
    const float a = A[p.ar[ks & 3] + 16 * (p.wave ^ (p.kq & 1)) + FNP * (16 * (ks >> 2) + (ks & 3))];
    s1 += a * x1[k];
    s2 += a * x2[k];
  }
  s2 *= scale2;
  s1 += __shfl_xor(s1, 16);
  s2 += __shfl_xor(s2, 16);
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  y1 = s1;
  y2 = s2;
}

__device__ __forceinline__ float strip_norm_bound(const fstrip& e, int N, fsmem32& sm, int& slot, const fpos& p) {
  float ss = 0;
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = e.v[ta][r];
      if (p.row(ta, r) < N && p.col < N) ss += v * v;
    }
  const float ws = wave_sum(ss * 1.0001f);
  if (p.lane == 0) sm.red[slot][p.wave] = ws;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < FNW; ++w) tot += sm.red[slot][w];
  slot ^= 1;
  return sqrtf(tot) * 1.001f;
}

__device__ __forceinline__ void gj_lds_strip(float* V, int N, gj_scratch<float, FNP>* sc) {
  using G = gj_cfg<FNP, FNT>;
  const int tr = threadIdx.x % G::TR, tc = threadIdx.x / G::TR;
  float g[G::RB][G::CB];
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      g[rb][cb] = (i < N && j < N) ? V[lidx32(i, j)] : ((i == j) ? 1.0f : 0.0f);
    }
  gj_invert<float, FNP, FNT, false>(g, N, *sc);
#pragma unroll
  for (int rb = 0; rb < G::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < G::CB; ++cb) {
      const int i = tr + G::TR * rb, j = tc * G::CB + cb;
      if (i < N && j < N) V[lidx32(i, sc->dst[j])] = g[rb][cb];
    }
  __syncthreads();
}

// G_s = strip of (I - E)^-1 (see invert_strip in vsm_strip.hip)
template <int KS>
__device__ __forceinline__ int invert_strip(fstrip& E, fstrip& G, float* W, int N, fsmem32& sm, int& slot, fpos& p) {
  const float nrm = strip_norm_bound(E, N, sm, slot, p);
  const float tol = num<float>::eps() * 0.25f;
  int K = 0;
  if (nrm < 0.3f) {
    const float lim = tol * (1.0f - nrm);
    const float n2 = nrm * nrm, n4 = n2 * n2, n8 = n4 * n4, n16 = n8 * n8;
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
  auto keep = [N](float a, int r, int c) { return (r < N && c < N) ? a : 0.0f; };
  if (K == 0) {
    store_strip(W, E, p, [=](float a, int r, int c) { return (r == c) ? 1.0f - keep(a, r, c) : -keep(a, r, c); });
    __syncthreads();
    gj_lds_strip(W, N, &sm.gj);
    load_strip(G, W, p);
    return 1;
  }
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const float e = keep(E.v[ta][r], row, p.col);
      E.v[ta][r] = e;
      G.v[ta][r] = (row == p.col && row < N) ? e + 1.0f : e;
    }
  if (K == 1) return 2;
  store_strip(W, E, p, [](float a, int, int) { return a; });
  __syncthreads();
  int cur = 1;
  for (;;) {
    fstrip W2;
    W2.zero();
    mm_ab<KS>(W2, W, E, p);
    cur *= 2;
    if (K == cur) {
#pragma unroll
      for (int ta = 0; ta < FTL; ++ta) G.v[ta] += W2.v[ta];
      break;
    }
    __syncthreads();
    store_strip(W, W2, p, [](float a, int, int) { return a; });
    __syncthreads();
    fstrip T;
    T.zero();
    mm_ab<KS>(T, W, G, p);
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta) G.v[ta] += T.v[ta];
    if (K == 2 * cur - 1) break;
    E = W2;
  }
  return 1 + K;
}

// ---------------------------------------------------------------------------
// elemental! + doubling! + apply_D!  (see ed_body in vsm_strip.hip; sources by mat-vec)
// On return: r_s = strip of r-+, t_s = strip of t++, sm.vec[0] = j0+, sm.vec[1] = j0-, all waves past a barrier.
// ---------------------------------------------------------------------------
template <int KS, bool MIX>
__device__ __forceinline__ void ed_body(fsmem32& sm, fpos& p, const quad<float>& q, int m, int ndoubl,
                                        const float* __restrict__ dtau, const float* __restrict__ varpi,
                                        const float* __restrict__ tau_sum, const float* __restrict__ F0,
                                        const zsrc<float>& z, fstrip& r_s, fstrip& t_s) {
  float* P = sm.P;
  float* Q = sm.Q;
  float* jp = sm.vec[0];
  float* jm = sm.vec[1];
  float* mus = sm.vec[2];
  float* wcs = sm.vec[3];
  float* xs = sm.vec[4];
  float* es = sm.vec[5];
  float* ems = sm.vec[6];
  float* va = sm.vec[2];  // after the elemental step the five helper vectors are free:  tt j0+, tt j1-, tmp j0+, tmp j1-
  float* vb = sm.vec[3];
  float* vc = sm.vec[4];
  float* vd = sm.vec[5];
  const int s = blockIdx.x;
  const int N = q.N, ns = q.n_stokes;
  const int tid = threadIdx.x;
  const float d = dtau[s], w = varpi[s];
  const int ncomp = MIX ? z.ncomp : 0;
  const long long NNz = (long long)q.N * q.N;
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

  if (tid < FNP) {
    mus[tid] = (tid < N) ? q.mu[tid] : 1.0f;
    const float wt = (tid < N) ? q.wt[tid] : 0.0f;
    wcs[tid] = (m == 0) ? wt / 2.0f : wt / 4.0f;
    const float x = d / mus[tid];
    xs[tid] = x;
    es[tid] = exp(-x);
    ems[tid] = expm1(-x);
  }
  __syncthreads();

  // ---- elemental (elemental.jl:289-334) ------------------------------------------------------------------------
  {
    const int j = p.col;
    const int jc = min(j, N - 1);
    const float mj = mus[j], wct = wcs[j], xj = xs[j], emj = ems[j], ej = es[j];
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta) {
      float zp[4], zm[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long zo = min(p.row(ta, r), N - 1) + (long long)N * jc;
        zp[r] = zget(Zp, zo);
        zm[r] = zget(Zm, zo);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = p.row(ta, r);
        float rr = 0.0f, tt = 0.0f;
        if (i < N && j < N) {
          const float mi = mus[i], xi = xs[i];
          if (wct > num<float>::eps()) {
            const float emi = ems[i];
            rr = w * zm[r] * (mj / (mi + mj)) * wct * (-(emi + emj + emi * emj));
            if (mi == mj) {
              if (i == j)
                tt = es[i] * (1.0f + w * zp[r] * xi * wct);
              else
                tt = ej * (w * zp[r] * xi * wct);
            } else {
              const float xm = fmax(xi, xj);
              const float ediff =
                  (xm < 0.5f && fabs(xi - xj) > 0.125f * xm) ? (emi - emj) : expdiff_neg<float>(xi, xj);
              tt = w * zp[r] * (mj / (mi - mj)) * wct * ediff;
            }
          } else {
            tt = (i == j) ? es[i] : 0.0f;
          }
          if (ndoubl >= 1 && is_uv_row(i, ns)) rr = -rr;
        }
        r_s.v[ta][r] = rr;
        t_s.v[ta][r] = tt;
      }
    }
  }
  // ---- SFI source (elemental.jl:348-392) -----------------------------------------------------------------------
  float vjp = 0.0f, vjm = 0.0f;
  if (tid < N) {
    const int i = tid;
    const int i_start = ns * q.i_mu0;
    const float wct02 = (m == 0) ? 0.5f : 0.25f;
    float zp = 0, zm = 0;
    for (int qq = 0; qq < ns; ++qq) {
      const long long zo = i + (long long)N * (i_start + qq);
      const float f = F0[qq + (long long)ns * s];
      zp += zget(Zp, zo) * f;
      zm += zget(Zm, zo) * f;
    }
    const float mi = mus[i], ms = mus[i_start];
    if (i >= i_start && i < i_start + ns)
      vjp = wct02 * w * zp * (d / mi) * exp(-d / mi);
    else
      vjp = wct02 * w * zp * (ms / (mi - ms)) * expdiff_neg<float>(d / mi, d / ms);
    vjm = wct02 * w * zm * (ms / (mi + ms)) * (-expm1(-d * ((1.0f / mi) + (1.0f / ms))));
    const float att = exp(-tau_sum[s] / ms);
    vjp *= att;
    vjm *= att;
    if (ndoubl >= 1 && is_uv_row(i, ns)) vjm = -vjm;
  }
  auto keepN = [N](float a, int r, int c) { return (r < N && c < N) ? a : 0.0f; };
  if (ndoubl > 0) {
    store_strip(P, r_s, p, keepN);
    store_strip(Q, t_s, p, keepN);
  }
  __syncthreads();  // (the helper vectors mus.. are dead from here on)
  if (tid < FNP) {
    jp[tid] = vjp;
    jm[tid] = vjm;
  }
  __syncthreads();

  // ---- doubling (rt_helpers.jl:102-166) ------------------------------------------------------------------------
  float expk = exp(-d / q.mu0);
  int slot = 0;
  const int mrow = 16 * p.wave + p.l15;   // row / lead lane of the mat-vecs
  const bool mlead = p.kq == 0;
  VSM_STAMP_DECL;
  VSM_STAMP(0);
  for (int n = 0; n < ndoubl; ++n) {
    fstrip G;
    {
      fstrip E;
      E.zero();
      mm_ab<KS>(E, P, r_s, p);
      VSM_STAMP(1);
      invert_strip<KS>(E, G, P, N, sm, slot, p);
      VSM_STAMP(2);
    }
    fstrip tt;
    tt.zero();
    mm_ab<KS>(tt, Q, G, p);
    load_strip(t_s, Q, p);
    __syncthreads();  // P (series powers) and Q (t) no longer read
    store_strip(P, tt, p, keepN);
    __syncthreads();  // tt complete in P
    VSM_STAMP(3);
    {
      float y1, y2;
      matvec2(P, jp, jm, expk, y1, y2, p);  // tt j0+ , tt j1-  (j1- = j0- expk)
      if (mlead) {
        va[mrow] = y1;
        vb[mrow] = y2;
      }
    }
    VSM_STAMP(4);
    fstrip tmp, tn;
    tmp.zero();
    tn.zero();
    mm_ab2<KS>(tmp, tn, P, r_s, t_s, p);
    store_strip(Q, tmp, p, keepN);
    __syncthreads();  // tmp complete in Q
    VSM_STAMP(5);
    {
      float y1, y2;
      matvec2(Q, jp, jm, expk, y1, y2, p);  // tmp j0+ , tmp j1-
      if (mlead) {
        vc[mrow] = y1;
        vd[mrow] = y2;
      }
    }
    VSM_STAMP(6);
    mm_ab<KS>(r_s, Q, t_s, p);  // r' = r + tmp t
    t_s = tn;
    __syncthreads();  // everybody finished reading P (tt), Q (tmp), jp, jm ; va..vd complete
    VSM_STAMP(7);
    // j0- <- j0- + tt j1- + tmp j0+ ; j0+ <- j1+ + tt j0+ + tmp j1-   (rt_helpers.jl:128-134)
    if (tid < FNP) {
      const float njm = jm[tid] + vb[tid] + vc[tid];
      const float njp = jp[tid] * expk + va[tid] + vd[tid];
      jm[tid] = (tid < N) ? njm : 0.0f;
      jp[tid] = (tid < N) ? njp : 0.0f;
    }
    expk = expk * expk;
    if (n + 1 < ndoubl) {
      store_strip(P, r_s, p, keepN);
      store_strip(Q, t_s, p, keepN);
    }
    __syncthreads();
    VSM_STAMP(8);
  }

  // ---- apply_D (doubling.jl:178-252) -----------------------------------------------------------------------------
  if (ndoubl >= 1) {
#pragma unroll
    for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (is_uv_row(p.row(ta, r), ns)) r_s.v[ta][r] = -r_s.v[ta][r];
    if (tid < FNP && is_uv_row(tid, ns)) jm[tid] = -jm[tid];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// interaction_helper!(::ScatteringInterface_11)  (see ia_body in vsm_strip.hip)
// ---------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void ia_body(fsmem32& sm, fpos& p, int N, int ns, const composite<float>& c, fstrip& r_s,
                                        fstrip& t_s, const float* __restrict__ r_pm, const float* __restrict__ t_mm) {
  float* P = sm.P;
  float* Q = sm.Q;
  float* vjp = sm.vec[0];
  float* vjm = sm.vec[1];
  float* vJp = sm.vec[2];
  float* vJm = sm.vec[3];
  float* vu = sm.vec[4];
  float* vz = sm.vec[5];
  const int s = blockIdx.x, tid = threadIdx.x;
  const long long NN = (long long)N * N;
  float* R_mp = c.R_mp + s * NN;
  float* R_pm = c.R_pm + s * NN;
  float* T_pp = c.T_pp + s * NN;
  float* T_mm = c.T_mm + s * NN;
  float* J0_p = c.J0_p + (long long)s * N;
  float* J0_m = c.J0_m + (long long)s * N;
  auto keepN = [N](float x, int r, int cc) { return (r < N && cc < N) ? x : 0.0f; };
  const int mrow = 16 * p.wave + p.l15;   // row / lead lane of the mat-vecs
  const bool mlead = p.kq == 0;
  int slot = 0;

  if (tid < FNP) {
    const bool in = tid < N;
    vJp[tid] = in ? J0_p[tid] : 0.0f;
    vJm[tid] = in ? J0_m[tid] : 0.0f;
  }
  fstrip X;
  load_strip_global(X, R_pm, N, p);  // R+- strip
  store_strip(P, r_s, p, keepN);     // [r-+] -> P
  VSM_STAMP_DECL;
  stage_aform(Q, T_mm, N, p);        // [T--] -> Q
  __syncthreads();
  VSM_STAMP(10);
  // u = r-+ J0+ + j0-
  {
    float y1, y2;
    matvec2(P, vJp, vJp, 1.0f, y1, y2, p);
    if (mlead) vu[mrow] = y1 + vjm[mrow];
  }
  // ---- G1 = (I - r-+ R+-)^-1 --------------------------------------------------------------------------------------
  fstrip G;
  {
    fstrip E;
    E.zero();
    mm_ab<KS>(E, P, X, p);
    VSM_STAMP(11);
    invert_strip<KS>(E, G, P, N, sm, slot, p);
  }
  __syncthreads();
  store_strip(P, G, p, keepN);  // [G1] -> P
  __syncthreads();
  VSM_STAMP(12);
  // ---- H = G1 r-+ ; T01 = T-- G1 ; T01 r-+ = T-- H -------------------------------------------------------------------
  fstrip H, A1;
  H.zero();
  mm_ab<KS>(H, P, r_s, p);
  A1.zero();
  mm_ab<KS>(A1, Q, G, p);
  X.zero();
  mm_ab<KS>(X, Q, H, p);
  __syncthreads();
  store_strip(P, X, p, keepN);   // [T01 r-+] -> P
  store_strip(Q, A1, p, keepN);  // [T01] -> Q
  __syncthreads();
  VSM_STAMP(13);
  // J0- += T01 u
  {
    float y1, y2;
    matvec2(Q, vu, vu, 1.0f, y1, y2, p);
    if (mlead && mrow < N) J0_m[mrow] = vJm[mrow] + y1;
  }
  // ---- R-+ += (T01 r-+) T++ -----------------------------------------------------------------------------------------
  {
    fstrip Tpp, acc;
    load_strip_global(Tpp, T_pp, N, p);
    load_strip_global(acc, R_mp, N, p);
    VSM_STAMP(14);
    mm_ab<KS>(acc, P, Tpp, p);
    store_strip_global(R_mp, acc, N, p);
  }
  VSM_STAMP(15);
  // ---- T-- = T01 t-- -------------------------------------------------------------------------------------------------
  {
    fstrip tmm, acc;
    if (ns) dsym_strip(tmm, t_s, ns, p); else load_strip_global(tmm, t_mm, N, p);
    acc.zero();
    mm_ab<KS>(acc, Q, tmm, p);
    store_strip_global(T_mm, acc, N, p);
  }
  __syncthreads();  // [T01 r-+] (P) and [T01] (Q) no longer read
  VSM_STAMP(16);
  // ---- G2 = I + R+- H  (push-through identity) ; z = J0+ + R+- j0- -----------------------------------------------------
  stage_aform(P, R_pm, N, p);      // [R+-] -> P
  store_strip(Q, t_s, p, keepN);   // [t++] -> Q
  __syncthreads();
  VSM_STAMP(17);
  G.zero();
  mm_ab<KS>(G, P, H, p);
  fstrip Rpm;
  load_strip(Rpm, P, p);  // R+- strip from its A-form
  {
    float y1, y2;
    matvec2(P, vjm, vjm, 1.0f, y1, y2, p);
    if (mlead) vz[mrow] = vJp[mrow] + y1;
  }
#pragma unroll
  for (int ta = 0; ta < FTL; ++ta)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = p.row(ta, r);
      const float g = keepN(G.v[ta][r], row, p.col);
      G.v[ta][r] = (row == p.col && row < N) ? g + 1.0f : g;
    }
  // ---- T21 = t++ G2 -----------------------------------------------------------------------------------------------------
  X.zero();
  mm_ab<KS>(X, Q, G, p);
  __syncthreads();  // [R+-] (P), [t++] (Q) no longer read ; z complete
  store_strip(P, X, p, keepN);  // [T21] -> P
  __syncthreads();
  VSM_STAMP(18);
  // J0+ = j0+ + T21 z
  {
    float y1, y2;
    matvec2(P, vz, vz, 1.0f, y1, y2, p);
    if (mlead && mrow < N) J0_p[mrow] = vjp[mrow] + y1;
  }
  // ---- T++ = T21 T++ ; tmp = T21 R+- -------------------------------------------------------------------------------------
  {
    fstrip Tpp, acc1, acc2;
    load_strip_global(Tpp, T_pp, N, p);
    acc1.zero();
    acc2.zero();
    mm_ab2<KS>(acc1, acc2, P, Tpp, Rpm, p);
    store_strip(Q, acc2, p, keepN);  // [T21 R+-] -> Q
    __syncthreads();
    store_strip_global(T_pp, acc1, N, p);
  }
  VSM_STAMP(19);
  // ---- R+- = r+- + tmp t-- -------------------------------------------------------------------------------------------------
  {
    fstrip tmm, acc;
    if (ns) {
      dsym_strip(tmm, t_s, ns, p);
      dsym_strip(acc, r_s, ns, p);
    } else {
      load_strip_global(tmm, t_mm, N, p);
      load_strip_global(acc, r_pm, N, p);
    }
    mm_ab<KS>(acc, Q, tmm, p);
    store_strip_global(R_pm, acc, N, p);
  }
  VSM_STAMP(20);
}

template <int KS>
__global__ __launch_bounds__(FNT, 2) void k_ia_strip32(int N, composite<float> c, added<float> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem32& sm = *reinterpret_cast<fsmem32*>(smem_raw);
  fpos p;
  const int s = blockIdx.x, tid = threadIdx.x;
  if (tid < FNP) {
    const bool in = tid < N;
    sm.vec[0][tid] = in ? a.j0_p[(long long)s * N + tid] : 0.0f;
    sm.vec[1][tid] = in ? a.j0_m[(long long)s * N + tid] : 0.0f;
  }
  fstrip r_s, t_s;
  load_strip_global(r_s, a.r_mp + s * a.mat_stride, N, p);
  load_strip_global(t_s, a.t_pp + s * a.mat_stride, N, p);
  __syncthreads();
  ia_body<KS>(sm, p, N, a.d_symmetric, c, r_s, t_s, a.d_symmetric ? nullptr : a.r_pm + s * a.mat_stride,
              a.d_symmetric ? nullptr : a.t_mm + s * a.mat_stride);
}

template <int KS, bool MIX>
__global__ __launch_bounds__(FNT, 2) void k_layer_strip32(quad<float> q, int m, int ndoubl, const float* __restrict__ dtau,
                                                          const float* __restrict__ varpi,
                                                          const float* __restrict__ tau_sum, const float* __restrict__ F0,
                                                          zsrc<float> z, int toa, composite<float> c) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fsmem32& sm = *reinterpret_cast<fsmem32*>(smem_raw);
  fpos p;
  fstrip r_s, t_s;
  ed_body<KS, MIX>(sm, p, q, m, ndoubl, dtau, varpi, tau_sum, F0, z, r_s, t_s);
  const int N = q.N, ns = q.n_stokes;
  if (toa) {
    const int s = blockIdx.x, tid = threadIdx.x;
    const long long NN = (long long)N * N;
    store_strip_global(c.R_mp + s * NN, r_s, N, p);
    store_strip_global(c.T_pp + s * NN, t_s, N, p);
    fstrip d;
    dsym_strip(d, r_s, ns, p);
    store_strip_global(c.R_pm + s * NN, d, N, p);
    dsym_strip(d, t_s, ns, p);
    store_strip_global(c.T_mm + s * NN, d, N, p);
    if (tid < N) {
      c.J0_p[(long long)s * N + tid] = sm.vec[0][tid];
      c.J0_m[(long long)s * N + tid] = sm.vec[1][tid];
    }
    return;
  }
  ia_body<KS>(sm, p, N, ns, c, r_s, t_s, nullptr, nullptr);
}

template <typename K>
static int enable_lds32(K kern, const char* what) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sizeof(fsmem32));
  return e == hipSuccess ? (int)VSM_OK : hip_fail(e, what);
}

template <int KS>
static int launch_layer32(const quad<float>& q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const zsrc<float>& z, int toa, const composite<float>& c,
                          hipStream_t st) {
  static int prepared = enable_lds32(k_layer_strip32<KS, false>, "hipFuncSetAttribute(k_layer_strip32)");
  static int prepared_mix = enable_lds32(k_layer_strip32<KS, true>, "hipFuncSetAttribute(k_layer_strip32 mix)");
  if (prepared) return prepared;
  if (prepared_mix) return prepared_mix;
  if (z.ncomp > 0)
    hipLaunchKernelGGL((k_layer_strip32<KS, true>), dim3(S), dim3(FNT), sizeof(fsmem32), st, q, m, ndoubl, dtau, varpi, tau_sum,
                       F0, z, toa, c);
  else
    hipLaunchKernelGGL((k_layer_strip32<KS, false>), dim3(S), dim3(FNT), sizeof(fsmem32), st, q, m, ndoubl, dtau, varpi, tau_sum,
                       F0, z, toa, c);
  VSM_LAUNCH_CHECK("k_layer_strip32");
  return VSM_OK;
}
template <int KS>
static int launch_ia32(int N, int S, const composite<float>& c, const added<float>& a, hipStream_t st) {
  static int prepared = enable_lds32(k_ia_strip32<KS>, "hipFuncSetAttribute(k_ia_strip32)");
  if (prepared) return prepared;
  hipLaunchKernelGGL(k_ia_strip32<KS>, dim3(S), dim3(FNT), sizeof(fsmem32), st, N, c, a);
  VSM_LAUNCH_CHECK("k_ia_strip32");
  return VSM_OK;
}

}  // namespace

bool strip32v1_supported(int N) {
  static const bool off = getenv("VSM_NO_STRIP") != nullptr || getenv("VSM_NO_STRIP32") != nullptr;
  return !off && N > 64 && N <= FNP;
}

int strip32v1_layer_forward(const quad<float>& q, int S, int m, int ndoubl, const float* dtau, const float* varpi,
                          const float* tau_sum, const float* F0, const zsrc<float>& z, int toa, const composite<float>& c,
                          hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (q.N > 80) return launch_layer32<24>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c, st);
  return launch_layer32<20>(q, S, m, ndoubl, dtau, varpi, tau_sum, F0, z, toa, c, st);
}

int strip32v1_interaction11(int N, int S, const composite<float>& c, const added<float>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (N > 80) return launch_ia32<24>(N, S, c, a, st);
  return launch_ia32<20>(N, S, c, a, st);
}

}  // namespace vsm

#ifdef VSM_PHASE_TIMING
extern "C" int vsm_debug_phase_cycles_strip32v1(unsigned long long* out_h, int reset) {
  if (out_h) (void)hipMemcpyFromSymbol(out_h, HIP_SYMBOL(vsm::vsm_phase_cycles_strip32v1), sizeof(unsigned long long) * 32);
  if (reset) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(vsm::vsm_phase_cycles_strip32v1), z, sizeof(z));
  }
  return 0;
}
#endif
