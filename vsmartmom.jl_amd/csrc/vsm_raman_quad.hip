// Rotational-Raman inelastic doubling step -- FOUR RAMAN LINES PER WAVE on v_mfma_f64_4x4x4 (FP64, 3 <= N <= 22; round 4).
//
// doubling_inelastic.jl:62-123 (the two `for dn` loops of doubling_helper!(::RRS, ...)).  Per (recipient point n1, line dn;
// donor n0 = n1 + shift[dn]) the step is ten N x N x N products and eight mat-vecs:
//     X    = ier r0 + r1 ier              W1 = iet + X gt0          WA = ier + X gr0        W3 = WA t0 + r1 iet
//     iet' = ttg1 W1 + iet gt0            ier' = ier + iet grt0 + ttg1 W3          (+ the source recurrences, riding in the
//                                                                                     spare columns N, N + 1)
// vsm_raman_wave.hip gives a line to a wave on 16 x 16 x 4 tiles: a 21 x 21 x 21 product executes 32 x 32 x 24 (38 % useful), and
// the kernel is bound by the MFMA pipe (67 % busy).  v_mfma_f64_4x4x4 runs FOUR independent 4 x 4 x 4 blocks per instruction at
// the same flop rate (tools/mfma_probe: 67 vs 70 TFLOP/s), and its layout gives each block its own operands:
//     lane = 16 q + 4 b + l :   A[i = l][k = q],  B[k = q][j = l],  D[i = q][j = l]   of block b          (tools/mfma44_probe)
// Here block b IS a Raman line: a wave walks four lines of one recipient at once, a matrix is NB x NB registers
// (NB = ceil((N + 2) / 4): N = 21 -> 24 x 24, 67 % useful), element [4 I + q][4 J + l] of line b in register (I, J) -- the layout
// of D and of B, so a product's result is the next product's right operand as it stands (as in every strip kernel of this
// library).  Left operands have the transposed lane map: those that come from memory are read that way from their LDS image
// (the interaction pass gathers three of them straight from memory, 32 contiguous bytes per four lanes), computed ones (X, WA, Y)
// are transposed in registers by ds_bpermute (no LDS storage).
//
// Operands that come from memory never sit in registers: a first version gathered them there and the allocator, at 512 registers,
// answered every load with "wait, spill" (4.1k points/s on C5 against the 6.0k of the wave-per-line kernel).  Now every N x N block
// is copied flat into LDS by LDS-DMA (global_load_lds: no registers, no address arithmetic, 256 B or 1 KB contiguous per
// instruction), one product ahead of its use, and the products stream their fragments from the images (36 + 36 ds_read_b64 per
// 216 MFMAs); both lane maps read the same flat image.  Two quad images (four lines each) alternate, the recipient's r1 / ttg1
// share a single-line image: 37 KB per wave, four single-wave workgroups per CU, no barrier.  Registers hold only what the products
// produce (peak: four matrices of 72 registers).
//
// Per line 540 MFMAs of 16 cycles (8.6 k cycles; the 16 x 16 x 4 form: 240 of 64 = 15.4 k) and ~ 900 other instructions (~ 860).
// Results leave through the idle images as full lines (q_img_put / q_img_out).  Measured: DESIGN.md 4.6b -- the kernels are bound by
// the bytes of a step (4 TB/s), not by the MFMA pipe (43 % busy) or by latency (a second wave per SIMD did not help).
#include <type_traits>

#include "vsm_common.h"
#include "vsm_internal.h"

namespace vsm {
namespace {

template <int N>
struct qcfg {
  static constexpr int RB = (N + 3) / 4;        // row / contraction blocks
  static constexpr int NB = (N + 2 + 3) / 4;    // column blocks incl. the rider columns N, N + 1
  static constexpr int cA = N, cB = N + 1;
  static constexpr int REM = N - 4 * (RB - 1);  // valid rows (columns) of the last block
};

// v[I][J] = element [4 I + q][4 J + l] of line b   (accumulator layout = right-operand layout)
template <int R, int C>
struct qmat {
  double v[R][C];
};
// v[I][K] = element [4 I + l][4 K + q] of line b   (left-operand layout)
template <int R>
struct amat {
  double v[R][R];
};
template <int R>
struct cvec {
  double x[R];   // row 4 I + q (every lane of a (q, b) group holds the same rows)
};
struct qpos {
  int lane, q, b, l;
  int tr4;      // ds_bpermute address of the transposed lane (l, b, q)
};

template <int R, int C>
__device__ __forceinline__ void q_zero(qmat<R, C>& m) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) m.v[i][j] = 0.0;
}
// acc[:, 0:CU) += A B[:, 0:CU)
template <int CU, int R, int C, int CB>
__device__ __forceinline__ void q_mm(qmat<R, C>& acc, const amat<R>& A, const qmat<R, CB>& B) {
  static_assert(CU <= C && CU <= CB, "columns");
#pragma unroll
  for (int k = 0; k < R; ++k)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < CU; ++j) acc.v[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(A.v[i][k], B.v[k][j], acc.v[i][j], 0, 0, 0);
}
__device__ __forceinline__ double q_bperm(double x, int addr4) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr4, __double2loint(x));
  const int hi = __builtin_amdgcn_ds_bpermute(addr4, __double2hiint(x));
  return __hiloint2double(hi, lo);
}
// the N x N part of a matrix in the accumulator layout -> left-operand layout (4 x 4 transposes of the lane groups)
template <int R, int C>
__device__ __forceinline__ void q_transpose(amat<R>& a, const qmat<R, C>& m, const qpos& p) {
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int k = 0; k < R; ++k) a.v[i][k] = q_bperm(m.v[i][k], p.tr4);
}
// gathers from a column-major N x N block; rows / columns >= N read as zero (clamped addresses)
template <int N>
__device__ __forceinline__ void q_load_a(amat<qcfg<N>::RB>& a, const double* __restrict__ g, const qpos& p) {
  constexpr int RB = qcfg<N>::RB, REM = qcfg<N>::REM;
  const int lc = min(p.l, REM - 1), qc = min(p.q, REM - 1);
  const double* g00 = g + p.l + N * p.q;    // interior blocks
  const double* g10 = g + lc + N * p.q;     // last row block
  const double* g01 = g + p.l + N * qc;     // last column block
  const double* g11 = g + lc + N * qc;
  const bool rok = p.l < REM, cok = p.q < REM;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int off = 4 * i + N * 4 * k;
      const bool ie = (i == RB - 1) && REM < 4, ke = (k == RB - 1) && REM < 4;
      const double x = ie ? (ke ? g11[off] : g10[off]) : (ke ? g01[off] : g00[off]);
      a.v[i][k] = ((!ie || rok) && (!ke || cok)) ? x : 0.0;
    }
}
template <int N, int C>
__device__ __forceinline__ void q_load_b(qmat<qcfg<N>::RB, C>& m, const double* __restrict__ g, const qpos& p) {
  constexpr int RB = qcfg<N>::RB, REM = qcfg<N>::REM;
  const int lc = min(p.l, REM - 1), qc = min(p.q, REM - 1);
  const double* g00 = g + p.q + N * p.l;
  const double* g10 = g + qc + N * p.l;     // last row block
  const double* g01 = g + p.q + N * lc;     // last column block
  const double* g11 = g + qc + N * lc;
  const bool rok = p.q < REM, cok = p.l < REM;
#pragma unroll
  for (int k = 0; k < RB; ++k)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      if (j >= RB) {   // a rider-only block
        m.v[k][j] = 0.0;
        continue;
      }
      const int off = 4 * k + N * 4 * j;
      const bool re = (k == RB - 1) && REM < 4, ce = (j == RB - 1) && REM < 4;
      const double x = re ? (ce ? g11[off] : g10[off]) : (ce ? g01[off] : g00[off]);
      m.v[k][j] = ((!re || rok) && (!ce || cok)) ? x : 0.0;
    }
}
template <int N>
__device__ __forceinline__ void q_load_v(cvec<qcfg<N>::RB>& x, const double* __restrict__ g, const qpos& p) {
  constexpr int RB = qcfg<N>::RB, REM = qcfg<N>::REM;
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const bool e = (i == RB - 1) && REM < 4;
    const double v = g[4 * i + (e ? min(p.q, REM - 1) : p.q)];
    x.x[i] = (!e || p.q < REM) ? v : 0.0;
  }
}
// column Ccol of m, valid on the lanes that hold it (l == Ccol & 3)
template <int Ccol, int R, int C>
__device__ __forceinline__ cvec<R> q_col(const qmat<R, C>& m) {
  cvec<R> x;
#pragma unroll
  for (int i = 0; i < R; ++i) x.x[i] = m.v[i][Ccol >> 2];
  return x;
}
template <int Ccol, int R, int C>
__device__ __forceinline__ void q_set_col(qmat<R, C>& m, const cvec<R>& x, const qpos& p) {
  const bool mine = p.l == (Ccol & 3);
#pragma unroll
  for (int i = 0; i < R; ++i) m.v[i][Ccol >> 2] = mine ? x.x[i] : m.v[i][Ccol >> 2];
}
template <int Ccol, int R, int C>
__device__ __forceinline__ void q_clear_col(qmat<R, C>& m, const qpos& p) {
  const bool mine = p.l == (Ccol & 3);
#pragma unroll
  for (int i = 0; i < R; ++i) m.v[i][Ccol >> 2] = mine ? 0.0 : m.v[i][Ccol >> 2];
}
// the values held by the lanes of column Ccol -> every lane of the same (q, b) group
template <int Ccol, int R>
__device__ __forceinline__ cvec<R> q_bcast(const cvec<R>& x, const qpos& p) {
  cvec<R> y;
  const int src4 = 4 * ((p.lane & ~3) | (Ccol & 3));
#pragma unroll
  for (int i = 0; i < R; ++i) y.x[i] = q_bperm(x.x[i], src4);
  return y;
}

// ---- LDS images: a line's N x N block as it lies in memory (flat, column-major), copied by LDS-DMA ------------------------------
#ifndef RQ_DMAW
#define RQ_DMAW 16   // bytes per lane and DMA instruction (4 or 16)
#endif
template <int N>
struct qimg {
  static constexpr int W = RQ_DMAW;
  static constexpr int BYTES = 8 * N * N;
  static constexpr int CH = (BYTES + 64 * W - 1) / (64 * W);   // DMA instructions per line
  static constexpr int LS = CH * 8 * W + 4;                    // line stride in doubles ((LS mod 32) = 4: the four lines spread over the banks)
  // W = 16 and N odd: the lane whose 16 bytes would straddle the end of the block reads the LAST 16 bytes instead, so the last
  // element lands one slot late (and nothing past the block is ever read)
  static constexpr bool LATE = (W == 16) && ((N * N) & 1);
};
// chunk C of a line: the instruction's offset field moves the global AND the LDS address by 64 W C, so every chunk of a line uses
// the same uniform base (SGPR pair) and the same lane offset (one VGPR; a second one, clamped, for the chunk that crosses the end)
template <int N, int C>
__device__ __forceinline__ void q_dma_chunks(double* L, const double* __restrict__ g, unsigned lane_off, unsigned lane_off_last) {
  using IM = qimg<N>;
  if constexpr (C < IM::CH) {
    [[maybe_unused]] constexpr bool last = 64 * IM::W * (C + 1) > IM::BYTES;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass instantiates the kernel template's body as well and has no such builtin)
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g) + (last ? lane_off_last : lane_off)),
        (__attribute__((address_space(3))) void*)L, IM::W, 64 * IM::W * C, 0);
#else
    (void)L;
#endif
    q_dma_chunks<N, C + 1>(L, g, lane_off, lane_off_last);
  }
}
// Before an image is overwritten: every LDS read issued so far has returned (lgkmcnt(0)), and neither the compiler nor the machine
// scheduler moves a read across this point.  (The builtin's memory operand names W bytes at the image's base -- a DMA instruction
// writes 64 W bytes -- so alias analysis sees no conflict with reads elsewhere in the image and WOULD reorder them: N = 16 returned
// the next operand in place of the one being read.)
__device__ __forceinline__ void q_dma_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
  asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ void q_dma_line(double* L, const double* __restrict__ g, int lane) {
  using IM = qimg<N>;
  q_dma_fence();
  const unsigned lo = (unsigned)(IM::W * lane);
  // the last chunk: lanes past the end of the block re-read its last W bytes (they land in the image's padding)
  const unsigned lo_last = min(lo, (unsigned)(IM::BYTES - IM::W - 64 * IM::W * (IM::CH - 1)));
  q_dma_chunks<N, 0>(L, g, lo, lo_last);
}
template <int N>
__device__ __forceinline__ void q_dma_quad(double* L, const double* __restrict__ g, const long long (&off)[4], int lane) {
#pragma unroll
  for (int b = 0; b < 4; ++b) q_dma_line<N>(L + b * qimg<N>::LS, g + off[b], lane);
}
// every DMA issued so far has landed in LDS
__device__ __forceinline__ void q_dma_wait() {
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  asm volatile("" ::: "memory");
}

// ---- operand fetchers of the streaming product ---------------------------------------------------------------------------------
template <int R>
struct fa_reg {   // left operand in registers
  const amat<R>& A;
  __device__ __forceinline__ double operator()(int i, int k) const { return A.v[i][k]; }
};
template <int R, int C>
struct fb_reg {   // right operand in registers
  const qmat<R, C>& B;
  __device__ __forceinline__ double operator()(int k, int j) const { return B.v[k][j]; }
};
template <int N>
struct fa_lds {   // left operand from a flat image: element [4 i + l][4 k + q]
  const double* a0;    // image (of the lane's line) + l + N q
  bool rok, cok;       // l < REM, q < REM
  int late;            // 1 on the lane that holds the block's last element (LATE images)
  __device__ __forceinline__ fa_lds(const double* img, const qpos& p)
      : a0(img + p.l + N * p.q), rok(p.l < qcfg<N>::REM), cok(p.q < qcfg<N>::REM),
        late((qimg<N>::LATE && p.l == qcfg<N>::REM - 1 && p.q == qcfg<N>::REM - 1) ? 1 : 0) {}
  __device__ __forceinline__ double operator()(int i, int k) const {
    constexpr int RB = qcfg<N>::RB, REM = qcfg<N>::REM;
    const bool ie = (i == RB - 1) && REM < 4, ke = (k == RB - 1) && REM < 4;
    const bool corner = (i == RB - 1) && (k == RB - 1) && qimg<N>::LATE;
    const double x = corner ? a0[4 * i + N * 4 * k + late] : a0[4 * i + N * 4 * k];
    return ((!ie || rok) && (!ke || cok)) ? x : 0.0;
  }
};
// right operand from a flat image: element [4 k + q][4 j + l]; NR rider columns (N: r1, N + 1: r2) from registers
template <int N, int NR>
struct fb_lds {
  const double* b0;    // image + q + N l
  bool rok, cok, mA, mB;
  int late;
  const cvec<qcfg<N>::RB>* r1;
  const cvec<qcfg<N>::RB>* r2;
  __device__ __forceinline__ fb_lds(const double* img, const qpos& p, const cvec<qcfg<N>::RB>* r1_ = nullptr,
                                    const cvec<qcfg<N>::RB>* r2_ = nullptr)
      : b0(img + p.q + N * p.l), rok(p.q < qcfg<N>::REM), cok(p.l < qcfg<N>::REM), mA(p.l == (qcfg<N>::cA & 3)),
        mB(p.l == (qcfg<N>::cB & 3)),
        late((qimg<N>::LATE && p.l == qcfg<N>::REM - 1 && p.q == qcfg<N>::REM - 1) ? 1 : 0), r1(r1_), r2(r2_) {}
  __device__ __forceinline__ double operator()(int k, int j) const {
    constexpr int RB = qcfg<N>::RB, REM = qcfg<N>::REM, cA = qcfg<N>::cA, cB = qcfg<N>::cB;
    double x = 0.0;
    if (j < RB) {
      const bool re = (k == RB - 1) && REM < 4, ce = (j == RB - 1) && REM < 4;
      const bool corner = (k == RB - 1) && (j == RB - 1) && qimg<N>::LATE;
      const double y = corner ? b0[4 * k + N * 4 * j + late] : b0[4 * k + N * 4 * j];
      x = ((!re || rok) && (!ce || cok)) ? y : 0.0;
    }
    if (NR >= 1 && j == (cA >> 2)) x = mA ? r1->x[k] : x;
    if (NR >= 2 && j == (cB >> 2)) x = mB ? r2->x[k] : x;
    return x;
  }
};
// acc[:, 0:CU) += A B[:, 0:CU), the fragments of k-step k + 1 requested before the MFMAs of step k
template <int CU, int R, int C, typename FA, typename FB>
__device__ __forceinline__ void q_mm_f(qmat<R, C>& acc, const FA& fa, const FB& fb) {
  static_assert(CU <= C, "columns");
  double a[2][R], bb[2][CU];
#pragma unroll
  for (int i = 0; i < R; ++i) a[0][i] = fa(i, 0);
#pragma unroll
  for (int j = 0; j < CU; ++j) bb[0][j] = fb(0, j);
#pragma unroll
  for (int k = 0; k < R; ++k) {
    if (k + 1 < R) {
#pragma unroll
      for (int i = 0; i < R; ++i) a[(k + 1) & 1][i] = fa(i, k + 1);
#pragma unroll
      for (int j = 0; j < CU; ++j) bb[(k + 1) & 1][j] = fb(k + 1, j);
    }
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
      for (int j = 0; j < CU; ++j)
        acc.v[i][j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[k & 1][i], bb[k & 1][j], acc.v[i][j], 0, 0, 0);
  }
}
// a whole right operand from its image into registers
template <int N, int NR, int C>
__device__ __forceinline__ void q_read_b(qmat<qcfg<N>::RB, C>& m, const fb_lds<N, NR>& fb) {
#pragma unroll
  for (int k = 0; k < qcfg<N>::RB; ++k)
#pragma unroll
    for (int j = 0; j < C; ++j) m.v[k][j] = fb(k, j);
}

// ---- outputs through the (now idle) images: full-line stores -------------------------------------------------------------------
// A result in the accumulator lane map scatters as 16 pieces of 32 bytes per store instruction (36 instructions per matrix and
// quad, partial lines at the L2: the last doubling step of a layer, which writes four matrices, took 0.5 ms more than the others).
// Written flat into its line of an image first, a matrix leaves as 16 contiguous bytes per lane, 1 KB per instruction.
template <int N, int C, typename F>
__device__ __forceinline__ void q_img_put(double* img, const qmat<qcfg<N>::RB, C>& m, const qpos& p, F&& f) {
  constexpr int RB = qcfg<N>::RB;
  double* w0 = img + p.q + N * p.l;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < RB; ++j)
      if (4 * i + p.q < N && 4 * j + p.l < N) w0[4 * i + N * 4 * j] = f(m.v[i][j], i);
}
// sign masks of the D-mirror for the flat element pairs a lane copies: bit 2 c + h set <=> element 2 (64 c + lane) + h sits in a
// block position whose row and column differ in their U / V property
template <int N>
__device__ __forceinline__ unsigned q_flat_diff_mask(int ns, int lane) {
  constexpr int NN = N * N, CH = (NN + 127) / 128;
  unsigned m = 0u;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int e = min(2 * (64 * c + lane) + h, NN - 1), col = e / N, row = e - col * N;
      m |= ((is_uv_row(row, ns) != is_uv_row(col, ns)) ? 1u : 0u) << (2 * c + h);
    }
  return m;
}
// the lines of an image -> global blocks (and, MIRROR, their D-mirrors); `ok` = the lane group's line is stored at all
template <int N, bool MIRROR>
__device__ __forceinline__ void q_img_out(const double* img, double* __restrict__ g, double* __restrict__ gm, const long long (&off)[4],
                                          int nvalid, unsigned dmask, int lane) {
  constexpr int NN = N * N, CH = (NN + 127) / 128, LS = qimg<N>::LS;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (b >= nvalid) break;   // (wave-uniform)
    const double* src = img + b * LS;
    double* dst = g + off[b];
    double* dstm = MIRROR ? gm + off[b] : nullptr;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int e = 2 * (64 * c + lane);
      if (e + 1 < NN) {
        const double x0 = src[e], x1 = src[e + 1];
        dst[e] = x0;
        dst[e + 1] = x1;
        if (MIRROR) {
          dstm[e] = ((dmask >> (2 * c)) & 1u) ? -x0 : x0;
          dstm[e + 1] = ((dmask >> (2 * c + 1)) & 1u) ? -x1 : x1;
        }
      } else if (e < NN) {   // the odd tail
        const double x0 = src[e];
        dst[e] = x0;
        if (MIRROR) dstm[e] = ((dmask >> (2 * c)) & 1u) ? -x0 : x0;
      }
    }
  }
}

template <int N, bool LAST>
__global__ __launch_bounds__(64) void k_raman_doubling_quad(
    int S, int K, int NQ, const int* __restrict__ shift, const double* __restrict__ r, const double* __restrict__ t,
    const double* __restrict__ ttg, const double* __restrict__ gt, const double* __restrict__ gr,
    const double* __restrict__ grt, const double* __restrict__ jp, const double* __restrict__ j1m,
    const double* __restrict__ tmp1, const double* __restrict__ tmp2, const double* __restrict__ expk, double* ier,
    double* iet, double* ieJp, double* ieJm, int ns_arg, double* ier_pm, double* iet_mm) {
  using Q = qcfg<N>;
  using IM = qimg<N>;
  constexpr int RB = Q::RB, NB = Q::NB, NN = N * N, cA = Q::cA, cB = Q::cB;
  __shared__ __attribute__((aligned(16))) double QA[4 * IM::LS];   // quad images (alternating)
  __shared__ __attribute__((aligned(16))) double QB[4 * IM::LS];
  __shared__ __attribute__((aligned(16))) double S1[IM::LS];       // the recipient's r1, then ttg1
  const int ns = LAST ? ns_arg : 0;   // (compile-time zero in the steps before the last one: no D-mirror code at all)
  qpos p;
  p.lane = threadIdx.x;
  p.q = p.lane >> 4;
  p.b = (p.lane >> 2) & 3;
  p.l = p.lane & 3;
  p.tr4 = 4 * (16 * p.l + 4 * p.b + p.q);
  // Workgroup -> (recipient, quad rank).  Workgroup ids go round the eight XCDs (each with its own L2); XCD x sweeps a contiguous
  // eighth of the recipients, quad rank by quad rank: the waves in flight on an XCD are ~ 128 neighbouring recipients at the SAME
  // four line offsets, so a donor's blocks are fetched once into that L2 and hit by the other three lines that use them
  // (recipient-major order: every block of every line came from HBM, 6.4 GB per launch at 5 TB/s).
  const int per = (S + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qi = idx / per, n1 = xcd * per + (idx - qi * per);
  if (n1 >= S) return;

  // ---- the four lines of this wave: in-band lines of rank 4 qi .. 4 qi + 3 of the recipient (K <= 128) -----------------------
  int dsel[4] = {-1, -1, -1, -1};
  int cnt = 0;
  {
    int base = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = 64 * h + p.lane;
      const int sh = (i < K) ? shift[i] : 0;
      const int n0 = n1 + sh;
      const bool inb = i < K && n0 >= 0 && n0 < S;
      const unsigned long long bal = __ballot(inb);
      const int rank = base + __popcll(bal & ((1ull << p.lane) - 1ull));
      base += __popcll(bal);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const unsigned long long mb = __ballot(inb && rank == 4 * qi + bb);
        if (mb) dsel[bb] = 64 * h + __ffsll((long long)mb) - 1;
      }
    }
    cnt = base;
  }
  if (LAST && ns > 0 && qi == 0) {   // out-of-band lines keep zero D-mirrors (the operator-level apply_D! writes every block)
    for (int dz = 0; dz < K; ++dz) {
      const int n0 = n1 + shift[dz];
      if (n0 >= 0 && n0 < S) continue;
      const long long oz = ((long long)n1 + (long long)S * dz) * NN;
      for (int e = p.lane; e < NN; e += 64) {
        ier_pm[oz + e] = 0.0;
        iet_mm[oz + e] = 0.0;
      }
    }
  }
  if (4 * qi >= cnt) return;
  // wave-uniform block offsets of the four lines (a partial last quad repeats its first line; nothing of it is stored)
  long long o4s[4], e4s[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int db = __builtin_amdgcn_readfirstlane((4 * qi + b < cnt) ? dsel[b] : dsel[0]);
    const int n0b = n1 + __builtin_amdgcn_readfirstlane(shift[db]);
    o4s[b] = ((long long)n1 + (long long)S * db) * NN;
    e4s[b] = (long long)n0b * NN;
  }
  const bool valid = 4 * qi + p.b < cnt;
  const long long o4 = (p.b == 0) ? o4s[0] : ((p.b == 1) ? o4s[1] : ((p.b == 2) ? o4s[2] : o4s[3]));
  const long long e4 = (p.b == 0) ? e4s[0] : ((p.b == 1) ? e4s[1] : ((p.b == 2) ? e4s[2] : e4s[3]));
  const long long o4v = o4 / N, e1 = e4 / N, s4 = (long long)n1 * NN;
  const double* qa = QA + p.b * IM::LS;   // the lane's line in the quad images
  const double* qb = QB + p.b * IM::LS;

  // Schedule: a product reads its operands from an image (or from registers), then waits for the DMAs issued BEFORE it (they had
  // the whole product to land) and requests the image a later product needs into the buffer it has just freed.
  // ---- prologue: ier -> QA, r0 -> QB, r1 -> S1; the vectors of the lines
  q_dma_quad<N>(QA, ier, o4s, p.lane);
  q_dma_quad<N>(QB, r, e4s, p.lane);
  q_dma_line<N>(&S1[0], r + s4, p.lane);
  const double e0 = expk[e1 / N];
  // (the vectors of a line are loaded where they are used: six broadcast loads each, L2 hits the second time -- held for the
  // whole kernel they were the first thing the register allocator spilled, and a scratch reload waits for every DMA in flight)
  qmat<RB, NB> X;
  {
    cvec<RB> v1, v2;
    q_load_v<N>(v1, j1m + e1, p);
    q_load_v<N>(v2, jp + e1, p);
    q_dma_wait();
    // ---- X = ier r0      (rider columns of r0: j1-[n0], j0+[n0]  ->  X[:, cA] = ier j1-, X[:, cB] = ier j0+)
    q_zero(X);
    q_mm_f<NB>(X, fa_lds<N>(qa, p), fb_lds<N, 2>(qb, p, &v1, &v2));
  }
  q_dma_wait();
  q_dma_quad<N>(QB, iet, o4s, p.lane);
  // ---- X += r1 ier ;  WA <- ier (rider columns zero): the accumulator of WA = ier + X gr0
  qmat<RB, NB> WA, W1;
  q_read_b<N, 0, NB>(WA, fb_lds<N, 0>(qa, p));
  q_mm_f<RB>(X, fa_lds<N>(S1, p), fb_reg<RB, NB>{WA});
  q_dma_wait();
  q_dma_quad<N>(QA, gr, e4s, p.lane);
  q_read_b<N, 0, NB>(W1, fb_lds<N, 0>(qb, p));   // W1 <- iet: the accumulator of W1 = iet + X gt0
  q_dma_quad<N>(QB, gt, e4s, p.lane);
  // The source recurrences accumulate in the rider column cA of the matrices that live anyway (no vector outlives its phase):
  //   a3 = iej0+ + ier j1- + X tmp1 + r1 iej1-   in W1[:, cA]  (the first two now, X tmp1 by the product, r1 iej1- from W3)
  //   a4 = iej1- + ier j0+ + X tmp2 + r1 iej0+   in WA[:, cA]  (the last term is added when it goes to W3[:, cA])
  {
    cvec<RB> cJp, cJm;
    q_load_v<N>(cJp, ieJp + o4v, p);
    q_load_v<N>(cJm, ieJm + o4v, p);
    const cvec<RB> xA = q_col<cA>(X), xB = q_bcast<cB>(q_col<cB>(X), p);
    cvec<RB> u3, u4;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      u3.x[i] = cJp.x[i] + xA.x[i];
      u4.x[i] = cJm.x[i] * e0 + xB.x[i];
    }
    q_set_col<cA>(W1, u3, p);
    q_set_col<cA>(WA, u4, p);
  }
  {
    amat<RB> XA;
    q_transpose(XA, X, p);
    // ---- WA = ier + X gr0 ; WA[:, cA] += X tmp2
    cvec<RB> vt;
    q_load_v<N>(vt, tmp2 + e1, p);
    q_dma_wait();
    q_mm_f<NB>(WA, fa_reg<RB>{XA}, fb_lds<N, 1>(qa, p, &vt));
    q_load_v<N>(vt, tmp1 + e1, p);
    q_dma_wait();
    q_dma_quad<N>(QA, iet, o4s, p.lane);
    // ---- W1 = iet + X gt0 ; W1[:, cA] += X tmp1
    q_mm_f<NB>(W1, fa_reg<RB>{XA}, fb_lds<N, 1>(qb, p, &vt));
  }
  q_dma_wait();
  q_dma_quad<N>(QB, t, e4s, p.lane);
  // ---- R1IET = r1 [iet | iej1- | iej0+]
  qmat<RB, NB> W3;
  {
    cvec<RB> cJp, j1;
    q_load_v<N>(cJp, ieJp + o4v, p);
    q_load_v<N>(j1, ieJm + o4v, p);
#pragma unroll
    for (int i = 0; i < RB; ++i) j1.x[i] *= e0;
    q_zero(W3);
    q_mm_f<NB>(W3, fa_lds<N>(S1, p), fb_lds<N, 2>(qa, p, &j1, &cJp));
  }
  q_dma_wait();
  q_dma_line<N>(&S1[0], ttg + s4, p.lane);
  q_dma_quad<N>(QA, ier, o4s, p.lane);
  cvec<RB> a4;
  {
    const cvec<RB> q1 = q_col<cA>(W3), q3 = q_col<cA>(W1);
    cvec<RB> a3;
#pragma unroll
    for (int i = 0; i < RB; ++i) a3.x[i] = q3.x[i] + q1.x[i];
    q_set_col<cA>(W1, a3, p);
    const cvec<RB> sv = q_bcast<cB>(q_col<cB>(W3), p), q4 = q_col<cA>(WA);
#pragma unroll
    for (int i = 0; i < RB; ++i) a4.x[i] = q4.x[i] + sv.x[i];
  }
  // ---- W3 = WA t0 + r1 iet, column cA = a4
  {
    amat<RB> WAA;
    q_transpose(WAA, WA, p);
    q_mm_f<RB>(W3, fa_reg<RB>{WAA}, fb_lds<N, 0>(qb, p));
    q_set_col<cA>(W3, a4, p);
  }
  q_dma_wait();
  q_dma_quad<N>(QB, iet, o4s, p.lane);
  // ---- iet' = ttg1 W1 + iet gt0 ;  ier' = ier + iet grt0 + ttg1 W3   (columns cA: ttg1 a3 + iet tmp1, ttg1 a4 + iet tmp2)
  qmat<RB, NB> O1, O2;
  q_read_b<N, 0, NB>(O2, fb_lds<N, 0>(qa, p));
  q_dma_quad<N>(QA, grt, e4s, p.lane);
  q_zero(O1);
  q_mm_f<NB>(O1, fa_lds<N>(S1, p), fb_reg<RB, NB>{W1});
  q_dma_wait();
  qmat<RB, NB> grtB;
  {
    cvec<RB> vt;
    q_load_v<N>(vt, tmp2 + e1, p);
    q_read_b<N, 1, NB>(grtB, fb_lds<N, 1>(qa, p, &vt));
  }
  q_dma_quad<N>(QA, gt, e4s, p.lane);
  q_mm_f<NB>(O2, fa_lds<N>(S1, p), fb_reg<RB, NB>{W3});
  cvec<RB> vt1;
  q_load_v<N>(vt1, tmp1 + e1, p);
  q_dma_wait();
  q_mm_f<NB>(O1, fa_lds<N>(qb, p), fb_lds<N, 1>(qa, p, &vt1));
  q_mm_f<NB>(O2, fa_lds<N>(qb, p), fb_reg<RB, NB>{grtB});
  // ---- outputs: ieJ0+' = iej0+ expk0 + O1[:, cA] ;  ieJ0-' = iej0- + O2[:, cA] ;  apply_D! on the way out of the last step
  // (doubling_inelastic.jl:166-195: ier' rows flipped for U / V, D-mirrors ier+-, iet--)
  cvec<RB> cJp, cJm;
  q_load_v<N>(cJp, ieJp + o4v, p);
  q_load_v<N>(cJm, ieJm + o4v, p);
  if (valid && p.l == (cA & 3)) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = 4 * i + p.q;
      if (row < N) {
        ieJp[o4v + row] = O1.v[i][cA >> 2] + cJp.x[i] * e0;
        const double x = O2.v[i][cA >> 2] + cJm.x[i];
        ieJm[o4v + row] = (ns > 0 && is_uv_row(row, ns)) ? -x : x;
      }
    }
  }
  // iet' -> QA, ier' (rows flipped for U / V on the way out of the last step) -> QB, then flat to memory
  q_dma_fence();   // (every read of the images has returned)
  q_img_put<N>(QA + p.b * IM::LS, O1, p, [](double x, int) { return x; });
  unsigned uvrow = 0u;   // bit i: row 4 i + q is a U / V row
  if (ns > 0) {
#pragma unroll
    for (int i = 0; i < RB; ++i) uvrow |= (is_uv_row(4 * i + p.q, ns) ? 1u : 0u) << i;
  }
  q_img_put<N>(QB + p.b * IM::LS, O2, p, [uvrow](double x, int i) { return ((uvrow >> i) & 1u) ? -x : x; });
  const int nvalid = min(cnt - 4 * qi, 4);
  const unsigned dmask = LAST ? q_flat_diff_mask<N>(ns_arg > 0 ? ns_arg : 1, p.lane) : 0u;
  q_img_out<N, LAST>(QA, iet, iet_mm, o4s, nvalid, dmask, p.lane);
  q_img_out<N, LAST>(QB, ier, ier_pm, o4s, nvalid, dmask, p.lane);
}

// ---- interaction pass (interaction_inelastic.jl:319-521, one of its two `for dn` loops; the operand bindings of rs_ia_pass) -------
//     W1 = L1 E0 + L2 I1      W3 = L1 E3 + L2 I3      Y = YA + TI W1      A = ACCA + TI W3 + Y GX      B = TI I4 + Y GY
// (+ the source recurrence in the rider column N: V1 = L1 VE0 + L2 VI1 + VADD, VOUT = VACC + TI V1 + Y VV).  Four lines of a recipient
// per wave as in the doubling step.  Left operands that come from memory (L1, L2, TI) are gathered straight into registers in the
// left-operand lane map (32 contiguous bytes per four lanes; L2, TI the same block for the four lines), right operands and the
// accumulator seeds are staged by LDS-DMA into two alternating quad images one product ahead, Y is transposed in registers.
template <int N>
__global__ __launch_bounds__(64) void k_raman_interaction_quad(int S, int K, const int* __restrict__ shift, const rs_ia_pass<double> h) {
  using Q = qcfg<N>;
  using IM = qimg<N>;
  constexpr int RB = Q::RB, NB = Q::NB, NN = N * N, cA = Q::cA;
  __shared__ __attribute__((aligned(16))) double QA[4 * IM::LS];
  __shared__ __attribute__((aligned(16))) double QB[4 * IM::LS];
  qpos p;
  p.lane = threadIdx.x;
  p.q = p.lane >> 4;
  p.b = (p.lane >> 2) & 3;
  p.l = p.lane & 3;
  p.tr4 = 4 * (16 * p.l + 4 * p.b + p.q);
  // workgroup -> (recipient, quad rank): XCD-aware, quad rank by quad rank (see k_raman_doubling_quad)
  const int per = (S + 7) >> 3;
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int qi = idx / per, n1 = xcd * per + (idx - qi * per);
  if (n1 >= S) return;
  int dsel[4] = {-1, -1, -1, -1};
  int cnt = 0;
  {
    int base = 0;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int i = 64 * hh + p.lane;
      const int sh = (i < K) ? shift[i] : 0;
      const int n0 = n1 + sh;
      const bool inb = i < K && n0 >= 0 && n0 < S;
      const unsigned long long bal = __ballot(inb);
      const int rank = base + __popcll(bal & ((1ull << p.lane) - 1ull));
      base += __popcll(bal);
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const unsigned long long mb = __ballot(inb && rank == 4 * qi + bb);
        if (mb) dsel[bb] = 64 * hh + __ffsll((long long)mb) - 1;
      }
    }
    cnt = base;
  }
  if (4 * qi >= cnt) return;
  long long o4s[4], e0s[4], e3s[4], g4s[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int db = __builtin_amdgcn_readfirstlane((4 * qi + b < cnt) ? dsel[b] : dsel[0]);
    const int n0b = n1 + __builtin_amdgcn_readfirstlane(shift[db]);
    o4s[b] = ((long long)n1 + (long long)S * db) * NN;
    e0s[b] = (long long)n0b * h.sE0;
    e3s[b] = (long long)n0b * h.sE3;
    g4s[b] = (long long)n0b * NN;
  }
  const bool valid = 4 * qi + p.b < cnt;
  const long long o4 = (p.b == 0) ? o4s[0] : ((p.b == 1) ? o4s[1] : ((p.b == 2) ? o4s[2] : o4s[3]));
  const long long g4 = (p.b == 0) ? g4s[0] : ((p.b == 1) ? g4s[1] : ((p.b == 2) ? g4s[2] : g4s[3]));
  const long long o4v = o4 / N, e1 = g4 / N;
  const double* qa = QA + p.b * IM::LS;
  const double* qb = QB + p.b * IM::LS;

  // ---- prologue: E0 -> QA, I1 -> QB ;  L1, L2 in registers
  q_dma_quad<N>(QA, h.E0, e0s, p.lane);
  q_dma_quad<N>(QB, h.I1, o4s, p.lane);
  qmat<RB, NB> W1, W3;
  {
    amat<RB> L1A, L2A;
    q_load_a<N>(L1A, h.L1 + o4, p);
    q_load_a<N>(L2A, h.L2 + (long long)n1 * h.sL2, p);
    {
      cvec<RB> v;
      q_load_v<N>(v, h.VE0 + e1, p);
      q_dma_wait();
      // ---- W1 = L1 E0 (rider: L1 VE0[n0])
      q_zero(W1);
      q_mm_f<NB>(W1, fa_reg<RB>{L1A}, fb_lds<N, 1>(qa, p, &v));
    }
    q_dma_wait();
    q_dma_quad<N>(QA, h.E3, e3s, p.lane);
    {
      cvec<RB> v;
      q_load_v<N>(v, h.VI1 + o4v, p);
      // ---- W1 += L2 I1 (rider: L2 VI1)
      q_mm_f<NB>(W1, fa_reg<RB>{L2A}, fb_lds<N, 1>(qb, p, &v));
    }
    q_dma_wait();
    q_dma_quad<N>(QB, h.I3, o4s, p.lane);
    // ---- W3 = L1 E3 + L2 I3
    q_zero(W3);
    q_mm_f<RB>(W3, fa_reg<RB>{L1A}, fb_lds<N, 0>(qa, p));
    q_dma_wait();
    q_dma_quad<N>(QA, h.YA, o4s, p.lane);
    q_mm_f<RB>(W3, fa_reg<RB>{L2A}, fb_lds<N, 0>(qb, p));
  }
  q_dma_wait();
  q_dma_quad<N>(QB, h.ACCA, o4s, p.lane);
  {   // column N of W1 becomes V1 = L1 VE0 + L2 VI1 + VADD
    cvec<RB> v;
    q_load_v<N>(v, h.VADD + o4v, p);
    const cvec<RB> q1 = q_col<cA>(W1);
#pragma unroll
    for (int i = 0; i < RB; ++i) v.x[i] += q1.x[i];
    q_set_col<cA>(W1, v, p);
  }
  qmat<RB, NB> Y, A, B;
  cvec<RB> vt;
  {
    amat<RB> TIA;
    q_load_a<N>(TIA, h.TI + (long long)n1 * NN, p);
    // ---- Y = YA + TI W1 (column N: TI V1)
    q_read_b<N, 0, NB>(Y, fb_lds<N, 0>(qa, p));
    q_dma_quad<N>(QA, h.I4, o4s, p.lane);
    q_mm_f<NB>(Y, fa_reg<RB>{TIA}, fb_reg<RB, NB>{W1});
    q_dma_wait();
    // ---- A = ACCA + TI W3
    q_read_b<N, 0, NB>(A, fb_lds<N, 0>(qb, p));
    q_dma_quad<N>(QB, h.GX, g4s, p.lane);
    q_mm_f<RB>(A, fa_reg<RB>{TIA}, fb_reg<RB, NB>{W3});
    q_dma_wait();
    // ---- B = TI I4
    q_zero(B);
    q_mm_f<RB>(B, fa_reg<RB>{TIA}, fb_lds<N, 0>(qa, p));
  }
  q_dma_wait();
  q_dma_quad<N>(QA, h.GY, g4s, p.lane);
  vt = q_col<cA>(Y);
  {
    amat<RB> YT;
    q_transpose(YT, Y, p);   // (its column N = TI V1 meets the zero rows >= N of GX, GY)
    cvec<RB> v;
    q_load_v<N>(v, h.VV + e1, p);
    // ---- A += Y GX (rider: Y VV[n0]) ;  B += Y GY
    q_mm_f<NB>(A, fa_reg<RB>{YT}, fb_lds<N, 1>(qb, p, &v));
    q_dma_wait();
    q_mm_f<RB>(B, fa_reg<RB>{YT}, fb_lds<N, 0>(qa, p));
  }
  cvec<RB> vac;
  q_load_v<N>(vac, h.VACC + o4v, p);
  if (valid && p.l == (cA & 3)) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int row = 4 * i + p.q;
      if (row < N) h.VOUT[o4v + row] = A.v[i][cA >> 2] + vac.x[i] + vt.x[i];
    }
  }
  // A -> QA, B -> QB, then flat to memory (full-line stores: q_img_out)
  q_dma_fence();
  q_img_put<N>(QA + p.b * IM::LS, A, p, [](double x, int) { return x; });
  q_img_put<N>(QB + p.b * IM::LS, B, p, [](double x, int) { return x; });
  const int nvalid = min(cnt - 4 * qi, 4);
  q_img_out<N, false>(QA, h.OUTA, nullptr, o4s, nvalid, 0u, p.lane);
  q_img_out<N, false>(QB, h.OUTB, nullptr, o4s, nvalid, 0u, p.lane);
}

#ifndef RQ_N_LO
#define RQ_N_LO 3
#endif
#ifndef RQ_N_HI
#define RQ_N_HI 22
#endif

template <int N>
int launch_rq(int S, int K, const int* shift, const double* r, const double* t, const double* ttg, const double* gt,
              const double* gr, const double* grt, const double* jp, const double* j1m, const double* tmp1,
              const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp, double* ieJm, int ns,
              double* ier_pm, double* iet_mm, hipStream_t st) {
  const int NQ = (K + 3) / 4;
  const long long blocks = 8LL * ((S + 7) / 8) * NQ;
  if (blocks > 0x7fffffffLL) return VSM_ERR_UNSUPPORTED;
  if (ns > 0)
    hipLaunchKernelGGL((k_raman_doubling_quad<N, true>), dim3((unsigned)blocks), dim3(64), 0, st, S, K, NQ, shift, r, t, ttg, gt, gr,
                       grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm, ns, ier_pm, iet_mm);
  else
    hipLaunchKernelGGL((k_raman_doubling_quad<N, false>), dim3((unsigned)blocks), dim3(64), 0, st, S, K, NQ, shift, r, t, ttg, gt,
                       gr, grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm, ns, ier_pm, iet_mm);
  VSM_LAUNCH_CHECK("k_raman_doubling_quad");
  return VSM_OK;
}
template <int N, typename F>
int dispatch_rq(int n, F f) {
  if constexpr (N > RQ_N_HI) {
    return VSM_ERR_UNSUPPORTED;
  } else {
    if (n == N) return f(std::integral_constant<int, N>{});
    return dispatch_rq<N + 1>(n, f);
  }
}

}  // namespace

// FP64, RQ_N_LO <= N <= 22 (six column blocks incl. the riders), K <= 128; VSM_ERR_UNSUPPORTED otherwise (the caller goes on to
// raman_doubling_wave).  ns > 0 (n_stokes) marks the LAST doubling step of a layer: apply_D! of the inelastic operators happens on
// the way out (ier_pm, iet_mm are written).  The wave reads ier / iet of its own lines only and writes them at the end: in place.
int raman_doubling_quad(int N, int S, int K, const int* shift, const double* r, const double* t, const double* ttg,
                        const double* gt, const double* gr, const double* grt, const double* jp, const double* j1m,
                        const double* tmp1, const double* tmp2, const double* expk, double* ier, double* iet, double* ieJp,
                        double* ieJm, int ns, double* ier_pm, double* iet_mm, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_RAMAN_QUAD");
  if (off || N < RQ_N_LO || N > RQ_N_HI || K > 128) return VSM_ERR_UNSUPPORTED;
  if (S <= 0 || K <= 0) return VSM_OK;
  return dispatch_rq<RQ_N_LO>(N, [&](auto tag) {
    return launch_rq<decltype(tag)::value>(S, K, shift, r, t, ttg, gt, gr, grt, jp, j1m, tmp1, tmp2, expk, ier, iet, ieJp, ieJm, ns,
                                           ier_pm, iet_mm, st);
  });
}


// One pass of the inelastic interaction, FP64, RQ_N_LO <= N <= 22, K <= 128; VSM_ERR_UNSUPPORTED otherwise (the caller goes on to
// raman_interaction_wave).  The outputs alias operands of the same line only; a wave reads all of its lines' blocks before it writes.
int raman_interaction_quad(int N, int S, int K, const int* shift, const rs_ia_pass<double>& h, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_RAMAN_QUAD");
  if (off || N < RQ_N_LO || N > RQ_N_HI || K > 128) return VSM_ERR_UNSUPPORTED;
  if (S <= 0 || K <= 0) return VSM_OK;
  const long long blocks = 8LL * ((S + 7) / 8) * ((K + 3) / 4);
  if (blocks > 0x7fffffffLL) return VSM_ERR_UNSUPPORTED;
  return dispatch_rq<RQ_N_LO>(N, [&](auto tag) {
    hipLaunchKernelGGL((k_raman_interaction_quad<decltype(tag)::value>), dim3((unsigned)blocks), dim3(64), 0, st, S, K, shift, h);
    VSM_LAUNCH_CHECK("k_raman_interaction_quad");
    return (int)VSM_OK;
  });
}

}  // namespace vsm
