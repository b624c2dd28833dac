// Operator-level ("generic") kernels: any N up to 128, matrices streamed from HBM/L2.
// These are the drop-in replacements for the reference's batched operators and
// KernelAbstractions kernels one-for-one; the fused LDS-resident kernels in
// vsm_fused.hip are the fast path for N that fits on-chip.
#include "vsm_internal.h"
#include "vsm_inverse.h"
#include "vsm_gemm_lds.h"

namespace vsm {

// ---------------------------------------------------------------------------
// C = alpha * A*B + beta * D + gamma * I      (batched, column-major)
// Replaces CUBLAS.gemm_strided_batched (ext/gpu_batched_cuda.jl:208-233) plus the
// broadcast that the reference launches around it (e.g. `I_static .- r ⊠ r`).
// One wave per 16x16 output tile; operands read straight from global (L2-resident).
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_gemm(int M, int Nc, int K, const T* __restrict__ A, long long sa,
                                              const T* __restrict__ B, long long sb, T* C, long long sc,
                                              T alpha, const T* D, long long sd, T beta, T gamma,
                                              long long pa, long long pb, long long pc, long long pd) {
  // second batch level (blockIdx.z = parameter index of the linearized pass); stride 0 = shared
  A += (long long)blockIdx.z * pa;
  B += (long long)blockIdx.z * pb;
  C += (long long)blockIdx.z * pc;
  if (D) D += (long long)blockIdx.z * pd;
  const int s = blockIdx.x;   // (spectral axis on gridDim.x: no 65535 limit)
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tilesM = (M + 15) >> 4, tilesN = (Nc + 15) >> 4;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= tilesM * tilesN) return;  // wave-uniform
  const int ti = tile % tilesM, tj = tile / tilesM;
  const T* As = A + (long long)s * sa;
  const T* Bs = B + (long long)s * sb;
  const int ai = ti * 16 + (lane & 15);
  const int bj = tj * 16 + (lane & 15);
  const int kq = lane >> 4;
  typename mfma<T>::acc_t acc = acc_zero<T>();
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + kq;
    const T a = (ai < M && k < K) ? As[ai + (long long)M * k] : T(0);
    const T b = (bj < Nc && k < K) ? Bs[k + (long long)K * bj] : T(0);
    acc = mfma<T>::mma(a, b, acc);
  }
  T* Cs = C + (long long)s * sc;
  const T* Ds = D ? D + (long long)s * sd : nullptr;
  const int col = tj * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ti * 16 + mfma<T>::crow(lane, r);
    if (row < M && col < Nc) {
      T v = alpha * acc[r];
      if (Ds) v += beta * Ds[row + (long long)M * col];
      if (row == col) v += gamma;
      Cs[row + (long long)M * col] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// The same product for 16 < M <= 128, 16 <= Nc <= 128 with every operand crossing the memory system ONCE (k_gemm's
// tile-waves re-read A rows / B columns tilesN / tilesM times, uncoalesced: 1.1 TB/s in FP32, 2.4 TB/s in FP64 at N = 96).
// One workgroup of NW = 4 | 6 | 8 waves (one per 16-column tile of C) per (spectral point, parameter).  The contraction runs in
// chunks of KC = 16:
//   * A chunk [M x 16]: coalesced global loads of all threads -> registers (issued one chunk ahead) -> LDS, column-major
//     with leading dimension 16 MT + 4, which makes the A-fragment reads conflict-free in both precisions;
//   * B never touches LDS: wave w loads, per chunk, four CONSECUTIVE k of its column straight into the MFMA B-operand
//     registers (lane (j, kq) takes k = kc + 4 kq + t, t = 0..3 -- the contraction index is permuted consistently in the
//     A fragments), also one chunk ahead;
//   * C tiles (MT row tiles per wave) stay in accumulators; epilogue alpha/D/beta/gamma as k_gemm.
// C may alias A, B or D (every load of an operand a wave's stores could touch has completed before its first store).
// ---------------------------------------------------------------------------
template <typename T, int MT, int NW, int KC>
__global__ __launch_bounds__(64 * NW) void k_gemm_lds(int M, int Nc, int K, const T* __restrict__ A, long long sa, long long pa,
                                                      const T* __restrict__ B, long long sb, long long pb, T* C, long long sc,
                                                      long long pc, T alpha, const T* D, long long sd, long long pd, T beta,
                                                      T gamma) {
  __shared__ __attribute__((aligned(16))) T As_lds[gemm_lds_cfg<MT, KC>::LDS_ELEMS];
  const long long s = blockIdx.x, pp = blockIdx.y;
  gemm_lds_body<T, MT, NW, KC>(M, Nc, K, A + s * sa + pp * pa, B + s * sb + pp * pb, C + s * sc + pp * pc,
                               D ? D + s * sd + pp * pd : nullptr, alpha, beta, gamma, As_lds);
}
// returns false when the shape is left to k_gemm (mat-vecs, M <= 16, anything past 128)
template <typename T>
static bool gemm_lds(int M, int Nc, int K, int S, int P, const T* A, long long sa, long long pa, const T* B, long long sb,
                     long long pb, T* C, long long sc, long long pc, T alpha, const T* D, long long sd, long long pd, T beta,
                     T gamma, hipStream_t st) {
  static const bool off = ab_switch("VSM_NO_GEMM_LDS");
  if (off || M <= 16 || M > 128 || Nc < 16 || Nc > 128 || K < 8 || P > 65535) return false;
  const dim3 grid(S, P);
  // chunks of 16 contraction indices: 32 measured 8..15 % slower in both precisions (fewer barriers, but half the chunks in flight)
#define VSM_GL(MT_, NW_)                                                                                              \
  hipLaunchKernelGGL((k_gemm_lds<T, MT_, NW_, 16>), grid, dim3(64 * NW_), 0, st, M, Nc, K, A, sa, pa, B, sb, pb, C, sc, pc, alpha, \
                     D, sd, pd, beta, gamma)
  VSM_GEMM_LDS_DISPATCH(M, Nc, VSM_GL);
#undef VSM_GL
  return true;
}

template <typename T>
int gemm(int M, int Nc, int K, int S, const T* A, long long sa, const T* B, long long sb, T* C, long long sc,
         T alpha, const T* D, long long sd, T beta, T gamma, hipStream_t st) {
  if (S <= 0 || M <= 0 || Nc <= 0) return VSM_OK;
  if constexpr (sizeof(T) == 8) {
    const int rc = strip_gemm(M, Nc, K, S, 1, A, sa, 0, B, sb, 0, C, sc, 0, alpha, D, sd, 0, beta, gamma, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  if (gemm_lds<T>(M, Nc, K, S, 1, A, sa, 0, B, sb, 0, C, sc, 0, alpha, D, sd, 0, beta, gamma, st)) {
    VSM_LAUNCH_CHECK("k_gemm_lds");
    return VSM_OK;
  }
  const int tiles = ((M + 15) / 16) * ((Nc + 15) / 16);
  dim3 grid(S, (tiles + 3) / 4);
  hipLaunchKernelGGL(k_gemm<T>, grid, dim3(256), 0, st, M, Nc, K, A, sa, B, sb, C, sc, alpha, D, sd, beta, gamma, 0LL,
                     0LL, 0LL, 0LL);
  VSM_LAUNCH_CHECK("k_gemm");
  return VSM_OK;
}
// two-level batch: S spectral points x P parameters
template <typename T>
int gemm2(int M, int Nc, int K, int S, int P, const T* A, long long sa, long long pa, const T* B, long long sb,
          long long pb, T* C, long long sc, long long pc, T alpha, const T* D, long long sd, long long pd, T beta,
          T gamma, hipStream_t st) {
  if (S <= 0 || P <= 0 || M <= 0 || Nc <= 0) return VSM_OK;
  if constexpr (sizeof(T) == 8) {
    const int rc = strip_gemm(M, Nc, K, S, P, A, sa, pa, B, sb, pb, C, sc, pc, alpha, D, sd, pd, beta, gamma, st);
    if (rc != VSM_ERR_UNSUPPORTED) return rc;
  }
  if (gemm_lds<T>(M, Nc, K, S, P, A, sa, pa, B, sb, pb, C, sc, pc, alpha, D, sd, pd, beta, gamma, st)) {
    VSM_LAUNCH_CHECK("k_gemm_lds(P)");
    return VSM_OK;
  }
  const int tiles = ((M + 15) / 16) * ((Nc + 15) / 16);
  dim3 grid(S, (tiles + 3) / 4, P);
  hipLaunchKernelGGL(k_gemm<T>, grid, dim3(256), 0, st, M, Nc, K, A, sa, B, sb, C, sc, alpha, D, sd, beta, gamma, pa,
                     pb, pc, pd);
  VSM_LAUNCH_CHECK("k_gemm(P)");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// Z of a layer with several scatterers: Z[:,:,s] = sum_k f[k,s] Z_k   (`+` of CoreScatteringOpticalProperties,
// types.jl:1262-1292, with f_k = tau_k varpi_k / sum_j tau_j varpi_j).  Materialises Z for the kernels that do not
// mix on the fly.
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_mix_Z(long long NN, int ncomp, const T* __restrict__ Zc_pp, const T* __restrict__ Zc_mp,
                        const T* __restrict__ fcomp, T* Zpp, T* Zmp) {
  const long long e = (long long)blockIdx.y * 256 + threadIdx.x;
  if (e >= NN) return;
  const long long s = blockIdx.x;
  T ap = 0, am = 0;
  for (int k = 0; k < ncomp; ++k) {
    const T f = fcomp[s * ncomp + k];
    ap += f * Zc_pp[k * NN + e];
    am += f * Zc_mp[k * NN + e];
  }
  Zpp[s * NN + e] = ap;
  Zmp[s * NN + e] = am;
}
// The same for nm Fourier moments of ONE layer in one launch, the moments folded into the spectral axis: block (im S + s) of the
// output is the mix of moment im's component stack with point s's weights (the weights do not depend on the moment); ncomp = 0:
// the layer has a single scatterer -- block `single` of every stack, copied.  (A small batch walks the layers with its moments
// as one batch: this is the per-layer input of that walk, one launch instead of nm mixes and a concatenation.)
template <typename T>
struct mixm_args {
  const T* zp[VSM_MM_MAX];
  const T* zm[VSM_MM_MAX];
};
template <typename T>
__global__ void k_mix_Z_moments(long long NN, int S, int ncomp, int single, mixm_args<T> a, const T* __restrict__ fcomp, T* Zpp, T* Zmp) {
  const long long e = (long long)blockIdx.z * 256 + threadIdx.x;
  if (e >= NN) return;
  const long long s = blockIdx.x, im = blockIdx.y;
  const T* zp = a.zp[im];
  const T* zm = a.zm[im];
  T ap = 0, am = 0;
  if (ncomp == 0) {
    ap = zp[single * NN + e];
    am = zm[single * NN + e];
  } else {
    for (int k = 0; k < ncomp; ++k) {
      const T f = fcomp[s * ncomp + k];
      ap += f * zp[k * NN + e];
      am += f * zm[k * NN + e];
    }
  }
  Zpp[(im * S + s) * NN + e] = ap;
  Zmp[(im * S + s) * NN + e] = am;
}
template <typename T>
int mix_Z_moments(int N, int S, int ncomp, int nm, const T* const* Zpp_comp, const T* const* Zmp_comp, int single, const T* fcomp,
                  T* Zpp, T* Zmp, hipStream_t st) {
  if (S <= 0 || nm <= 0) return VSM_OK;
  if (nm > VSM_MM_MAX) {
    set_error("mix_Z_moments: %d moments per call (limit %d)", nm, VSM_MM_MAX);
    return VSM_ERR_INVALID_ARG;
  }
  mixm_args<T> a;
  for (int k = 0; k < VSM_MM_MAX; ++k) {
    a.zp[k] = Zpp_comp[k < nm ? k : 0];
    a.zm[k] = Zmp_comp[k < nm ? k : 0];
  }
  const long long NN = (long long)N * N;
  hipLaunchKernelGGL(k_mix_Z_moments<T>, dim3(S, nm, (unsigned)((NN + 255) / 256)), dim3(256), 0, st, NN, S, ncomp, single, a, fcomp, Zpp,
                     Zmp);
  VSM_LAUNCH_CHECK("k_mix_Z_moments");
  return VSM_OK;
}
template int mix_Z_moments<double>(int, int, int, int, const double* const*, const double* const*, int, const double*, double*, double*,
                                   hipStream_t);
template int mix_Z_moments<float>(int, int, int, int, const float* const*, const float* const*, int, const float*, float*, float*, hipStream_t);

template <typename T>
int mix_Z(int N, int S, int ncomp, const T* Zpp_comp, const T* Zmp_comp, const T* fcomp, T* Zpp, T* Zmp, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const long long NN = (long long)N * N;
  hipLaunchKernelGGL(k_mix_Z<T>, dim3(S, (unsigned)((NN + 255) / 256)), dim3(256), 0, st, NN, ncomp, Zpp_comp, Zmp_comp, fcomp,
                     Zpp, Zmp);
  VSM_LAUNCH_CHECK("k_mix_Z");
  return VSM_OK;
}
template int mix_Z<double>(int, int, int, const double*, const double*, const double*, double*, double*, hipStream_t);
template int mix_Z<float>(int, int, int, const float*, const float*, const float*, float*, float*, hipStream_t);

// ---------------------------------------------------------------------------
// batch_inv!
// ---------------------------------------------------------------------------
template <typename T, int NPAD, int NT>
__global__ __launch_bounds__(NT) void k_batch_inv(int N, const T* A, T* X, int* info) {
  using C = gj_cfg<NPAD, NT>;
  __shared__ gj_scratch<T, NPAD> sc;
  const int s = blockIdx.x;
  const T* As = A + (long long)s * N * N;
  T* Xs = X + (long long)s * N * N;
  const int tr = threadIdx.x % C::TR, tc = threadIdx.x / C::TR;
  T a[C::RB][C::CB];
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < C::CB; ++cb) {
      const int i = tr + C::TR * rb, j = tc * C::CB + cb;
      a[rb][cb] = (i < N && j < N) ? As[i + (long long)N * j] : (i == j ? T(1) : T(0));
    }
  gj_invert<T, NPAD, NT>(a, N, sc);
#pragma unroll
  for (int rb = 0; rb < C::RB; ++rb)
#pragma unroll
    for (int cb = 0; cb < C::CB; ++cb) {
      const int i = tr + C::TR * rb, j = tc * C::CB + cb;
      if (i < N && j < N) Xs[i + (long long)N * sc.dst[j]] = a[rb][cb];
    }
  if (info && threadIdx.x == 0) info[s] = sc.info;
}

// N > 128: the same in-place Gauss-Jordan with partial pivoting (same pivot rule), the matrix in global memory (X, L2-resident:
// one workgroup of 1024 threads per matrix), pivot row / column / the displaced row k published through LDS per step.  A is
// copied to X first (A is not clobbered).  A fallback for sizes past every on-chip kernel, not a fast path.
template <typename T>
__global__ __launch_bounds__(1024) void k_batch_inv_big(int N, const T* __restrict__ A, T* X, int* info) {
  extern __shared__ __attribute__((aligned(16))) unsigned char inv_smem[];
  T* col = reinterpret_cast<T*>(inv_smem);   // column k before the step
  T* rowP = col + N;                         // pivot row (row p) before the step
  T* rowK = rowP + N;                        // row k before the step (needed where the rows are exchanged)
  int* piv = reinterpret_cast<int*>(rowK + N);
  __shared__ T red_v[16];
  __shared__ int red_i[16];
  __shared__ int s_p, s_info;
  const long long NN = (long long)N * N;
  const T* As = A + (long long)blockIdx.x * NN;
  T* Xs = X + (long long)blockIdx.x * NN;
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  for (long long e = tid; e < NN; e += nt) Xs[e] = As[e];
  if (tid == 0) s_info = 0;
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    // pivot: largest |X[i,k]|, i >= k, first occurrence
    T best = T(-1);
    int bi = k;
    for (int i = k + tid; i < N; i += nt) {
      const T v = fabs(Xs[i + (long long)N * k]);
      if (v > best) {
        best = v;
        bi = i;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const T ov = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      if (ov > best || (ov == best && oi < bi)) {
        best = ov;
        bi = oi;
      }
    }
    if (lane == 0) {
      red_v[wave] = best;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      T b = red_v[0];
      int p = red_i[0];
      for (int w = 1; w < nw; ++w)
        if (red_v[w] > b || (red_v[w] == b && red_i[w] < p)) {
          b = red_v[w];
          p = red_i[w];
        }
      s_p = p;
      piv[k] = p;
    }
    __syncthreads();
    const int p = s_p;
    for (int i = tid; i < N; i += nt) {
      col[i] = Xs[i + (long long)N * k];
      rowP[i] = Xs[p + (long long)N * i];
      rowK[i] = Xs[k + (long long)N * i];
    }
    __syncthreads();
    const T pv = rowP[k];
    if (tid == 0 && pv == T(0) && s_info == 0) s_info = k + 1;
    const T d = T(1) / pv;
    // after the exchange row p carries the old row k (column-k entry col[k]); every other row keeps its own
    for (long long e = tid; e < NN; e += nt) {
      const int i = (int)(e % N), j = (int)(e / N);
      const T u = (j == k) ? d : rowP[j] * d;     // new row k
      T v;
      if (i == k) {
        v = u;
      } else {
        const T f = (i == p) ? col[k] : col[i];
        const T src = (i == p) ? rowK[j] : Xs[e];
        v = (j == k) ? -f * d : src - f * u;
      }
      Xs[e] = v;
    }
    __syncthreads();
  }
  // undo the row exchanges as column exchanges of the inverse, last first
  for (int k = N - 1; k >= 0; --k) {
    const int p = piv[k];
    if (p != k) {
      for (int i = tid; i < N; i += nt) {
        const T a = Xs[i + (long long)N * k], b = Xs[i + (long long)N * p];
        Xs[i + (long long)N * k] = b;
        Xs[i + (long long)N * p] = a;
      }
      __syncthreads();
    }
  }
  if (info && tid == 0) info[blockIdx.x] = s_info;
}

template <typename T>
int batch_inv(int N, int S, const T* A, T* X, int* info, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  if (N > 128) {
    const size_t bytes = 3 * (size_t)N * sizeof(T) + (size_t)N * sizeof(int);
    if (bytes > 60000) {
      set_error("batch_inv: N=%d is past the LDS budget of the global-memory Gauss-Jordan kernel", N);
      return VSM_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_batch_inv_big<T>, dim3(S), dim3(1024), bytes, st, N, A, X, info);
    VSM_LAUNCH_CHECK("k_batch_inv_big");
    return VSM_OK;
  }
  if (N <= 32)
    hipLaunchKernelGGL((k_batch_inv<T, 32, 256>), dim3(S), dim3(256), 0, st, N, A, X, info);
  else if (N <= 64)
    hipLaunchKernelGGL((k_batch_inv<T, 64, 256>), dim3(S), dim3(256), 0, st, N, A, X, info);
  else if (N <= 96)
    hipLaunchKernelGGL((k_batch_inv<T, 96, 256>), dim3(S), dim3(256), 0, st, N, A, X, info);
  else   // 512 threads: 32 elements per thread (with 256 the 64-element register block spilled: 5.2 ms per 2048 matrices)
    hipLaunchKernelGGL((k_batch_inv<T, 128, 512>), dim3(S), dim3(512), 0, st, N, A, X, info);
  VSM_LAUNCH_CHECK("k_batch_inv");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// elemental!  (get_elem_rt! + apply_D_elemental!, elemental.jl:289-334,403-422)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_elemental(int N, int n_stokes, int m, int ndoubl, const T* __restrict__ dtau,
                                                   const T* __restrict__ varpi, const T* __restrict__ Zpp,
                                                   const T* __restrict__ Zmp, long long zs, const T* __restrict__ mu,
                                                   const T* __restrict__ wt, T* r_mp, T* t_pp, T* r_pm, T* t_mm) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const int i = e % N, j = e / N;
  const T wct = (m == 0) ? wt[j] / T(2) : wt[j] / T(4);
  const T mi = mu[i], mj = mu[j];
  const T d = dtau[s], w = varpi[s];
  const long long zo = (long long)s * zs + e;
  T r, t;
  if (wct > num<T>::eps()) {
    r = w * Zmp[zo] * (mj / (mi + mj)) * wct * (-expm1(-d * ((T(1) / mi) + (T(1) / mj))));
    if (mi == mj) {
      if (i == j)
        t = exp(-d / mi) * (T(1) + w * Zpp[zo] * (d / mi) * wct);
      else
        t = exp(-d / mj) * (w * Zpp[zo] * (d / mi) * wct);
    } else {
      t = w * Zpp[zo] * (mj / (mi - mj)) * wct * expdiff_neg<T>(d / mi, d / mj);
    }
  } else {
    r = T(0);
    t = (i == j) ? exp(-d / mi) : T(0);
  }
  const long long o = (long long)s * N * N + e;
  if (ndoubl < 1) {
    const bool same = is_uv_row(i, n_stokes) == is_uv_row(j, n_stokes);
    r_mp[o] = r;
    t_pp[o] = t;
    r_pm[o] = same ? r : -r;
    t_mm[o] = same ? t : -t;
  } else {
    r_mp[o] = is_uv_row(i, n_stokes) ? -r : r;
    t_pp[o] = t;
  }
}

// get_elem_rt_SFI! (elemental.jl:348-392)
template <typename T>
__global__ __launch_bounds__(256) void k_elemental_sfi(int N, int n_stokes, int S, int m, int ndoubl, int i_mu0,
                                                       const T* __restrict__ dtau, const T* __restrict__ varpi,
                                                       const T* __restrict__ tau_sum, const T* __restrict__ F0,
                                                       const T* __restrict__ Zpp, const T* __restrict__ Zmp,
                                                       long long zs, const T* __restrict__ mu, T* j0_p, T* j0_m) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * S) return;
  const int i = e % N, s = e / N;
  const int i_start = n_stokes * i_mu0;
  const T wct02 = (m == 0) ? T(0.5) : T(0.25);
  T zp = 0, zm = 0;
  for (int q = 0; q < n_stokes; ++q) {
    const long long zo = (long long)s * zs + i + (long long)N * (i_start + q);
    const T f = F0[q + (long long)n_stokes * s];
    zp += Zpp[zo] * f;
    zm += Zmp[zo] * f;
  }
  const T d = dtau[s], w = varpi[s];
  const T mi = mu[i], ms = mu[i_start];
  T jp;
  if (i >= i_start && i < i_start + n_stokes)
    jp = wct02 * w * zp * (d / mi) * exp(-d / mi);
  else
    jp = wct02 * w * zp * (ms / (mi - ms)) * expdiff_neg<T>(d / mi, d / ms);
  T jm = wct02 * w * zm * (ms / (mi + ms)) * (-expm1(-d * ((T(1) / mi) + (T(1) / ms))));
  const T att = exp(-tau_sum[s] / ms);
  jp *= att;
  jm *= att;
  if (ndoubl >= 1 && is_uv_row(i, n_stokes)) jm = -jm;
  j0_p[e] = jp;
  j0_m[e] = jm;
}

template <typename T>
int elemental(const quad<T>& q, int S, int m, int ndoubl, const T* dtau, const T* varpi, const T* tau_sum,
              const T* F0, const T* Zpp, const T* Zmp, long long zs, const added<T>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int N = q.N;
  hipLaunchKernelGGL(k_elemental<T>, dim3(S, (N * N + 255) / 256), dim3(256), 0, st, N, q.n_stokes, m, ndoubl, dtau,
                     varpi, Zpp, Zmp, zs, q.mu, q.wt, a.r_mp, a.t_pp, a.r_pm, a.t_mm);
  VSM_LAUNCH_CHECK("k_elemental");
  hipLaunchKernelGGL(k_elemental_sfi<T>, dim3((N * S + 255) / 256), dim3(256), 0, st, N, q.n_stokes, S, m, ndoubl,
                     q.i_mu0, dtau, varpi, tau_sum, F0, Zpp, Zmp, zs, q.mu, a.j0_p, a.j0_m);
  VSM_LAUNCH_CHECK("k_elemental_sfi");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// small elementwise kernels
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_scale2(int N, int S, const T* __restrict__ expk, const T* __restrict__ a, const T* __restrict__ b,
                         T* oa, T* ob) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * S) return;
  const T k = expk[e / N];
  oa[e] = a[e] * k;
  ob[e] = b[e] * k;
}
template <typename T>
__global__ void k_square(int S, T* x) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < S) x[e] = x[e] * x[e];
}
// dst[N*N*S] <- src with slice stride ss (0 = broadcast one matrix)
template <typename T>
__global__ void k_copy_strided(long long per, int S, const T* __restrict__ src, long long ss, T* dst) {
  const long long e = (long long)blockIdx.y * 256 + threadIdx.x;
  if (e >= per) return;
  const int s = blockIdx.x;
  dst[(long long)s * per + e] = src[(long long)s * ss + e];
}
template <typename T>
int copy_strided(long long per, int S, const T* src, long long ss, T* dst, hipStream_t st) {
  if (S <= 0 || per <= 0) return VSM_OK;
  hipLaunchKernelGGL(k_copy_strided<T>, dim3(S, (unsigned)((per + 255) / 256)), dim3(256), 0, st, per, S, src, ss, dst);
  VSM_LAUNCH_CHECK("k_copy_strided");
  return VSM_OK;
}

// apply_D! + apply_D_SFI! after doubling (doubling.jl:178-252)
template <typename T>
__global__ void k_apply_D(int N, int n_stokes, T* r_mp, const T* __restrict__ t_pp, T* r_pm, T* t_mm, T* j0_m) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const int i = e % N, j = e / N;
  const long long o = (long long)s * N * N + e;
  T r = r_mp[o];
  const T t = t_pp[o];
  const bool ui = is_uv_row(i, n_stokes), uj = is_uv_row(j, n_stokes);
  if (ui) r = -r;
  r_mp[o] = r;
  r_pm[o] = (ui == uj) ? r : -r;
  t_mm[o] = (ui == uj) ? t : -t;
  if (j == 0 && ui) j0_m[(long long)s * N + i] = -j0_m[(long long)s * N + i];
}

template <typename T>
int doubling(int N, int n_stokes, int S, int ndoubl, T* expk, const added<T>& a, T* work, hipStream_t st) {
  if (ndoubl == 0 || S <= 0) return VSM_OK;  // doubling.jl:50
  if constexpr (sizeof(T) == 8) {
    static const bool no_strip = ab_switch("VSM_NO_STRIP128");
    if (!no_strip && strip128_supported(N)) return strip128_doubling(N, n_stokes, S, ndoubl, expk, a, st);
  } else {
    // Float32, 96 < N <= 128 (beyond the FP32 strip kernels): the FP64 kernels over the FP32 arrays
    if (strip128_f32_supported(N)) return strip128_doubling<T>(N, n_stokes, S, ndoubl, expk, a, st);
  }
  const long long NN = (long long)N * N, per = NN * S, pv = (long long)N * S;
  T* W1 = work;
  T* W2 = W1 + per;
  T* W3 = W2 + per;
  T* v1 = W3 + per;  // j1+
  T* v2 = v1 + pv;   // j1-
  T* v3 = v2 + pv;
  T* v4 = v3 + pv;
  int rc;
  const T* nul = nullptr;
  for (int n = 0; n < ndoubl; ++n) {
    // G = (I - r r)^-1 ; tt = t G            (rt_helpers.jl:102-107)
    if ((rc = gemm<T>(N, N, N, S, a.r_mp, NN, a.r_mp, NN, W1, NN, T(-1), nul, 0, T(0), T(1), st))) return rc;
    if ((rc = batch_inv<T>(N, S, W1, W1, nullptr, st))) return rc;
    if ((rc = gemm<T>(N, N, N, S, a.t_pp, NN, W1, NN, W2, NN, T(1), nul, 0, T(0), T(0), st))) return rc;
    // sources                                 (rt_helpers.jl:128-134)
    hipLaunchKernelGGL(k_scale2<T>, dim3((unsigned)((pv + 255) / 256)), dim3(256), 0, st, N, S, expk, a.j0_p, a.j0_m, v1, v2);
    VSM_LAUNCH_CHECK("k_scale2");
    if ((rc = gemm<T>(N, 1, N, S, a.r_mp, NN, a.j0_p, N, v3, N, T(1), v2, N, T(1), T(0), st))) return rc;   // j1- + r j0+
    if ((rc = gemm<T>(N, 1, N, S, W2, NN, v3, N, v4, N, T(1), a.j0_m, N, T(1), T(0), st))) return rc;        // j0- new
    if ((rc = gemm<T>(N, 1, N, S, a.r_mp, NN, v2, N, v3, N, T(1), a.j0_p, N, T(1), T(0), st))) return rc;    // j0+ + r j1-
    if ((rc = gemm<T>(N, 1, N, S, W2, NN, v3, N, a.j0_p, N, T(1), v1, N, T(1), T(0), st))) return rc;        // j0+ new
    if ((rc = copy_strided<T>(pv, 1, v4, 0, a.j0_m, st))) return rc;
    // r <- r + tt r t ; t <- tt t ; expk <- expk^2   (rt_helpers.jl:161-166)
    if ((rc = gemm<T>(N, N, N, S, W2, NN, a.r_mp, NN, W3, NN, T(1), nul, 0, T(0), T(0), st))) return rc;
    if ((rc = gemm<T>(N, N, N, S, W3, NN, a.t_pp, NN, a.r_mp, NN, T(1), a.r_mp, NN, T(1), T(0), st))) return rc;
    if ((rc = gemm<T>(N, N, N, S, W2, NN, a.t_pp, NN, W3, NN, T(1), nul, 0, T(0), T(0), st))) return rc;
    if ((rc = copy_strided<T>(per, 1, W3, 0, a.t_pp, st))) return rc;
    hipLaunchKernelGGL(k_square<T>, dim3((S + 255) / 256), dim3(256), 0, st, S, expk);
    VSM_LAUNCH_CHECK("k_square");
  }
  hipLaunchKernelGGL(k_apply_D<T>, dim3(S, (unsigned)((NN + 255) / 256)), dim3(256), 0, st, N, n_stokes, a.r_mp, a.t_pp,
                     a.r_pm, a.t_mm, a.j0_m);
  VSM_LAUNCH_CHECK("k_apply_D");
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// non-scattering layer, TOA copy
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_noscat(int N, const T* __restrict__ tau, const T* __restrict__ mu, T* r_mp, T* r_pm, T* t_pp,
                         T* t_mm, T* j0_m) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const int i = e % N, j = e / N;
  const long long o = (long long)s * N * N + e;
  const T t = (i == j) ? exp(-tau[s] / mu[i]) : T(0);
  r_mp[o] = T(0);
  r_pm[o] = T(0);
  t_pp[o] = t;
  t_mm[o] = t;
  if (j == 0) j0_m[(long long)s * N + i] = T(0);
}
template <typename T>
int noscat_layer(const quad<T>& q, int S, const T* tau, const added<T>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const int N = q.N;
  hipLaunchKernelGGL(k_noscat<T>, dim3(S, (N * N + 255) / 256), dim3(256), 0, st, N, tau, q.mu, a.r_mp, a.r_pm, a.t_pp,
                     a.t_mm, a.j0_m);
  VSM_LAUNCH_CHECK("k_noscat");
  return VSM_OK;
}

// contribute!(::PreparedThermalEmission, ...) (Sources/thermal_emission.jl:241-301): the `:thermal` source slot of an elemental
// layer, j0+ = j0- = 2 pi (1 - varpi) B (1 - exp(-dtau / mu_i)) on the Stokes-I rows, zero elsewhere.  One thread per (row, point).
template <typename T>
__global__ void k_thermal_source(int N, int ns, int S, const T* __restrict__ dtau, const T* __restrict__ varpi,
                                 const T* __restrict__ B, const T* __restrict__ mu, T* j0_p, T* j0_m) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)N * S) return;
  const int i = (int)(e % N);
  const long long s = e / N;
  T v = T(0);
  if (i % ns == 0 && mu[i] > num<T>::eps())
    v = T(6.283185307179586476925286766559) * (T(1) - varpi[s]) * B[s] * (-expm1(-dtau[s] / mu[i]));
  j0_p[e] = v;
  j0_m[e] = v;
}
template <typename T>
int thermal_source(const quad<T>& q, int S, const T* dtau, const T* varpi, const T* B, const added<T>& a, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const long long n = (long long)q.N * S;
  hipLaunchKernelGGL(k_thermal_source<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, q.N, q.n_stokes, S, dtau, varpi, B,
                     q.mu, a.j0_p, a.j0_m);
  VSM_LAUNCH_CHECK("k_thermal_source");
  return VSM_OK;
}

// dst = D src D  (r+- from r-+, t-- from t++; doubling.jl:178-201)
template <typename T>
__global__ void k_copy_dsym(int N, int ns, const T* __restrict__ src, long long ss, T* dst) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int s = blockIdx.x;
  const T x = src[(long long)s * ss + e];
  dst[(long long)s * N * N + e] = (is_uv_row(e % N, ns) == is_uv_row(e / N, ns)) ? x : -x;
}
template <typename T>
int copy_added_to_composite(int N, int S, const added<T>& a, const composite<T>& c, hipStream_t st) {
  const long long NN = (long long)N * N;
  int rc;
  if ((rc = copy_strided<T>(NN, S, a.t_pp, a.mat_stride, c.T_pp, st))) return rc;
  if ((rc = copy_strided<T>(NN, S, a.r_mp, a.mat_stride, c.R_mp, st))) return rc;
  if (a.d_symmetric) {
    if (S > 0) {
      dim3 grid(S, (unsigned)((NN + 255) / 256));
      hipLaunchKernelGGL(k_copy_dsym<T>, grid, dim3(256), 0, st, N, a.d_symmetric, a.t_pp, a.mat_stride, c.T_mm);
      hipLaunchKernelGGL(k_copy_dsym<T>, grid, dim3(256), 0, st, N, a.d_symmetric, a.r_mp, a.mat_stride, c.R_pm);
      VSM_LAUNCH_CHECK("k_copy_dsym");
    }
    if ((rc = copy_strided<T>((long long)N * S, 1, a.j0_p, 0, c.J0_p, st))) return rc;
    if ((rc = copy_strided<T>((long long)N * S, 1, a.j0_m, 0, c.J0_m, st))) return rc;
    return VSM_OK;
  }
  if ((rc = copy_strided<T>(NN, S, a.t_mm, a.mat_stride, c.T_mm, st))) return rc;
  if ((rc = copy_strided<T>(NN, S, a.r_pm, a.mat_stride, c.R_pm, st))) return rc;
  if ((rc = copy_strided<T>((long long)N * S, 1, a.j0_p, 0, c.J0_p, st))) return rc;
  if ((rc = copy_strided<T>((long long)N * S, 1, a.j0_m, 0, c.J0_m, st))) return rc;
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// interaction!  (interaction.jl:52-266) -- operator-for-operator, reference order
// ---------------------------------------------------------------------------
template <typename T>
int interaction_generic(int iface, int N, int S, const composite<T>& c, const added<T>& a, T* work, hipStream_t st) {
  if (S <= 0) return VSM_OK;
  const long long NN = (long long)N * N, per = NN * S, pv = (long long)N * S;
  const long long as = a.mat_stride;
  T* W1 = work;
  T* W2 = W1 + per;
  T* W3 = W2 + per;
  T* v1 = W3 + per;
  T* v2 = v1 + pv;
  const T* nul = nullptr;
  const T one = T(1), zero = T(0);
  int rc;
#define G(...)                                   \
  if ((rc = gemm<T>(__VA_ARGS__, st))) return rc
  switch (iface) {
    case VSM_IFACE_00:
      G(N, 1, N, S, a.t_pp, as, c.J0_p, N, v1, N, one, a.j0_p, N, one, zero);      // J0+ = j0+ + t++ J0+
      if ((rc = copy_strided<T>(pv, 1, v1, 0, c.J0_p, st))) return rc;
      G(N, 1, N, S, c.T_mm, NN, a.j0_m, N, c.J0_m, N, one, c.J0_m, N, one, zero);  // J0- += T-- j0-
      G(N, N, N, S, a.t_mm, as, c.T_mm, NN, W3, NN, one, nul, 0, zero, zero);      // T-- = t-- T--
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_mm, st))) return rc;
      G(N, N, N, S, a.t_pp, as, c.T_pp, NN, W3, NN, one, nul, 0, zero, zero);      // T++ = t++ T++
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_pp, st))) return rc;
      break;
    case VSM_IFACE_01:
      G(N, 1, N, S, a.r_mp, as, c.J0_p, N, v1, N, one, a.j0_m, N, one, zero);      // r-+ J0+ + j0-
      G(N, 1, N, S, c.T_mm, NN, v1, N, c.J0_m, N, one, c.J0_m, N, one, zero);      // J0- += T-- (..)
      G(N, 1, N, S, a.t_pp, as, c.J0_p, N, v1, N, one, a.j0_p, N, one, zero);      // J0+ = j0+ + t++ J0+
      if ((rc = copy_strided<T>(pv, 1, v1, 0, c.J0_p, st))) return rc;
      G(N, N, N, S, c.T_mm, NN, a.r_mp, as, W3, NN, one, nul, 0, zero, zero);      // T-- r-+
      G(N, N, N, S, W3, NN, c.T_pp, NN, c.R_mp, NN, one, nul, 0, zero, zero);      // R-+ = (T-- r-+) T++
      if ((rc = copy_strided<T>(NN, S, a.r_pm, as, c.R_pm, st))) return rc;        // R+- = r+-
      G(N, N, N, S, a.t_pp, as, c.T_pp, NN, W3, NN, one, nul, 0, zero, zero);      // T++ = t++ T++
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_pp, st))) return rc;
      G(N, N, N, S, c.T_mm, NN, a.t_mm, as, W3, NN, one, nul, 0, zero, zero);      // T-- = T-- t--
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_mm, st))) return rc;
      break;
    case VSM_IFACE_10:
      G(N, 1, N, S, c.R_pm, NN, a.j0_m, N, v1, N, one, c.J0_p, N, one, zero);      // J0+ + R+- j0-
      G(N, 1, N, S, a.t_pp, as, v1, N, c.J0_p, N, one, a.j0_p, N, one, zero);      // J0+ = j0+ + t++ (..)
      G(N, 1, N, S, c.T_mm, NN, a.j0_m, N, c.J0_m, N, one, c.J0_m, N, one, zero);  // J0- += T-- j0-
      G(N, N, N, S, a.t_pp, as, c.T_pp, NN, W3, NN, one, nul, 0, zero, zero);      // T++ = t++ T++
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_pp, st))) return rc;
      G(N, N, N, S, c.T_mm, NN, a.t_mm, as, W3, NN, one, nul, 0, zero, zero);      // T-- = T-- t--
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_mm, st))) return rc;
      G(N, N, N, S, a.t_pp, as, c.R_pm, NN, W3, NN, one, nul, 0, zero, zero);      // t++ R+-
      G(N, N, N, S, W3, NN, a.t_mm, as, c.R_pm, NN, one, nul, 0, zero, zero);      // R+- = (t++ R+-) t--
      break;
    case VSM_IFACE_11:
      G(N, N, N, S, a.r_mp, as, c.R_pm, NN, W1, NN, -one, nul, 0, zero, one);      // I - r-+ R+-
      if ((rc = batch_inv<T>(N, S, W1, W1, nullptr, st))) return rc;
      G(N, N, N, S, c.T_mm, NN, W1, NN, W2, NN, one, nul, 0, zero, zero);          // T01_inv = T-- G1
      G(N, 1, N, S, a.r_mp, as, c.J0_p, N, v1, N, one, a.j0_m, N, one, zero);      // r-+ J0+ + j0-
      G(N, 1, N, S, W2, NN, v1, N, c.J0_m, N, one, c.J0_m, N, one, zero);          // J0- += T01_inv (..)
      G(N, N, N, S, W2, NN, a.r_mp, as, W3, NN, one, nul, 0, zero, zero);          // T01_inv r-+
      G(N, N, N, S, W3, NN, c.T_pp, NN, c.R_mp, NN, one, c.R_mp, NN, one, zero);   // R-+ += (..) T++
      G(N, N, N, S, W2, NN, a.t_mm, as, W3, NN, one, nul, 0, zero, zero);          // T-- = T01_inv t--
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_mm, st))) return rc;
      G(N, N, N, S, c.R_pm, NN, a.r_mp, as, W1, NN, -one, nul, 0, zero, one);      // I - R+- r-+
      if ((rc = batch_inv<T>(N, S, W1, W1, nullptr, st))) return rc;
      G(N, N, N, S, a.t_pp, as, W1, NN, W2, NN, one, nul, 0, zero, zero);          // T21_inv = t++ G2
      G(N, 1, N, S, c.R_pm, NN, a.j0_m, N, v1, N, one, c.J0_p, N, one, zero);      // J0+ + R+- j0-
      G(N, 1, N, S, W2, NN, v1, N, c.J0_p, N, one, a.j0_p, N, one, zero);          // J0+ = j0+ + T21_inv (..)
      G(N, N, N, S, W2, NN, c.T_pp, NN, W3, NN, one, nul, 0, zero, zero);          // T++ = T21_inv T++
      if ((rc = copy_strided<T>(per, 1, W3, 0, c.T_pp, st))) return rc;
      G(N, N, N, S, W2, NN, c.R_pm, NN, W3, NN, one, nul, 0, zero, zero);          // T21_inv R+-
      G(N, N, N, S, W3, NN, a.t_mm, as, c.R_pm, NN, one, a.r_pm, as, one, zero);   // R+- = r+- + (..) t--
      break;
    default:
      set_error("interaction: unknown scattering interface %d", iface);
      return VSM_ERR_INVALID_ARG;
  }
#undef G
  (void)v2;
  return VSM_OK;
}

// ---------------------------------------------------------------------------
// Lambertian surface (lambertian_surface.jl:41-95), postprocessing (postprocessing_vza.jl:23-94)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void k_lambertian_mats(int N, int n_stokes, int m, T albedo, const T* __restrict__ mu,
                                  const T* __restrict__ wt, T* r_mp, T* r_pm, T* t_pp, T* t_mm) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * N) return;
  const int i = e % N, j = e / N;
  T r = T(0);
  if (m == 0 && (i % n_stokes) == 0 && (j % n_stokes) == 0) r = T(2) * albedo * (mu[j] * wt[j]);
  r_mp[e] = r;
  r_pm[e] = T(0);
  const T t = (i == j) ? T(1) : T(0);
  t_pp[e] = t;
  t_mm[e] = t;
}
template <typename T>
__global__ void k_lambertian_src(int N, int n_stokes, int S, int m, T albedo, int i_mu0, T mu0,
                                 const T* __restrict__ tau_sum, T* j0_p, T* j0_m) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N * S) return;
  const int i = e % N, s = e / N;
  T jp = T(0), jm = T(0);
  if (m == 0) {
    const T att = exp(-tau_sum[s] / mu0);
    const int i_start = n_stokes * i_mu0;
    if (i == i_start) jp = att;                                   // I0 = e1 on the SZA stream
    if ((i % n_stokes) == 0) jm = mu0 * (T(2) * albedo) * att;    // mu0 * (R_surf * I0_N)
  }
  j0_p[e] = jp;
  j0_m[e] = jm;
}
template <typename T>
int lambertian_surface(const quad<T>& q, int S, int m, T albedo, const T* tau_sum, const added<T>& a, hipStream_t st) {
  if (a.mat_stride != 0) {
    set_error("lambertian_surface: added.mat_stride must be 0 (one shared surface block)");
    return VSM_ERR_INVALID_ARG;
  }
  const int N = q.N;
  hipLaunchKernelGGL(k_lambertian_mats<T>, dim3((N * N + 255) / 256), dim3(256), 0, st, N, q.n_stokes, m, albedo, q.mu,
                     q.wt, a.r_mp, a.r_pm, a.t_pp, a.t_mm);
  VSM_LAUNCH_CHECK("k_lambertian_mats");
  if (S > 0) {
    hipLaunchKernelGGL(k_lambertian_src<T>, dim3((N * S + 255) / 256), dim3(256), 0, st, N, q.n_stokes, S, m, albedo,
                       q.i_mu0, q.mu0, tau_sum, a.j0_p, a.j0_m);
    VSM_LAUNCH_CHECK("k_lambertian_src");
  }
  return VSM_OK;
}

struct pp_args {
  int row0[64];
  double w[256];
};
template <typename T>
__global__ void k_postprocess(int N, int n_stokes, int S, int nV, int nVtot, int v0, pp_args pa, const T* __restrict__ J0_m,
                              const T* __restrict__ J0_p, T* R, T* Tt) {
  // one chunk of <= 64 viewing angles [v0, v0 + nV) of the nVtot of the output arrays [S][n_stokes][nVtot]
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long tot = (long long)nV * n_stokes * S;
  if (e >= tot) return;
  const int v = (int)(e % nV);
  const int k = (int)((e / nV) % n_stokes);
  const long long s = e / ((long long)nV * n_stokes);
  const T w = (T)pa.w[v + nV * k];
  const long long src = s * N + pa.row0[v] + k;
  const long long dst = (s * n_stokes + k) * nVtot + v0 + v;
  R[dst] += w * J0_m[src];
  Tt[dst] += w * J0_p[src];
}
template <typename T>
int postprocess_vza(int N, int n_stokes, int S, int nV, const int* row0_h, const T* w_h, const T* J0_m, const T* J0_p,
                    T* R, T* Tt, hipStream_t st) {
  if (n_stokes > 4) {
    set_error("postprocess_vza: n_stokes <= 4 (got %d)", n_stokes);
    return VSM_ERR_UNSUPPORTED;
  }
  if (S <= 0 || nV <= 0) return VSM_OK;
  for (int v0 = 0; v0 < nV; v0 += 64) {   // the launch arguments carry 64 viewing angles: any number, in chunks
    const int nc = nV - v0 < 64 ? nV - v0 : 64;
    pp_args pa;
    for (int v = 0; v < nc; ++v) pa.row0[v] = row0_h[v0 + v];
    for (int k = 0; k < n_stokes; ++k)
      for (int v = 0; v < nc; ++v) pa.w[v + nc * k] = (double)w_h[v0 + v + nV * k];
    const long long tot = (long long)nc * n_stokes * S;
    hipLaunchKernelGGL(k_postprocess<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, n_stokes, S, nc, nV, v0, pa,
                       J0_m, J0_p, R, Tt);
    VSM_LAUNCH_CHECK("k_postprocess");
  }
  return VSM_OK;
}

// explicit instantiations ------------------------------------------------------
#define VSM_INST(T)                                                                                                    \
  template int gemm2<T>(int, int, int, int, int, const T*, long long, long long, const T*, long long, long long, T*,  \
                        long long, long long, T, const T*, long long, long long, T, T, hipStream_t);                   \
  template int gemm<T>(int, int, int, int, const T*, long long, const T*, long long, T*, long long, T, const T*,      \
                       long long, T, T, hipStream_t);                                                                  \
  template int batch_inv<T>(int, int, const T*, T*, int*, hipStream_t);                                                \
  template int elemental<T>(const quad<T>&, int, int, int, const T*, const T*, const T*, const T*, const T*, const T*, \
                            long long, const added<T>&, hipStream_t);                                                  \
  template int doubling<T>(int, int, int, int, T*, const added<T>&, T*, hipStream_t);                                  \
  template int noscat_layer<T>(const quad<T>&, int, const T*, const added<T>&, hipStream_t);                           \
  template int thermal_source<T>(const quad<T>&, int, const T*, const T*, const T*, const added<T>&, hipStream_t);      \
  template int copy_added_to_composite<T>(int, int, const added<T>&, const composite<T>&, hipStream_t);                \
  template int interaction_generic<T>(int, int, int, const composite<T>&, const added<T>&, T*, hipStream_t);           \
  template int lambertian_surface<T>(const quad<T>&, int, int, T, const T*, const added<T>&, hipStream_t);             \
  template int postprocess_vza<T>(int, int, int, int, const int*, const T*, const T*, const T*, T*, T*, hipStream_t);  \
  template int copy_strided<T>(long long, int, const T*, long long, T*, hipStream_t);
VSM_INST(double)
VSM_INST(float)

}  // namespace vsm
