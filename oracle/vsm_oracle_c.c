/* CPU ORACLE / CPU BASELINE in C + OpenMP -- test infrastructure, NOT the product path.
 *
 * A C restatement of the elastic forward hot path of vSmartMOM.jl's rt_run for scenes whose layers all scatter
 * (ScatteringInterface_11 throughout) over a Lambertian surface, structured like the reference's CPU path
 * (src/CoreRT/tools/cpu_batched.jl:25-82): every batched operator is a dense N x N product or an LU solve per spectral
 * point, and the spectral axis is spread over threads.  Statement order follows
 *   elemental!            src/CoreRT/CoreKernel/elemental.jl:289-392 (get_elem_rt!, get_elem_rt_SFI!, apply_D_elemental!)
 *   doubling_helper!      src/CoreRT/CoreKernel/doubling.jl:38-131, rt_helpers.jl:102-166, apply_D doubling.jl:178-252
 *   interaction_helper!   src/CoreRT/CoreKernel/interaction.jl:207-266 (ScatteringInterface_11)
 *   create_surface_layer! src/CoreRT/Surfaces/lambertian_surface.jl:41-95
 *   postprocessing_vza!   src/CoreRT/tools/postprocessing_vza.jl:23-94
 * Inputs that the reference prepares on the host (streams, Z moments, tau/varpi/tau_sum, ndoubl) are handed in by the
 * Python oracle (oracle/vsm_oracle.py), which is pinned by the reference's golden tables; tests/test_oracle_c.py checks this
 * file against it.  Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load the library built from it.
 *
 * Matrices are row-major N x N (A[i*N + j], i = outgoing stream*Stokes, j = incoming); one thread owns one spectral point
 * for the whole run (all moments, all layers), so no synchronisation is needed.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void mm(int N, const double* restrict A, const double* restrict B, double* restrict C) { /* C = A B */
  int i = 0;
  for (; i + 4 <= N; i += 4) {   /* four rows of C share every row of B (register blocking; j is the SIMD axis) */
    double* restrict c0 = C + (size_t)i * N;
    double* restrict c1 = c0 + N;
    double* restrict c2 = c1 + N;
    double* restrict c3 = c2 + N;
    for (int j = 0; j < N; ++j) c0[j] = c1[j] = c2[j] = c3[j] = 0.0;
    for (int k = 0; k < N; ++k) {
      const double a0 = A[(size_t)i * N + k], a1 = A[(size_t)(i + 1) * N + k], a2 = A[(size_t)(i + 2) * N + k],
                   a3 = A[(size_t)(i + 3) * N + k];
      const double* restrict b = B + (size_t)k * N;
#pragma omp simd
      for (int j = 0; j < N; ++j) {
        const double bj = b[j];
        c0[j] += a0 * bj;
        c1[j] += a1 * bj;
        c2[j] += a2 * bj;
        c3[j] += a3 * bj;
      }
    }
  }
  for (; i < N; ++i) {
    double* restrict c = C + (size_t)i * N;
    for (int j = 0; j < N; ++j) c[j] = 0.0;
    for (int k = 0; k < N; ++k) {
      const double a = A[(size_t)i * N + k];
      const double* restrict b = B + (size_t)k * N;
#pragma omp simd
      for (int j = 0; j < N; ++j) c[j] += a * b[j];
    }
  }
}
static void mv(int N, const double* restrict A, const double* restrict x, double* restrict y) { /* y = A x */
  for (int i = 0; i < N; ++i) {
    double s = 0.0;
    const double* restrict a = A + (size_t)i * N;
#pragma omp simd reduction(+ : s)
    for (int j = 0; j < N; ++j) s += a[j] * x[j];
    y[i] = s;
  }
}
/* X = A^-1 by LU with partial pivoting + N solves (the reference's `A \ I`, cpu_batched.jl:32-36); A is destroyed */
static int inv_lu(int N, double* restrict A, double* restrict X, int* restrict piv) {
  for (int k = 0; k < N; ++k) {
    int p = k;
    double best = fabs(A[(size_t)k * N + k]);
    for (int i = k + 1; i < N; ++i) {
      const double v = fabs(A[(size_t)i * N + k]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    piv[k] = p;
    if (best == 0.0) return k + 1;
    if (p != k)
      for (int j = 0; j < N; ++j) {
        const double t = A[(size_t)k * N + j];
        A[(size_t)k * N + j] = A[(size_t)p * N + j];
        A[(size_t)p * N + j] = t;
      }
    const double d = 1.0 / A[(size_t)k * N + k];
    for (int i = k + 1; i < N; ++i) {
      const double f = A[(size_t)i * N + k] * d;
      A[(size_t)i * N + k] = f;
      double* restrict ai = A + (size_t)i * N;
      const double* restrict ak = A + (size_t)k * N;
#pragma omp simd
      for (int j = k + 1; j < N; ++j) ai[j] -= f * ak[j];
    }
  }
  /* X = P (identity with the row interchanges applied), then forward / back substitution on all columns at once */
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) X[(size_t)i * N + j] = (i == j) ? 1.0 : 0.0;
  for (int k = 0; k < N; ++k)
    if (piv[k] != k)
      for (int j = 0; j < N; ++j) {
        const double t = X[(size_t)k * N + j];
        X[(size_t)k * N + j] = X[(size_t)piv[k] * N + j];
        X[(size_t)piv[k] * N + j] = t;
      }
  for (int k = 0; k < N; ++k)
    for (int i = k + 1; i < N; ++i) {
      const double f = A[(size_t)i * N + k];
      double* restrict xi = X + (size_t)i * N;
      const double* restrict xk = X + (size_t)k * N;
#pragma omp simd
      for (int j = 0; j < N; ++j) xi[j] -= f * xk[j];
    }
  for (int k = N - 1; k >= 0; --k) {
    const double d = 1.0 / A[(size_t)k * N + k];
    double* restrict xk = X + (size_t)k * N;
#pragma omp simd
    for (int j = 0; j < N; ++j) xk[j] *= d;
    for (int i = 0; i < k; ++i) {
      const double f = A[(size_t)i * N + k];
      double* restrict xi = X + (size_t)i * N;
#pragma omp simd
      for (int j = 0; j < N; ++j) xi[j] -= f * xk[j];
    }
  }
  return 0;
}
static double expdiff_neg(double a, double b) { /* exp(-a) - exp(-b), rt_helpers.jl:32-40 */
  if (a == b) return 0.0;
  if (a < b) return exp(-a) * (-expm1(-(b - a)));
  return -exp(-b) * (-expm1(-(a - b)));
}
static int is_uv(int i, int ns) { return (i % ns) >= 2; }

typedef struct {
  double *r, *t, *rpm, *tmm, *jp, *jm;                                   /* added layer */
  double *R, *Rpm, *T, *Tmm, *Jp, *Jm;                                   /* composite */
  double *w1, *w2, *w3, *w4, *w5, *v1, *v2, *v3, *v4;                    /* work */
  int* piv;
} ws_t;

/* elemental + doubling of one layer for one spectral point */
static void layer(int N, int ns, const double* mu, const double* wt, int i_mu0, double mu0, int m, int nd, double dtau,
                  double varpi, double tau_sum, const double* F0, const double* Zpp, const double* Zmp, ws_t* w) {
  double *r = w->r, *t = w->t, *jp = w->jp, *jm = w->jm;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const double wct = (m == 0) ? wt[j] / 2 : wt[j] / 4;
      const double mi = mu[i], mj = mu[j];
      double rr, tt;
      if (wct > 2.220446049250313e-16) {
        rr = varpi * Zmp[(size_t)i * N + j] * (mj / (mi + mj)) * wct * (-expm1(-dtau * ((1 / mi) + (1 / mj))));
        if (mi == mj) {
          if (i == j)
            tt = exp(-dtau / mi) * (1 + varpi * Zpp[(size_t)i * N + j] * (dtau / mi) * wct);
          else
            tt = exp(-dtau / mj) * (varpi * Zpp[(size_t)i * N + j] * (dtau / mi) * wct);
        } else {
          tt = varpi * Zpp[(size_t)i * N + j] * (mj / (mi - mj)) * wct * expdiff_neg(dtau / mi, dtau / mj);
        }
      } else {
        rr = 0.0;
        tt = (i == j) ? exp(-dtau / mi) : 0.0;
      }
      if (nd >= 1 && is_uv(i, ns)) rr = -rr;
      r[(size_t)i * N + j] = rr;
      t[(size_t)i * N + j] = tt;
    }
  const int i0 = ns * i_mu0;
  const double wct02 = (m == 0) ? 0.5 : 0.25, ms = mu[i0], att = exp(-tau_sum / ms);
  for (int i = 0; i < N; ++i) {
    double zp = 0, zm = 0;
    for (int q = 0; q < ns; ++q) {
      zp += Zpp[(size_t)i * N + i0 + q] * F0[q];
      zm += Zmp[(size_t)i * N + i0 + q] * F0[q];
    }
    const double mi = mu[i];
    double vp, vm;
    if (i >= i0 && i < i0 + ns)
      vp = wct02 * varpi * zp * (dtau / mi) * exp(-dtau / mi);
    else
      vp = wct02 * varpi * zp * (ms / (mi - ms)) * expdiff_neg(dtau / mi, dtau / ms);
    vm = wct02 * varpi * zm * (ms / (mi + ms)) * (-expm1(-dtau * ((1 / mi) + (1 / ms))));
    vp *= att;
    vm *= att;
    if (nd >= 1 && is_uv(i, ns)) vm = -vm;
    jp[i] = vp;
    jm[i] = vm;
  }
  if (nd < 1) { /* apply_D_elemental!: fill r+-, t-- by Stokes parity */
    for (int i = 0; i < N; ++i)
      for (int j = 0; j < N; ++j) {
        const int same = is_uv(i, ns) == is_uv(j, ns);
        w->rpm[(size_t)i * N + j] = same ? r[(size_t)i * N + j] : -r[(size_t)i * N + j];
        w->tmm[(size_t)i * N + j] = same ? t[(size_t)i * N + j] : -t[(size_t)i * N + j];
      }
    return;
  }
  /* doubling (rt_helpers.jl:102-166) */
  double expk = exp(-dtau / mu0);
  double *G = w->w1, *tt = w->w2, *tmp = w->w3, *E = w->w4, *x = w->w5;
  double *j1p = w->v1, *j1m = w->v2, *u = w->v3, *y = w->v4;
  const size_t NN = (size_t)N * N;
  for (int n = 0; n < nd; ++n) {
    mm(N, r, r, E);
    for (size_t e = 0; e < NN; ++e) E[e] = -E[e];
    for (int i = 0; i < N; ++i) E[(size_t)i * N + i] += 1.0;
    inv_lu(N, E, G, w->piv);                 /* G = (I - r r)^-1 */
    mm(N, t, G, tt);                         /* tt = t G */
    for (int i = 0; i < N; ++i) {
      j1p[i] = jp[i] * expk;
      j1m[i] = jm[i] * expk;
    }
    /* j0- <- j0- + tt (j1- + r j0+) ;  j0+ <- j1+ + tt (j0+ + r j1-) */
    mv(N, r, jp, u);
    for (int i = 0; i < N; ++i) u[i] += j1m[i];
    mv(N, tt, u, y);
    mv(N, r, j1m, u);
    for (int i = 0; i < N; ++i) {
      jm[i] += y[i];
      u[i] += jp[i];
    }
    mv(N, tt, u, y);
    for (int i = 0; i < N; ++i) jp[i] = j1p[i] + y[i];
    expk *= expk;
    mm(N, tt, r, tmp);                       /* tmp = tt r */
    mm(N, tmp, t, x);                        /* r' = r + tmp t */
    for (size_t e = 0; e < NN; ++e) r[e] += x[e];
    mm(N, tt, t, x);                         /* t' = tt t */
    memcpy(t, x, NN * sizeof(double));
  }
  /* apply_D! / apply_D_SFI! */
  for (int i = 0; i < N; ++i) {
    if (is_uv(i, ns)) {
      for (int j = 0; j < N; ++j) r[(size_t)i * N + j] = -r[(size_t)i * N + j];
      jm[i] = -jm[i];
    }
  }
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      const int same = is_uv(i, ns) == is_uv(j, ns);
      w->rpm[(size_t)i * N + j] = same ? r[(size_t)i * N + j] : -r[(size_t)i * N + j];
      w->tmm[(size_t)i * N + j] = same ? t[(size_t)i * N + j] : -t[(size_t)i * N + j];
    }
}

/* interaction_helper!(::ScatteringInterface_11) (interaction.jl:207-266): composite (above) + added (below), in place */
static void interaction11(int N, const double* r, const double* t, const double* rpm, const double* tmm, const double* jp,
                          const double* jm, ws_t* w) {
  double *R = w->R, *Rpm = w->Rpm, *T = w->T, *Tmm = w->Tmm, *Jp = w->Jp, *Jm = w->Jm;
  double *A = w->w1, *G = w->w2, *T01 = w->w3, *x = w->w4, *y = w->w5, *u = w->v1, *v = w->v2;
  const size_t NN = (size_t)N * N;
  /* T01 = T-- (I - r-+ R+-)^-1 */
  mm(N, r, Rpm, A);
  for (size_t e = 0; e < NN; ++e) A[e] = -A[e];
  for (int i = 0; i < N; ++i) A[(size_t)i * N + i] += 1.0;
  inv_lu(N, A, G, w->piv);
  mm(N, Tmm, G, T01);
  /* J0- = J0- + T01 (r-+ J0+ + j0-) */
  mv(N, r, Jp, u);
  for (int i = 0; i < N; ++i) u[i] += jm[i];
  mv(N, T01, u, v);
  for (int i = 0; i < N; ++i) Jm[i] += v[i];
  /* R-+ = R-+ + T01 r-+ T++ ;  T-- = T01 t-- */
  mm(N, T01, r, x);
  mm(N, x, T, y);
  for (size_t e = 0; e < NN; ++e) R[e] += y[e];
  mm(N, T01, tmm, x);
  memcpy(Tmm, x, NN * sizeof(double));
  /* T21 = t++ (I - R+- r-+)^-1 */
  mm(N, Rpm, r, A);
  for (size_t e = 0; e < NN; ++e) A[e] = -A[e];
  for (int i = 0; i < N; ++i) A[(size_t)i * N + i] += 1.0;
  inv_lu(N, A, G, w->piv);
  mm(N, t, G, T01);                         /* T21 */
  /* J0+ = j0+ + T21 (J0+ + R+- j0-) */
  mv(N, Rpm, jm, u);
  for (int i = 0; i < N; ++i) u[i] += Jp[i];
  mv(N, T01, u, v);
  for (int i = 0; i < N; ++i) Jp[i] = jp[i] + v[i];
  /* T++ = T21 T++ ;  R+- = r+- + T21 R+- t-- */
  mm(N, T01, T, x);
  memcpy(T, x, NN * sizeof(double));
  mm(N, T01, Rpm, x);
  mm(N, x, tmm, y);
  for (size_t e = 0; e < NN; ++e) Rpm[e] = rpm[e] + y[e];
}

/* rt_run for S spectral points.  Layouts: tau, varpi [S, L] and tau_sum [S, L+1] row-major; Zpp / Zmp [M, N, N] (one
 * scatterer: shared by all points); ndoubl [L]; F0 [S, ns]; row0 [nV], wgt [M, nV, ns]; outputs R, T [S, ns, nV] (+=).
 * Returns 0, or -1 if a workspace allocation failed. */
int vsm_oracle_c_rt_run(int N, int ns, const double* mu, const double* wt, int i_mu0, double mu0, int S, int L, int M,
                        const double* tau, const double* varpi, const double* tau_sum, const int* ndoubl, const double* Zpp,
                        const double* Zmp, const double* F0, double albedo, int nV, const int* row0, const double* wgt,
                        int nthreads, double* Rout, double* Tout) {
  int fail = 0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    const size_t NN = (size_t)N * N;
    ws_t w;
    double* buf = (double*)malloc(sizeof(double) * (15 * NN + 12 * (size_t)N));
    w.piv = (int*)malloc(sizeof(int) * (size_t)N);
    if (!buf || !w.piv) {
#pragma omp atomic write
      fail = 1;
    } else {
      double* p = buf;
      double** mats[] = {&w.r, &w.t, &w.rpm, &w.tmm, &w.R, &w.Rpm, &w.T, &w.Tmm, &w.w1, &w.w2, &w.w3, &w.w4, &w.w5};
      for (unsigned k = 0; k < sizeof(mats) / sizeof(mats[0]); ++k) {
        *mats[k] = p;
        p += NN;
      }
      double* sr = p;
      p += NN;
      double* st = p;
      p += NN;
      double** vecs[] = {&w.jp, &w.jm, &w.Jp, &w.Jm, &w.v1, &w.v2, &w.v3, &w.v4};
      for (unsigned k = 0; k < sizeof(vecs) / sizeof(vecs[0]); ++k) {
        *vecs[k] = p;
        p += N;
      }
      double *sjp = p, *sjm = p + N;
#pragma omp for schedule(dynamic, 1)
      for (int s = 0; s < S; ++s) {
        for (int m = 0; m < M; ++m) {
          const double* zp = Zpp + (size_t)m * NN;
          const double* zm = Zmp + (size_t)m * NN;
          for (int l = 0; l < L; ++l) {
            const double tl = tau[(size_t)s * L + l];
            layer(N, ns, mu, wt, i_mu0, mu0, m, ndoubl[l], tl / ldexp(1.0, ndoubl[l]), varpi[(size_t)s * L + l],
                  tau_sum[(size_t)s * (L + 1) + l], F0 + (size_t)s * ns, zp, zm, &w);
            if (l == 0) { /* copy_added_to_composite! */
              memcpy(w.R, w.r, NN * sizeof(double));
              memcpy(w.Rpm, w.rpm, NN * sizeof(double));
              memcpy(w.T, w.t, NN * sizeof(double));
              memcpy(w.Tmm, w.tmm, NN * sizeof(double));
              memcpy(w.Jp, w.jp, N * sizeof(double));
              memcpy(w.Jm, w.jm, N * sizeof(double));
            } else {
              interaction11(N, w.r, w.t, w.rpm, w.tmm, w.jp, w.jm, &w);
            }
          }
          /* Lambertian surface (lambertian_surface.jl:41-95) + last interaction */
          memset(sr, 0, NN * sizeof(double));
          memset(st, 0, NN * sizeof(double));
          for (int i = 0; i < N; ++i) st[(size_t)i * N + i] = 1.0;
          memset(sjp, 0, N * sizeof(double));
          memset(sjm, 0, N * sizeof(double));
          if (m == 0) {
            const double att = exp(-tau_sum[(size_t)s * (L + 1) + L] / mu0);
            for (int i = 0; i < N; i += ns)
              for (int j = 0; j < N; j += ns) sr[(size_t)i * N + j] = 2 * albedo * (mu[j] * wt[j]);
            sjp[ns * i_mu0] = att;
            for (int i = 0; i < N; i += ns) sjm[i] = mu0 * (2 * albedo) * att;
          }
          memset(w.w5, 0, NN * sizeof(double)); /* r+- of the surface = 0 */
          {
            double* zero = (double*)calloc(NN, sizeof(double));
            interaction11(N, sr, st, zero, st, sjp, sjm, &w);
            free(zero);
          }
          for (int v = 0; v < nV; ++v)
            for (int k = 0; k < ns; ++k) {
              const double ww = wgt[((size_t)m * nV + v) * ns + k];
              Rout[((size_t)s * ns + k) * nV + v] += ww * w.Jm[row0[v] + k];
              Tout[((size_t)s * ns + k) * nV + v] += ww * w.Jp[row0[v] + k];
            }
        }
      }
    }
    free(buf);
    free(w.piv);
  }
  return fail ? -1 : 0;
}
