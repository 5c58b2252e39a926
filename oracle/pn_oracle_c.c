/* CPU ORACLE, C restatement (test infrastructure, NOT product code): dense EKF1 filter step in square-root
 * form with dynamic calibration + PI-controlled adaptive loop, for the Brusselator headline configuration.
 * Same algorithm as oracle/pn_oracle.py (`step`, `rejection_loop`), which is pinned to the reference's golden
 * vectors; this file is checked against pn_oracle.py in tests/test_oracle_c.py and serves as the OpenMP
 * "port" CPU baseline of bench.py (one trajectory per thread).
 *
 * Follows (call sites; the arithmetic lives in un-vendored probdiffeq):
 *   strategy_filter + correction_ts1 + solver_dynamic   /root/reference/experiments/1_van_der_pol/vdp.py:62-66
 *   control_proportional_integral + adaptive            /root/reference/src/odecheckpts/ivpsolvers.py:52-53
 *   brusselator                                         /root/reference/src/odecheckpts/ivps.py:124-156
 */
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXN 9

/* Optional LAPACK / BLAS back end for the CPU baseline of bench.py: the Householder QR and the dense products go through the
 * OpenBLAS that scipy ships (blocked dgeqrf, dgemm, dtrsm; one BLAS thread per OpenMP thread) instead of the plain loops below.
 * pnoc_use_lapack(path) dlopens it; without it (or with path = NULL) the restatement runs on its own loops. Results agree to
 * rounding (tests/test_oracle_c.py). */
typedef void (*dgeqrf_t)(const int*, const int*, double*, const int*, double*, double*, const int*, int*);
typedef void (*dgemm_t)(const char*, const char*, const int*, const int*, const int*, const double*, const double*, const int*,
                        const double*, const int*, const double*, double*, const int*);
typedef void (*dtrsm_t)(const char*, const char*, const char*, const char*, const int*, const int*, const double*, const double*,
                        const int*, double*, const int*);
static dgeqrf_t p_dgeqrf = NULL;
static dgemm_t p_dgemm = NULL;
static dtrsm_t p_dtrsm = NULL;

int pnoc_use_lapack(const char* path) {
  p_dgeqrf = NULL; p_dgemm = NULL; p_dtrsm = NULL;
  if (!path) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return -1;
  const char* pre[2] = {"scipy_", ""};
  for (int i = 0; i < 2 && !(p_dgeqrf && p_dgemm && p_dtrsm); ++i) {
    char name[64];
    snprintf(name, sizeof(name), "%sdgeqrf_", pre[i]); p_dgeqrf = (dgeqrf_t)dlsym(h, name);
    snprintf(name, sizeof(name), "%sdgemm_", pre[i]); p_dgemm = (dgemm_t)dlsym(h, name);
    snprintf(name, sizeof(name), "%sdtrsm_", pre[i]); p_dtrsm = (dtrsm_t)dlsym(h, name);
    snprintf(name, sizeof(name), "%sopenblas_set_num_threads", pre[i]);
    void (*setn)(int) = (void (*)(int))dlsym(h, name);
    if (setn) setn(1);
  }
  if (!(p_dgeqrf && p_dgemm && p_dtrsm)) { p_dgeqrf = NULL; p_dgemm = NULL; p_dtrsm = NULL; return -2; }
  return 1;
}

/* row-major C (r x c) = alpha op(A) op(B) + beta C through column-major dgemm (a row-major array IS the transposed column-major one) */
static void gemm_rm(char ta, char tb, int r, int c, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                    double beta, double* C, int ldc) {
  p_dgemm(&tb, &ta, &c, &r, &k, &alpha, B, &ldb, A, &lda, &beta, C, &ldc);
}

typedef struct {
  int nu, n, d, D;
  double a[MAXN][MAXN];   /* binom(nu-i, nu-j) */
  double qs[MAXN][MAXN];  /* lower Cholesky of the flipped Hilbert matrix */
  double fact[MAXN];
} iwp_t;

static void iwp_init(iwp_t* w, int nu, int d) {
  int n = nu + 1;
  w->nu = nu; w->n = n; w->d = d; w->D = n * d;
  double q[MAXN][MAXN];
  memset(w->a, 0, sizeof(w->a)); memset(w->qs, 0, sizeof(w->qs));
  for (int i = 0; i < n; ++i) {
    double f = 1.0;
    for (int m = 2; m <= nu - i; ++m) f *= m;
    w->fact[i] = f;
    for (int j = 0; j < n; ++j) {
      if (j >= i) { /* binom(nu-i, nu-j) */
        double r = 1.0; int A = nu - i, Bq = nu - j;
        for (int t = 1; t <= Bq; ++t) r = r * (A - Bq + t) / t;
        w->a[i][j] = floor(r + 0.5);
      }
      q[i][j] = 1.0 / (2 * nu + 1 - i - j);
    }
  }
  for (int j = 0; j < n; ++j) {
    double s = q[j][j];
    for (int k = 0; k < j; ++k) s -= w->qs[j][k] * w->qs[j][k];
    w->qs[j][j] = sqrt(s);
    for (int i = j + 1; i < n; ++i) {
      double t = q[i][j];
      for (int k = 0; k < j; ++k) t -= w->qs[i][k] * w->qs[j][k];
      w->qs[i][j] = t / w->qs[j][j];
    }
  }
}

static void precond(const iwp_t* w, double dt, double* p, double* pinv) {
  double adt = fabs(dt), pw = sqrt(adt);
  for (int i = w->n - 1; i >= 0; --i) { p[i] = pw / w->fact[i]; pinv[i] = w->fact[i] / pw; pw *= adt; }
}

/* Householder LQ of the row-major n x m matrix A (m >= n), in place: afterwards the lower triangle of the first
 * n columns holds L with L L^T = A A^T (no sign normalisation); this is the transpose of the reference's
 * "R factor of a QR of A^T", laid out so that every inner loop runs over contiguous memory. */
static void lq_inplace(double* A, int n, int m) {
  if (p_dgeqrf) {
    /* the row-major n x m array is the column-major m x n matrix A^T: its QR's R^T is the L of A, in the same memory cells */
    int info = 0, lwork = 64 * n;
    double* tau = (double*)malloc(sizeof(double) * (size_t)(n + lwork));
    p_dgeqrf(&m, &n, A, &m, tau, tau + n, &lwork, &info);
    free(tau);
    return;
  }
  for (int c = 0; c < n; ++c) {
    double* x = A + (size_t)c * m;
    double xx = 0.0;
    for (int q = c + 1; q < m; ++q) xx += x[q] * x[q];
    if (xx == 0.0) continue;
    double alpha = x[c];
    double beta = -copysign(sqrt(alpha * alpha + xx), alpha);
    double tau = (beta - alpha) / beta, scale = 1.0 / (alpha - beta);
    for (int q = c + 1; q < m; ++q) x[q] *= scale;
    x[c] = beta;
    for (int j = c + 1; j < n; ++j) {
      double* y = A + (size_t)j * m;
      double dot = y[c];
      for (int q = c + 1; q < m; ++q) dot += y[q] * x[q];
      dot *= tau;
      y[c] -= dot;
      for (int q = c + 1; q < m; ++q) y[q] -= dot * x[q];
    }
  }
}

static void brusselator_f_jac(int N, const double* y, double* f, double* J /* d x d or NULL */) {
  const int d = 2 * N;
  const double c = (1.0 / 50.0) * (N + 1) * (N + 1);
  if (J) memset(J, 0, sizeof(double) * d * d);
  for (int j = 0; j < N; ++j) {
    double u = y[j], v = y[N + j];
    double ul = j > 0 ? y[j - 1] : 1.0, ur = j < N - 1 ? y[j + 1] : 1.0;
    double vl = j > 0 ? y[N + j - 1] : 3.0, vr = j < N - 1 ? y[N + j + 1] : 3.0;
    f[j] = 1.0 + u * u * v - 4.0 * u + c * (ul - 2.0 * u + ur);
    f[N + j] = 3.0 * u - u * u * v + c * (vl - 2.0 * v + vr);
    if (J) {
      J[j * d + j] = 2.0 * u * v - 4.0 - 2.0 * c;
      J[j * d + N + j] = u * u;
      J[(N + j) * d + j] = 3.0 - 2.0 * u * v;
      J[(N + j) * d + N + j] = -u * u - 2.0 * c;
      if (j > 0) { J[j * d + j - 1] = c; J[(N + j) * d + N + j - 1] = c; }
      if (j < N - 1) { J[j * d + j + 1] = c; J[(N + j) * d + N + j + 1] = c; }
    }
  }
}

typedef struct {
  double* work;  /* (2D x D) QR buffer + D x D + d x D x 2 + d x d x 2 + misc */
} scratch_t;

/* One dense EKF1 (order-1 ODE) step: (mean[n*d], L[D*D] row-major, Sigma = L L^T) -> new mean / L.
 * Returns the dynamic output scale; err[d] receives dt * sigma_hat * sqrt(diag S1). */
double pnoc_dense_step(const iwp_t* w, int N, const double* mean, const double* L, double dt, double* mean_out,
                       double* L_out, double* err, double* work) {
  const int n = w->n, d = w->d, D = w->D;
  double p[MAXN], pinv[MAXN];
  precond(w, dt, p, pinv);
  double* stack = work;                       /* 2D x D */
  double* Lm = stack + (size_t)2 * D * D;     /* D x D : L^- */
  double* J = Lm + (size_t)D * D;             /* d x d */
  double* F = J + (size_t)d * d;              /* d x D */
  double* Ft = F + (size_t)d * D;             /* D x d, also QR buffer for S */
  double* K = Ft + (size_t)D * d;             /* D x d */
  double* f = K + (size_t)D * d;              /* d */
  double* z = f + d;                          /* d */
  double* mext = z + d;                       /* D */
  double* tmp = mext + D;                     /* d */
  /* mean extrapolation */
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < d; ++j) {
      double acc = 0.0;
      for (int k = i; k < n; ++k) acc += w->a[i][k] * pinv[k] * mean[k * d + j];
      mext[i * d + j] = p[i] * acc;
    }
  brusselator_f_jac(N, mext, f, J);
  for (int j = 0; j < d; ++j) z[j] = mext[d + j] - f[j];
  /* S1 = F1 F1^T, F1 = H (P chol(Qbar) (x) I): only columns j = 0, 1 of chol(Qbar) reach rows 0, 1 */
  {
    double lq00 = p[0] * w->qs[0][0], lq10 = p[1] * w->qs[1][0], lq11 = p[1] * w->qs[1][1];
    /* F1 (d x 2d) = [ lq10 I - lq00 J | lq11 I ] -> C1 lower with S1 = C1 C1^T */
    double* Q1 = Ft;
    for (int a = 0; a < d; ++a)
      for (int c = 0; c < d; ++c) {
        Q1[(size_t)a * 2 * d + c] = ((a == c) ? lq10 : 0.0) - lq00 * J[a * d + c];
        Q1[(size_t)a * 2 * d + d + c] = (a == c) ? lq11 : 0.0;
      }
    for (int a = 0; a < d; ++a) {
      double s = 0.0;
      for (int c = 0; c < 2 * d; ++c) s += Q1[(size_t)a * 2 * d + c] * Q1[(size_t)a * 2 * d + c];
      err[a] = sqrt(s);
    }
    lq_inplace(Q1, d, 2 * d);
    /* white = C1^-1 z */
    for (int a = 0; a < d; ++a) {
      double acc = z[a];
      for (int k = 0; k < a; ++k) acc -= Q1[(size_t)a * 2 * d + k] * tmp[k];
      tmp[a] = acc / Q1[(size_t)a * 2 * d + a];
    }
  }
  double ww = 0.0;
  for (int a = 0; a < d; ++a) ww += tmp[a] * tmp[a];
  const double sigma = sqrt(ww / d);
  for (int a = 0; a < d; ++a) err[a] *= dt * sigma;
  /* time update: L^- = lq([ A L~ | sigma chol(Qbar) (x) I ]) in preconditioned coordinates
   * (= transpose of the reference's qr_r of the stacked transposes) */
  for (int r = 0; r < D; ++r) {
    const int i = r / d, q = r % d;
    double* row = stack + (size_t)r * 2 * D;
    memset(row, 0, sizeof(double) * 2 * (size_t)D);
    for (int k = i; k < n; ++k) {  /* (k outermost: unit-stride, vectorisable inner loop) */
      const double ck = w->a[i][k] * pinv[k];
      const double* lrow = L + (size_t)(k * d + q) * D;
      for (int c = 0; c < D; ++c) row[c] += ck * lrow[c];
    }
    for (int ic = 0; ic <= i; ++ic) row[D + ic * d + q] = sigma * w->qs[i][ic];
  }
  lq_inplace(stack, D, 2 * D);
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) Lm[(size_t)r * D + c] = (c <= r) ? p[r / d] * stack[(size_t)r * 2 * D + c] : 0.0;
  /* correction: F = H L^-, S^(T/2) = qr_r(F^T), K = L^- F^T S^-1, m = m^- - K z, L = L^- - K F */
  if (p_dgemm) {
    memcpy(F, Lm + (size_t)d * D, sizeof(double) * (size_t)d * D);
    gemm_rm('N', 'N', d, D, d, -1.0, J, d, Lm, D, 1.0, F, D);
    gemm_rm('N', 'T', D, d, D, 1.0, Lm, D, F, D, 0.0, K, d);
  } else {
  for (int a = 0; a < d; ++a)
    for (int c = 0; c < D; ++c) {
      double acc = Lm[(size_t)(d + a) * D + c];
      for (int q = 0; q < d; ++q) acc -= J[a * d + q] * Lm[(size_t)q * D + c];
      F[(size_t)a * D + c] = acc;
    }
  /* cross = L^- F^T  (D x d) */
  for (int r = 0; r < D; ++r)
    for (int a = 0; a < d; ++a) {
      double acc = 0.0;
      for (int c = 0; c <= r; ++c) acc += Lm[(size_t)r * D + c] * F[(size_t)a * D + c];
      K[(size_t)r * d + a] = acc;
    }
  }
  memcpy(Ft, F, sizeof(double) * (size_t)d * D);
  lq_inplace(Ft, d, D);  /* C_s (d x d lower, leading dimension D): S = C_s C_s^T */
  if (p_dtrsm) {
    /* K^T (column-major d x D) <- C_s^-T C_s^-1 K^T; the row-major lower C_s is the column-major upper U = C_s^T */
    const double one = 1.0;
    p_dtrsm("L", "U", "T", "N", &d, &D, &one, Ft, &D, K, &d);
    p_dtrsm("L", "U", "N", "N", &d, &D, &one, Ft, &D, K, &d);
  } else
  for (int r = 0; r < D; ++r) {  /* K <- cross C_s^-T C_s^-1 */
    double* row = K + (size_t)r * d;
    for (int a = 0; a < d; ++a) {
      double acc = row[a];
      for (int k = 0; k < a; ++k) acc -= row[k] * Ft[(size_t)a * D + k];
      row[a] = acc / Ft[(size_t)a * D + a];
    }
    for (int a = d - 1; a >= 0; --a) {
      double acc = row[a];
      for (int k = a + 1; k < d; ++k) acc -= row[k] * Ft[(size_t)k * D + a];
      row[a] = acc / Ft[(size_t)a * D + a];
    }
  }
  for (int r = 0; r < D; ++r) {
    double acc = mext[r];
    for (int a = 0; a < d; ++a) acc -= K[(size_t)r * d + a] * z[a];
    mean_out[r] = acc;
    if (p_dgemm) continue;
    for (int c = 0; c < D; ++c) {
      double v = Lm[(size_t)r * D + c];
      for (int a = 0; a < d; ++a) v -= K[(size_t)r * d + a] * F[(size_t)a * D + c];
      L_out[(size_t)r * D + c] = v;
    }
  }
  if (p_dgemm) {
    memcpy(L_out, Lm, sizeof(double) * (size_t)D * D);
    gemm_rm('N', 'N', D, D, d, -1.0, K, d, F, D, 1.0, L_out, D);
  }
  return sigma;
}

size_t pnoc_work_doubles(int nu, int d) {
  size_t n = nu + 1, D = n * d;
  return 2 * D * D + D * D + (size_t)d * d + 3 * (size_t)d * D + 3 * (size_t)d + D + 64;
}

/* Single step entry for tests (Brusselator, order 1). */
double pnoc_step_brusselator(int N, int nu, const double* mean, const double* L, double dt, double* mean_out, double* L_out,
                             double* err) {
  iwp_t w;
  iwp_init(&w, nu, 2 * N);
  double* work = (double*)malloc(sizeof(double) * pnoc_work_doubles(nu, 2 * N));
  double s = pnoc_dense_step(&w, N, mean, L, dt, mean_out, L_out, err, work);
  free(work);
  return s;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* Adaptive solves (PI control, accept iff error norm <= 1) of B trajectories, one per OpenMP thread at a time,
 * each running until `seconds` of wall time have passed. tcoeffs: [B][n][d]. Returns elapsed seconds. */
double pnoc_bench_brusselator(int N, int nu, int B, const double* tcoeffs, double tol, double dt0, double seconds,
                              int nthreads, long* accepted_out, long* attempts_out) {
  iwp_t w;
  const int d = 2 * N;
  iwp_init(&w, nu, d);
  const int n = w.n, D = w.D;
  long accepted = 0, attempts = 0;
  const double t_start = now_s();
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : accepted, attempts)
  for (int b = 0; b < B; ++b) {
    double* work = (double*)malloc(sizeof(double) * pnoc_work_doubles(nu, d));
    double* mean = (double*)malloc(sizeof(double) * D);
    double* mnew = (double*)malloc(sizeof(double) * D);
    double* L = (double*)calloc((size_t)D * D, sizeof(double));
    double* Lnew = (double*)malloc(sizeof(double) * (size_t)D * D);
    double* err = (double*)malloc(sizeof(double) * d);
    memcpy(mean, tcoeffs + (size_t)b * n * d, sizeof(double) * D);
    double dt = dt0, err_prev = 1.0;
    const double rate = nu + 1;
    while (now_s() - t_start < seconds) {
      pnoc_dense_step(&w, N, mean, L, dt, mnew, Lnew, err, work);
      ++attempts;
      double acc = 0.0;
      for (int j = 0; j < d; ++j) {
        double rel = err[j] / (tol + tol * fabs(mnew[j]));
        acc += rel * rel;
      }
      const double e = sqrt(acc / d);
      double fac = 0.95 * pow(1.0 / e, 0.3 / rate) * pow(err_prev / e, 0.4 / rate);
      fac = fmax(0.2, fmin(fac, 10.0));
      if (e <= 1.0) {
        err_prev = e;
        double* t1 = mean; mean = mnew; mnew = t1;
        double* t2 = L; L = Lnew; Lnew = t2;
        ++accepted;
      }
      dt *= fac;
    }
    free(work); free(mean); free(mnew); free(L); free(Lnew); free(err);
  }
  *accepted_out = accepted;
  *attempts_out = attempts;
  return now_s() - t_start;
}
