// Dense EKF1/EKF0 stepping kernels (generic path: any d, ode order 1 or 2, any nu).
//
// State per trajectory: mean (n x d, derivative-major) and a D x D factor L (row-major) with
// Sigma = L L^T, D = n*d (impl.select("dense"), /root/reference/experiments/1_van_der_pol/vdp.py:61).
// Everything is done with orthogonal transformations / Gram-of-factor forms, as the reference's
// square-root implementation does (strategy_filter + correction_ts1 + solver_dynamic, vdp.py:62-66):
//   begin   : m^- = A m ; f, J at m^- ; z ; S1 = F1 F1^T with F1 = H (P chol(Qbar) (x) I) ;
//             sigma_hat^2 = z^T S1^{-1} z / d ; err = dt sigma_hat sqrt(diag S1)
//   lq      : [Mt | T] -> L^-  (row Householder LQ; Mt = (Abar P^{-1} (x) I) L, T = sigma chol(Qbar) (x) I)
//             so that L^- L^-T = A Sigma A^T + sigma^2 Q, in preconditioned coordinates
//   correct : F = H L^- ; S = F F^T = C C^T ; W = C^{-1} F ; Y = L^- W^T ;
//             m = m^- - Y C^{-1} z ; L = L^- - Y W   (projector form, no re-triangularisation)
// One workgroup (256 threads) per trajectory; matrices stay in global memory (L2) in this generic
// path. The MFMA-tiled fast path for d % 16 == 0 lives in pnode_dense_mfma.hip.
#include "pnode_device.hpp"

namespace pnode {

constexpr int kThreads = 256;

__device__ __forceinline__ double block_sum(double v, double* red /* >= 4 doubles of LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ void preconditioner_rt(int n, double dt, const IwpConst& K, double* p, double* pinv) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);
  for (int i = n - 1; i >= 0; --i) {
    p[i] = pw / K.fact[i];
    pinv[i] = K.fact[i] / pw;
    pw *= adt;
  }
}

// In-place lower Cholesky of the d x d matrix A (leading dimension ld), whole workgroup.
__device__ void wg_cholesky(double* A, int d, int ld) {
  for (int c = 0; c < d; ++c) {
    __syncthreads();
    const double piv = sqrt(overflow_to_nan(A[c * ld + c]));
    __syncthreads();
    if (threadIdx.x == 0) A[c * ld + c] = piv;
    for (int r = c + 1 + threadIdx.x; r < d; r += kThreads) A[r * ld + c] /= piv;
    __syncthreads();
    const int m = d - c - 1;
    for (int idx = threadIdx.x; idx < m * m; idx += kThreads) {
      const int r = c + 1 + idx / m, q = c + 1 + idx % m;
      if (q <= r) A[r * ld + q] -= A[r * ld + c] * A[q * ld + c];
    }
  }
  __syncthreads();
}

// x <- C^{-1} x for lower-triangular C (d x d), x a vector in LDS/global; whole workgroup.
__device__ void wg_forward_vec(const double* C, int d, int ld, double* x) {
  for (int a = 0; a < d; ++a) {
    __syncthreads();
    const double xa = x[a] / C[a * ld + a];
    __syncthreads();
    if (threadIdx.x == 0) x[a] = xa;
    for (int r = a + 1 + threadIdx.x; r < d; r += kThreads) x[r] -= C[r * ld + a] * xa;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// begin: mean extrapolation, vector field + Jacobian, local calibration and error estimate
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void dense_begin_kernel(DenseView v, const StepArg* __restrict__ args,
                                                               const IwpConst* __restrict__ Kc) {
  extern __shared__ double smem[];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const IwpConst& K = *Kc;
  const int d = v.d, k = v.k, n = v.n, kd = k * d;
  double* ybuf = smem;            // [k][d]
  double* zbuf = ybuf + kd;       // [d]
  double* red = zbuf + d;         // [4]
  double* lq = red + 4;           // [n][n]  P chol(Qbar) (lower)
  double p[kMaxN], pinv[kMaxN];
  preconditioner_rt(n, arg.dt, K, p, pinv);
  for (int idx = threadIdx.x; idx < n * n; idx += kThreads) {
    const int i = idx / n, j = idx % n;
    lq[idx] = (j <= i) ? p[i] * K.qsT[j][i] : 0.0;
  }
  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(arg.src) * sB + b) * n * d;
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * n * d;
  for (int j = threadIdx.x; j < d; j += kThreads) {
    for (int i = 0; i < n; ++i) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * msrc[kk * d + j];
      const double e = p[i] * acc;
      mdst[i * d + j] = e;
      if (v.fixedpoint) v.mext[size_t(b) * n * d + i * d + j] = e;
      if (i < k) ybuf[i * d + j] = e;
    }
  }
  __syncthreads();
  VfCtx ctx{v.vf_id, d, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * k * d : nullptr};
  const double tnew = arg.t + arg.dt;
  double* zg = v.z + size_t(b) * d;
  for (int j = threadIdx.x; j < d; j += kThreads) {
    const double z = mdst[k * d + j] - vf_f(ctx, ybuf, j, tnew);
    zbuf[j] = z;
    zg[j] = z;
  }
  double* Jg = v.J + size_t(b) * d * kd;
  if (v.ts1) {
    for (int idx = threadIdx.x; idx < d * kd; idx += kThreads) {
      const int a = idx / kd, rest = idx % kd;
      Jg[idx] = vf_jac(ctx, ybuf, a, rest / d, rest % d, tnew);
    }
  }
  __syncthreads();
  // S1 = sum_j B_j B_j^T,  B_j = lq[k][j] I - sum_{i<k} lq[i][j] J_i   (Gram of the factor F1)
  double* S = v.S + size_t(b) * d * d;
  double* errg = v.err + size_t(b) * d;
  for (int idx = threadIdx.x; idx < d * d; idx += kThreads) {
    const int a = idx / d, a2 = idx % d;
    if (a2 > a) continue;
    double acc = 0.0;
    if (!v.ts1) {
      if (a == a2)
        for (int j = 0; j < n; ++j) acc += lq[k * n + j] * lq[k * n + j];
    } else {
      for (int j = 0; j <= k; ++j) {  // lq is lower triangular: B_j = 0 for j > k
        const double lk = lq[k * n + j];
        for (int c = 0; c < d; ++c) {
          double ba = (a == c) ? lk : 0.0, ba2 = (a2 == c) ? lk : 0.0;
          for (int i = 0; i < k; ++i) {
            const double l = lq[i * n + j];
            ba -= l * Jg[a * kd + i * d + c];
            ba2 -= l * Jg[a2 * kd + i * d + c];
          }
          acc += ba * ba2;
        }
      }
    }
    S[a * d + a2] = acc;
    if (a == a2) errg[a] = acc;  // diag(S1), scaled below
  }
  __syncthreads();
  wg_cholesky(S, d, d);
  wg_forward_vec(S, d, d, zbuf);
  double part = 0.0;
  for (int j = threadIdx.x; j < d; j += kThreads) part += zbuf[j] * zbuf[j];
  const double ww = block_sum(part, red);
  const double sigma_hat = sqrt(ww / double(d));
  for (int j = threadIdx.x; j < d; j += kThreads) errg[j] = arg.dt * sigma_hat * sqrt(errg[j]);
  if (threadIdx.x == 0)
    v.sig_new[b] = v.dynamic ? sigma_hat : v.sigma[size_t(arg.src) * sB + b];
}

// ---------------------------------------------------------------------------------------------
// LQ input: Mt = (Abar P^{-1} (x) I) L   and   T = sigma chol(Qbar) (x) I   (preconditioned coords)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void dense_lq_input_kernel(DenseView v, const StepArg* __restrict__ args,
                                                                  const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const StepArg arg = args[b];
  if (!arg.active) {
    if (blockIdx.x == 0 && threadIdx.x == 0) v.act[b] = 0;
    return;
  }
  const IwpConst& K = *Kc;
  const int d = v.d, n = v.n, D = v.D;
  double p[kMaxN], pinv[kMaxN];
  preconditioner_rt(n, arg.dt, K, p, pinv);
  const double sigma = v.sig_new[b];
  const double* L = v.U + (size_t(arg.src) * v.B + b) * D * D;
  double* Mt = v.M + size_t(b) * v.mr_rows * D * D;
  double* T = v.R + size_t(b) * v.mr_rows * D * D;
  if (blockIdx.x == 0 && threadIdx.x == 0) v.act[b] = 1;
  const bool tiled = v.u_tiled_now != 0;  // (checkpoint factors of an MFMA-path handle: the state factor is in strip order)
  for (int r = blockIdx.x; r < D; r += gridDim.x) {
    const int i = r / d, q = r % d;
    for (int c = threadIdx.x; c < D; c += kThreads) {
      double acc = 0.0;
      for (int kk = i; kk < n; ++kk)
        acc += K.a[i][kk] * pinv[kk] * (tiled ? tiled_factor_entry(L, kk * d + q, c, D, d) : L[size_t(kk * d + q) * D + c]);
      Mt[size_t(r) * D + c] = acc;
      const int j = c / d, q2 = c % d;
      T[size_t(r) * D + c] = (q2 == q && j <= i) ? sigma * K.qsT[j][i] : 0.0;
      if (v.fixedpoint) {  // extra (never pivoted) rows [P^-1 L | 0]: they receive the backward transition
        Mt[size_t(D + r) * D + c] = pinv[i] * L[size_t(r) * D + c];
        T[size_t(D + r) * D + c] = 0.0;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row-Householder LQ of [T | Mt] (T lower triangular): afterwards T T^T = T0 T0^T + Mt Mt^T.
// Generic, unblocked: step c annihilates row c of Mt against the pivot T[c][c] and applies the
// reflector to the rows below. One wave per row in the update.
// ---------------------------------------------------------------------------------------------
// rows_total >= D: rows D .. rows_total-1 are updated by every reflector but never pivoted.
__global__ __launch_bounds__(kThreads) void dense_lq_kernel(DenseView v, int rows_total) {
  __shared__ double red[4];
  __shared__ double sh_tau, sh_scale;
  const int b = blockIdx.x;
  if (!v.act[b]) return;
  const int D = v.D;
  double* Mt = v.M + size_t(b) * v.mr_rows * D * D;
  double* T = v.R + size_t(b) * v.mr_rows * D * D;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = 0; c < D; ++c) {
    double* x = Mt + size_t(c) * D;
    double part = 0.0;
    for (int q = threadIdx.x; q < D; q += kThreads) part += x[q] * x[q];
    const double xx = block_sum(part, red);
    if (threadIdx.x == 0) {
      const double alpha = T[size_t(c) * D + c];
      double tau = 0.0, scale = 0.0;
      if (xx != 0.0) {
        const double beta = -copysign(sqrt(alpha * alpha + xx), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
        T[size_t(c) * D + c] = beta;
      }
      sh_tau = tau;
      sh_scale = scale;
    }
    __syncthreads();
    const double tau = sh_tau, scale = sh_scale;
    if (tau != 0.0) {
      // v = [1 ; scale * x]; rows j > c:  w = tau (T[j][c] + scale * Mt[j].x)
      for (int j = c + 1 + wv; j < rows_total; j += 4) {
        double* y = Mt + size_t(j) * D;
        double dot = 0.0;
        for (int q = lane; q < D; q += kWave) dot += y[q] * x[q];
        dot = wave_sum(dot);
        const double tjc = T[size_t(j) * D + c];
        const double w = tau * (tjc + scale * dot);
        const double ws = w * scale;
        for (int q = lane; q < D; q += kWave) y[q] -= ws * x[q];
        if (lane == 0) T[size_t(j) * D + c] = tjc - w;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// correct: measurement update on z = 0 in projector form + normalised error norm
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void dense_correct_kernel(DenseView v, const StepArg* __restrict__ args,
                                                                 const IwpConst* __restrict__ Kc) {
  extern __shared__ double smem[];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const IwpConst& K = *Kc;
  const int d = v.d, k = v.k, n = v.n, D = v.D, kd = k * d;
  double* wz = smem;       // [d]
  double* red = wz + d;    // [4]
  double p[kMaxN], pinv[kMaxN];
  preconditioner_rt(n, arg.dt, K, p, pinv);
  const double* T = v.R + size_t(b) * v.mr_rows * D * D;   // L^- in preconditioned coordinates (lower)
  const double* Jg = v.J + size_t(b) * d * kd;
  double* F = v.Ft + size_t(b) * d * D;        // F, then W  (d x D)
  double* Y = v.X + size_t(b) * D * d;         // Y = L^- W^T (D x d)
  double* S = v.S + size_t(b) * d * d;
  // F = H L^-  with L^-[r] = p[r / d] * T[r]
  for (int idx = threadIdx.x; idx < d * D; idx += kThreads) {
    const int a = idx / D, c = idx % D;
    double acc = p[k] * T[size_t(k * d + a) * D + c];
    if (v.ts1) {
      for (int i = 0; i < k; ++i) {
        const double pi = p[i];
        for (int q = 0; q < d; ++q) acc -= Jg[a * kd + i * d + q] * pi * T[size_t(i * d + q) * D + c];
      }
    }
    F[idx] = acc;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < d * d; idx += kThreads) {
    const int a = idx / d, a2 = idx % d;
    if (a2 > a) continue;
    double acc = 0.0;
    for (int c = 0; c < D; ++c) acc += F[size_t(a) * D + c] * F[size_t(a2) * D + c];
    S[idx] = acc;
  }
  __syncthreads();
  wg_cholesky(S, d, d);
  // W = C^{-1} F (rows), wz = C^{-1} z
  for (int j = threadIdx.x; j < d; j += kThreads) wz[j] = v.z[size_t(b) * d + j];
  wg_forward_vec(S, d, d, wz);
  for (int a = 0; a < d; ++a) {
    const double inv = 1.0 / S[a * d + a];
    for (int c = threadIdx.x; c < D; c += kThreads) F[size_t(a) * D + c] *= inv;
    __syncthreads();
    const int m = d - a - 1;
    for (int idx = threadIdx.x; idx < m * D; idx += kThreads) {
      const int r = a + 1 + idx / D, c = idx % D;
      F[size_t(r) * D + c] -= S[r * d + a] * F[size_t(a) * D + c];
    }
    __syncthreads();
  }
  // Y = L^- W^T
  for (int idx = threadIdx.x; idx < D * d; idx += kThreads) {
    const int r = idx / d, a = idx % d;
    const double pr = p[r / d];
    double acc = 0.0;
    for (int c = 0; c <= r; ++c) acc += T[size_t(r) * D + c] * F[size_t(a) * D + c];
    Y[idx] = pr * acc;
  }
  __syncthreads();
  const size_t sB = size_t(v.B);
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * n * d;
  for (int r = threadIdx.x; r < D; r += kThreads) {
    double acc = mdst[r];
    for (int a = 0; a < d; ++a) acc -= Y[size_t(r) * d + a] * wz[a];
    mdst[r] = acc;
  }
  double* Ln = v.U + (size_t(arg.dst) * sB + b) * D * D;
  for (int idx = threadIdx.x; idx < D * D; idx += kThreads) {
    const int r = idx / D, c = idx % D;
    double acc = (c <= r) ? p[r / d] * T[idx] : 0.0;
    for (int a = 0; a < d; ++a) acc -= Y[size_t(r) * d + a] * F[size_t(a) * D + c];
    Ln[idx] = acc;
  }
  __syncthreads();
  const double atol = v.atol[b], rtol = v.rtol[b];
  double part = 0.0;
  for (int j = threadIdx.x; j < d; j += kThreads) {
    const double rel = v.err[size_t(b) * d + j] / (atol + rtol * fabs(mdst[j]));
    part += rel * rel;
  }
  const double tot = block_sum(part, red);
  if (threadIdx.x == 0) {
    v.errnorm[b] = sqrt(tot / double(d));
    v.sigma[size_t(arg.dst) * sB + b] = v.sig_new[b];
  }
}

// Covariance factors at the save_at points of the dense FILTER (solution.posterior of the reference carries one Cholesky factor
// per checkpoint: /root/reference/src/odecheckpts/ivpsolvers.py:71-81, experiments/4_brusselator/run.py:136-138).
//   prepare: sig_new <- output scale of STEP_FROM (the scale the interpolation uses);
//   store  : at_step -> copy L of STEP_FROM (de-tiled for the MFMA path); otherwise L^-(t) = diag(p) T with T the LQ factor of
//            [sigma chol(Qbar) (x) I | (Abar P^-1 (x) I) L(INTERP_FROM)] that dense_lq_input_kernel + dense_lq_kernel left in v.R.
__global__ void dense_ckfac_prepare_kernel(DenseView v, const InterpArg* __restrict__ args) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= v.B) return;
  const InterpArg a = args[b];
  if (a.ckpt >= 0 && !a.at_step) v.sig_new[b] = v.sigma[size_t(a.s1) * v.B + b];
}

__global__ __launch_bounds__(kThreads) void dense_ckfac_store_kernel(DenseView v, const InterpArg* __restrict__ args,
                                                                     const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.y;
  const InterpArg a = args[b];
  if (a.ckpt < 0) return;
  const int d = v.d, n = v.n, D = v.D;
  double* out = v.ck_fac + (size_t(b) * v.K + a.ckpt) * D * D;
  if (a.at_step) {
    const double* L = v.U + (size_t(a.s1) * v.B + b) * D * D;
    const bool tiled = v.u_tiled_now != 0;
    for (int r = blockIdx.x; r < D; r += gridDim.x)
      for (int c = threadIdx.x; c < D; c += kThreads) out[size_t(r) * D + c] = tiled ? tiled_factor_entry(L, r, c, D, d) : L[size_t(r) * D + c];
    return;
  }
  double p[kMaxN], pinv[kMaxN];
  preconditioner_rt(n, a.t - a.t0, *Kc, p, pinv);
  const double* T = v.R + size_t(b) * v.mr_rows * D * D;
  for (int r = blockIdx.x; r < D; r += gridDim.x)
    for (int c = threadIdx.x; c < D; c += kThreads) out[size_t(r) * D + c] = (c <= r) ? p[r / d] * T[size_t(r) * D + c] : 0.0;
}

// Filter checkpoint: the solution at t is the prediction of the mean from INTERP_FROM (Chapman-
// Kolmogorov: predicting t0 -> t -> t' equals t0 -> t', so INTERP_FROM itself is left untouched).
__global__ __launch_bounds__(kThreads) void dense_interp_kernel(DenseView v, const InterpArg* __restrict__ args,
                                                                const IwpConst* __restrict__ Kc) {
  const int b = blockIdx.x;
  const InterpArg arg = args[b];
  if (arg.ckpt < 0) return;
  const IwpConst& K = *Kc;
  const int d = v.d, n = v.n;
  const size_t sB = size_t(v.B);
  double* cm = v.ck_mean + (size_t(b) * v.K + arg.ckpt) * n * d;
  if (arg.at_step) {
    const double* m1 = v.mean + (size_t(arg.s1) * sB + b) * n * d;
    for (int idx = threadIdx.x; idx < n * d; idx += kThreads) cm[idx] = m1[idx];
  } else {
    double p[kMaxN], pinv[kMaxN];
    preconditioner_rt(n, arg.t - arg.t0, K, p, pinv);
    const double* m0 = v.mean + (size_t(arg.s0) * sB + b) * n * d;
    for (int j = threadIdx.x; j < d; j += kThreads)
      for (int i = 0; i < n; ++i) {
        double acc = 0.0;
        for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * pinv[kk] * m0[kk * d + j];
        cm[i * d + j] = p[i] * acc;
      }
  }
  if (threadIdx.x == 0) v.ck_steps[size_t(b) * v.K + arg.ckpt] = arg.steps;
}

// The handles whose steps run on the register-resident MFMA kernels keep the state factor in the register order of the
// 16-row strips (pnode_mfma_common.hpp); every other path stores L row-major. Same predicate as the dispatch below.
bool dense_state_factor_tiled(const DenseView& v) {
  if (dense_small_supported(v) && !v.force_generic) return false;
  if (dense_use_large(v)) return false;
  return dense_mfma_supported(v) && !v.force_generic;
}

// The register-resident MFMA kernels of the EKF1: their measurement-update kernel can carry the controller's decision.
bool dense_fused_control_supported(const DenseView& v) {
  if (dense_small_supported(v) && !v.force_generic) return false;
  if (dense_use_large(v)) return false;
  return dense_mfma_supported(v) && !v.force_generic && v.ts1 && !v.fixedpoint;
}

void dense_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s,
                       const PhaseEvents* ev, const FusedControl* fc) {
  auto mark = [&](int i) {
    if (ev) (void)hipEventRecord(ev->ev[i], s);
  };
  mark(0);
  if (dense_small_supported(v) && !v.force_generic) {
    dense_small_launch_step(v, args, K, s);
    mark(1); mark(2); mark(3); mark(4);
    return;
  }
  const bool fast = dense_mfma_supported(v) && !v.force_generic;
  if (dense_use_large(v)) {
    dense_large_launch_step(v, args, K, s, ev);
    return;
  }
  const size_t shm_begin = (size_t(v.k) * v.d + v.d + 4 + size_t(v.n) * v.n) * sizeof(double);
  if (fast && v.ts1) {
    dense_launch_mfma_begin(v, args, K, s);
  } else
    hipLaunchKernelGGL(dense_begin_kernel, dim3(v.B), dim3(kThreads), shm_begin, s, v, args, K);
  mark(1);
  if (fast) {
    dense_launch_mfma_tail(v, args, K, s, ev, (fc && v.ts1) ? fc : nullptr);
    return;
  }
  const int rows_per = v.D < 64 ? v.D : 64;
  hipLaunchKernelGGL(dense_lq_input_kernel, dim3(rows_per, v.B), dim3(kThreads), 0, s, v, args, K);
  mark(2);
  hipLaunchKernelGGL(dense_lq_kernel, dim3(v.B), dim3(kThreads), 0, s, v, v.fixedpoint ? 2 * v.D : v.D);
  mark(3);
  const size_t shm_corr = (size_t(v.d) + 4) * sizeof(double);
  hipLaunchKernelGGL(dense_correct_kernel, dim3(v.B), dim3(kThreads), shm_corr, s, v, args, K);
  mark(4);
}

void dense_launch_lq_input(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  const int rows_per = v.D < 64 ? v.D : 64;
  hipLaunchKernelGGL(dense_lq_input_kernel, dim3(rows_per, v.B), dim3(kThreads), 0, s, v, args, K);
}

void dense_launch_lq_generic(const DenseView& v, int rows_total, hipStream_t s) {
  hipLaunchKernelGGL(dense_lq_kernel, dim3(v.B), dim3(kThreads), 0, s, v, rows_total);
}

void dense_launch_interp(const DenseView& v, const InterpArg* args, const IwpConst* K, hipStream_t s) {
  hipLaunchKernelGGL(dense_interp_kernel, dim3(v.B), dim3(kThreads), 0, s, v, args, K);
}

// `pargs`: per trajectory {dt = t - t0, src = INTERP_FROM slot, active = interpolated checkpoint} (built by the caller).
void dense_launch_checkpoint_factors(const DenseView& v_, const InterpArg* args, const StepArg* pargs, const IwpConst* K, hipStream_t s) {
  DenseView v = v_;
  v.u_tiled_now = dense_state_factor_tiled(v_) ? 1 : 0;
  hipLaunchKernelGGL(dense_ckfac_prepare_kernel, dim3((v.B + 255) / 256), dim3(256), 0, s, v, args);
  const int rows_per = v.D < 64 ? v.D : 64;
  hipLaunchKernelGGL(dense_lq_input_kernel, dim3(rows_per, v.B), dim3(kThreads), 0, s, v, pargs, K);
  hipLaunchKernelGGL(dense_lq_kernel, dim3(v.B), dim3(kThreads), 0, s, v, v.D);
  hipLaunchKernelGGL(dense_ckfac_store_kernel, dim3(rows_per, v.B), dim3(kThreads), 0, s, v, args, K);
}

}  // namespace pnode
