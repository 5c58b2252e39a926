// Dense EKF1/EKF0 for SMALL state dimensions (D = n*d <= 32): one wavefront per trajectory, the whole attempted
// step fused in ONE kernel with every matrix in LDS (the north star's "stage the (nu+1)d x (nu+1)d blocks in LDS").
// This is the shape of the reference's own dense run — stiff Van der Pol, d = 1, nu = 4, D = 5
// (/root/reference/experiments/1_van_der_pol/vdp.py:61-66) — and of BASELINE config 2 (tolerance sweep, B = 256).
// Same algorithm as pnode_dense.hip (generic path): row-Householder LQ time update, Gram-of-factor innovation,
// projector-form measurement update. Lane j owns row j in the O(D^3) loops, so there are no cross-lane reductions
// except the pivot-row norm. Also provides the device-resident controller loop (dense_small_advance_kernel).
#include "pnode_lanes.hpp"
#include "pnode_vf.hpp"

namespace pnode {

struct SmallLds {
  double *L, *Mt, *T, *F, *J, *S, *mext, *z, *wz, *errv, *y, *lq, *p, *pinv, *tmp;
};

__host__ __device__ inline size_t small_lds_doubles(int D, int d, int k, int n) {
  const int LD = D + 1;
  return size_t(3) * D * LD + size_t(d) * LD + size_t(d) * k * d + size_t(d) * (d + 1) + D + 4 * size_t(d) + size_t(k) * d +
         size_t(n) * n + 2 * size_t(n) + 8;
}

__device__ __forceinline__ SmallLds small_carve(double* base, int D, int d, int k, int n) {
  const int LD = D + 1;
  SmallLds s;
  s.L = base;
  s.Mt = s.L + D * LD;
  s.T = s.Mt + D * LD;
  s.F = s.T + D * LD;
  s.J = s.F + d * LD;
  s.S = s.J + d * k * d;
  s.mext = s.S + d * (d + 1);
  s.z = s.mext + D;
  s.wz = s.z + d;
  s.errv = s.wz + d;
  s.tmp = s.errv + d;
  s.y = s.tmp + d;
  s.lq = s.y + k * d;
  s.p = s.lq + n * n;
  s.pinv = s.p + n;
  return s;
}

__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------
// d = 1 (a scalar ODE of order k <= 2: the reference's stiff Van der Pol run, vdp.py:61-66, BASELINE config 2): the dense model
// is the isotropic one with one column, the EKF1 observation row is h = e_k - [J_0 .. J_{k-1}], and everything is n x n --
// no LDS, no wavefront barrier (the general path below: about 40 barriers and 16 k cycles per attempt at D = 5).
// r04: the n x n algebra in the lane layout of pnode_lanes.hpp (one column of R = L^T per lane, Householder time update: n
// reciprocal square roots per attempt where the Givens form of r03 took n^2); the mean, the observation row and every scalar
// are wave-uniform. Measurement update in projector form; calibration from s1^2 = h (P Qbar P) h^T. The state factor is
// stored as L = R^T (lower triangular, non-negative diagonal), row-major.
// ---------------------------------------------------------------------------------------------
template <int N>
struct D1Prior {
  double arow[N], qcol[N], qrow[N], fact;  // per lane l: row l of Abar, column l and row l of chol(Qbar)^T, (nu - l)!
};
template <int N>
__device__ __forceinline__ void d1_load_prior(const IwpConst* __restrict__ Kc, D1Prior<N>& P, int l) {
  ln::load_prior<N>(Kc, P.arow, P.qcol, l);
  ln::load_rows<N, kMaxN>(&Kc->qsT[0][0], P.qrow, l);
  P.fact = Kc->fact[l];
}

// p_i = dt^(nu-i+1/2) / (nu-i)!, pinv_i = 1 / p_i: lane i forms its own pair, the uniform arrays come from row broadcasts
// (the multiplication sequence of preconditioner(), pnode_common.hpp)
template <int N>
__device__ __forceinline__ void d1_preconditioner(double dt, double fact, int l, double (&p)[N], double (&pinv)[N], double& pl, double& pil) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);
#pragma unroll
  for (int s = 0; s < N - 1; ++s) pw = (N - 1 - l > s) ? pw * adt : pw;
  pl = pw / fact;
  pil = fact / pw;
  ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    p[i] = ln::bcast<i>(pl);
    pinv[i] = ln::bcast<i>(pil);
  });
}

// state in: m (uniform), R (= L^T, col-lane), sigma_prev; out: mnew, Rnew, sigma_out; returns the normalised error norm (uniform)
template <int N>
__device__ __forceinline__ double dense_d1_core(const DenseView& v, const IwpConst& K, const D1Prior<N>& P, int b, int l, double dt, double t,
                                                const double (&m)[N], const double (&Rprev)[N], double sigma_prev, double (&mnew)[N],
                                                double (&Rnew)[N], double& sigma_out, double atol, double rtol) {
  const int k = v.k;
  double p[N], pinv[N], pl, pil;
  d1_preconditioner<N>(dt, P.fact, l, p, pinv, pl, pil);
  // m^- = P Abar P^-1 m
  double me[N];
  {
    double mp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) mp[i] = pinv[i] * m[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * mp[kk];
      me[i] = p[i] * acc;
    }
  }
  // residual and observation row at y = m^-[0 .. k-1]
  double y[2] = {me[0], me[1]};
  VfCtx ctx{v.vf_id, 1, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) : nullptr, v.user_J ? v.user_J + size_t(b) * k : nullptr};
  const double tnew = t + dt;
  double mek = me[1];
#pragma unroll
  for (int i = 2; i < N; ++i) mek = (i == k) ? me[i] : mek;
  const double z = mek - vf_f(ctx, y, 0, tnew);
  double h[N];
#pragma unroll
  for (int i = 0; i < N; ++i) h[i] = (i == k) ? 1.0 : 0.0;
  if (v.ts1) {
    h[0] = -vf_jac(ctx, y, 0, 0, 0, tnew);
    if (k > 1) h[1] = -vf_jac(ctx, y, 0, 1, 0, tnew);
  }
  // calibration + error estimate: s1^2 = h (P Qbar P) h^T = sum_c (sum_i chol(Qbar)^T[c][i] p_i h_i)^2, lane c forms term c
  double s1sq = 0.0;
  {
    double b0 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) b0 += P.qrow[i] * (p[i] * h[i]);
    ln::sfor<0, N>([&](auto cc) PNODE_LAMBDA_INLINE {
      constexpr int c = decltype(cc)::value;
      const double bc = ln::bcast<c>(b0);
      s1sq += bc * bc;
    });
  }
  const double s1 = sqrt(overflow_to_nan(s1sq));
  const double sigma_hat = fabs(z) / s1;
  const double err = dt * sigma_hat * s1;
  const double sigma = v.dynamic ? sigma_hat : sigma_prev;
  double Gu[N], RLu[N];
  ln::time_update<N, false>(P.arow, P.qcol, Rprev, pinv, pl, pil, sigma, l, Rnew, Gu, RLu);
  // measurement update on z = 0: F = h L^- = (Rm h)^T, gain = Sigma^- h^T / (F F^T), projector form
  double hl = h[0];  // my lane's entry of h
#pragma unroll
  for (int i = 1; i < N; ++i) hl = (l == i) ? h[i] : hl;
  double F[N];
  double s2 = 0.0;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const double tc = Rnew[c] * hl;
    double f = 0.0;
    ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
      constexpr int i = decltype(ic)::value;
      f += ln::bcast<i>(tc);
    });
    F[c] = f;
    s2 += f * f;
  }
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < N; ++c) acc += Rnew[c] * F[c];
  const double gain = acc / overflow_to_nan(s2);  // lane i: gain_i
  ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    mnew[i] = me[i] - ln::bcast<i>(gain) * z;
  });
#pragma unroll
  for (int c = 0; c < N; ++c) Rnew[c] -= F[c] * gain;
  sigma_out = sigma;
  const double rel = err / (atol + rtol * fabs(mnew[0]));
  return ln::uniform(fabs(rel));
}

template <int N>
__device__ __forceinline__ void dense_d1_load(const DenseView& v, int b, int l, int slot, double (&m)[N], double (&R)[N], double& sigma) {
  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(slot) * sB + b) * N;
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = msrc[i];
  ln::load_rows<N>(v.U + (size_t(slot) * sB + b) * N * N, R, l);  // lane l: row l of L = column l of R = L^T
  sigma = v.sigma[size_t(slot) * sB + b];
}

template <int N>
__device__ __forceinline__ void dense_d1_store(const DenseView& v, int b, int l, int slot, const double (&m)[N], const double (&R)[N], double sigma) {
  const size_t sB = size_t(v.B);
  double* mdst = v.mean + (size_t(slot) * sB + b) * N;
#pragma unroll
  for (int i = 0; i < N; ++i) mdst[i] = m[i];  // (uniform values: every lane writes them)
  ln::store_rows<N>(v.U + (size_t(slot) * sB + b) * N * N, R, l);
  v.sigma[size_t(slot) * sB + b] = sigma;
}

template <int N>
__device__ __forceinline__ double dense_d1_attempt(const DenseView& v, const IwpConst* __restrict__ Kc, int b, int lane, const StepArg& arg) {
  const int l = ln::col_index<N>(lane);
  D1Prior<N> P;
  d1_load_prior<N>(Kc, P, l);
  double m[N], R[N], sig, mn[N], Rn[N], sign;
  dense_d1_load<N>(v, b, l, arg.src, m, R, sig);
  const double e = dense_d1_core<N>(v, *Kc, P, b, l, arg.dt, arg.t, m, R, sig, mn, Rn, sign, v.atol[b], v.rtol[b]);
  dense_d1_store<N>(v, b, l, arg.dst, mn, Rn, sign);
  return e;
}

// One attempted step by one wavefront; returns the normalised error norm (wave-uniform).
// (WITH_D1 = false: the caller never sees a scalar ODE -- the device-resident controller has its own kernel for it -- and does not
// carry that code.)
template <bool WITH_D1>
__device__ double dense_small_attempt(const DenseView& v, const IwpConst& K, int b, int lane, const StepArg& arg, double* lds) {
  if (WITH_D1 && v.d == 1 && v.k <= 2 && v.n >= 3) {  // scalar ODE: n x n in registers (above)
    switch (v.n) {
      case 3: return dense_d1_attempt<3>(v, &K, b, lane, arg);
      case 4: return dense_d1_attempt<4>(v, &K, b, lane, arg);
      case 5: return dense_d1_attempt<5>(v, &K, b, lane, arg);
      case 6: return dense_d1_attempt<6>(v, &K, b, lane, arg);
      case 7: return dense_d1_attempt<7>(v, &K, b, lane, arg);
      case 8: return dense_d1_attempt<8>(v, &K, b, lane, arg);
      default: break;
    }
  }
  const int d = v.d, k = v.k, n = v.n, D = v.D, kd = k * d, LD = D + 1, LS = d + 1;
  SmallLds s = small_carve(lds, D, d, k, n);
  const size_t sB = size_t(v.B);
  // preconditioner + P chol(Qbar)
  if (lane == 0) {
    const double adt = fabs(arg.dt);
    double pw = sqrt(adt);
    for (int i = n - 1; i >= 0; --i) {
      s.p[i] = pw / K.fact[i];
      s.pinv[i] = K.fact[i] / pw;
      pw *= adt;
    }
  }
  wsync();
  for (int idx = lane; idx < n * n; idx += kWave) {
    const int i = idx / n, j = idx % n;
    s.lq[idx] = (j <= i) ? s.p[i] * K.qsT[j][i] : 0.0;
  }
  const double* msrc = v.mean + (size_t(arg.src) * sB + b) * D;
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * D;
  const double* Lsrc = v.U + (size_t(arg.src) * sB + b) * D * D;
  for (int idx = lane; idx < D * D; idx += kWave) s.L[(idx / D) * LD + idx % D] = Lsrc[idx];
  for (int r = lane; r < D; r += kWave) {
    const int i = r / d, j = r % d;
    double acc = 0.0;
    for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * s.pinv[kk] * msrc[kk * d + j];
    const double e = s.p[i] * acc;
    s.mext[r] = e;
    if (i < k) s.y[r] = e;
  }
  wsync();
  VfCtx ctx{v.vf_id, d, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * k * d : nullptr};
  const double tnew = arg.t + arg.dt;
  for (int j = lane; j < d; j += kWave) {
    const double z = s.mext[k * d + j] - vf_f(ctx, s.y, j, tnew);
    s.z[j] = z;
    s.tmp[j] = z;
  }
  if (v.ts1)
    for (int idx = lane; idx < d * kd; idx += kWave) s.J[idx] = vf_jac(ctx, s.y, idx / kd, (idx % kd) / d, idx % d, tnew);
  wsync();
  // S1 = sum_{j <= k} B_j B_j^T,  B_j = lq[k][j] I - sum_{i<k} lq[i][j] J_i
  for (int idx = lane; idx < d * d; idx += kWave) {
    const int a = idx / d, a2 = idx % d;
    if (a2 > a) continue;
    double acc = 0.0;
    for (int j = 0; j <= k; ++j) {
      const double lk = s.lq[k * n + j];
      if (!v.ts1) {
        if (a == a2) acc += lk * lk;
        continue;
      }
      for (int c = 0; c < d; ++c) {
        double ba = (a == c) ? lk : 0.0, ba2 = (a2 == c) ? lk : 0.0;
        for (int i = 0; i < k; ++i) {
          const double l = s.lq[i * n + j];
          ba -= l * s.J[a * kd + i * d + c];
          ba2 -= l * s.J[a2 * kd + i * d + c];
        }
        acc += ba * ba2;
      }
    }
    s.S[a * LS + a2] = acc;
    if (a == a2) s.errv[a] = acc;
  }
  wsync();
  // Cholesky of S1 (lower, in place) + forward solve of z; lane r owns row r
  for (int c = 0; c < d; ++c) {
    const double piv = sqrt(overflow_to_nan(s.S[c * LS + c]));
    wsync();
    for (int r = c + lane; r < d; r += kWave) s.S[r * LS + c] = (r == c) ? piv : s.S[r * LS + c] / piv;
    wsync();
    for (int r = c + 1 + lane; r < d; r += kWave)
      for (int q = c + 1; q <= r; ++q) s.S[r * LS + q] -= s.S[r * LS + c] * s.S[q * LS + c];
    wsync();
  }
  for (int a = 0; a < d; ++a) {
    const double xa = s.tmp[a] / s.S[a * LS + a];
    wsync();
    if (lane == 0) s.tmp[a] = xa;
    for (int r = a + 1 + lane; r < d; r += kWave) s.tmp[r] -= s.S[r * LS + a] * xa;
    wsync();
  }
  double ww = 0.0;
  for (int j = lane; j < d; j += kWave) ww += s.tmp[j] * s.tmp[j];
  ww = wave_sum(ww);
  const double sigma_hat = sqrt(ww / double(d));
  const double sigma = v.dynamic ? sigma_hat : v.sigma[size_t(arg.src) * sB + b];
  for (int j = lane; j < d; j += kWave) s.errv[j] = arg.dt * sigma_hat * sqrt(s.errv[j]);
  // LQ input: Mt = (Abar P^-1 (x) I) L ,  T = sigma chol(Qbar) (x) I
  for (int idx = lane; idx < D * D; idx += kWave) {
    const int r = idx / D, c = idx % D;
    const int i = r / d, q = r % d;
    double acc = 0.0;
    for (int kk = i; kk < n; ++kk) acc += K.a[i][kk] * s.pinv[kk] * s.L[(kk * d + q) * LD + c];
    s.Mt[r * LD + c] = acc;
    const int j = c / d, q2 = c % d;
    s.T[r * LD + c] = (q2 == q && j <= i) ? sigma * K.qsT[j][i] : 0.0;
  }
  wsync();
  // row-Householder LQ of [T | Mt]; lane j owns row j in the update
  for (int c = 0; c < D; ++c) {
    double part = 0.0;
    for (int q = lane; q < D; q += kWave) part += s.Mt[c * LD + q] * s.Mt[c * LD + q];
    const double xx = wave_sum(part);
    if (xx != 0.0) {
      const double alpha = s.T[c * LD + c];
      const double beta = -copysign(sqrt(alpha * alpha + xx), alpha);
      const double tau = (beta - alpha) / beta, scale = 1.0 / (alpha - beta);
      for (int j = c + 1 + lane; j < D; j += kWave) {
        double dot = 0.0;
        for (int q = 0; q < D; ++q) dot += s.Mt[j * LD + q] * s.Mt[c * LD + q];
        const double tjc = s.T[j * LD + c];
        const double w = tau * (tjc + scale * dot);
        const double ws = w * scale;
        for (int q = 0; q < D; ++q) s.Mt[j * LD + q] -= ws * s.Mt[c * LD + q];
        s.T[j * LD + c] = tjc - w;
      }
      if (lane == 0) s.T[c * LD + c] = beta;
    }
    wsync();
  }
  // L^- = diag(p) T  (into s.L), F = H L^-
  for (int idx = lane; idx < D * D; idx += kWave) {
    const int r = idx / D, c = idx % D;
    s.L[r * LD + c] = (c <= r) ? s.p[r / d] * s.T[r * LD + c] : 0.0;
  }
  wsync();
  for (int idx = lane; idx < d * D; idx += kWave) {
    const int a = idx / D, c = idx % D;
    double acc = s.L[(k * d + a) * LD + c];
    if (v.ts1)
      for (int i = 0; i < k; ++i)
        for (int q = 0; q < d; ++q) acc -= s.J[a * kd + i * d + q] * s.L[(i * d + q) * LD + c];
    s.F[a * LD + c] = acc;
  }
  wsync();
  for (int idx = lane; idx < d * d; idx += kWave) {
    const int a = idx / d, a2 = idx % d;
    if (a2 > a) continue;
    double acc = 0.0;
    for (int c = 0; c < D; ++c) acc += s.F[a * LD + c] * s.F[a2 * LD + c];
    s.S[a * LS + a2] = acc;
  }
  for (int j = lane; j < d; j += kWave) s.wz[j] = s.z[j];
  wsync();
  for (int c = 0; c < d; ++c) {
    const double piv = sqrt(overflow_to_nan(s.S[c * LS + c]));
    wsync();
    for (int r = c + lane; r < d; r += kWave) s.S[r * LS + c] = (r == c) ? piv : s.S[r * LS + c] / piv;
    wsync();
    for (int r = c + 1 + lane; r < d; r += kWave)
      for (int q = c + 1; q <= r; ++q) s.S[r * LS + q] -= s.S[r * LS + c] * s.S[q * LS + c];
    wsync();
  }
  // W = C^-1 F (lane = column), wz = C^-1 z (lane D)
  for (int c = lane; c < D + 1; c += kWave) {
    double* col = (c < D) ? (s.F + c) : s.wz;
    const int ld = (c < D) ? LD : 1;
    for (int a = 0; a < d; ++a) {
      const double w = col[a * ld] / s.S[a * LS + a];
      col[a * ld] = w;
      for (int a2 = a + 1; a2 < d; ++a2) col[a2 * ld] -= s.S[a2 * LS + a] * w;
    }
  }
  wsync();
  // Y = L^- W^T (row per lane, kept in Mt[:, 0:d]); mean; L_new
  for (int r = lane; r < D; r += kWave) {
    double mnew = s.mext[r];
    for (int a = 0; a < d; ++a) {
      double acc = 0.0;
      for (int c = 0; c <= r; ++c) acc += s.L[r * LD + c] * s.F[a * LD + c];
      s.Mt[r * LD + a] = acc;
      mnew -= acc * s.wz[a];
    }
    mdst[r] = mnew;
    s.mext[r] = mnew;
  }
  wsync();
  double* Ln = v.U + (size_t(arg.dst) * sB + b) * D * D;
  for (int idx = lane; idx < D * D; idx += kWave) {
    const int r = idx / D, c = idx % D;
    double acc = s.L[r * LD + c];
    for (int a = 0; a < d; ++a) acc -= s.Mt[r * LD + a] * s.F[a * LD + c];
    Ln[idx] = acc;
  }
  const double atol = v.atol[b], rtol = v.rtol[b];
  double part = 0.0;
  for (int j = lane; j < d; j += kWave) {
    const double rel = s.errv[j] / (atol + rtol * fabs(s.mext[j]));
    part += rel * rel;
  }
  part = wave_sum(part);
  if (lane == 0) v.sigma[size_t(arg.dst) * sB + b] = sigma;
  return sqrt(part / double(d));
}

__global__ __launch_bounds__(64) void dense_small_step_kernel(DenseView v, const StepArg* __restrict__ args,
                                                              const IwpConst* __restrict__ Kc) {
  extern __shared__ double smem[];
  const int b = blockIdx.x;
  const StepArg arg = args[b];
  if (!arg.active) return;
  const double e = dense_small_attempt<true>(v, *Kc, b, threadIdx.x, arg, smem);
  if (threadIdx.x == 0) v.errnorm[b] = e;
}

// The controller's bookkeeping after an attempt with error norm e and step dt: returns whether the attempt is accepted; c.status = 2
// ends the loop (same arithmetic as the host loop of pnode_solve_adaptive_save_at).
__device__ __forceinline__ bool small_after_attempt(TrajCtl& c, const ControlParams& cp, double e, double dt) {
  c.attempts += 1;
  bool accept;
  if (e != e) {
    c.status = 2;
    accept = true;
  } else {
    const double le = ln::uniform(log(e));
    const double raw = cp.safety * exp(cp.n2 * c.log_prev - (cp.n1 + cp.n2) * le);
    const double factor = fmax(cp.factor_min, fmin(raw, cp.factor_max));
    if (e <= 1.0) c.log_prev = le;
    c.dt = ln::uniform(factor * dt);
    accept = !(e > 1.0);
  }
  if (accept) {
    c.s_if = c.s_sf;
    c.t_if = c.t_sf;
    c.s_sf = c.s_pr;
    c.t_sf = ln::uniform(c.t_sf + dt);
    c.steps += 1;
    int fr = 0;
    while (fr == c.s_if || fr == c.s_sf) ++fr;
    c.s_pr = fr;
  }
  if (c.status != 2 && !(isfinite(c.dt) && isfinite(c.t_sf) && c.dt > 0.0)) c.status = 2;
  return accept;
}

// Device-resident controller (opt-in; same contract as iso_advance_kernel).
__global__ __launch_bounds__(64) void dense_small_advance_kernel(DenseView v, TrajCtl* __restrict__ ctl,
                                                                 const IwpConst* __restrict__ Kc, ControlParams cp) {
  extern __shared__ double smem[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  TrajCtl c = ctl[b];
  if (c.status < 0) return;
  int iters = 0;
  c.status = 0;
  while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
    double dt = c.dt;
    if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
    StepArg arg;
    arg.dt = dt; arg.t = c.t_sf; arg.src = c.s_sf; arg.dst = c.s_pr; arg.active = 1; arg.pad = 0;
    const double e = dense_small_attempt<false>(v, *Kc, b, lane, arg, smem);
    __threadfence_block();
    ++iters;
    small_after_attempt(c, cp, e, dt);
    if (c.status == 2) break;
  }
  if (c.status == 0) c.status = (c.t_sf + cp.numerical_zero >= c.t_stop) ? 1 : 3;
  if (lane == 0) ctl[b] = c;
}

// The same for a scalar ODE (d = 1): the last accepted state stays in registers over the attempts of a launch, accepted states
// are written to their slots (as iso_advance_kernel does for the isotropic filter).
template <int N>
__global__ __launch_bounds__(64) void dense_d1_advance_kernel(DenseView v, TrajCtl* __restrict__ ctl, const IwpConst* __restrict__ Kc,
                                                              ControlParams cp) {
  const int b = blockIdx.x;
  const int l = ln::col_index<N>(threadIdx.x);
  TrajCtl c = ctl[b];
  if (c.status < 0) return;
  const IwpConst& K = *Kc;
  D1Prior<N> P;
  d1_load_prior<N>(Kc, P, l);
  int iters = 0;
  c.status = 0;
  const double atol = v.atol[b], rtol = v.rtol[b];
  double cm[N], cR[N], csig;
  dense_d1_load<N>(v, b, l, c.s_sf, cm, cR, csig);
  while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
    double dt = c.dt;
    if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
    double nm[N], nR[N], nsig;
    const double e = dense_d1_core<N>(v, K, P, b, l, dt, c.t_sf, cm, cR, csig, nm, nR, nsig, atol, rtol);
    ++iters;
    const int dst = c.s_pr;
    if (small_after_attempt(c, cp, e, dt)) {
      dense_d1_store<N>(v, b, l, dst, nm, nR, nsig);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        cm[i] = nm[i];
        cR[i] = nR[i];
      }
      csig = nsig;
    }
    if (c.status == 2) break;
  }
  if (c.status == 0) c.status = (c.t_sf + cp.numerical_zero >= c.t_stop) ? 1 : 3;
  ctl[b] = c;
}

// Fixed grid (pnode_solve_fixed_grid, the reference's solve_fixed_grid, vdp.py:88-105) for a scalar ODE: the whole replay in ONE
// launch, a wavefront per trajectory walking over its G - 1 steps with the state in registers; u at every grid point goes to
// `out` [B][G][1], the states of the last two steps to the slots the host's rotation assigns them (args[g * B + b]: dt, t, src, dst
// of step g; entry 0 carries the initial slot). Launch per step: 10 us for 2 us of work (725 k steps of the reference's
// "fixed, accurate" run: 7.2 s; its JAX CPU loop 6.5 s).
template <int N>
__global__ __launch_bounds__(64) void dense_d1_fixed_grid_kernel(DenseView v, const StepArg* __restrict__ args, const IwpConst* __restrict__ Kc,
                                                                 double* __restrict__ out, int G) {
  const int b = blockIdx.x;
  const int l = ln::col_index<N>(threadIdx.x);
  const IwpConst& K = *Kc;
  D1Prior<N> P;
  d1_load_prior<N>(Kc, P, l);
  const size_t B = size_t(v.B);
  double cm[N], cR[N], csig;
  dense_d1_load<N>(v, b, l, args[b].dst, cm, cR, csig);
  out[size_t(b) * G] = cm[0];
  const double atol = v.atol[b], rtol = v.rtol[b];
#pragma unroll 1
  for (int g = 1; g < G; ++g) {
    const StepArg a = args[size_t(g) * B + b];
    double nm[N], nR[N], nsig;
    (void)dense_d1_core<N>(v, K, P, b, l, a.dt, a.t, cm, cR, csig, nm, nR, nsig, atol, rtol);
    out[size_t(b) * G + g] = nm[0];
    if (g >= G - 2) dense_d1_store<N>(v, b, l, a.dst, nm, nR, nsig);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      cm[i] = nm[i];
      cR[i] = nR[i];
    }
    csig = nsig;
  }
}

bool dense_d1_fixed_grid_supported(const DenseView& v) {
  return v.d == 1 && v.k <= 2 && v.n >= 3 && v.n <= 8 && !v.fixedpoint && !v.force_generic;
}

void dense_d1_launch_fixed_grid(const DenseView& v, const StepArg* args, const IwpConst* K, double* out, int G, hipStream_t s) {
  switch (v.n) {
    case 3: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<3>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    case 4: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<4>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    case 5: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<5>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    case 6: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<6>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    case 7: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<7>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    case 8: hipLaunchKernelGGL(dense_d1_fixed_grid_kernel<8>, dim3(v.B), dim3(64), 0, s, v, args, K, out, G); break;
    default: break;
  }
}

static size_t small_shm_bytes(const DenseView& v) { return small_lds_doubles(v.D, v.d, v.k, v.n) * sizeof(double); }

bool dense_small_supported(const DenseView& v) { return v.D <= 32 && !v.fixedpoint; }

void dense_small_launch_step(const DenseView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  hipLaunchKernelGGL(dense_small_step_kernel, dim3(v.B), dim3(64), small_shm_bytes(v), s, v, args, K);
}

void dense_small_launch_advance(const DenseView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s) {
  if (v.d == 1 && v.k <= 2) {  // scalar ODE: the register-resident loop
    switch (v.n) {
      case 3: hipLaunchKernelGGL(dense_d1_advance_kernel<3>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      case 4: hipLaunchKernelGGL(dense_d1_advance_kernel<4>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      case 5: hipLaunchKernelGGL(dense_d1_advance_kernel<5>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      case 6: hipLaunchKernelGGL(dense_d1_advance_kernel<6>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      case 7: hipLaunchKernelGGL(dense_d1_advance_kernel<7>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      case 8: hipLaunchKernelGGL(dense_d1_advance_kernel<8>, dim3(v.B), dim3(64), 0, s, v, ctl, K, cp); return;
      default: break;
    }
  }
  hipLaunchKernelGGL(dense_small_advance_kernel, dim3(v.B), dim3(64), small_shm_bytes(v), s, v, ctl, K, cp);
}

}  // namespace pnode
