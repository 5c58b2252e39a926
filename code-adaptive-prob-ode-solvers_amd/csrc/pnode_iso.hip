// Isotropic (Kronecker) EKF0 stepping kernels for gfx950: one wavefront per trajectory.
//
// State per trajectory: mean (n x d) and ONE n x n factor R with Sigma = (R^T R) (x) I_d
// (impl.select("isotropic"), /root/reference/src/odecheckpts/ivpsolvers.py:32-33). The n x n
// algebra (IWP transition, square-root time update, backward transition of the fixed-point
// smoother, gain) is spread over the lanes of the wavefront, one column (or row) of a matrix per lane
// (pnode_lanes.hpp); the d-dimensional part (mean, residual, error norm) has one ODE dimension per
// lane, with wave-level reductions. The whole attempted step -- extrapolate, evaluate the vector
// field, calibrate, correct, normalise the error -- is ONE kernel.
//
// Every lane stays enabled from the first to the last instruction of a kernel (r04): loops over the ODE
// dimensions run a wave-uniform number of times on clamped indices, only stores are predicated. The row
// broadcasts of pnode_lanes.hpp need it, and it removes the `lane < d` regions in whose join blocks the
// compiler had placed register spills in front of the exec restore.
//
// Follows (arithmetic re-derived, see oracle/pn_oracle.py for the CPU restatement):
//   correction_ts0 + strategy_filter/fixedpoint + solver[_dynamic]   ivpsolvers.py:37-50
//   experiments/4_brusselator/run.py:51-61 (the configuration of the golden vectors)
#include "pnode_lanes.hpp"

namespace pnode {

// ---------------------------------------------------------------------------------------------
// The fused attempted step
// ---------------------------------------------------------------------------------------------

// Sum over the W wavefronts that share a trajectory (W = 1: plain wave reduction).
template <int W>
__device__ __forceinline__ double team_sum(double v, double* red, int wteam) {
  v = wave_sum(v);
  if constexpr (W == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wteam] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < W; ++i) t += red[i];
  return t;
}

// A wave-uniform value that this kernel itself may have written a moment ago (the output scale of a slot): loaded through the VECTOR
// memory path -- the scalar data cache is not coherent with the wavefront's own vector stores.
__device__ __forceinline__ double load_written(const double* p) {
  asm volatile("" : "+v"(p));
  return ln::uniform(*p);
}

// Per-lane constants of a kernel: row l of Abar, column l of chol(Qbar)^T, (nu - l)!  (l = ln::col_index<N>(lane))
template <int N>
struct LanePrior {
  double arow[N], qcol[N], fact;
};
template <int N>
__device__ __forceinline__ void load_lane_prior(const IwpConst* __restrict__ Kc, LanePrior<N>& P, int l) {
  ln::load_prior<N>(Kc, P.arow, P.qcol, l);
  P.fact = Kc->fact[l];
}

// p_i = dt^(nu-i+1/2) / (nu-i)!,  pinv_i = 1/p_i   (probdiffeq preconditioner_diagonal): lane i forms its own pair (one
// division each, all lanes side by side; the same sequence of multiplications as preconditioner() of pnode_common.hpp), the
// uniform arrays come from row broadcasts. pl / pil: my lane's entries.
template <int N>
__device__ __forceinline__ void preconditioner_lanes(double dt, double fact, int l, double (&p)[N], double (&pinv)[N], double& pl,
                                                     double& pil) {
  const double adt = fabs(dt);
  double pw = sqrt(adt);  // dt^(0.5 + m), m = nu - l = N - 1 - l
#pragma unroll
  for (int s = 0; s < N - 1; ++s) pw = (N - 1 - l > s) ? pw * adt : pw;
  const double mine = pw / fact, mine_inv = fact / pw;
  ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    p[i] = ln::bcast<i>(mine);
    pinv[i] = ln::bcast<i>(mine_inv);
  });
  pl = mine;
  pil = mine_inv;
}

// Correction on z = 0 (isotropic EKF0): F = column k of Rm (row k of L^-), gain, projector-form update of the factor in place.
// Returns the gain, one entry per lane (lane i: gain_i).
template <int N>
__device__ __forceinline__ double correct_factor(double (&Rm)[N], int k) {
  double F[N];
  double s2 = 0.0;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    F[c] = ln::lane_value(Rm[c], k);
    s2 += F[c] * F[c];
  }
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < N; ++c) acc += Rm[c] * F[c];
  const double gain = acc / overflow_to_nan(s2);
#pragma unroll
  for (int c = 0; c < N; ++c) Rm[c] -= F[c] * gain;
  return gain;
}

// ---------------------------------------------------------------------------------------------
// Filter, d <= 64, one wavefront per trajectory: ONE ODE dimension per lane, and the state of a trajectory fits the registers
// of its wavefront -- the attempted step as a function from registers to registers. iso_attempt (one launch per attempt, host
// controller) wraps it in a load and a store; the device-resident controller (iso_advance_kernel) keeps the last accepted
// state in registers over all the attempts of a launch and writes a state to its slot only when it is accepted: an attempt
// then waits for no global memory at all.
// State: m[N]: my ODE dimension of the mean (lanes beyond d repeat dimension d - 1), R[N]: the factor, col-lane, sigma.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void iso_copy_regs(double (&dm)[N], double (&dR)[N], double& dsig, const double (&sm)[N],
                                              const double (&sR)[N], double ssig) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    dm[i] = sm[i];
    dR[i] = sR[i];
  }
  dsig = ssig;
}

template <int N>
__device__ __forceinline__ void iso_load_regs(const IsoView& v, int b, int lane, int slot, double (&m)[N], double (&R)[N], double& sigma) {
  const int d = v.d;
  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(slot) * sB + b) * N * d;
  const int jc = lane < d ? lane : d - 1;
#pragma unroll
  for (int i = 0; i < N; ++i) m[i] = msrc[i * d + jc];
  ln::load_cols<N>(v.fac + (size_t(slot) * sB + b) * N * N, R, ln::col_index<N>(lane));
  sigma = load_written(v.sigma + size_t(slot) * sB + b);
}

template <int N>
__device__ __forceinline__ void iso_store_regs(const IsoView& v, int b, int lane, int slot, const double (&m)[N], const double (&R)[N],
                                               double sigma) {
  const int d = v.d;
  const size_t sB = size_t(v.B);
  double* mdst = v.mean + (size_t(slot) * sB + b) * N * d;
  const int jc = lane < d ? lane : d - 1;
#pragma unroll
  for (int i = 0; i < N; ++i) mdst[i * d + jc] = m[i];
  ln::store_cols<N>(v.fac + (size_t(slot) * sB + b) * N * N, R, ln::col_index<N>(lane));
  v.sigma[size_t(slot) * sB + b] = sigma;
}

// The running backward transition of the fixed-point smoother as register state: G (row-lane), RL (col-lane), xi (my ODE dimension)
template <int N>
__device__ __forceinline__ void iso_load_cond(const IsoView& v, int b, int lane, int slot, double (&G)[N], double (&RL)[N], double (&xi)[N]) {
  const int d = v.d, l = ln::col_index<N>(lane);
  const size_t sB = size_t(v.B);
  ln::load_rows<N>(v.cG + (size_t(slot) * sB + b) * N * N, G, l);
  ln::load_cols<N>(v.cLam + (size_t(slot) * sB + b) * N * N, RL, l);
  const double* x = v.cxi + (size_t(slot) * sB + b) * N * d;
  const int jc = lane < d ? lane : d - 1;
#pragma unroll
  for (int i = 0; i < N; ++i) xi[i] = x[i * d + jc];
}
template <int N>
__device__ __forceinline__ void iso_store_cond(const IsoView& v, int b, int lane, int slot, const double (&G)[N], const double (&RL)[N],
                                               const double (&xi)[N]) {
  const int d = v.d, l = ln::col_index<N>(lane);
  const size_t sB = size_t(v.B);
  ln::store_rows<N>(v.cG + (size_t(slot) * sB + b) * N * N, G, l);
  ln::store_cols<N>(v.cLam + (size_t(slot) * sB + b) * N * N, RL, l);
  double* x = v.cxi + (size_t(slot) * sB + b) * N * d;
  const int jc = lane < d ? lane : d - 1;
#pragma unroll
  for (int i = 0; i < N; ++i) x[i * d + jc] = xi[i];
}
template <int N>
__device__ __forceinline__ void iso_copy_cond(double (&dG)[N], double (&dRL)[N], double (&dxi)[N], const double (&sG)[N], const double (&sRL)[N],
                                              const double (&sxi)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    dG[i] = sG[i];
    dRL[i] = sRL[i];
    dxi[i] = sxi[i];
  }
}

// One attempted step from the state `s`. Returns the normalised error norm (a scalar the compiler knows to be uniform).
// FP: the fixed-point smoother's attempt -- also the backward transition of the step, composed with the running one
// (sG0, sRL0, sxi0) into (oG, oRL, oxi); the factor comes out of the same reflections as in the filter.
template <int N, bool FP = false>
__device__ __forceinline__ double iso_attempt_regs(const IsoView& v, const IwpConst& K, const LanePrior<N>& P, int b, int lane, double dt,
                                                   double t, const double (&sm)[N], const double (&sR)[N], double ssigma, double (&om)[N],
                                                   double (&oR)[N], double& osigma, double atol, double rtol, double* ybuf, double (&me)[N],
                                                   double (*sG0)[N] = nullptr, double (*sRL0)[N] = nullptr, const double (*sxi0)[N] = nullptr,
                                                   double (*oG)[N] = nullptr, double (*oRL)[N] = nullptr, double (*oxi)[N] = nullptr) {
  const int d = v.d, k = v.k, l = ln::col_index<N>(lane);
  const bool on = lane < d;
  const int jc = on ? lane : d - 1;
  double p[N], pinv[N], pl, pil;
  preconditioner_lanes<N>(dt, P.fact, l, p, pinv, pl, pil);

  // --- mean extrapolation  m^- = P Abar P^{-1} m  (handed back: the fixed-point smoother's backward offset needs it)
  {
    double mp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) mp[i] = pinv[i] * sm[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * mp[kk];
      const double e = p[i] * acc;
      me[i] = e;
      if (i < k) ybuf[i * d + jc] = e;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0);  // LDS writes of this wave visible to all its lanes

  // --- residual z = m^-[k] - f(m^-[<k])  and its norm
  VfCtx ctx{v.vf_id, d, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * k * d : nullptr};
  const double tnew = t + dt;
  // (entry k of a register array by selection: a run-time index sends the whole array to scratch memory)
  double mek = me[0], pk = p[0];
#pragma unroll
  for (int i = 1; i < N; ++i) {
    mek = (i == k) ? me[i] : mek;
    pk = (i == k) ? p[i] : pk;
  }
  const double z = mek - vf_f(ctx, ybuf, jc, tnew);
  const double zz = ln::uniform(wave_sum(on ? z * z : 0.0));  // (uniform by construction; said to the compiler as well)
  __builtin_amdgcn_wave_barrier();  // (ybuf is rewritten by the next attempt)

  // --- calibration + error estimate
  const double s1 = pk * K.qdiag_sqrt[k];
  const double sigma_hat = sqrt(zz) / (s1 * sqrt(double(d)));
  const double err = dt * sigma_hat * s1;
  const double sigma = v.dynamic ? sigma_hat : ssigma;

  // --- n x n square-root time update (+ backward transition), correction on z = 0
  if constexpr (!FP) {
    double Gunused[N], RLunused[N];
    ln::time_update<N, false>(P.arow, P.qcol, sR, pinv, pl, pil, sigma, l, oR, Gunused, RLunused);
  } else {
    ln::time_update<N, true>(P.arow, P.qcol, sR, pinv, pl, pil, sigma, l, oR, *oG, *oRL);
    // xi = m - G m^-, merged with the running backward transition
    double xi[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      xi[i] = sm[i];
      (*oxi)[i] = (*sxi0)[i];
    }
    ln::matvec_rows<N, -1>(*oG, me, xi);
    ln::matvec_rows<N, +1>(*sG0, xi, *oxi);
    ln::merge<N>(*sG0, *sRL0, *oG, *oRL, l);
  }
  const double gain = correct_factor<N>(oR, k);

  // --- corrected mean + normalised error norm
  ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    om[i] = me[i] - ln::bcast<i>(gain) * z;
  });
  const double rel = err / (atol + rtol * fabs(om[0]));
  const double acc2 = wave_sum(on ? rel * rel : 0.0);
  osigma = sigma;
  return ln::uniform(sqrt(acc2 / double(d)));
}

// One attempted step of trajectory b by a team of W wavefronts (W = 1: one wave per trajectory, four
// trajectories per workgroup; W = 4: the whole 256-thread workgroup works on one trajectory, used for large d):
// reads slot arg.src, writes slot arg.dst, returns the normalised error norm (team-uniform). `lane` is the
// index inside the team (0 .. 64 W - 1); ybuf [k][d], zbuf [d], red [W] are the team's LDS scratch. Every wavefront of a team
// carries the n x n algebra (identical copies); the first one writes it.
// REGS (W = 1, d <= 64; chosen by the launcher): one ODE dimension per lane, the register form above between a load and a store.
template <int N, bool FP, int W, bool REGS>
__device__ __forceinline__ double iso_attempt(const IsoView& v, const IwpConst& K, const LanePrior<N>& P, int b, int lane,
                                              const StepArg& arg, double* ybuf, double* zbuf, double* red) {
  constexpr int kTeam = kWave * W;
  const int wteam = lane >> 6, l = ln::col_index<N>(lane);
  const int d = v.d, k = v.k;
  if constexpr (REGS) {
    static_assert(W == 1, "register form: one wavefront per trajectory");
    {
      double sm[N], sR[N], ssig, om[N], oR[N], osig;
      iso_load_regs<N>(v, b, lane, arg.src, sm, sR, ssig);
      double me[N];
      if constexpr (!FP) {
        const double e = iso_attempt_regs<N>(v, K, P, b, lane, arg.dt, arg.t, sm, sR, ssig, om, oR, osig, v.atol[b], v.rtol[b], ybuf, me);
        iso_store_regs<N>(v, b, lane, arg.dst, om, oR, osig);
        return e;
      } else {
        double G0[N], RL0[N], xi0[N], oG[N], oRL[N], oxi[N];
        iso_load_cond<N>(v, b, lane, arg.src, G0, RL0, xi0);
        const double e = iso_attempt_regs<N, true>(v, K, P, b, lane, arg.dt, arg.t, sm, sR, ssig, om, oR, osig, v.atol[b], v.rtol[b], ybuf,
                                                   me, &G0, &RL0, &xi0, &oG, &oRL, &oxi);
        iso_store_regs<N>(v, b, lane, arg.dst, om, oR, osig);
        iso_store_cond<N>(v, b, lane, arg.dst, oG, oRL, oxi);
        return e;
      }
    }
  }

  // Lanes beyond d repeat the work of dimension d - 1 on clamped indices (no lane is ever disabled). That is harmless only while the
  // repeating lanes run in lockstep with the lane that owns d - 1, i.e. inside ONE wavefront: a wavefront of a team (W = 4) whose 64
  // dimensions all lie beyond d would repeat another wavefront's read-modify-write of m[d - 1] at a different time (found by
  // scripts/soak_iso_shapes.py at d = 130 .. 140: the smoother's offsets and the corrected mean of the last dimension were updated twice).
  // Such a wavefront skips the loop bodies -- a wave-uniform branch -- and still takes part in the barriers and sums.
  const int wbase = __builtin_amdgcn_readfirstlane(lane & ~(kWave - 1));
  const double dt = arg.dt;
  double p[N], pinv[N], pl, pil;
  preconditioner_lanes<N>(dt, P.fact, l, p, pinv, pl, pil);

  const size_t sB = size_t(v.B);
  const double* msrc = v.mean + (size_t(arg.src) * sB + b) * N * d;
  double* mdst = v.mean + (size_t(arg.dst) * sB + b) * N * d;

  // --- pass A: mean extrapolation  m^- = P Abar P^{-1} m
  for (int j0 = 0; j0 < d; j0 += kTeam) {
    if (W > 1 && j0 + wbase >= d) continue;  // (see wbase)
    const int j = j0 + lane;
    const bool on = j < d;
    const int jc = on ? j : d - 1;
    double mp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) mp[i] = pinv[i] * msrc[i * d + jc];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * mp[kk];
      const double e = p[i] * acc;
      mdst[i * d + jc] = e;
      if (i < k) ybuf[i * d + jc] = e;
    }
  }
  if constexpr (W == 1) {
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);  // LDS writes of this wave visible to all its lanes
  } else {
    __syncthreads();
  }

  // --- pass B: residual z = m^-[k] - f(m^-[<k])  and its norm
  VfCtx ctx{v.vf_id, d, k, v.prm ? v.prm + (v.prm_per_traj ? size_t(b) * v.P : 0) : nullptr,
            v.user_f ? v.user_f + size_t(b) * d : nullptr, v.user_J ? v.user_J + size_t(b) * d * k * d : nullptr};
  const double tnew = arg.t + dt;
  double zz = 0.0;
  for (int j0 = 0; j0 < d; j0 += kTeam) {
    if (W > 1 && j0 + wbase >= d) continue;  // (see wbase)
    const int j = j0 + lane;
    const bool on = j < d;
    const int jc = on ? j : d - 1;
    const double z = mdst[k * d + jc] - vf_f(ctx, ybuf, jc, tnew);
    zbuf[jc] = z;
    zz += on ? z * z : 0.0;
  }
  zz = ln::uniform(team_sum<W>(zz, red, wteam));

  // --- calibration + error estimate (isotropic EKF0: S1 = s1^2, s1 = p_k * sqrt(Qbar_kk))
  double pk = p[0];  // (entry k by selection: a run-time index sends the whole array to scratch memory)
#pragma unroll
  for (int i = 1; i < N; ++i) pk = (i == k) ? p[i] : pk;
  const double s1 = pk * K.qdiag_sqrt[k];
  const double sigma_hat = sqrt(zz) / (s1 * sqrt(double(d)));
  const double err = dt * sigma_hat * s1;
  const double sigma_prev = load_written(v.sigma + size_t(arg.src) * sB + b);
  const double sigma = v.dynamic ? sigma_hat : sigma_prev;

  // --- n x n square-root time update (+ backward transition for the fixed-point smoother)
  double Rprev[N], Rm[N], G[N], RL[N];
  ln::load_cols<N>(v.fac + (size_t(arg.src) * sB + b) * N * N, Rprev, l);
  ln::time_update<N, FP>(P.arow, P.qcol, Rprev, pinv, pl, pil, sigma, l, Rm, G, RL);

  if constexpr (FP) {
    // xi = m - G m^-, merged with the running backward transition of the source slot
    double G0[N], RL0[N];
    ln::load_rows<N>(v.cG + (size_t(arg.src) * sB + b) * N * N, G0, l);
    ln::load_cols<N>(v.cLam + (size_t(arg.src) * sB + b) * N * N, RL0, l);
    const double* xi0 = v.cxi + (size_t(arg.src) * sB + b) * N * d;
    double* xid = v.cxi + (size_t(arg.dst) * sB + b) * N * d;
    for (int j0 = 0; j0 < d; j0 += kTeam) {
      if (W > 1 && j0 + wbase >= d) continue;  // (see wbase)
      const int j = j0 + lane;
      const bool on = j < d;
      const int jc = on ? j : d - 1;
      double xi[N], me[N], acc[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        me[i] = mdst[i * d + jc];
        xi[i] = msrc[i * d + jc];
        acc[i] = xi0[i * d + jc];
      }
      ln::opaque(G);
      ln::opaque(G0);
      ln::matvec_rows<N, -1>(G, me, xi);
      ln::matvec_rows<N, +1>(G0, xi, acc);
#pragma unroll
      for (int i = 0; i < N; ++i) xid[i * d + jc] = acc[i];
    }
    ln::merge<N>(G0, RL0, G, RL, l);
    ln::store_rows<N>(v.cG + (size_t(arg.dst) * sB + b) * N * N, G, l);
    ln::store_cols<N>(v.cLam + (size_t(arg.dst) * sB + b) * N * N, RL, l);
  }

  // --- correction on z = 0: F = row k of L^- = column k of Rm; gain; projector-form update
  const double gain = correct_factor<N>(Rm, k);
  ln::store_cols<N>(v.fac + (size_t(arg.dst) * sB + b) * N * N, Rm, l);
  double gu[N];
  ln::sfor<0, N>([&](auto ic) PNODE_LAMBDA_INLINE {
    constexpr int i = decltype(ic)::value;
    gu[i] = ln::bcast<i>(gain);
  });

  // --- pass C: corrected mean + normalised error norm
  const double atol = v.atol[b], rtol = v.rtol[b];
  double acc2 = 0.0;
  for (int j0 = 0; j0 < d; j0 += kTeam) {
    if (W > 1 && j0 + wbase >= d) continue;  // (see wbase)
    const int j = j0 + lane;
    const bool on = j < d;
    const int jc = on ? j : d - 1;
    const double z = zbuf[jc];
    double u0 = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const double mnew = mdst[i * d + jc] - gu[i] * z;
      mdst[i * d + jc] = mnew;
      if (i == 0) u0 = mnew;
    }
    const double rel = err / (atol + rtol * fabs(u0));
    acc2 += on ? rel * rel : 0.0;
  }
  acc2 = team_sum<W>(acc2, red, wteam);
  v.sigma[size_t(arg.dst) * sB + b] = sigma;
  return ln::uniform(sqrt(acc2 / double(d)));
}

template <int N, bool FP, int W, bool REGS>
__global__ __launch_bounds__(256) void iso_step_kernel(IsoView v, const StepArg* __restrict__ args,
                                                       const IwpConst* __restrict__ Kc) {
  extern __shared__ double smem[];
  constexpr int TPB = 4 / W;  // trajectories per 256-thread workgroup
  const int lane = threadIdx.x % (kWave * W);
  const int team = threadIdx.x / (kWave * W);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * TPB + team);
  if (b >= v.B) return;
  const StepArg arg = args[b];
  if (!arg.active) return;
  double* ybuf = smem + size_t(team) * ((size_t(v.k) + 1) * v.d + 4);  // [k][d] staged state for the vector field
  double* zbuf = ybuf + size_t(v.k) * v.d;                            // [d] residual
  double* red = zbuf + v.d;                                           // [W]
  LanePrior<N> P;
  load_lane_prior<N>(Kc, P, ln::col_index<N>(lane));
  const double e = iso_attempt<N, FP, W, REGS>(v, *Kc, P, b, lane, arg, ybuf, zbuf, red);
  v.errnorm[b] = e;
}

// ---------------------------------------------------------------------------------------------
// Checkpoint interpolation (probdiffeq "interpolate" for filter / fixedpoint) and right corner
// ---------------------------------------------------------------------------------------------

// One checkpoint of trajectory b by ONE wavefront (lane = 0 .. 63, every lane enabled): the body of iso_interp_kernel, also called
// by the device-resident controller when it walks over the save_at points itself (iso_advance_kernel with a CkptPlan).
template <int N, bool FP>
__device__ __forceinline__ void iso_interp_body(const IsoView& v, const IwpConst* __restrict__ Kc, int b, int lane, const InterpArg& arg) {
  const int l = ln::col_index<N>(lane);
  const IwpConst& K = *Kc;
  const int d = v.d;
  const size_t sB = size_t(v.B);
  const size_t ck = size_t(b) * v.K + arg.ckpt;
  double* m0 = v.mean + (size_t(arg.s0) * sB + b) * N * d;
  double* m1 = v.mean + (size_t(arg.s1) * sB + b) * N * d;
  double* cm = v.ck_mean + ck * N * d;

  if (arg.at_step) {
    // checkpoint coincides with STEP_FROM: store it; fixedpoint resets the backward transition
    // (wave-uniform trip counts, clamped indices, unpredicated stores: the duplicate lanes repeat their neighbour's copy)
    const int nd = N * d;
    for (int e0 = 0; e0 < nd; e0 += kWave) {
      const int e = e0 + lane < nd ? e0 + lane : nd - 1;
      cm[e] = m1[e];
    }
    const double* f1 = v.fac + (size_t(arg.s1) * sB + b) * N * N;
    for (int e0 = 0; e0 < N * N; e0 += kWave) {
      const int e = e0 + lane < N * N ? e0 + lane : N * N - 1;
      v.ck_fac[ck * N * N + e] = f1[e];
    }
    if constexpr (FP) {
      double* g1 = v.cG + (size_t(arg.s1) * sB + b) * N * N;
      double* l1 = v.cLam + (size_t(arg.s1) * sB + b) * N * N;
      for (int e0 = 0; e0 < N * N; e0 += kWave) {
        const int e = e0 + lane < N * N ? e0 + lane : N * N - 1;
        const double gv = g1[e], lv = l1[e];
        v.ck_G[ck * N * N + e] = gv;
        v.ck_Lam[ck * N * N + e] = lv;
        __builtin_amdgcn_wave_barrier();  // (the duplicate lanes have read the entry before anyone resets it)
        g1[e] = (e / N == e % N) ? 1.0 : 0.0;
        l1[e] = 0.0;
      }
      double* xi1 = v.cxi + (size_t(arg.s1) * sB + b) * N * d;
      double* cxi = v.ck_xi + ck * N * d;
      for (int e0 = 0; e0 < nd; e0 += kWave) {
        const int e = e0 + lane < nd ? e0 + lane : nd - 1;
        const double x = xi1[e];
        cxi[e] = x;
        __builtin_amdgcn_wave_barrier();
        xi1[e] = 0.0;
      }
    }
    v.ck_steps[ck] = arg.steps;
    v.ck_sigma[ck] = load_written(v.sigma + size_t(arg.s1) * sB + b);
    return;
  }

  LanePrior<N> P;
  load_lane_prior<N>(Kc, P, l);
  const double sigma = load_written(v.sigma + size_t(arg.s1) * sB + b);
  const double dt0 = arg.t - arg.t0, dt1 = arg.t1 - arg.t;
  double p[N], pinv[N], pl, pil;
  preconditioner_lanes<N>(dt0, P.fact, l, p, pinv, pl, pil);

  double R0[N], Rt[N], Gt[N], RLt[N];
  ln::load_cols<N>(v.fac + (size_t(arg.s0) * sB + b) * N * N, R0, l);
  ln::time_update<N, FP>(P.arow, P.qcol, R0, pinv, pl, pil, sigma, l, Rt, Gt, RLt);

  if constexpr (!FP) {
    // filter: the solution at t is the prediction from INTERP_FROM; it becomes the new INTERP_FROM
    for (int j0 = 0; j0 < d; j0 += kWave) {
      const int j = j0 + lane;
      const bool on = j < d;
      const int jc = on ? j : d - 1;
      double mp[N];
#pragma unroll
      for (int i = 0; i < N; ++i) mp[i] = pinv[i] * m0[i * d + jc];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * mp[kk];
        const double e = p[i] * acc;
        m0[i * d + jc] = e;
        cm[i * d + jc] = e;
      }
    }
    ln::store_cols<N>(v.fac + (size_t(arg.s0) * sB + b) * N * N, Rt, l);
    ln::store_cols<N>(v.ck_fac + ck * N * N, Rt, l);
    v.sigma[size_t(arg.s0) * sB + b] = sigma;
    v.ck_steps[ck] = arg.steps;
    v.ck_sigma[ck] = sigma;
    return;
  } else {
    // fixedpoint: (i) t0 -> t with merged backward transition, (ii) t -> t1 from an identity
    // transition, (iii) marginalise STEP_FROM through the t1 -> t transition.
    double q[N], qinv[N], ql, qil;
    preconditioner_lanes<N>(dt1, P.fact, l, q, qinv, ql, qil);
    double Runused[N], G1[N], RL1[N];
    ln::time_update<N, true>(P.arow, P.qcol, Rt, qinv, ql, qil, sigma, l, Runused, G1, RL1);

    double G0[N], RL0[N];
    ln::load_rows<N>(v.cG + (size_t(arg.s0) * sB + b) * N * N, G0, l);
    ln::load_cols<N>(v.cLam + (size_t(arg.s0) * sB + b) * N * N, RL0, l);
    double* xi0 = v.cxi + (size_t(arg.s0) * sB + b) * N * d;
    double* xi1 = v.cxi + (size_t(arg.s1) * sB + b) * N * d;
    double* cxi = v.ck_xi + ck * N * d;
    for (int j0 = 0; j0 < d; j0 += kWave) {
      const int j = j0 + lane;
      const bool on = j < d;
      const int jc = on ? j : d - 1;
      double mo[N], mt[N], mt1[N], tmp[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        mo[i] = m0[i * d + jc];
        tmp[i] = pinv[i] * mo[i];
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * tmp[kk];
        mt[i] = p[i] * acc;  // filtering mean at t
      }
#pragma unroll
      for (int i = 0; i < N; ++i) tmp[i] = qinv[i] * mt[i];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * tmp[kk];
        mt1[i] = q[i] * acc;  // prediction of t1 from t
      }
      // xi of (t0 -> t), merged with the running transition of INTERP_FROM
      double xit[N], cx[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        xit[i] = mo[i];
        cx[i] = xi0[i * d + jc];
      }
      ln::opaque(Gt);
      ln::opaque(G0);
      ln::opaque(G1);
      ln::matvec_rows<N, -1>(Gt, mt, xit);
      ln::matvec_rows<N, +1>(G0, xit, cx);
      // xi of (t -> t1); smoothed mean at t
      double m1v[N], x1[N], ms[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        m1v[i] = m1[i * d + jc];
        x1[i] = mt[i];
        ms[i] = 0.0;
      }
      ln::matvec_rows<N, -1>(G1, mt1, x1);
      ln::matvec_rows<N, +1>(G1, m1v, ms);
#pragma unroll
      for (int i = 0; i < N; ++i) {
        cxi[i * d + jc] = cx[i];
        xi1[i * d + jc] = x1[i];        // STEP_FROM now remembers how to get back to t
        cm[i * d + jc] = ms[i] + x1[i];  // solution at the checkpoint
        m0[i * d + jc] = mt[i];         // INTERP_FROM <- filtering state at t
        xi0[i * d + jc] = 0.0;          //   ... with an identity backward transition
      }
    }
    // checkpoint factor: qr_r([R1 G1^T ; RL1])
    {
      double R1[N], X[N], Ts[N], PT[N], PB[N];
      ln::load_cols<N>(v.fac + (size_t(arg.s1) * sB + b) * N * N, R1, l);
#pragma unroll
      for (int c = 0; c < N; ++c) {  // row c of R1 G1^T, col-lane
        double acc = 0.0;
        ln::sfor<0, N>([&](auto kc) PNODE_LAMBDA_INLINE {
          constexpr int kk = decltype(kc)::value;
          acc += ln::bcast<kk>(R1[c]) * G1[kk];
        });
        X[c] = acc;
      }
#pragma unroll
      for (int r = 0; r < N; ++r) Ts[r] = (l >= r) ? RL1[r] : 0.0;
      ln::qr_stack<N, false>(Ts, X, PT, PB, l);
      ln::store_cols<N>(v.ck_fac + ck * N * N, Ts, l);
    }
    // checkpoint backward transition = merged (t0 -> t) transition
    ln::merge<N>(G0, RL0, Gt, RLt, l);
    ln::store_rows<N>(v.ck_G + ck * N * N, Gt, l);
    ln::store_cols<N>(v.ck_Lam + ck * N * N, RLt, l);
    // STEP_FROM transition <- (t1 -> t); INTERP_FROM <- (filter state at t, identity)
    ln::store_rows<N>(v.cG + (size_t(arg.s1) * sB + b) * N * N, G1, l);
    ln::store_cols<N>(v.cLam + (size_t(arg.s1) * sB + b) * N * N, RL1, l);
    ln::store_cols<N>(v.fac + (size_t(arg.s0) * sB + b) * N * N, Rt, l);
    {
      double* g0 = v.cG + (size_t(arg.s0) * sB + b) * N * N;
      double* l0 = v.cLam + (size_t(arg.s0) * sB + b) * N * N;
#pragma unroll
      for (int c = 0; c < N; ++c) {
        g0[l * N + c] = (l == c) ? 1.0 : 0.0;
        l0[l * N + c] = 0.0;
      }
    }
    v.sigma[size_t(arg.s0) * sB + b] = sigma;
    v.ck_steps[ck] = arg.steps;
    v.ck_sigma[ck] = sigma;
  }
}

// Device-resident controller (opt-in): every wavefront runs the accept/reject loop and the PI
// step-size law for ITS trajectory until the next save_at point is reached (or an iteration budget is spent), so
// the host synchronises once per checkpoint interval instead of once per attempted step. Same arithmetic as the
// host controller in pnode_solve_adaptive_save_at. The controller's state is wave-uniform and known to the compiler as such
// (scalar loads, the error norm through ln::uniform): its branches are scalar branches, no lane is ever disabled by them.
// The controller's bookkeeping after an attempt with error norm e and step dt (both loops of iso_advance_kernel): returns
// whether the state of the attempt is accepted; c.status = 2 ends the loop.
__device__ __forceinline__ bool iso_after_attempt(TrajCtl& c, const ControlParams& cp, double e, double dt) {
  c.attempts += 1;
  bool accept;
  if (e != e) {
    c.status = 2;  // NaN error estimate: stop (the reference accepts the NaN state and fails later)
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

// With a CkptPlan the launch also does what the host does between two launches otherwise: every save_at point STEP_FROM has reached
// or passed is stored (interpolation between INTERP_FROM and STEP_FROM, or the right corner), the next one becomes t_stop.
// Returns false when the last point is stored (c.status = 1).
template <int N, bool FP>
__device__ __forceinline__ bool iso_walk_checkpoints(const IsoView& v, const IwpConst* __restrict__ Kc, const CkptPlan& plan, const ControlParams& cp,
                                                     int b, int lane, TrajCtl& c, int& cursor, bool& touched) {
  const double* sa = plan.save_at + (plan.per_traj ? size_t(b) * (plan.K + 1) : 0);
  while (cursor <= plan.K) {
    const double nxt = sa[cursor];
    if (!(c.t_sf + cp.numerical_zero >= nxt)) break;
    InterpArg a;
    a.t0 = c.t_if; a.t1 = c.t_sf; a.s0 = c.s_if; a.s1 = c.s_sf; a.ckpt = cursor - 1; a.steps = c.steps;
    if (c.t_sf > nxt + cp.numerical_zero) {  // passed: interpolate, INTERP_FROM moves to the checkpoint
      a.t = nxt; a.at_step = 0;
      iso_interp_body<N, FP>(v, Kc, b, lane, a);
      c.t_if = nxt;
    } else {  // reached: STEP_FROM is the checkpoint
      a.t = c.t_sf; a.at_step = 1;
      iso_interp_body<N, FP>(v, Kc, b, lane, a);
      c.s_if = c.s_sf;
      c.t_if = c.t_sf;
      int fr = 0;
      while (fr == c.s_sf) ++fr;
      c.s_pr = fr;
    }
    plan.num_steps[size_t(b) * (plan.K + 1) + cursor] = c.steps;
    ++cursor;
    touched = true;
    __threadfence_block();  // (the slots the checkpoint rewrote are read by the next checkpoint / the next attempt)
  }
  if (cursor > plan.K) {
    c.status = 1;
    return false;
  }
  c.t_stop = sa[cursor];
  return true;
}

// WALK: the kernel contains the checkpoint code and honours the CkptPlan (instantiated for nu <= 5: with the smoother's checkpoint
// inlined, the kernels of higher order would need scratch memory; they end at t_stop and leave the checkpoint to the host).
template <int N, bool FP, int W, bool REGS, bool WALK>
__global__ __launch_bounds__(256) void iso_advance_kernel(IsoView v, TrajCtl* __restrict__ ctl, const IwpConst* __restrict__ Kc,
                                                          ControlParams cp, CkptPlan plan) {
  extern __shared__ double smem[];
  constexpr int TPB = 4 / W;
  const int lane = threadIdx.x % (kWave * W);
  const int team = threadIdx.x / (kWave * W);
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * TPB + team);
  if (b >= v.B) return;
  TrajCtl c = ctl[b];
  if (c.status < 0) return;
  double* ybuf = smem + size_t(team) * ((size_t(v.k) + 1) * v.d + 4);
  double* zbuf = ybuf + size_t(v.k) * v.d;
  double* red = zbuf + v.d;
  int iters = 0;
  c.status = 0;
  // (Nothing of the stepping loop is live across a checkpoint: the per-lane constants and the register state are loaded again
  // after it -- once per checkpoint interval -- so that the register budget is the larger of the two phases, not their sum.)
  const bool walk = WALK && plan.save_at != nullptr;
  int cursor = walk ? plan.cursor[b] : 0;
  // the checkpoints are one wavefront's work (the first of a team); the others wait at the barrier
  auto checkpoints = [&](bool& touched) {
    bool more = true;
    if constexpr (!WALK) {
      return more;
    } else if constexpr (W == 1) {
      more = iso_walk_checkpoints<N, FP>(v, Kc, plan, cp, b, lane, c, cursor, touched);
    } else {
      __syncthreads();
      // (every wavefront of the team walks the same uniform bookkeeping; only the first one touches memory)
      if ((lane >> 6) == 0) {
        more = iso_walk_checkpoints<N, FP>(v, Kc, plan, cp, b, lane & 63, c, cursor, touched);
        __threadfence_block();
      }
      __syncthreads();
      // the other wavefronts repeat the bookkeeping without the memory work
      if ((lane >> 6) != 0) {
        const double* sa = plan.save_at + (plan.per_traj ? size_t(b) * (plan.K + 1) : 0);
        while (cursor <= plan.K) {
          const double nxt = sa[cursor];
          if (!(c.t_sf + cp.numerical_zero >= nxt)) break;
          if (c.t_sf > nxt + cp.numerical_zero) {
            c.t_if = nxt;
          } else {
            c.s_if = c.s_sf;
            c.t_if = c.t_sf;
            int fr = 0;
            while (fr == c.s_sf) ++fr;
            c.s_pr = fr;
          }
          ++cursor;
          touched = true;
        }
        if (cursor > plan.K) {
          c.status = 1;
          more = false;
        } else {
          c.t_stop = sa[cursor];
        }
      }
    }
    return more;
  };
  if constexpr (REGS) {
    static_assert(W == 1, "register form: one wavefront per trajectory");
    const IwpConst& K = *Kc;
    const double atol = v.atol[b], rtol = v.rtol[b];
    if constexpr (FP) {
      // the smoother: the state AND the running backward transition stay in registers
      for (;;) {
        if (walk) {
          bool touched = false;
          if (!checkpoints(touched)) break;
        }
        LanePrior<N> P;
        load_lane_prior<N>(Kc, P, ln::col_index<N>(lane));
        double cm[N], cR[N], csig, cG[N], cRL[N], cxi[N];
        iso_load_regs<N>(v, b, lane, c.s_sf, cm, cR, csig);
        iso_load_cond<N>(v, b, lane, c.s_sf, cG, cRL, cxi);
        while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
          double dt = c.dt;
          if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
          double nm[N], nR[N], nsig, nG[N], nRL[N], nxi[N];
          double me[N];
          const double e = iso_attempt_regs<N, true>(v, K, P, b, lane, dt, c.t_sf, cm, cR, csig, nm, nR, nsig, atol, rtol, ybuf, me, &cG, &cRL,
                                                     &cxi, &nG, &nRL, &nxi);
          ++iters;
          const int dst = c.s_pr;
          if (iso_after_attempt(c, cp, e, dt)) {
            iso_store_regs<N>(v, b, lane, dst, nm, nR, nsig);
            iso_store_cond<N>(v, b, lane, dst, nG, nRL, nxi);
            iso_copy_regs<N>(cm, cR, csig, nm, nR, nsig);
            iso_copy_cond<N>(cG, cRL, cxi, nG, nRL, nxi);
          }
          if (c.status == 2) break;
        }
        if (!walk || c.status == 2 || !(c.t_sf + cp.numerical_zero >= c.t_stop)) break;
        __threadfence_block();  // (the accepted states are in their slots before the checkpoint reads them)
      }
    } else {
      // the last accepted state stays in registers; a state goes to its slot when it is accepted (the slots STEP_FROM and
      // INTERP_FROM are what the host -- or the checkpoint -- reads), a rejected attempt leaves no trace.
      for (;;) {
        if (walk) {
          bool touched = false;
          if (!checkpoints(touched)) break;
        }
        LanePrior<N> P;
        load_lane_prior<N>(Kc, P, ln::col_index<N>(lane));
        double cm[N], cR[N], csig;
        iso_load_regs<N>(v, b, lane, c.s_sf, cm, cR, csig);
        while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
          double dt = c.dt;
          if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
          double nm[N], nR[N], nsig;
          double me[N];
          const double e = iso_attempt_regs<N>(v, K, P, b, lane, dt, c.t_sf, cm, cR, csig, nm, nR, nsig, atol, rtol, ybuf, me);
          ++iters;
          const int dst = c.s_pr;
          if (iso_after_attempt(c, cp, e, dt)) {
            iso_store_regs<N>(v, b, lane, dst, nm, nR, nsig);
            iso_copy_regs<N>(cm, cR, csig, nm, nR, nsig);
          }
          if (c.status == 2) break;
        }
        if (!walk || c.status == 2 || !(c.t_sf + cp.numerical_zero >= c.t_stop)) break;
        __threadfence_block();
      }
    }
  } else {
    for (;;) {
      if (walk) {
        bool touched = false;
        if (!checkpoints(touched)) break;
      }
      LanePrior<N> P;
      load_lane_prior<N>(Kc, P, ln::col_index<N>(lane));
      while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
        double dt = c.dt;
        if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
        StepArg arg;
        arg.dt = dt; arg.t = c.t_sf; arg.src = c.s_sf; arg.dst = c.s_pr; arg.active = 1; arg.pad = 0;
        const double e = iso_attempt<N, FP, W, false>(v, *Kc, P, b, lane, arg, ybuf, zbuf, red);
        if constexpr (W > 1) __syncthreads();
        __threadfence_block();  // the slot written by this attempt is read by the next one (other lanes of the team)
        ++iters;
        iso_after_attempt(c, cp, e, dt);
        if (c.status == 2) break;
      }
      if (!walk || c.status == 2 || !(c.t_sf + cp.numerical_zero >= c.t_stop)) break;
    }
  }
  if (c.status == 0) c.status = (c.t_sf + cp.numerical_zero >= c.t_stop) ? 1 : 3;
  ctl[b] = c;
  if (walk) plan.cursor[b] = cursor;
}

// ---------------------------------------------------------------------------------------------
// Device-resident controller of the FIXED-POINT SMOOTHER, d <= 64: one workgroup of four wavefronts per trajectory.
// An attempt of the smoother is the filter's attempt plus the backward transition of the step and its composition with the
// running transition (ivpsolvers.py:43), none of which the controller's decision needs.
// Wavefront 0 runs the filter and the controller exactly like iso_advance_kernel (state in registers); every
// ACCEPTED step goes into a ring in LDS (previous factor and mean, extrapolated mean, dt, sigma, destination slot). The
// wavefronts 1..3 take the steps a = w - 1, w + 2, .. : the transition of a step (ln::time_update<N, true>) depends on that
// step alone, so three are in flight; the compositions are chained through the running transition in LDS (mergedS), which is
// written to the step's slot as it goes: at the end of the launch STEP_FROM and INTERP_FROM hold theirs, as the host-driven
// path leaves them. Same arithmetic, expression by expression, as iso_attempt<N, true, 1>. Every wait is a bounded spin.
// ---------------------------------------------------------------------------------------------
constexpr int kFpRing = 4;
constexpr int kFpSpin = 1 << 22;
using lds_word = __attribute__((address_space(3))) int;

template <int N>
__host__ __device__ constexpr int iso_fp_entry_doubles() { return N * N + 2 * N * kWave + 4; }
__host__ __device__ inline size_t iso_fp_shm_doubles(int n, int k, int d) {
  return size_t(k) * d + size_t(kFpRing) * (size_t(n) * n + 2 * size_t(n) * kWave + 4) + 2 * size_t(n) * n + size_t(n) * kWave + 8;
}

template <int N>
__global__ __launch_bounds__(256) void iso_advance_fp_kernel(IsoView v, TrajCtl* __restrict__ ctl, const IwpConst* __restrict__ Kc,
                                                             ControlParams cp) {
  extern __shared__ double smem[];
  constexpr int EN = iso_fp_entry_doubles<N>();
  const int tid = threadIdx.x, lane = tid & 63, l = ln::col_index<N>(lane);
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x;
  if (b >= v.B) return;
  TrajCtl c = ctl[b];
  if (c.status < 0) return;  // (uniform over the workgroup)
  const IwpConst& K = *Kc;
  const int d = v.d, k = v.k;
  const size_t sB = size_t(v.B);
  double* ybuf = smem;                         // [k][d]
  double* ring = ybuf + size_t(k) * d;         // [kFpRing][EN]: R_prev [N][N] | m_prev [N][64] | m^- [N][64] | dt, sigma, dst
  double* runG = ring + kFpRing * EN;          // running transition of the last merged step: G [N][N], RL [N][N], xi [N][64]
  double* runL = runG + N * N;
  double* runX = runL + N * N;
  int* flags = reinterpret_cast<int*>(runX + N * kWave);  // pushed, merged, total (-1: still running), abort
  volatile lds_word* pushedS = (volatile lds_word*)&flags[0];
  volatile lds_word* mergedS = (volatile lds_word*)&flags[1];
  volatile lds_word* totalS = (volatile lds_word*)&flags[2];
  volatile lds_word* abortS = (volatile lds_word*)&flags[3];
  if (tid == 0) {
    flags[0] = 0;
    flags[1] = 0;
    flags[2] = -1;
    flags[3] = 0;
  }
  if (wv == 1) {  // the running transition of STEP_FROM
    for (int e = lane; e < N * N; e += kWave) {
      runG[e] = v.cG[(size_t(c.s_sf) * sB + b) * N * N + e];
      runL[e] = v.cLam[(size_t(c.s_sf) * sB + b) * N * N + e];
    }
    const double* xi0 = v.cxi + (size_t(c.s_sf) * sB + b) * N * d;
    const int jc = lane < d ? lane : d - 1;
#pragma unroll
    for (int i = 0; i < N; ++i) runX[i * kWave + lane] = xi0[i * d + jc];
  }
  __syncthreads();  // the only workgroup barrier
  LanePrior<N> P;
  load_lane_prior<N>(Kc, P, l);
  auto lds_done = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  const int spin_limit = v.fp_spin_limit > 0 ? v.fp_spin_limit : kFpSpin;
  // A wait that runs into its bound is a synchronisation failure, not a result: reported through the handle's fault words (the
  // call that synchronises returns PNODE_EDEVICE); every wait of the workgroup then falls through.
  auto timed_out = [&](int which) {
    if (lane == 0 && v.fault && !__builtin_amdgcn_readfirstlane(*abortS)) {
      v.fault[1] = b;
      v.fault[3] = which;
      __atomic_store_n(v.fault, 2, __ATOMIC_RELEASE);
    }
    *abortS = 1;
  };

  if (wv == 0) {
    // ---------------- filter + controller (iso_advance_kernel's loop), accepted steps handed to the ring
    int iters = 0, acc = 0;
    c.status = 0;
    const double atol = v.atol[b], rtol = v.rtol[b];
    double cm[N], cR[N], csig;
    iso_load_regs<N>(v, b, lane, c.s_sf, cm, cR, csig);
    while (c.t_sf + cp.numerical_zero < c.t_stop && iters < cp.max_iters) {
      double dt = c.dt;
      if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
      double nm[N], nR[N], nsig;
      double me[N];
      const double e = iso_attempt_regs<N>(v, K, P, b, lane, dt, c.t_sf, cm, cR, csig, nm, nR, nsig, atol, rtol, ybuf, me);
      ++iters;
      const int dst = c.s_pr;
      if (iso_after_attempt(c, cp, e, dt)) {
        int n = 0;
        while (acc - __builtin_amdgcn_readfirstlane(*mergedS) >= kFpRing) {  // ring full: the smoother is behind
          __builtin_amdgcn_s_sleep(4);
          if (++n > spin_limit || __builtin_amdgcn_readfirstlane(*abortS)) {
            if (n > spin_limit) timed_out(0);
            break;
          }
        }
        double* ent = ring + (acc % kFpRing) * EN;
        ln::store_cols<N>(ent, cR, l);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          ent[N * N + i * kWave + lane] = cm[i];
          ent[N * N + N * kWave + i * kWave + lane] = me[i];
        }
        ent[EN - 4] = dt;  // (uniform values: every lane writes them)
        ent[EN - 3] = nsig;
        ent[EN - 2] = double(dst);
        lds_done();
        ++acc;
        *pushedS = acc;
        iso_store_regs<N>(v, b, lane, dst, nm, nR, nsig);
        iso_copy_regs<N>(cm, cR, csig, nm, nR, nsig);
      }
      if (c.status == 2) break;
      if (__builtin_amdgcn_readfirstlane(*abortS)) {
        c.status = 2;
        break;
      }
    }
    if (c.status == 0) c.status = (c.t_sf + cp.numerical_zero >= c.t_stop) ? 1 : 3;
    *totalS = acc;
    ctl[b] = c;
  } else {
    // ---------------- backward transitions of the accepted steps a = wv - 1, wv + 2, .. and their composition in order
#pragma unroll 1
    for (int a = wv - 1;; a += 3) {
      int n = 0;
      bool have = true;
      while (__builtin_amdgcn_readfirstlane(*pushedS) <= a) {
        const int tot = __builtin_amdgcn_readfirstlane(*totalS);
        if (tot >= 0 && a >= tot) {
          have = false;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
        if (++n > spin_limit || __builtin_amdgcn_readfirstlane(*abortS)) {
          if (n > spin_limit) timed_out(1);
          have = false;
          break;
        }
      }
      if (!have) break;
      asm volatile("" ::: "memory");
      const double* ent = ring + (a % kFpRing) * EN;
      double Rprev[N], Rm[N], G[N], RL[N];
      ln::load_cols<N>(ent, Rprev, l);
      const double dt = ln::uniform(ent[EN - 4]), sigma = ln::uniform(ent[EN - 3]);
      const int dst = __builtin_amdgcn_readfirstlane(int(ent[EN - 2]));
      double mprev[N], me[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        mprev[i] = ent[N * N + i * kWave + lane];
        me[i] = ent[N * N + N * kWave + i * kWave + lane];
      }
      double p[N], pinv[N], pl, pil;
      preconditioner_lanes<N>(dt, P.fact, l, p, pinv, pl, pil);
      ln::time_update<N, true>(P.arow, P.qcol, Rprev, pinv, pl, pil, sigma, l, Rm, G, RL);
      double xi[N];
#pragma unroll
      for (int i = 0; i < N; ++i) xi[i] = mprev[i];
      ln::matvec_rows<N, -1>(G, me, xi);
      // composition with the running transition: in the order of the steps
      n = 0;
      bool ok = true;
      while (__builtin_amdgcn_readfirstlane(*mergedS) < a) {
        __builtin_amdgcn_s_sleep(2);
        if (++n > spin_limit || __builtin_amdgcn_readfirstlane(*abortS)) {
          if (n > spin_limit) timed_out(2);
          ok = false;
          break;
        }
      }
      if (!ok) break;
      asm volatile("" ::: "memory");
      double G0[N], RL0[N];
      ln::load_rows<N>(runG, G0, l);
      ln::load_cols<N>(runL, RL0, l);
      double xn[N];
#pragma unroll
      for (int i = 0; i < N; ++i) xn[i] = runX[i * kWave + lane];
      ln::matvec_rows<N, +1>(G0, xi, xn);
      ln::merge<N>(G0, RL0, G, RL, l);
      lds_done();  // (my reads of the running transition are complete before it is overwritten)
      ln::store_rows<N>(runG, G, l);
      ln::store_cols<N>(runL, RL, l);
#pragma unroll
      for (int i = 0; i < N; ++i) runX[i * kWave + lane] = xn[i];
      double* xid = v.cxi + (size_t(dst) * sB + b) * N * d;
      const int jc = lane < d ? lane : d - 1;
#pragma unroll
      for (int i = 0; i < N; ++i) xid[i * d + jc] = xn[i];
      ln::store_rows<N>(v.cG + (size_t(dst) * sB + b) * N * N, G, l);
      ln::store_cols<N>(v.cLam + (size_t(dst) * sB + b) * N * N, RL, l);
      lds_done();
      *mergedS = a + 1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Checkpoint interpolation (probdiffeq "interpolate" for filter / fixedpoint) and right corner
// ---------------------------------------------------------------------------------------------

template <int N, bool FP>
__global__ __launch_bounds__(256) void iso_interp_kernel(IsoView v, const InterpArg* __restrict__ args,
                                                         const IwpConst* __restrict__ Kc) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
  if (b >= v.B) return;
  const InterpArg arg = args[b];
  if (arg.ckpt < 0) return;
  iso_interp_body<N, FP>(v, Kc, b, lane, arg);
}

// Backward pass over the stored checkpoint transitions (stats.markov_marginals(reverse=True),
// /root/reference/src/odecheckpts/ivpsolvers.py:80-81). out: [B][K+1][n][d]; out[:, c+1] is the
// smoothed mean at checkpoint c, out[:, 0] the smoothed mean at the initial time.
template <int N>
__global__ __launch_bounds__(256) void iso_smooth_kernel(IsoView v, const int32_t* __restrict__ num_filled,
                                                         double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
  if (b >= v.B) return;
  const int nf = num_filled[b];
  if (nf <= 0) return;
  const int d = v.d;
  double* ob = out + size_t(b) * (v.K + 1) * N * d;
  for (int j = lane; j < d; j += kWave) {
    double m[N];
    const double* last = v.ck_mean + (size_t(b) * v.K + (nf - 1)) * N * d;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      m[i] = last[i * d + j];
      ob[size_t(nf) * N * d + i * d + j] = m[i];
    }
    for (int c = nf - 1; c >= 0; --c) {
      const double* Gc = v.ck_G + (size_t(b) * v.K + c) * N * N;
      const double* xc = v.ck_xi + (size_t(b) * v.K + c) * N * d;
      double mn[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = xc[i * d + j];
#pragma unroll
        for (int kk = 0; kk < N; ++kk) acc += Gc[i * N + kk] * m[kk];
        mn[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        m[i] = mn[i];
        ob[size_t(c) * N * d + i * d + j] = mn[i];
      }
    }
  }
}

// Off-grid marginal means of a stored (smoother / fixed-point) posterior: stats.offgrid_marginals_searchsorted,
// /root/reference/src/odecheckpts/ivpsolvers.py:118,144. Query q of trajectory b lies in (t_lo, t_hi], the interval that ends at
// store entry c = idx (c = 0: it starts at the initial state, whose mean is init[b] and whose covariance is zero). Same
// arithmetic as the checkpoint interpolation above: extrapolate the filtering state at t_lo to t, form the backward transition
// t_hi -> t from a second extrapolation t -> t_hi (output scale of the state at t_hi), and push the SMOOTHED mean at t_hi
// (backward pass, `smoothed`) through it. One wavefront per (trajectory, query).
template <int N>
__global__ __launch_bounds__(256) void iso_offgrid_kernel(IsoView v, const OffgridArg* __restrict__ qs, int Q,
                                                          const double* __restrict__ init, const double* __restrict__ smoothed,
                                                          const IwpConst* __restrict__ Kc, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, l = ln::col_index<N>(lane);
  const int wv = threadIdx.x >> 6;
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
  if (item >= v.B * Q) return;
  const int b = item / Q;
  const OffgridArg a = qs[item];
  if (a.idx < 0) return;
  const IwpConst& K = *Kc;
  const int d = v.d, c = a.idx;
  const double* ms = smoothed + (size_t(b) * (v.K + 1) + c + 1) * N * d;  // smoothed mean at t_hi
  double* o = out + size_t(item) * N * d;
  if (a.at_grid) {
    const int nd = N * d;
    for (int e0 = 0; e0 < nd; e0 += kWave) {
      const int e = e0 + lane < nd ? e0 + lane : nd - 1;
      o[e] = ms[e];
    }
    return;
  }
  const double* m0 = (c == 0) ? init + size_t(b) * N * d : v.ck_mean + (size_t(b) * v.K + c - 1) * N * d;
  const double sigma = v.ck_sigma[size_t(b) * v.K + c];
  LanePrior<N> P;
  load_lane_prior<N>(Kc, P, l);
  double p[N], pinv[N], pl, pil, q[N], qinv[N], ql, qil;
  preconditioner_lanes<N>(a.t - a.t_lo, P.fact, l, p, pinv, pl, pil);
  preconditioner_lanes<N>(a.t_hi - a.t, P.fact, l, q, qinv, ql, qil);
  double R0[N], Rt[N], Gt[N], RLt[N], Ru[N], G1[N], RL1[N];
  ln::load_cols<N>(v.ck_fac + (size_t(b) * v.K + (c > 0 ? c - 1 : 0)) * N * N, R0, l);
  if (c == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) R0[i] = 0.0;
  }
  ln::time_update<N, false>(P.arow, P.qcol, R0, pinv, pl, pil, sigma, l, Rt, Gt, RLt);
  ln::time_update<N, true>(P.arow, P.qcol, Rt, qinv, ql, qil, sigma, l, Ru, G1, RL1);
  for (int j0 = 0; j0 < d; j0 += kWave) {
    const int j = j0 + lane;
    const bool on = j < d;
    const int jc = on ? j : d - 1;
    double mt[N], mt1[N], tmp[N];
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i] = pinv[i] * m0[i * d + jc];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * tmp[kk];
      mt[i] = p[i] * acc;  // filtering mean at t
    }
#pragma unroll
    for (int i = 0; i < N; ++i) tmp[i] = qinv[i] * mt[i];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = 0.0;
#pragma unroll
      for (int kk = i; kk < N; ++kk) acc += K.a[i][kk] * tmp[kk];
      mt1[i] = q[i] * acc;  // prediction of t_hi from t
    }
    double diff[N], acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      diff[i] = ms[i * d + jc] - mt1[i];
      acc[i] = mt[i];
    }
    ln::opaque(G1);
    ln::matvec_rows<N, +1>(G1, diff, acc);
#pragma unroll
    for (int i = 0; i < N; ++i) o[i * d + jc] = acc[i];
  }
}

// Joint samples of the stored posterior, backwards from the terminal marginal (stats.markov_sample(..., reverse=True),
// /root/reference/experiments/5_vs_interpolation/measure.py:69-75): x_T = m_T + R_T^T e_T, x_c = G_c x_{c+1} + xi_c + Lam_c^T e_c.
// eps / out: [B][S][K+1][n][d] (entry 0 = initial time, entry c+1 = store entry c); the host supplies the standard-normal base
// samples, so a run is reproducible against the oracle. One wavefront per (trajectory, sample).
template <int N>
__global__ __launch_bounds__(256) void iso_sample_kernel(IsoView v, const int32_t* __restrict__ num_filled, int S, int P,
                                                         const double* __restrict__ eps, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv);
  if (item >= v.B * S) return;
  const int b = item / S;
  const int nf = num_filled[b];
  if (nf <= 0) return;
  const int d = v.d;
  const double* e = eps + size_t(item) * P * N * d;  // P points per sample (>= num_filled + 1): the caller's stride, not the store's capacity
  double* o = out + size_t(item) * P * N * d;
  const double* RT = v.ck_fac + (size_t(b) * v.K + nf - 1) * N * N;  // (wave-uniform address: scalar loads)
  for (int j = lane; j < d; j += kWave) {
    double x[N];
    const double* mT = v.ck_mean + (size_t(b) * v.K + nf - 1) * N * d;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double acc = mT[i * d + j];
#pragma unroll
      for (int kk = 0; kk <= i; ++kk) acc += RT[kk * N + i] * e[(size_t(nf) * N + kk) * d + j];  // R upper: (R^T e)_i = sum_{k <= i} R[k][i] e_k
      x[i] = acc;
      o[(size_t(nf) * N + i) * d + j] = acc;
    }
    for (int c = nf - 1; c >= 0; --c) {
      const double* Gc = v.ck_G + (size_t(b) * v.K + c) * N * N;
      const double* Lc = v.ck_Lam + (size_t(b) * v.K + c) * N * N;
      const double* xc = v.ck_xi + (size_t(b) * v.K + c) * N * d;
      double xn[N];
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double acc = xc[i * d + j];
#pragma unroll
        for (int kk = 0; kk < N; ++kk) acc += Gc[i * N + kk] * x[kk];
#pragma unroll
        for (int kk = 0; kk <= i; ++kk) acc += Lc[kk * N + i] * e[(size_t(c) * N + kk) * d + j];
        xn[i] = acc;
      }
#pragma unroll
      for (int i = 0; i < N; ++i) {
        x[i] = xn[i];
        o[(size_t(c) * N + i) * d + j] = xn[i];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Host-side launchers (dispatch on n = nu + 1 and strategy)
// ---------------------------------------------------------------------------------------------

// one wave per trajectory up to d = 128; the whole workgroup on one trajectory beyond (latency of a step
// falls ~3x for d >= 256, which is what a single long trajectory is bound by)
static inline bool iso_wide(const IsoView& v) { return v.d > 128; }
static inline size_t iso_shm(const IsoView& v, int tpb) { return size_t(tpb) * ((size_t(v.k) + 1) * v.d + 4) * sizeof(double); }

template <int N>
static void launch_step_n(const IsoView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
  if (iso_wide(v)) {
    if (v.fixedpoint)
      hipLaunchKernelGGL((iso_step_kernel<N, true, 4, false>), dim3(v.B), dim3(256), iso_shm(v, 1), s, v, args, K);
    else
      hipLaunchKernelGGL((iso_step_kernel<N, false, 4, false>), dim3(v.B), dim3(256), iso_shm(v, 1), s, v, args, K);
    return;
  }
  const int blocks = (v.B + 3) / 4;
  const bool regs = v.d <= kWave;  // one ODE dimension per lane: the state of a trajectory in the registers of its wavefront
  if (v.fixedpoint) {
    if (regs)
      hipLaunchKernelGGL((iso_step_kernel<N, true, 1, true>), dim3(blocks), dim3(256), iso_shm(v, 4), s, v, args, K);
    else
      hipLaunchKernelGGL((iso_step_kernel<N, true, 1, false>), dim3(blocks), dim3(256), iso_shm(v, 4), s, v, args, K);
  } else {
    if (regs)
      hipLaunchKernelGGL((iso_step_kernel<N, false, 1, true>), dim3(blocks), dim3(256), iso_shm(v, 4), s, v, args, K);
    else
      hipLaunchKernelGGL((iso_step_kernel<N, false, 1, false>), dim3(blocks), dim3(256), iso_shm(v, 4), s, v, args, K);
  }
}
template <int N>
static void launch_interp_n(const IsoView& v, const InterpArg* args, const IwpConst* K, hipStream_t s) {
  const int blocks = (v.B + 3) / 4;
  if (v.fixedpoint)
    hipLaunchKernelGGL((iso_interp_kernel<N, true>), dim3(blocks), dim3(256), 0, s, v, args, K);
  else
    hipLaunchKernelGGL((iso_interp_kernel<N, false>), dim3(blocks), dim3(256), 0, s, v, args, K);
}

#define PNODE_DISPATCH_N(n, CALL)      \
  switch (n) {                         \
    case 2: CALL(2); break;            \
    case 3: CALL(3); break;            \
    case 4: CALL(4); break;            \
    case 5: CALL(5); break;            \
    case 6: CALL(6); break;            \
    case 7: CALL(7); break;            \
    case 8: CALL(8); break;            \
    case 9: CALL(9); break;            \
    default: break;                    \
  }

void iso_launch_step(const IsoView& v, const StepArg* args, const IwpConst* K, hipStream_t s) {
#define CALL(NN) launch_step_n<NN>(v, args, K, s)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}
void iso_launch_interp(const IsoView& v, const InterpArg* args, const IwpConst* K, hipStream_t s) {
#define CALL(NN) launch_interp_n<NN>(v, args, K, s)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}
constexpr int kWalkMaxN = 6;  // checkpoints inside the controller kernel up to nu = 5 (beyond: scratch memory, see iso_advance_kernel)
template <int N, bool FP, int W, bool REGS>
static void launch_advance_one(const IsoView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s, const CkptPlan& plan,
                               dim3 grid, size_t shm) {
  if constexpr (N <= kWalkMaxN) {
    if (plan.save_at) {
      hipLaunchKernelGGL((iso_advance_kernel<N, FP, W, REGS, true>), grid, dim3(256), shm, s, v, ctl, K, cp, plan);
      return;
    }
  }
  hipLaunchKernelGGL((iso_advance_kernel<N, FP, W, REGS, false>), grid, dim3(256), shm, s, v, ctl, K, cp, plan);
}
template <int N>
static void launch_advance_n(const IsoView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s, const CkptPlan& plan) {
  if (iso_wide(v)) {
    if (v.fixedpoint)
      launch_advance_one<N, true, 4, false>(v, ctl, K, cp, s, plan, dim3(v.B), iso_shm(v, 1));
    else
      launch_advance_one<N, false, 4, false>(v, ctl, K, cp, s, plan, dim3(v.B), iso_shm(v, 1));
    return;
  }
  const int blocks = (v.B + 3) / 4;
  // Fixed-point smoother, d <= 64, on request (pnode_set_smoother_pipeline): a workgroup per trajectory (filter wavefront + three
  // smoother wavefronts) shortens the chain of a trajectory but holds more of a CU; with more trajectories than fit
  // at once and equal work per trajectory the one-wavefront kernel (four trajectories per workgroup, every SIMD busy) has the
  // higher throughput. Both take the same decisions bit for bit (the filter arithmetic is the same code); the smoothed means agree
  // to rounding (the compiler contracts the multiply-adds of the transitions differently in the two kernels) -- which is why the
  // choice is the caller's and never made from the batch size: a shard of a batch gives the bits of the unsharded solve.
  if (v.fixedpoint && v.d <= kWave && v.fp_pipeline == 1) {
    hipLaunchKernelGGL((iso_advance_fp_kernel<N>), dim3(v.B), dim3(256), iso_fp_shm_doubles(N, v.k, v.d) * sizeof(double), s, v, ctl, K, cp);
    return;
  }
  const bool regs = v.d <= kWave;  // one ODE dimension per lane: the state of a trajectory in the registers of its wavefront
  const size_t shm = iso_shm(v, 4);
  if (v.fixedpoint) {
    if (regs)
      launch_advance_one<N, true, 1, true>(v, ctl, K, cp, s, plan, dim3(blocks), shm);
    else
      launch_advance_one<N, true, 1, false>(v, ctl, K, cp, s, plan, dim3(blocks), shm);
  } else {
    if (regs)
      launch_advance_one<N, false, 1, true>(v, ctl, K, cp, s, plan, dim3(blocks), shm);
    else
      launch_advance_one<N, false, 1, false>(v, ctl, K, cp, s, plan, dim3(blocks), shm);
  }
}
bool iso_advance_walks_checkpoints(const IsoView& v) { return v.n <= kWalkMaxN && !(v.fixedpoint && v.d <= kWave && v.fp_pipeline == 1 && !iso_wide(v)); }

void iso_launch_advance(const IsoView& v, TrajCtl* ctl, const IwpConst* K, const ControlParams& cp, hipStream_t s, const CkptPlan& plan) {
#define CALL(NN) launch_advance_n<NN>(v, ctl, K, cp, s, plan)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}
void iso_launch_offgrid(const IsoView& v, const OffgridArg* qs, int Q, const double* init, const double* smoothed, const IwpConst* K,
                        double* out, hipStream_t s) {
  const int blocks = (v.B * Q + 3) / 4;
#define CALL(NN) hipLaunchKernelGGL((iso_offgrid_kernel<NN>), dim3(blocks), dim3(256), 0, s, v, qs, Q, init, smoothed, K, out)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}
void iso_launch_sample(const IsoView& v, const int32_t* num_filled, int S, int P, const double* eps, double* out, hipStream_t s) {
  const int blocks = (v.B * S + 3) / 4;
#define CALL(NN) hipLaunchKernelGGL((iso_sample_kernel<NN>), dim3(blocks), dim3(256), 0, s, v, num_filled, S, P, eps, out)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}
void iso_launch_smooth(const IsoView& v, const int32_t* num_filled, double* out, hipStream_t s) {
  const int blocks = (v.B + 3) / 4;
#define CALL(NN) hipLaunchKernelGGL((iso_smooth_kernel<NN>), dim3(blocks), dim3(256), 0, s, v, num_filled, out)
  PNODE_DISPATCH_N(v.n, CALL)
#undef CALL
}

}  // namespace pnode
