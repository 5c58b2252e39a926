// Shared device-side definitions for the pnode kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pnode_vf.hpp"

namespace pnode {

constexpr int kWave = 64;
constexpr int kMaxN = 9;  // nu <= 8

// Per-trajectory arguments of one batched attempt (uploaded by the host every step).
struct StepArg {
  double dt;
  double t;        // time of the source state
  int32_t src;     // state slot read (STEP_FROM)
  int32_t dst;     // state slot written (PROPOSED)
  int32_t active;  // 0: trajectory skipped
  int32_t pad;
};

// Per-trajectory arguments of one batched checkpoint interpolation.
struct InterpArg {
  double t;        // checkpoint time
  double t0;       // INTERP_FROM time
  double t1;       // STEP_FROM time
  int32_t s0;      // INTERP_FROM slot (updated in place)
  int32_t s1;      // STEP_FROM slot (backward model updated in place for fixedpoint)
  int32_t ckpt;    // checkpoint slot, < 0: skip
  int32_t at_step; // 1: STEP_FROM.t == checkpoint (no interpolation)
  int64_t steps;   // num_steps recorded with the checkpoint
};

// One off-grid query of a stored posterior (iso_offgrid_kernel).
struct OffgridArg {
  double t, t_lo, t_hi;  // query time and the stored interval (t_lo, t_hi] that contains it
  int32_t idx;           // store entry at t_hi (0-based); < 0: skip
  int32_t at_grid;       // 1: t coincides with t_hi, the smoothed mean there is the answer
};

// One "predict with backward transition" operation of the dense fixed-point smoother, per trajectory.
struct FpArg {
  double dt;
  int32_t src;         // slot providing mean / L (and, if merge, the running backward transition)
  int32_t dst;         // slot receiving the (merged) transition; mean / L too when the op writes the state
  int32_t merge;       // 1: compose with the transition of `src`; 0: start from the identity
  int32_t active;
  int32_t pad0, pad1;
};

// Device-resident controller state of one trajectory (iso_advance_kernel) and its parameters.
struct TrajCtl {
  double t_sf, t_if, dt, log_prev, t_stop;
  int64_t steps, attempts;
  int32_t s_sf, s_if, s_pr;
  int32_t status;  // in: < 0 skip; out: 1 reached t_stop, 2 failed (non-finite), 3 iteration budget spent
};
struct ControlParams {
  double safety, factor_min, factor_max, n1, n2, numerical_zero;
  int32_t clip, max_iters;
};
// The save_at points handed to the device-resident controller of the isotropic model: the launch then walks over them itself
// (interpolation / right corner of every checkpoint its trajectory passes, /root/reference/src/odecheckpts/ivpsolvers.py:71-77)
// and ends when the last one is stored, a trajectory has failed or the iteration budget is spent.
struct CkptPlan {
  const double* save_at;  // device: [K + 1] shared or [B][K + 1]; nullptr: the launch ends at TrajCtl::t_stop, the host does the checkpoint
  int32_t per_traj, K;
  int32_t* cursor;        // device [B]: index of the next save_at point (1 .. K + 1), in / out
  int64_t* num_steps;     // device [B][K + 1]: accepted steps when the checkpoint was stored
};

// One controller decision of a trajectory between two attempts of the multi-kernel dense paths (dense_ctl_kernel, and the tail of
// the fused measurement-update kernel): takes the error norm of the attempt just made (unless `first`), accepts / rejects,
// applies the PI law, rotates the state slots and forms the arguments of the next attempt -- same arithmetic and order as the host
// loop of pnode_solve_adaptive_save_at (control_proportional_integral + adaptive,
// /root/reference/src/odecheckpts/ivpsolvers.py:52-53).
__device__ __forceinline__ void controller_decision(TrajCtl& c, StepArg& a, double e, const ControlParams& cp, bool first) {
  if (c.status != 0) {  // < 0: not taking part; 1, 2: done
    a.active = 0;
    return;
  }
  if (!first && a.active) {
    const double dt = a.dt;
    c.attempts += 1;
    bool accept;
    if (e != e) {
      c.status = 2;  // NaN error estimate: stop (the reference accepts the NaN state and fails later)
      accept = true;
    } else {
      const double le = log(e);
      const double raw = cp.safety * exp(cp.n2 * c.log_prev - (cp.n1 + cp.n2) * le);
      const double factor = fmax(cp.factor_min, fmin(raw, cp.factor_max));
      if (e <= 1.0) c.log_prev = le;
      c.dt = factor * dt;
      accept = !(e > 1.0);
    }
    if (accept) {
      c.s_if = c.s_sf;
      c.t_if = c.t_sf;
      c.s_sf = c.s_pr;
      c.t_sf += dt;
      c.steps += 1;
      int fr = 0;
      while (fr == c.s_if || fr == c.s_sf) ++fr;
      c.s_pr = fr;
    }
    if (c.status == 0 && !(isfinite(c.dt) && isfinite(c.t_sf) && c.dt > 0.0)) c.status = 2;
  }
  if (c.status == 0 && !(c.t_sf + cp.numerical_zero < c.t_stop)) c.status = 1;
  double dt = c.dt;
  if (cp.clip) dt = fmin(dt, c.t_stop - c.t_sf);
  a.active = c.status == 0 ? 1 : 0;
  a.dt = a.active ? dt : 1.0;
  a.t = c.t_sf;
  a.src = c.s_sf;
  a.dst = c.s_pr;
  a.pad = 0;
}

// IWP(nu) constants in preconditioned coordinates, computed on the host in fp64.
//   a[i][k]   = binom(nu-i, nu-k)           (upper triangular)
//   qsT[c][i] = chol(Qbar)^T                (upper triangular),  Qbar[i][j] = 1/(2nu+1-i-j)
//   fact[i]   = (nu-i)!
struct IwpConst {
  double a[kMaxN][kMaxN];
  double qsT[kMaxN][kMaxN];
  double fact[kMaxN];
  double qdiag_sqrt[kMaxN];  // sqrt(Qbar[i][i]) = || row i of chol(Qbar) ||
};

// Butterfly sum over the wavefront: every lane gets the SAME bits (each stage adds two rounded values, and the addition commutes).
// The argument is made opaque first: with HIP's default -ffp-contract=fast a product in the caller (wave_sum(z * z)) would be
// fused into the first stage, fma(z, z, partner's rounded z * z), and lanes i and i ^ 32 would get sums that differ in the last
// bit (r04: found as an output scale that differed between the halves of a wavefront).
__device__ __forceinline__ double wave_sum(double v) {
  asm volatile("" : "+v"(v));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// p_i = dt^(nu-i+1/2) / (nu-i)!,  pinv_i = 1/p_i   (probdiffeq preconditioner_diagonal)
template <int N>
__device__ __forceinline__ void preconditioner(double dt, const IwpConst& K, double (&p)[N], double (&pinv)[N]) {
  const double adt = fabs(dt);
  const double sq = sqrt(adt);
  double pw = sq;  // dt^(0.5 + m), m = nu - i
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    p[i] = pw / K.fact[i];
    pinv[i] = K.fact[i] / pw;
    pw *= adt;
  }
}

// A sum of squares that has overflowed must not vanish from the result: finite / inf = 0 would silently switch a measurement
// update off (r04: on the reference's too-coarse Van der Pol grid, experiments/1_van_der_pol/vdp.py:94-108, the innovation
// variance overflowed at |u| ~ 1e54, the gain became 0, and the solution crept on, finite, to 1e61 -- where the reference, the
// oracle and exact arithmetic (tests/test_oracle_golden.py) all blow up). +-inf becomes NaN, everything else passes.
__device__ __forceinline__ double overflow_to_nan(double x) { return (fabs(x) <= 1.7976931348623157e308) ? x : __builtin_nan(""); }

// 1/sqrt(x) and 1/x to full fp64 precision from the hardware estimates + Newton steps: short dependency chains
// instead of the ~30-instruction IEEE sqrt / divide expansions (the small-matrix algebra is latency-bound).
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const double e = fma(-x * y, y, 1.0);  // 1 - x y^2
    y = fma(0.5 * y, e, y);
  }
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int i = 0; i < 2; ++i) y = fma(fma(-x, y, 1.0), y, y);
  return y;
}

// State factor of the handles that run the register-resident MFMA kernels (pnode_mfma_common.hpp). The measurement update on a
// noise-free observation leaves a covariance of rank D - d, and the MFMA path stores a factor with that many columns:
//      Sigma = Ls Ls^T,   Ls (D x (D - d)) = [ L^- Q2 (d dense columns) | columns >= 2d of the lower-triangular L^- ],
// Q2 an orthonormal basis of the null space of F = H L^- inside its 2d non-zero columns. The column count is padded to whole
// 64-column super-tiles (zero columns), DC = number of 16-column tiles; element (row, col) of Ls sits at tiled_index_rt(row, col, DC):
// the register order of the 16-row strips.
__host__ __device__ inline int tiled_col_tiles(int D, int d) { return ((D - d + 63) / 64) * 4; }
__host__ __device__ inline size_t tiled_index_rt(int row, int col, int DC) {
  const int s = row >> 4, r = row & 15, T = col >> 6, g = (col >> 4) & 3, e = col & 15;
  return ((size_t(s) * (2 * DC) + 8 * T + (e >> 1)) * 64 + r + 16 * g) * 2 + (e & 1);
}
// The same factor seen as a D x D matrix [Ls | 0] (what the generic kernels and the read-back hand out).
__host__ __device__ inline double tiled_factor_entry(const double* L, int row, int col, int D, int d) {
  return col < D - d ? L[tiled_index_rt(row, col, tiled_col_tiles(D, d))] : 0.0;
}

}  // namespace pnode
