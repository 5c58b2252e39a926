/* pnode.h — C ABI of the MI355X-native EKF predict/correct stepping path.
 *
 * The reference (pnkraemer/code-adaptive-prob-ode-solvers) has NO native boundary: the hot path is
 * entered through Python calls into probdiffeq (third party). This header therefore DEFINES the
 * boundary a probdiffeq-style host binds to; each entry point cites the reference call site whose
 * work it replaces (paths relative to /root/reference).
 *
 * Model: one opaque handle (`pnode_batch`) owns the device state of B independent IVPs on ONE GPU
 * and one HIP stream. The adaptive PI controller and the save_at bookkeeping stay on the host
 * (sequential, per trajectory); every call below moves whole batches. All host buffers are
 * caller-owned, contiguous, C-order doubles / int32 / uint8. Every function returns 0 on success
 * and a negative PNODE_E* code otherwise; `pnode_last_error()` has the text. A handle is not
 * thread-safe; use one handle per GPU (per process, under torch.distributed / RCCL).
 *
 * State slots per trajectory (the reference carries the same three copies, experiments/
 * 4_brusselator/run.py:68-70): STEP_FROM (latest accepted), INTERP_FROM (previous accepted or last
 * checkpoint), PROPOSED (target of the running attempt). Checkpoint storage is separate and sized
 * at creation (`num_checkpoints`).
 */
#ifndef PNODE_H
#define PNODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNODE_OK 0
#define PNODE_EINVAL (-1)   /* bad argument / unsupported configuration  -> Python ValueError   */
#define PNODE_EDEVICE (-2)  /* HIP runtime failure                      -> Python RuntimeError */
#define PNODE_ESTATE (-3)   /* call order violated (e.g. step before set_initial)              */

/* impl.select("isotropic" | "dense", ode_shape=(d,))  — src/odecheckpts/ivpsolvers.py:32-33,
 * experiments/1_van_der_pol/vdp.py:61 */
#define PNODE_MODEL_ISOTROPIC 0
#define PNODE_MODEL_DENSE 1
/* ivpsolvers.correction_ts0 / correction_ts1 — ivpsolvers.py:37, vdp.py:64 */
#define PNODE_CORR_TS0 0
#define PNODE_CORR_TS1 1
/* ivpsolvers.strategy_filter / strategy_fixedpoint — vdp.py:65, ivpsolvers.py:43 */
#define PNODE_STRATEGY_FILTER 0
#define PNODE_STRATEGY_FIXEDPOINT 1
/* ivpsolvers.solver / solver_dynamic — ivpsolvers.py:45-50 */
#define PNODE_CALIB_NONE 0
#define PNODE_CALIB_DYNAMIC 1

/* Built-in vector fields (device code + analytic Jacobians); replaces the JAX callables of
 * src/odecheckpts/ivps.py. `vf_params` meaning per id is given on the right. */
#define PNODE_VF_BRUSSELATOR 0   /* ivps.py:124-156; d = 2N, order 1; params: [alpha] (default 1/50) */
#define PNODE_VF_VDP_ORDER2 1    /* ivps.py:159-167, vdp.py:52-54; d = 1, order 2; params: [mu]       */
#define PNODE_VF_VDP_ORDER1 2    /* experiments/old/7_node/node.py:23-37; d = 2, order 1; [mu]        */
#define PNODE_VF_RIGID_BODY 3    /* ivps.py:20-29 (diffeqzoo defaults); d = 3, order 1; [a, b, c]     */
#define PNODE_VF_PLEIADES_ORDER2 4 /* ivps.py:66-99; d = 14, order 2; no params                       */
#define PNODE_VF_PLEIADES_ORDER1 5 /* ivps.py:44-56; d = 28, order 1; no params                       */
#define PNODE_VF_LOGISTIC 6      /* ivps.py:8-17 (diffeqzoo logistic); d = 1, order 1; [a, b]         */
#define PNODE_VF_LINEAR 7        /* test problem u' = lam * u (diagonal); d arbitrary; [lam]          */
#define PNODE_VF_THREE_BODY 8    /* ivps.py:32-41 (diffeqzoo three_body_restricted, experiment 5); d = 2, order 2; [moon mass] */

/* Escape hatch for vector fields that are not built in: the HOST evaluates f (and, for correction_ts1, the Jacobian)
 * for the whole batch at the extrapolated means and hands the values over before each attempt
 * (pnode_user_extrapolate -> user code -> pnode_user_set_vf -> pnode_attempt_step). Any d, order 1 or 2. */
#define PNODE_VF_USER 100

#define PNODE_MAX_VF_PARAMS 8

/* pnode_config.reserved, bit 0: the dense filter also stores the covariance factor at every save_at point ([B][K][D][D]
 * doubles) -- solution.posterior of the reference carries one per checkpoint (src/odecheckpts/ivpsolvers.py:71-81,
 * experiments/4_brusselator/run.py:136-138). The isotropic model always stores its (n x n) factors. */
#define PNODE_FLAG_CHECKPOINT_FACTORS 1

typedef struct pnode_config {
  int32_t batch;            /* B: number of independent IVPs on this GPU                     */
  int32_t d;                /* ODE dimension                                                 */
  int32_t nu;               /* prior_ibm(num_derivatives=nu), 1..8                           */
  int32_t ode_order;        /* 1 or 2                                                        */
  int32_t model;            /* PNODE_MODEL_*                                                 */
  int32_t correction;       /* PNODE_CORR_*                                                  */
  int32_t strategy;         /* PNODE_STRATEGY_*                                              */
  int32_t calibration;      /* PNODE_CALIB_*                                                 */
  int32_t vf_id;            /* PNODE_VF_*                                                    */
  int32_t num_vf_params;    /* doubles per trajectory in vf_params                           */
  int32_t vf_params_per_trajectory; /* 0: vf_params is [P] shared, 1: [B][P]                 */
  int32_t num_checkpoints;  /* capacity K of the checkpoint store (save_at points after t0)  */
  int32_t device;           /* HIP device ordinal                                            */
  int32_t reserved;         /* flags: PNODE_FLAG_*                                           */
  const double* vf_params;  /* host pointer, copied at creation                              */
} pnode_config;

typedef struct pnode_batch pnode_batch;

const char* pnode_last_error(void);
int pnode_version(void);
/* Number of visible HIP devices (negative code on failure). */
int pnode_device_count(void);

/* Build the solver state for B trajectories.
 * Replaces: impl.select + prior_ibm + correction_ts* + strategy_* + solver[_dynamic]
 * (src/odecheckpts/ivpsolvers.py:32-50, experiments/4_brusselator/run.py:51-61, vdp.py:61-66). */
int pnode_create(const pnode_config* cfg, pnode_batch** out);
int pnode_destroy(pnode_batch* h);

/* solver.initial_condition(tcoeffs, output_scale) — ivpsolvers.py:68, run.py:66, vdp.py:71.
 * tcoeffs: [B][nu+1][d] exact derivatives at t0 (zero initial covariance); t0, output_scale: [B]. */
int pnode_set_initial(pnode_batch* h, const double* tcoeffs, const double* t0, const double* output_scale);

/* ivpsolve.adaptive(solver, atol=, rtol=) — ivpsolvers.py:53; per-trajectory so that tolerance
 * sweeps (experiments/2_workprec_simple/run_simple.py:150-156) become the batch axis. */
int pnode_set_tolerances(pnode_batch* h, const double* atol, const double* rtol);

/* One attempted solver step for every active trajectory: extrapolate (IWP transition), linearise
 * the vector field, innovation + local error estimate, dynamic calibration, square-root covariance
 * time/measurement update, and the normalised error norm  ||dt*err / (atol + rtol*|u|)||_rms.
 * Reads STEP_FROM, writes PROPOSED. dt, active: [B]; err_norm_out: [B] (untouched where inactive).
 * Replaces the body of the rejection loop inside ivpsolve.solve_adaptive_* (ivpsolvers.py:71-77,
 * run.py:82-90,122-129, vdp.py:77-79) up to — not including — the PI controller. Synchronous. */
int pnode_attempt_step(pnode_batch* h, const double* dt, const uint8_t* active, double* err_norm_out);

/* PNODE_VF_USER only. Extrapolated means m^-[0..k-1] of the attempt about to be made with these dt:
 * y_out [B][ode_order][d] (rows of inactive trajectories are left untouched). */
int pnode_user_extrapolate(pnode_batch* h, const double* dt, const uint8_t* active, double* y_out);
/* PNODE_VF_USER only. f: [B][d] vector-field values at y_out; jac: [B][d][ode_order*d] with
 * jac[b][a][i*d + c] = d f_a / d u^(i)_c (may be NULL with correction_ts0). Consumed by the next attempt_step. */
int pnode_user_set_vf(pnode_batch* h, const double* f, const double* jac);

/* Accept/reject decision taken by the host controller: where accept[b] != 0,
 * INTERP_FROM <- STEP_FROM, STEP_FROM <- PROPOSED (slot rotation, no copies), t += dt, steps += 1. */
int pnode_commit(pnode_batch* h, const uint8_t* accept);

/* Checkpoint handling after an overstep (probdiffeq "interpolate"): for every b with slot[b] >= 0,
 * compute the solution at t[b] in (INTERP_FROM.t, STEP_FROM.t) and store it in checkpoint slot[b].
 * filter: prediction from INTERP_FROM; fixedpoint: prediction + one-step smoothing, the running
 * backward transition is stored with the checkpoint and reset (fixed-memory smoothing).
 * Replaces the save_at branch of solve_adaptive_save_at (ivpsolvers.py:71-77, run.py:119-138). */
int pnode_interpolate(pnode_batch* h, const double* t, const int32_t* slot);
/* Same, for the case STEP_FROM.t == save_at point (no interpolation needed). */
int pnode_checkpoint_at_step(pnode_batch* h, const int32_t* slot);

/* ivpsolve.control_proportional_integral(...) parameters (defaults: 0.95, 0.2, 10, 0.3, 0.4, no clipping). */
typedef struct pnode_control {
  double safety, factor_min, factor_max, power_integral_unscaled, power_proportional_unscaled;
  int32_t clip;      /* 1: clip dt to the next save_at point instead of overstepping + interpolating */
  int32_t reserved;
} pnode_control;

/* The whole adaptive driver loop of ivpsolve.solve_adaptive_save_at (ivpsolvers.py:71-77, run.py:122-129) in
 * native host code: per batched attempt one pnode_attempt_step, the PI controller + accept/reject for every
 * trajectory (sequential host arithmetic, as in the reference), pnode_commit, and checkpoint interpolation
 * whenever an accepted step reaches the next save_at point. Exactly the sequence the Python driver issues,
 * without the interpreter between launches.
 * save_at: [num_save] shared (per_trajectory = 0) or [B][num_save]; save_at[..., 0] must equal the time given to
 * pnode_set_initial and num_save - 1 <= num_checkpoints. dt0: [B]. error_contraction_rate = nu + 1.
 * num_steps_out: [B][num_save] accepted steps at each save_at point; num_attempts_out: [B];
 * failed_out: [B] (1: a non-finite step size or time stopped the trajectory). max_attempts <= 0: unlimited. */
int pnode_solve_adaptive_save_at(pnode_batch* h, const double* save_at, int32_t num_save, int32_t per_trajectory,
                                 const double* dt0, const pnode_control* control, double numerical_zero,
                                 int64_t max_attempts, int64_t* num_steps_out, int64_t* num_attempts_out,
                                 uint8_t* failed_out);

/* Same contract as pnode_solve_adaptive_save_at, with the accept/reject loop and the PI law running ON THE
 * DEVICE (one wavefront per trajectory loops until its next save_at point; SURVEY.md 8(f)-1, opt-in; isotropic
 * model, or dense model with (nu+1)*d <= 32; built-in vector fields only). The host synchronises once per
 * checkpoint interval (or every `iters_per_launch` attempts, <= 0: 4096) and still performs the checkpoint
 * interpolation calls. */
int pnode_solve_adaptive_save_at_device(pnode_batch* h, const double* save_at, int32_t num_save, int32_t per_trajectory,
                                        const double* dt0, const pnode_control* control, double numerical_zero,
                                        int32_t iters_per_launch, int64_t* num_steps_out, int64_t* num_attempts_out,
                                        uint8_t* failed_out);

/* Device-side replay of a prescribed grid without controller: ivpsolve.solve_fixed_grid
 * (vdp.py:88-103). grid: [G] shared by all trajectories (grid_per_trajectory = 0) or [B][G].
 * u_out: [B][G][d] zeroth-derivative means (u_out[:,0] = initial value). All launches are enqueued
 * without host synchronisation. */
int pnode_solve_fixed_grid(pnode_batch* h, const double* grid, int32_t num_grid, int32_t grid_per_trajectory,
                           double* u_out);

/* Read-back. which: 0 STEP_FROM, 1 INTERP_FROM, 2 PROPOSED. */
int pnode_read_mean(pnode_batch* h, int32_t which, double* mean_out /* [B][nu+1][d] */);
int pnode_read_time(pnode_batch* h, double* t_out /* [B] */, int64_t* num_steps_out /* [B] */);
int pnode_read_output_scale(pnode_batch* h, int32_t which, double* out /* [B] */);
/* Covariance factor of a state slot, as the upper-triangular-or-square R with Sigma = R^T R:
 * isotropic [B][n][n], dense [B][D][D]. */
int pnode_read_factor(pnode_batch* h, int32_t which, double* out);
/* Checkpoint store: means [B][K][nu+1][d]; num_steps at each checkpoint [B][K]. */
int pnode_read_checkpoint_means(pnode_batch* h, double* out);
/* Covariance factors at the checkpoints: out [B][K][m][m], a factor R with Sigma = R^T R (upper triangular for the isotropic
 * model and for interpolated checkpoints, a general square factor where a checkpoint coincides with a step; m = nu+1 for the isotropic
 * model, Sigma = R^T R (x) I_d; m = (nu+1) d for the dense filter created with PNODE_FLAG_CHECKPOINT_FACTORS). */
int pnode_read_checkpoint_factors(pnode_batch* h, double* out);
/* Handles with a backward model (strategy_fixedpoint, strategy_smoother): the stored transitions themselves,
 * x_{c-1} | x_c ~ N(G_c x_c + xi_c, Lambda_c) for store entry c (x_{-1}: the initial time) -- what probdiffeq's solution object
 * carries as `posterior.conditional` (src/odecheckpts/ivpsolvers.py:80 reads it through markov_select_terminal).
 * Isotropic model: G [B][K][nu+1][nu+1], xi [B][K][nu+1][d], Lam [B][K][nu+1][nu+1] an UPPER triangular factor with non-negative
 * diagonal, Lambda = (Lam^T Lam) (x) I_d. Dense model (D = (nu+1) d, state index i d + j): G [B][K][D][D], xi [B][K][D],
 * Lam [B][K][D][D] a LOWER triangular factor, Lambda = Lam Lam^T (the factors as the two models' kernels keep them). */
int pnode_read_checkpoint_transitions(pnode_batch* h, double* G, double* xi, double* Lam);
/* Backward pass over the stored checkpoint transitions (fixedpoint only):
 * stats.markov_select_terminal + stats.markov_marginals(reverse=True) — ivpsolvers.py:80-81.
 * num_filled[b] = number of checkpoints stored for trajectory b; the terminal marginal is the
 * last stored checkpoint. out: [B][K][nu+1][d] smoothed means. */
int pnode_smooth_checkpoint_means(pnode_batch* h, const int32_t* num_filled, double* out);

/* Off-grid marginal means of the stored posterior (isotropic model, smoother / fixedpoint): stats.offgrid_marginals_searchsorted
 * as called at /root/reference/src/odecheckpts/ivpsolvers.py:118,144. The backward pass runs first (as in
 * pnode_smooth_checkpoint_means); query (b, q) at time t[b][q] lies in the stored interval (t_lo, t_hi] that ENDS at store entry
 * idx[b][q] (0-based; entry 0 ends the interval that starts at the initial state, whose means are init_means[B][nu+1][d]);
 * at_grid = 1: t equals t_hi, the smoothed mean there is returned; idx < 0: skipped (zeros). out: [B][Q][nu+1][d]. */
int pnode_offgrid_means(pnode_batch* h, const int32_t* num_filled, const double* init_means, int32_t num_queries, const double* t,
                        const double* t_lo, const double* t_hi, const int32_t* idx, const uint8_t* at_grid, double* out);
/* Joint samples of the stored posterior, backwards from the terminal marginal (isotropic model, smoother / fixedpoint):
 * stats.markov_sample(key, markov_seq, shape=(S,), reverse=True), /root/reference/experiments/5_vs_interpolation/measure.py:69-75.
 * eps: standard-normal base samples [B][S][K+1][nu+1][d] from the caller (entry 0 = initial time, entry c+1 = store entry c);
 * out: the samples, same layout. */
int pnode_sample_checkpoints(pnode_batch* h, const int32_t* num_filled, int32_t num_samples, const double* eps, double* out);
/* The same with eps / out of num_points (>= max num_filled + 1) entries per sample instead of the store's capacity K + 1: a
 * store-every-step smoother has a capacity of thousands of entries of which a solve fills a few hundred. */
int pnode_sample_checkpoints_n(pnode_batch* h, const int32_t* num_filled, int32_t num_points, int32_t num_samples, const double* eps,
                               double* out);

/* Timing of the device work of the most recent attempt_step (HIP events on the handle's stream),
 * in milliseconds; used by bench.py for the roofline line. Needs pnode_set_phase_timing(h, 1). */
int pnode_last_step_ms(pnode_batch* h, float* ms_out);
/* Dense model only, A/B testing of the device paths (all of them HIP, none a CPU fallback): 0 = dispatch by shape
 * (fused one-wavefront kernel for (nu+1)*d <= 32, register-resident MFMA kernels, HBM-tiled large-D kernels, generic
 * kernels otherwise), 1 = generic kernels, 2 = large-D kernels (PNODE_EINVAL if the shape is not supported there),
 * 3 = MFMA kernels with the two-launch LQ (32-row passes, csrc/pnode_dense_mfma.hip) instead of the one-launch LQ with
 * overlapping 64-row passes (csrc/pnode_dense_lq4.hip), which is what 0 selects for (nu+1)*d = 128, 192, 256.
 * 4 = large-D kernels with every batched GEMM on the 64 x 64 tile (the 128 x 128 double-buffered tile of r04 switched off; identical bits).
 * The choice must be made before the first step after pnode_set_initial (PNODE_ESTATE afterwards): the paths keep the state
 * factor in different layouts.
 * No environment variable influences the library: the choice is per handle, through this call only. */
int pnode_set_generic_path(pnode_batch* h, int32_t enabled);
/* Isotropic model, strategy_fixedpoint (src/odecheckpts/ivpsolvers.py:43), d <= 64, device-resident controller
 * (pnode_solve_adaptive_save_at_device): which of two kernels runs the attempts of a checkpoint interval. 0 (default) = one
 * wavefront per trajectory (highest throughput when the batch fills the device with equal work); 1 = one workgroup per trajectory,
 * the filter and the controller on one wavefront and the backward transitions of the accepted steps on three others (the chain of
 * one trajectory is 1.6x shorter: single trajectories, tolerance sweeps, anything bound by its slowest member). Both kernels
 * evaluate the same expressions in the same order: the controller's decisions (accepted steps, step sizes) are identical bit for
 * bit, the smoothed means agree to rounding (1e-13 relative: multiply-adds of the backward transitions are contracted differently
 * by the compiler) -- hence a per-handle choice of the caller, never one the library makes from the batch size. Both kernels
 * exist for every nu <= 8; beyond d = 64 the setting is ignored (one wavefront, or one workgroup for d > 128, loops over the
 * ODE dimensions). */
int pnode_set_smoother_pipeline(pnode_batch* h, int32_t mode);
/* Diagnostics of the one-launch LQ of the dense MFMA path (csrc/pnode_dense_lq4.hip): its wavefronts hand panels over through
 * flag words in LDS, and every wait is a bounded spin. A wait that runs into its bound is reported by the call that synchronises
 * with the device (pnode_attempt_step, pnode_advance_device, pnode_solve_*) as PNODE_EDEVICE with the trajectory and the awaited
 * value in pnode_last_error -- never as NaN results; the batch must then be re-initialised. polls > 0 replaces the default
 * bound (2^21 polls) so that tests can provoke the failure; 0 restores it. On a handle of the isotropic model the call sets the
 * bound of the waits of the pipelined smoother controller (pnode_set_smoother_pipeline; ring and composition hand-shakes in LDS,
 * default 2^22 polls), reported the same way. */
int pnode_debug_set_lq_spin_limit(pnode_batch* h, int32_t polls);
/* Enables HIP-event timing of attempt_step (all models). Dense model: also record HIP events between the kernels of a step (on the handle's stream) and
 * read the durations of the last attempt, ms_out[4] = {begin, lq_input, lq, correct}. */
int pnode_set_phase_timing(pnode_batch* h, int32_t enabled);
int pnode_last_phase_ms(pnode_batch* h, float* ms_out);
/* The same four durations averaged over EVERY attempt of the most recent pnode_advance_device call (one set of HIP events per
 * round on the handle's stream); attempts_out (may be null) receives the number of attempts averaged. */
int pnode_phase_ms_mean(pnode_batch* h, float* ms_out, int64_t* attempts_out);
/* Raw stream handle (hipStream_t) for callers that need to order their own work after ours. */
void* pnode_stream(pnode_batch* h);
/* Device-resident controller as a building block (/root/reference/src/odecheckpts/ivpsolvers.py:52-53 is the
 * controller it restates): `rounds` batched attempts are enqueued back to back, each followed by a controller kernel that accepts
 * or rejects, applies the PI law and prepares the next attempt ON THE DEVICE; the host synchronises once, at the end.
 * status_out[b]: in < 0 = trajectory does not take part; out 0 = still short of t_stop[b], 1 = STEP_FROM.t + nz >= t_stop[b],
 * 2 = failed (non-finite). dt_inout / log_err_prev_inout carry the controller state (log of the previous accepted error norm,
 * 0 initially); attempts_out[b] = attempts made in this call. pnode_solve_adaptive_save_at_device uses it for the dense model
 * beyond (nu+1) d = 32 (strategy_filter and strategy_fixedpoint); bench.py uses it for the timed region of the headline. */
int pnode_advance_device(pnode_batch* h, const double* t_stop, double* dt_inout, double* log_err_prev_inout, const pnode_control* ctrl,
                         double numerical_zero, int32_t rounds, int32_t* status_out, int64_t* attempts_out);

#ifdef __cplusplus
}
#endif
#endif /* PNODE_H */
