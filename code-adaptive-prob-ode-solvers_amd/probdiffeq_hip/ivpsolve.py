"""Adaptive and fixed-grid IVP solution routines (probdiffeq.ivpsolve surface) for batches of IVPs.

Reference call sites: control_proportional_integral / adaptive (src/odecheckpts/ivpsolvers.py:52-53),
solve_adaptive_save_at (ivpsolvers.py:71-77, experiments/4_brusselator/run.py:122-129),
solve_adaptive_terminal_values (run.py:82-90), solve_adaptive_save_every_step and solve_fixed_grid
(experiments/1_van_der_pol/vdp.py:77-79,90).

Division of labour (BASELINE north star): the predict/correct step, the error norm and the checkpoint
interpolation run on the GPU for the whole batch at once; the PI step-size controller, the
accept/reject decision and the save_at cursor stay HERE, sequential per trajectory, vectorised over the
batch with NumPy. One host<->device round trip per attempted batched step.
"""

from __future__ import annotations

import dataclasses
import warnings

import numpy as np

from . import _backend
from .ivpsolvers import InitialCondition, Solver
from .vector_fields import UserVectorField, VectorField

# accepted steps per trajectory that the device-side store of strategy_smoother holds by default (solve_adaptive_save_every_step)
SMOOTHER_MAX_STEPS = 16384


@dataclasses.dataclass(frozen=True)
class PIControl:
    """Proportional-integral step-size control, same defaults as the reference's controller."""

    clip: bool = False
    safety: float = 0.95
    factor_min: float = 0.2
    factor_max: float = 10.0
    power_integral_unscaled: float = 0.3
    power_proportional_unscaled: float = 0.4

    def apply(self, dt, err_prev, err, rate):
        """Vectorised over the batch; `err` is the normalised error of the attempt just made."""
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            a1 = (1.0 / err) ** (self.power_integral_unscaled / rate)
            a2 = (err_prev / err) ** (self.power_proportional_unscaled / rate)
            factor = np.maximum(self.factor_min, np.minimum(self.safety * a1 * a2, self.factor_max))
        err_prev_new = np.where(err <= 1.0, err, err_prev)
        return factor * dt, err_prev_new


def control_proportional_integral(*, clip=False, safety=0.95, factor_min=0.2, factor_max=10.0,
                                  power_integral_unscaled=0.3, power_proportional_unscaled=0.4) -> PIControl:
    return PIControl(clip, safety, factor_min, factor_max, power_integral_unscaled, power_proportional_unscaled)


@dataclasses.dataclass(frozen=True)
class AdaptiveSolver:
    solver: Solver
    atol: object
    rtol: object
    control: PIControl
    numerical_zero: float = 1e-10


def adaptive(solver: Solver, atol=1e-4, rtol=1e-2, control=None, norm_ord=None, numerical_zero=1e-10):
    """atol / rtol may be scalars or arrays of shape (batch,): a tolerance sweep is a batch."""
    if norm_ord is not None:
        raise ValueError("only the default (RMS) error norm is implemented")
    if control is None:
        control = control_proportional_integral()
    return AdaptiveSolver(solver, atol, rtol, control, numerical_zero)


@dataclasses.dataclass
class Posterior:
    """Means of the posterior at the returned time points (+ the live device batch for more reads).

    For the strategies that store a backward model (fixedpoint: one transition per save_at interval; smoother: one per
    accepted step) `grid`, `num_filled` and `init_means` describe the store on the device, which
    `stats.offgrid_marginals_searchsorted` and `stats.markov_sample` evaluate there."""

    means: np.ndarray  # (B, T, n, d)
    kind: str  # "filter" | "fixedpoint" | "smoother"
    batch: object = None
    grid: np.ndarray = None  # (B, K+1) times of the stored entries, grid[:, 0] = t0 (NaN beyond num_filled)
    num_filled: np.ndarray = None  # (B,) stored entries per trajectory
    init_means: np.ndarray = None  # (B, n, d) mean of the initial condition
    index: int = None  # set when this object describes ONE trajectory of a batched solve

    def checkpoint_factors(self):
        """Upper covariance factors R at the checkpoints after t0 (Sigma = R^T R; isotropic: Sigma = R^T R (x) I): what
        `solution.posterior` of the reference carries per save_at point (ivpsolvers.py:71-81). Filter: the filtering
        covariance; fixedpoint (isotropic): the one-step-smoothed one of the checkpoint interpolation. (B, K, m, m), or
        (K, m, m) for one trajectory."""
        f = self.batch.read_checkpoint_factors()
        return f if self.index is None else f[self.index]


@dataclasses.dataclass
class Solution:
    t: np.ndarray
    u: np.ndarray
    posterior: Posterior
    output_scale: np.ndarray
    num_steps: np.ndarray
    num_attempts: np.ndarray = None


def _as_vf(vf):
    if not isinstance(vf, (VectorField, UserVectorField)):
        raise TypeError(
            "vector_field must be a probdiffeq_hip.vector_fields.VectorField (device code + analytic Jacobian "
            "selected by id) or a vector_fields.user(fun, d, ode_order) wrapper (host-evaluated escape hatch); a bare "
            "Python callable cannot run inside the HIP step kernel."
        )
    return vf


# The dense filter stores a D x D covariance factor per save_at point (Posterior.checkpoint_factors) by default up to this state
# dimension; beyond it on request (`checkpoint_factors=True`): the factor of an interpolated checkpoint costs a time update on
# the generic kernels (69 ms for 1024 trajectories at D = 256) and B x K x D^2 x 8 bytes.
CHECKPOINT_FACTORS_UP_TO_D = 64


def _make_batch(vf, init: InitialCondition, solver: Solver, num_checkpoints, device=None, checkpoint_factors=None):
    vf = _as_vf(vf)
    st = solver.strategy
    B, n, d = init.tcoeffs.shape
    if n != st.prior.num_derivatives + 1:
        raise ValueError("initial condition does not match the prior's number of derivatives")
    if d != vf.d or st.correction.ode_order != vf.ode_order:
        raise ValueError(f"vector field {vf.name!r} has d={vf.d}, order={vf.ode_order}; solver was built for "
                         f"d={d}, order={st.correction.ode_order}")
    if st.ode_shape not in ((), (d,)):
        raise ValueError(f"impl.select ode_shape {st.ode_shape} does not match the initial condition (d={d})")
    kind = st.kind
    if kind == "smoother":
        # one stored backward transition per accepted step: the fixed-point machinery with a checkpoint at every step
        # (solve_adaptive_save_every_step drives it); the store holds num_checkpoints steps
        if st.model != "isotropic":
            raise NotImplementedError("strategy_smoother is implemented for the isotropic model (the reference's use, "
                                      "src/odecheckpts/ivpsolvers.py:102-112); use strategy_fixedpoint with the dense model")
        kind = "fixedpoint"
    prm = vf.param_array()
    batch = _backend.Batch(
        batch=B, d=d, nu=st.prior.num_derivatives, ode_order=st.correction.ode_order, model=st.model,
        correction=st.correction.kind, strategy=kind, calibration=solver.calibration, vf_id=vf.vf_id,
        vf_params=prm, num_checkpoints=num_checkpoints, device=device,
        checkpoint_factors=(st.model == "dense" and kind == "filter" and num_checkpoints > 0
                            and (n * d <= CHECKPOINT_FACTORS_UP_TO_D if checkpoint_factors is None else bool(checkpoint_factors))),
    )
    return batch


class _AdaptiveLoop:
    """Host-sequential controller state for one batch."""

    def __init__(self, batch, asolver: AdaptiveSolver, t0, dt0, vf=None):
        B = batch.B
        self.batch = batch
        self.user_vf = vf if isinstance(vf, UserVectorField) else None
        self.needs_jac = asolver.solver.strategy.correction.kind == "ts1"
        self.ctrl = asolver.control
        self.rate = asolver.solver.error_contraction_rate
        self.nz = asolver.numerical_zero
        self.t = np.array(np.broadcast_to(t0, (B,)), dtype=np.float64)
        self.t_prev = self.t.copy()
        self.dt = np.array(np.broadcast_to(dt0, (B,)), dtype=np.float64)
        self.err_prev = np.ones(B)
        self.steps = np.zeros(B, dtype=np.int64)
        self.attempts = np.zeros(B, dtype=np.int64)
        self.failed = np.zeros(B, dtype=bool)
        self._err = np.zeros(B)
        self.time_device = False
        self.device_ms = 0.0
        self.num_launches = 0

    def attempt(self, active, t1=None):
        """One batched attempted step for the trajectories in `active`; returns the accept mask."""
        dt = self.dt
        if self.ctrl.clip and t1 is not None:
            dt = np.where(active, np.minimum(dt, t1 - self.t), dt)
        dt_used = np.where(active, dt, 1.0)
        if self.user_vf is not None:
            # host-evaluated vector field: fetch the extrapolated means, evaluate f (and J), hand them to the device
            k = self.user_vf.ode_order
            y = self.batch.user_extrapolate(dt_used, active, k)
            ys = [np.ascontiguousarray(y[:, i]) for i in range(k)]
            tq = self.t + dt_used
            f = np.asarray(self.user_vf(*ys, t=tq[:, None]), dtype=np.float64)
            jac = self.user_vf.jacobian(ys, tq[:, None]) if self.needs_jac else None
            self.batch.user_set_vf(f, jac)
        err = self.batch.attempt_step(dt_used, active, self._err)
        if self.time_device:
            self.device_ms += self.batch.last_step_ms()
        self.num_launches += 1
        self.attempts += active
        new_dt, new_prev = self.ctrl.apply(dt_used, self.err_prev, err, self.rate)
        # the reference's rejection loop runs `while error_norm > 1`: a NaN error leaves the loop
        accept = active & ~(err > 1.0)
        self.dt = np.where(active, new_dt, self.dt)
        self.err_prev = np.where(active, new_prev, self.err_prev)
        self.batch.commit(accept)
        self.t_prev = np.where(accept, self.t, self.t_prev)
        self.t = np.where(accept, self.t + dt_used, self.t)
        self.steps += accept
        bad = active & ~(np.isfinite(self.dt) & np.isfinite(self.t) & (self.dt > 0))
        self.failed |= bad
        return accept


def _finish(sol_means, init, kind, batch, t, num_steps, loop, squeeze):
    u = sol_means[:, :, 0, :]
    out_scale = batch.read_output_scale(0)
    post = Posterior(sol_means, kind, batch)
    if squeeze:
        return Solution(t if t.ndim == 1 else t[0], u[0], post, out_scale[0], num_steps[0], loop.attempts[0])
    return Solution(t, u, post, out_scale, num_steps, loop.attempts.copy())


def _python_save_at_loop(batch, loop, sa_b, K, nz):
    """The driver loop in Python/NumPy (reference implementation of the native pnode_solve_adaptive_save_at)."""
    B = batch.B
    cursor = np.ones(B, dtype=np.int64)
    num_steps = np.zeros((B, K + 1), dtype=np.int64)
    rows = np.arange(B)
    while True:
        while True:
            live = (cursor <= K) & ~loop.failed
            nxt = sa_b[rows, np.minimum(cursor, K)]
            pend = live & (loop.t + nz >= nxt)
            if not pend.any():
                break
            over = pend & (loop.t > nxt + nz)
            corner = pend & ~over
            if over.any():
                batch.interpolate(np.where(over, nxt, 0.0), np.where(over, cursor - 1, -1))
            if corner.any():
                batch.checkpoint_at_step(np.where(corner, cursor - 1, -1))
            num_steps[rows[pend], cursor[pend]] = loop.steps[pend]
            cursor = cursor + pend
        active = (cursor <= K) & ~loop.failed
        if not active.any():
            break
        t_next = sa_b[rows, np.minimum(cursor, K)]
        loop.attempt(active, t1=t_next)
    return cursor, num_steps


def solve_adaptive_save_at(vector_field, initial_condition: InitialCondition, save_at, adaptive_solver: AdaptiveSolver,
                           dt0, *, device=None, python_controller=False, device_controller=False,
                           checkpoint_factors=None) -> Solution:
    """Adaptive solve that stores the solution at `save_at` only (fixed memory, O(len(save_at))).

    save_at: (K+1,) shared or (B, K+1) per trajectory; save_at[..., 0] is the initial time. Steps are
    not clipped (unless control.clip): an accepted step that oversteps a checkpoint is interpolated.
    With strategy_fixedpoint, `Solution.u` holds the smoothed means (backward pass over the stored
    checkpoint transitions), as probdiffeq's solution object does.

    `checkpoint_factors`: the dense filter also stores the covariance factor at every save_at point
    (`solution.posterior.checkpoint_factors()`); default: yes up to (nu+1) d = 64, on request beyond.

    Controller placement: by default the PI controller and the accept/reject decisions run on the host (native
    loop inside the C-ABI; `python_controller=True` runs the same loop in NumPy). `device_controller=True`
    moves them to the device: one host synchronisation per checkpoint interval (isotropic model, dense model with (nu+1) d <= 32: inside
    the step kernel; the other dense paths: a controller kernel between the attempts). For the isotropic fixed-point smoother with
    d <= 64 there are two such kernels with identical decisions (smoothed means equal to rounding): `device_controller="pipelined"` (a workgroup per trajectory, the backward
    transitions beside the filter: shortest chain per trajectory -- single trajectories, tolerance sweeps) and `"one-wavefront"` (= `True`,
    the default: highest throughput for a batch of equal trajectories that fills the device); include/pnode.h, pnode_set_smoother_pipeline.
    """
    solver = adaptive_solver.solver
    if not solver.strategy.is_suitable_for_save_at:
        raise ValueError("strategy_smoother stores every step: use solve_adaptive_save_every_step (+ "
                         "stats.offgrid_marginals_searchsorted), or strategy_fixedpoint with solve_adaptive_save_at")
    init = initial_condition
    B, n, d = init.tcoeffs.shape
    sa = np.asarray(save_at, dtype=np.float64)
    sa_b = np.ascontiguousarray(np.broadcast_to(sa, (B, sa.shape[-1])))
    K = sa_b.shape[1] - 1
    if K < 1 or np.any(np.diff(sa_b, axis=1) <= 0):
        raise ValueError("save_at must be strictly increasing with at least two points")
    batch = _make_batch(vector_field, init, solver, K, device, checkpoint_factors)
    batch.set_initial(init.tcoeffs, sa_b[:, 0], init.output_scale)
    batch.set_tolerances(adaptive_solver.atol, adaptive_solver.rtol)
    if isinstance(device_controller, str):
        if device_controller not in ("pipelined", "one-wavefront"):
            raise ValueError("device_controller: True, False, 'pipelined' or 'one-wavefront'")
        if solver.strategy.model == "isotropic":
            batch.set_smoother_pipeline(1 if device_controller == "pipelined" else 0)
        device_controller = True
    loop = _AdaptiveLoop(batch, adaptive_solver, sa_b[:, 0], dt0, vf=vector_field)
    if python_controller or loop.user_vf is not None:
        cursor, num_steps = _python_save_at_loop(batch, loop, sa_b, K, adaptive_solver.numerical_zero)
    else:
        # native driver: same controller arithmetic and call sequence, in C++ inside libpnode_hip.so
        num_steps, attempts, failed = batch.solve_adaptive_save_at(sa if sa.ndim == 1 else sa_b, loop.dt, adaptive_solver.control,
                                                                   adaptive_solver.numerical_zero,
                                                                   device_controller=device_controller)
        loop.attempts, loop.failed = attempts, failed
        cursor = np.where(failed, 1 + np.argmax(np.concatenate([num_steps[:, 1:] == 0, np.ones((B, 1), bool)], axis=1), axis=1),
                          K + 1)
    if solver.strategy.kind == "fixedpoint":
        means = batch.smooth_checkpoint_means(np.minimum(cursor - 1, K))
        # the smoothed value at t0 equals the initial mean up to rounding; keep the backward-pass value
    else:
        means = np.concatenate([init.tcoeffs[:, None], batch.read_checkpoint_means()], axis=1)
    if loop.failed.any():
        for b in np.nonzero(loop.failed)[0]:
            means[b, cursor[b]:] = np.nan
            if cursor[b] <= 1:  # failed before its first checkpoint: the backward pass had nothing to start from and left
                means[b, 0] = init.tcoeffs[b]  # the zero-initialised output; the value at t0 is the initial mean
    sol = _finish(means, init, solver.strategy.kind, batch, sa if sa.ndim == 1 else sa_b, num_steps, loop,
                  not init.batched)
    sol.posterior.index = None if init.batched else 0
    if solver.strategy.kind == "fixedpoint":
        sol.posterior.grid = sa_b
        sol.posterior.num_filled = np.minimum(cursor - 1, K).astype(np.int32)
        sol.posterior.init_means = init.tcoeffs
        sol.posterior.index = None if init.batched else 0
    return sol


def solve_adaptive_terminal_values(vector_field, initial_condition: InitialCondition, t0, t1,
                                   adaptive_solver: AdaptiveSolver, dt0, *, device=None) -> Solution:
    """Adaptive solve returning only the state at t1 (experiments/4_brusselator/run.py:82-90)."""
    B = initial_condition.tcoeffs.shape[0]
    sa = np.stack([np.broadcast_to(np.asarray(t0, dtype=np.float64), (B,)),
                   np.broadcast_to(np.asarray(t1, dtype=np.float64), (B,))], axis=1)
    sol = solve_adaptive_save_at(vector_field, dataclasses.replace(initial_condition, batched=True), sa,
                                 adaptive_solver, dt0, device=device)
    post = Posterior(sol.posterior.means[:, -1:], sol.posterior.kind, sol.posterior.batch)
    if not initial_condition.batched:
        return Solution(sa[0, 1], sol.u[0, -1], post, sol.output_scale[0], sol.num_steps[0, -1], sol.num_attempts[0])
    return Solution(sa[:, 1], sol.u[:, -1], post, sol.output_scale, sol.num_steps[:, -1], sol.num_attempts)


def _save_every_step_smoother(vector_field, init, t0b, t1b, adaptive_solver, dt0, device, max_steps):
    """strategy_smoother: every accepted step is stored on the device with its own backward transition (memory O(#steps),
    the 'textbook' baseline of experiments/4_brusselator/run.py:102-117 and src/odecheckpts/ivpsolvers.py:112,133-142);
    the solution is the backward pass over all of them."""
    solver = adaptive_solver.solver
    B = init.tcoeffs.shape[0]
    batch = _make_batch(vector_field, init, solver, max_steps, device)
    batch.set_initial(init.tcoeffs, t0b, init.output_scale)
    batch.set_tolerances(adaptive_solver.atol, adaptive_solver.rtol)
    loop = _AdaptiveLoop(batch, adaptive_solver, t0b, dt0, vf=vector_field)
    grid = np.full((B, max_steps + 1), np.nan)
    grid[:, 0] = t0b
    count = np.zeros(B, dtype=np.int64)
    done = np.zeros(B, dtype=bool)
    rows = np.arange(B)
    while True:
        active = ~done & ~loop.failed
        if not active.any():
            break
        if np.any(count[active] >= max_steps):
            batch.close()  # (the store of max_steps transitions is device memory: do not leave it to the garbage collector)
            raise RuntimeError(f"strategy_smoother stores every accepted step and max_steps={max_steps} is used up; "
                               "raise max_steps or use strategy_fixedpoint with solve_adaptive_save_at (fixed memory)")
        acc = loop.attempt(active, t1=t1b)
        over = acc & (loop.t > t1b)  # (same end-point rule as the filter's driver below: exact comparisons)
        corner = acc & ~over
        if corner.any():
            batch.checkpoint_at_step(np.where(corner, count, -1))
            grid[rows[corner], count[corner] + 1] = loop.t[corner]
        if over.any():  # the last step oversteps t1: the solution at t1 is interpolated, the overshooting state dropped
            batch.interpolate(np.where(over, t1b, 0.0), np.where(over, count, -1))
            grid[rows[over], count[over] + 1] = t1b[over]
        count = count + acc
        done |= over | (corner & (loop.t >= t1b))
    means = batch.smooth_checkpoint_means(count.astype(np.int32))
    scale = batch.read_output_scale(0)
    sols = []
    for b in range(B):
        T = int(count[b]) + 1
        m = means[b, :T].copy()
        if loop.failed[b]:
            m[:] = np.nan
        post = Posterior(m[None], "smoother", batch, grid, count.astype(np.int32), init.tcoeffs, b)
        sols.append(Solution(grid[b, :T].copy(), m[:, 0, :], post, scale[b], np.arange(T).clip(max=int(loop.steps[b])),
                             int(loop.attempts[b])))
    return sols if init.batched else sols[0]


def solve_adaptive_save_every_step(vector_field, initial_condition: InitialCondition, t0, t1,
                                   adaptive_solver: AdaptiveSolver, dt0, *, device=None, max_steps=None):
    """Adaptive solve that returns every accepted step (vdp.py:77-79). Ragged over the batch: returns a
    Solution for an unbatched initial condition and a list of Solutions otherwise. Meant for small
    problems: each accepted batched step copies the means back to the host (strategy_filter) or stores the step with its
    backward transition on the device (strategy_smoother, at most `max_steps` accepted steps per trajectory)."""
    solver = adaptive_solver.solver
    if solver.strategy.kind == "smoother":
        B0 = initial_condition.tcoeffs.shape[0]
        return _save_every_step_smoother(vector_field, initial_condition, np.array(np.broadcast_to(t0, (B0,)), dtype=np.float64),
                                         np.array(np.broadcast_to(t1, (B0,)), dtype=np.float64), adaptive_solver, dt0, device,
                                         int(SMOOTHER_MAX_STEPS if max_steps is None else max_steps))
    if solver.strategy.kind != "filter":
        raise NotImplementedError("solve_adaptive_save_every_step is implemented for strategy_filter (vdp.py:65) and "
                                  "strategy_smoother (ivpsolvers.py:112); use solve_adaptive_save_at with fixedpoint")
    init = initial_condition
    B, n, d = init.tcoeffs.shape
    t0b = np.array(np.broadcast_to(t0, (B,)), dtype=np.float64)
    t1b = np.array(np.broadcast_to(t1, (B,)), dtype=np.float64)
    batch = _make_batch(vector_field, init, solver, 1, device)
    batch.set_initial(init.tcoeffs, t0b, init.output_scale)
    batch.set_tolerances(adaptive_solver.atol, adaptive_solver.rtol)
    loop = _AdaptiveLoop(batch, adaptive_solver, t0b, dt0, vf=vector_field)
    ts = [[t0b[b]] for b in range(B)]
    ms = [[init.tcoeffs[b]] for b in range(B)]
    while True:
        active = (loop.t < t1b) & ~loop.failed
        if not active.any():
            break
        acc = loop.attempt(active, t1=t1b)
        inside = acc & (loop.t < t1b)
        if inside.any():
            means = batch.read_mean(0)
            for b in np.nonzero(inside)[0]:
                ts[b].append(loop.t[b])
                ms[b].append(means[b])
    over = (loop.t > t1b) & ~loop.failed
    corner = ~over & ~loop.failed
    if over.any():
        batch.interpolate(np.where(over, t1b, 0.0), np.where(over, 0, -1))
    if corner.any():
        batch.checkpoint_at_step(np.where(corner, 0, -1))
    last = batch.read_checkpoint_means()[:, 0]
    scale = batch.read_output_scale(0)
    sols = []
    for b in range(B):
        ts[b].append(t1b[b])
        ms[b].append(last[b] if not loop.failed[b] else np.full_like(last[b], np.nan))
        means = np.stack(ms[b])
        sols.append(Solution(np.asarray(ts[b]), means[:, 0, :], Posterior(means[None], "filter", batch), scale[b],
                             np.arange(len(ts[b])).clip(max=int(loop.steps[b])), int(loop.attempts[b])))
    return sols if init.batched else sols[0]


def _fixed_grid_smoother(vector_field, init, gb, solver, device):
    """strategy_smoother on a prescribed grid (experiments/5_vs_interpolation/measure.py:129-140: the 'textbook' runner re-solves
    on the union of its adaptive grid and save_at): every step is stored with its backward transition, the solution is the
    backward pass over all of them; `solution.posterior` feeds stats.markov_sample. Isotropic model (the store is the fixed-point
    kernels' without the merge), one host round trip per step."""
    B, G = gb.shape
    if np.any(np.diff(gb, axis=1) <= 0):
        raise ValueError("grid must be strictly increasing")
    batch = _make_batch(vector_field, init, solver, G - 1, device)
    batch.set_initial(init.tcoeffs, np.ascontiguousarray(gb[:, 0]), init.output_scale)
    err, one = np.zeros(B), np.ones(B, np.uint8)
    count = np.zeros(B, dtype=np.int64)
    for g in range(1, G):
        batch.attempt_step(np.ascontiguousarray(gb[:, g] - gb[:, g - 1]), one, err)
        batch.commit(one)
        batch.checkpoint_at_step(count)
        count = count + 1
    means = batch.smooth_checkpoint_means(count.astype(np.int32))
    scale = batch.read_output_scale(0)
    sols = []
    for b in range(B):
        m = means[b, :G].copy()
        post = Posterior(m[None], "smoother", batch, gb.copy(), count.astype(np.int32), init.tcoeffs, b)
        sols.append(Solution(gb[b].copy(), m[:, 0, :], post, scale[b], np.arange(G), G - 1))
    return sols if init.batched else sols[0]


def solve_fixed_grid(vector_field, initial_condition: InitialCondition, grid, solver: Solver, *, device=None) -> Solution:
    """Pure predict/correct loop on a prescribed grid, no controller (vdp.py:88-103). All launches are
    enqueued on the device stream without host synchronisation. grid: (G,) shared or (B, G)."""
    init = initial_condition
    B, n, d = init.tcoeffs.shape
    grid = np.asarray(grid, dtype=np.float64)
    gb = np.broadcast_to(grid, (B, grid.shape[-1]))
    if solver.strategy.kind == "smoother":
        return _fixed_grid_smoother(vector_field, init, np.ascontiguousarray(gb), solver, device)
    batch = _make_batch(vector_field, init, solver, 0, device)
    batch.set_initial(init.tcoeffs, np.ascontiguousarray(gb[:, 0]), init.output_scale)
    u = batch.solve_fixed_grid(grid if grid.ndim == 1 else np.ascontiguousarray(gb))
    G = gb.shape[1]
    steps = np.broadcast_to(np.arange(G), (B, G))
    post = Posterior(batch.read_mean(0)[:, None], solver.strategy.kind, batch)
    scale = batch.read_output_scale(0)
    if not init.batched:
        return Solution(grid, u[0], post, scale[0], steps[0], G - 1)
    return Solution(grid, u, post, scale, steps, np.full(B, G - 1))
