"""CPU ORACLE (test infrastructure, NOT product code) for the EKF predict/correct stepping path.

A plain NumPy fp64 restatement of the algorithm the reference runs through `probdiffeq`
(third-party, un-vendored, version unpinned: /root/reference/pyproject.toml:10). The arithmetic
is re-derived from the published design (Kraemer 2024; Kraemer & Hennig 2020/2024; Bosch et al.
2021) and anchored on the reference's call sites:

  prior_ibm / correction_ts0 / strategy_fixedpoint / solver_dynamic / adaptive
                               /root/reference/src/odecheckpts/ivpsolvers.py:32-53
  solve_adaptive_save_at       /root/reference/src/odecheckpts/ivpsolvers.py:71-77
                               /root/reference/experiments/4_brusselator/run.py:119-138
  solve_adaptive_terminal_values  /root/reference/experiments/4_brusselator/run.py:81-94
  dense / correction_ts1 / strategy_filter / solve_adaptive_save_every_step / solve_fixed_grid
                               /root/reference/experiments/1_van_der_pol/vdp.py:61-91

PARITY PIN: this file is validated by tests/test_oracle_golden.py against the reference's own
committed outputs (tests/golden/*.npz, produced by tests/golden/make_fixtures.py):
  * Van der Pol mu=1e3, dense EKF1 nu=4: the 2 912-point accepted grid and filtered means;
  * Brusselator N=2..64, isotropic EKF0 nu=4 fixed-point smoother: `num_steps` and the 200
    checkpoint means.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Conventions: nu derivatives, n = nu + 1, d = ODE dimension. Isotropic states hold a mean of
shape (n, d) and ONE n x n Cholesky factor shared by all d dimensions (Sigma = L L^T (x) I_d).
Dense states hold the same (n, d) mean, flattened derivative-major (index i*d + j), and a
D x D factor, D = n*d.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Callable, NamedTuple

import numpy as np

# --------------------------------------------------------------------------------------------
# Integrated Wiener process prior (ivpsolvers.prior_ibm: ivpsolvers.py:42, run.py:57, vdp.py:63)
# --------------------------------------------------------------------------------------------


def iwp_system_matrices(nu: int):
    """Preconditioned 1-D IWP(nu) transition and process-noise Cholesky factor.

    A[i, j] = binom(nu - i, nu - j) (flipped Pascal, upper triangular),
    Q[i, j] = 1 / (2 nu + 1 - i - j) (flipped Hilbert). Both are dt-free: the step size
    lives in the diagonal preconditioner (`iwp_preconditioner`).
    """
    n = nu + 1
    a = np.zeros((n, n))
    q = np.zeros((n, n))
    for i in range(n):
        for j in range(n):
            if j >= i:
                a[i, j] = math.comb(nu - i, nu - j)
            q[i, j] = 1.0 / (2 * nu + 1 - i - j)
    return a, np.linalg.cholesky(q)


def iwp_preconditioner(dt: float, nu: int):
    """p_i = |dt|^(nu - i + 1/2) / (nu - i)!  and its inverse."""
    powers = np.arange(nu, -1.0, -1.0)
    scales = np.array([math.factorial(int(k)) for k in powers], dtype=np.float64)
    p = np.abs(dt) ** (powers + 0.5) / scales
    p_inv = np.abs(dt) ** (-(powers + 0.5)) * scales
    return p, p_inv


def qr_r(mat):
    """Upper-triangular factor of a (tall) matrix. Signs of rows are not normalised."""
    return np.linalg.qr(mat, mode="r")


# --------------------------------------------------------------------------------------------
# State containers
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class Conditional:
    """Backward transition x_prev | x_next ~ N(G x_next + xi, Lam Lam^T).

    Isotropic: G, Lam are n x n and xi is (n, d). Dense: D x D and (D,).
    """

    G: np.ndarray
    xi: np.ndarray
    Lam: np.ndarray


@dataclasses.dataclass
class State:
    t: float
    mean: np.ndarray  # (n, d)
    chol: np.ndarray  # (n, n) isotropic  |  (D, D) dense
    output_scale: float
    num_steps: int
    cond: Conditional | None = None  # fixed-point / smoother backward model


class Problem(NamedTuple):
    """Vector field for an ODE of order k: u^(k) = f(u, ..., u^(k-1), t).

    f(y[k, d], t) -> (d,);   jac(y[k, d], t) -> (d, k, d)  with jac[a, i, b] = d f_a / d y[i, b].
    """

    f: Callable
    jac: Callable | None
    ode_order: int
    d: int


# --------------------------------------------------------------------------------------------
# Solver configuration (mirrors the probdiffeq construction sequence of ivpsolvers.py:32-53)
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class SolverConfig:
    nu: int
    model: str = "isotropic"  # impl.select(...)            "isotropic" | "dense"
    correction: str = "ts0"  # correction_ts0 / correction_ts1
    strategy: str = "filter"  # strategy_filter / strategy_fixedpoint / strategy_smoother
    calibration: str = "dynamic"  # solver_dynamic / solver ("none")

    def __post_init__(self):
        if self.model not in ("isotropic", "dense"):
            raise ValueError(self.model)
        if self.correction not in ("ts0", "ts1"):
            raise ValueError(self.correction)
        if self.strategy not in ("filter", "fixedpoint", "smoother"):
            raise ValueError(self.strategy)
        if self.calibration not in ("dynamic", "none"):
            raise ValueError(self.calibration)
        if self.model == "isotropic" and self.correction == "ts1":
            raise ValueError("ts1 needs the dense model (Jacobian couples the dimensions)")
        self.a, self.q_sqrt = iwp_system_matrices(self.nu)

    @property
    def n(self):
        return self.nu + 1


def initial_state(cfg: SolverConfig, tcoeffs, t0: float, output_scale: float = 1.0) -> State:
    """solver.initial_condition(tcoeffs, output_scale): exact derivatives, zero covariance."""
    mean = np.stack([np.atleast_1d(np.asarray(c, dtype=np.float64)) for c in tcoeffs])
    n, d = mean.shape
    assert n == cfg.n, (n, cfg.n)
    size = n if cfg.model == "isotropic" else n * d
    cond = None
    if cfg.strategy != "filter":
        cond = _identity_conditional(cfg, d)
    return State(t0, mean, np.zeros((size, size)), float(output_scale), 0, cond)


def _identity_conditional(cfg, d):
    size = cfg.n if cfg.model == "isotropic" else cfg.n * d
    xi = np.zeros((cfg.n, d)) if cfg.model == "isotropic" else np.zeros(cfg.n * d)
    return Conditional(np.eye(size), xi, np.zeros((size, size)))


# --------------------------------------------------------------------------------------------
# One solver step: extrapolate -> linearise -> error estimate / calibrate -> complete -> correct
# --------------------------------------------------------------------------------------------


def _kron_apply(mat_nn, x, d, dense):
    """Apply (mat (x) I_d) to the rows of a factor / a mean."""
    if not dense:
        return mat_nn @ x
    n = x.shape[0] // d
    return (mat_nn @ x.reshape(n, -1)).reshape(x.shape)


def _scale_rows(vec_n, x, d, dense):
    if not dense:
        return vec_n[:, None] * x
    return np.repeat(vec_n, d)[:, None] * x


def _linearise(cfg, prob, m_ext, t):
    """Observation model z = H x + b at the extrapolated mean; returns residual z and (k, J)."""
    k = prob.ode_order
    fx = prob.f(m_ext[:k], t)
    z = m_ext[k] - fx
    jac = prob.jac(m_ext[:k], t) if cfg.correction == "ts1" else None
    return z, jac


def _observe_factor(cfg, prob, jac, factor, d):
    """F = H @ factor for H = E_k - J E_{<k}.  Isotropic: (n_cols,) row; dense: (d, cols)."""
    k = prob.ode_order
    if cfg.model == "isotropic":
        return factor[k]
    f = factor[k * d : (k + 1) * d].copy()
    if jac is not None:
        for i in range(k):
            f -= jac[:, i, :] @ factor[i * d : (i + 1) * d]
    return f


def _extrapolate_cov(cfg, chol_prev, p, p_inv, sigma, d, want_cond):
    """Square-root time update in preconditioned coordinates (+ backward model)."""
    dense = cfg.model == "dense"
    l_p = _scale_rows(p_inv, chol_prev, d, dense)
    al = _kron_apply(cfg.a, l_p, d, dense)  # A L
    q = sigma * (np.kron(cfg.q_sqrt, np.eye(d)) if dense else cfg.q_sqrt)
    size = al.shape[0]
    if not want_cond:
        r = qr_r(np.concatenate([al.T, q.T], axis=0))
        return _scale_rows(p, r.T, d, dense), None
    # revert_conditional: QR of [[Q^T, 0], [(A L)^T, L^T]] -> [[R_ext, G-ish], [0, R_bw]]
    top = np.concatenate([q.T, np.zeros((size, size))], axis=1)
    bot = np.concatenate([al.T, l_p.T], axis=1)
    r = qr_r(np.concatenate([top, bot], axis=0))
    r_ext = r[:size, :size]
    r12 = r[:size, size:]
    r_bw = r[size : 2 * size, size:]
    # gain G_p solves R_ext^T R_ext G_p^T = A Sigma  =>  G_p = (R_ext^{-1} r12)^T
    g_p = np.linalg.solve(r_ext, r12).T if np.all(np.diag(r_ext) != 0) else np.linalg.lstsq(r_ext, r12, rcond=None)[0].T
    l_ext = _scale_rows(p, r_ext.T, d, dense)
    pv = np.repeat(p, d) if dense else p
    piv = np.repeat(p_inv, d) if dense else p_inv
    g = pv[:, None] * g_p * piv[None, :]
    lam = pv[:, None] * r_bw.T
    return l_ext, (g, lam)


def _merge_conditionals(cfg, bw0: Conditional, bw1: Conditional, d) -> Conditional:
    """Compose x_ckpt | x_prev (bw0) with x_prev | x_new (bw1) into x_ckpt | x_new."""
    g = bw0.G @ bw1.G
    if cfg.model == "isotropic":
        xi = bw0.G @ bw1.xi + bw0.xi
    else:
        xi = bw0.G @ bw1.xi + bw0.xi
    lam = qr_r(np.concatenate([(bw0.G @ bw1.Lam).T, bw0.Lam.T], axis=0)).T
    return Conditional(g, xi, lam)


class StepResult(NamedTuple):
    error: np.ndarray  # dt * sigma_hat * std  (scalar for isotropic, (d,) for dense)
    state: State


def _predict(cfg, prob, mean, chol, cond, dt, sigma, d, t_new):
    """Mean + factor extrapolation with a given output scale (used by interpolation too)."""
    dense = cfg.model == "dense"
    p, p_inv = iwp_preconditioner(dt, cfg.nu)
    m_p = p_inv[:, None] * mean
    m_ext_p = cfg.a @ m_p
    m_ext = p[:, None] * m_ext_p
    want_cond = cond is not None
    l_ext, bw = _extrapolate_cov(cfg, chol, p, p_inv, sigma, d, want_cond)
    new_cond = None
    if want_cond:
        g, lam = bw
        if dense:
            xi = mean.reshape(-1) - g @ m_ext.reshape(-1)
        else:
            xi = mean - g @ m_ext
        new_cond = _merge_conditionals(cfg, cond, Conditional(g, xi, lam), d)
    return m_ext, l_ext, new_cond


def step(cfg: SolverConfig, prob: Problem, state: State, dt: float) -> StepResult:
    """solver.step / solver_dynamic.step (probdiffeq), dense or isotropic, ts0 or ts1."""
    d = prob.d
    k = prob.ode_order
    dense = cfg.model == "dense"
    t_new = state.t + dt

    # begin(): mean extrapolation; covariance = unit-scale process noise only
    p, p_inv = iwp_preconditioner(dt, cfg.nu)
    m_ext = p[:, None] * (cfg.a @ (p_inv[:, None] * state.mean))
    l_begin = p[:, None] * cfg.q_sqrt
    if dense:
        l_begin = np.kron(l_begin, np.eye(d))

    # linearise + error estimate with local calibration
    z, jac = _linearise(cfg, prob, m_ext, t_new)
    f1 = _observe_factor(cfg, prob, jac, l_begin, d)
    if dense:
        s1_sqrt_t = qr_r(f1.T)  # upper; S1 = R^T R
        white = np.linalg.solve(s1_sqrt_t.T, z)
        sigma_hat = np.linalg.norm(white) / math.sqrt(d)
        std = np.sqrt(np.einsum("ij,ij->i", f1, f1))
    else:
        s1 = np.linalg.norm(f1)
        sigma_hat = np.linalg.norm(z / s1) / math.sqrt(d)
        std = s1
    error = dt * sigma_hat * std

    sigma = sigma_hat if cfg.calibration == "dynamic" else state.output_scale

    # complete(): full square-root extrapolation (+ backward model for fixedpoint / smoother)
    cond_in = state.cond if cfg.strategy == "fixedpoint" else (
        _identity_conditional(cfg, d) if cfg.strategy == "smoother" else None
    )
    m_ext2, l_ext, new_cond = _predict(cfg, prob, state.mean, state.chol, cond_in, dt, sigma, d, t_new)
    assert np.array_equal(m_ext2, m_ext, equal_nan=True)

    # correction: condition on z = 0 (noise-free)
    f = _observe_factor(cfg, prob, jac, l_ext, d)
    if dense:
        r_s = qr_r(f.T)
        cross = l_ext @ f.T  # D x d
        gain = np.linalg.solve(r_s, np.linalg.solve(r_s.T, cross.T)).T
        mean = (m_ext.reshape(-1) - gain @ z).reshape(m_ext.shape)
        chol = qr_r((l_ext - gain @ f).T).T
    else:
        s2 = f @ f
        gain = (l_ext @ f) / s2
        mean = m_ext - np.outer(gain, z)
        chol = qr_r((l_ext - np.outer(gain, f)).T).T
    new = State(t_new, mean, chol, sigma, state.num_steps + 1, new_cond)
    return StepResult(error, new)


# --------------------------------------------------------------------------------------------
# Adaptive step-size selection (ivpsolve.adaptive + control_proportional_integral)
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass
class PIControl:
    """ivpsolve.control_proportional_integral() defaults (no clipping)."""

    safety: float = 0.95
    factor_min: float = 0.2
    factor_max: float = 10.0
    power_integral_unscaled: float = 0.3
    power_proportional_unscaled: float = 0.4
    clip: bool = False

    def apply(self, dt, err_prev_accepted, error_normalised, rate):
        n1 = self.power_integral_unscaled / rate
        n2 = self.power_proportional_unscaled / rate
        a1 = (1.0 / error_normalised) ** n1
        a2 = (err_prev_accepted / error_normalised) ** n2
        factor = max(self.factor_min, min(self.safety * a1 * a2, self.factor_max))
        if error_normalised <= 1.0:
            err_prev_accepted = error_normalised
        return factor * dt, err_prev_accepted


@dataclasses.dataclass
class AdaptiveConfig:
    atol: float
    rtol: float
    control: PIControl = dataclasses.field(default_factory=PIControl)
    numerical_zero: float = 1e-10
    # which states enter the error normalisation  max(|u_proposed|, |u_ref|):
    #   "proposed": u_ref = u_proposed  (what reproduces the reference's committed grids)
    #   "step_from": u_ref = u at the start of the step
    reference_state: str = "proposed"


@dataclasses.dataclass
class AdaptiveState:
    step_from: State
    interp_from: State
    dt: float
    err_prev: float
    num_attempts: int = 0


def normalise_error(acfg: AdaptiveConfig, error, u_prop, u_from):
    if acfg.reference_state == "proposed":
        u = np.abs(u_prop)
    else:
        u = np.maximum(np.abs(u_prop), np.abs(u_from))
    rel = error / (acfg.atol + acfg.rtol * u)
    return float(np.linalg.norm(np.atleast_1d(rel)) / math.sqrt(u.size))


def adaptive_init(state: State, dt0: float) -> AdaptiveState:
    return AdaptiveState(state, state, float(dt0), 1.0)


def rejection_loop(cfg, acfg, prob, ast: AdaptiveState, t1: float) -> AdaptiveState:
    """Attempt steps from ast.step_from until one is accepted (error_norm <= 1)."""
    dt, err_prev = ast.dt, ast.err_prev
    attempts = ast.num_attempts
    rate = cfg.nu + 1
    while True:
        if acfg.control.clip:
            dt = min(dt, t1 - ast.step_from.t)
        err, proposed = step(cfg, prob, ast.step_from, dt)
        attempts += 1
        e = normalise_error(acfg, err, proposed.mean[0], ast.step_from.mean[0])
        if not np.isfinite(e):
            raise FloatingPointError(f"non-finite error norm at t={ast.step_from.t}, dt={dt}")
        dt, err_prev = acfg.control.apply(dt, err_prev, e, rate)
        if e <= 1.0:
            return AdaptiveState(proposed, ast.step_from, dt, err_prev, attempts)


# --------------------------------------------------------------------------------------------
# Interpolation at checkpoints (filter: predict; fixedpoint: predict + one-step smoothing)
# --------------------------------------------------------------------------------------------


def interpolate(cfg, prob, ast: AdaptiveState, t: float):
    """extract_after_overstep: returns (new adaptive state, solution state at t)."""
    s0, s1 = ast.interp_from, ast.step_from
    d = prob.d
    dense = cfg.model == "dense"
    sigma = s1.output_scale
    dt0, dt1 = t - s0.t, s1.t - t
    if cfg.strategy == "filter":
        m, l, _ = _predict(cfg, prob, s0.mean, s0.chol, None, dt0, sigma, d, t)
        sol = State(t, m, l, sigma, s1.num_steps, None)
        return AdaptiveState(s1, sol, ast.dt, ast.err_prev, ast.num_attempts), sol
    # fixed-point smoother (strategy_fixedpoint): t0 -> t (merged backward model), t -> t1
    m_t, l_t, cond_t = _predict(cfg, prob, s0.mean, s0.chol, s0.cond, dt0, sigma, d, t)
    ident = _identity_conditional(cfg, d)
    _, _, cond_t1_to_t = _predict(cfg, prob, m_t, l_t, ident, dt1, sigma, d, s1.t)
    # marginalise the t1-marginal through the backward model t1 -> t
    if dense:
        mean_t = (cond_t1_to_t.G @ s1.mean.reshape(-1) + cond_t1_to_t.xi).reshape(m_t.shape)
    else:
        mean_t = cond_t1_to_t.G @ s1.mean + cond_t1_to_t.xi
    chol_t = qr_r(np.concatenate([(cond_t1_to_t.G @ s1.chol).T, cond_t1_to_t.Lam.T], axis=0)).T
    sol = State(t, mean_t, chol_t, sigma, s1.num_steps, cond_t)
    if cfg.strategy == "smoother":
        step_from = State(s1.t, s1.mean, s1.chol, s1.output_scale, s1.num_steps, cond_t1_to_t)
        interp_from = State(t, mean_t, chol_t, sigma, s1.num_steps, cond_t)
    else:
        step_from = State(s1.t, s1.mean, s1.chol, s1.output_scale, s1.num_steps, cond_t1_to_t)
        interp_from = State(t, m_t, l_t, sigma, s1.num_steps, ident)
    return AdaptiveState(step_from, interp_from, ast.dt, ast.err_prev, ast.num_attempts), sol


def _right_corner(cfg, prob, ast: AdaptiveState):
    s1 = ast.step_from
    sol = s1
    if cfg.strategy == "fixedpoint":
        reset = State(s1.t, s1.mean, s1.chol, s1.output_scale, s1.num_steps, _identity_conditional(cfg, prob.d))
        return AdaptiveState(reset, reset, ast.dt, ast.err_prev, ast.num_attempts), sol
    return AdaptiveState(s1, s1, ast.dt, ast.err_prev, ast.num_attempts), sol


# --------------------------------------------------------------------------------------------
# Driver loops (ivpsolve.solve_*)
# --------------------------------------------------------------------------------------------


class Solution(NamedTuple):
    t: np.ndarray
    u: np.ndarray  # (len(t), d): zeroth derivative means
    states: list  # per-time State (posterior marginal + backward model where applicable)
    num_steps: np.ndarray
    num_attempts: int


def solve_adaptive_save_at(cfg, acfg, prob, init: State, save_at, dt0) -> Solution:
    """ivpsolve.solve_adaptive_save_at (ivpsolvers.py:71-77, run.py:122-129)."""
    save_at = np.asarray(save_at, dtype=np.float64)
    assert init.t == save_at[0]
    ast = adaptive_init(init, dt0)
    sols = [init]
    for t_next in save_at[1:]:
        while ast.step_from.t + acfg.numerical_zero < t_next:
            ast = rejection_loop(cfg, acfg, prob, ast, t_next)
        if ast.step_from.t > t_next + acfg.numerical_zero:
            ast, sol = interpolate(cfg, prob, ast, float(t_next))
        else:
            ast, sol = _right_corner(cfg, prob, ast)
        sols.append(sol)
    return Solution(
        save_at,
        np.stack([s.mean[0] for s in sols]),
        sols,
        np.asarray([s.num_steps for s in sols]),
        ast.num_attempts,
    )


def solve_adaptive_terminal_values(cfg, acfg, prob, init: State, t1, dt0) -> Solution:
    """ivpsolve.solve_adaptive_terminal_values (run.py:82-90)."""
    sol = solve_adaptive_save_at(cfg, acfg, prob, init, np.asarray([init.t, t1]), dt0)
    return Solution(sol.t[-1:], sol.u[-1:], sol.states[-1:], sol.num_steps[-1:], sol.num_attempts)


def solve_adaptive_save_every_step(cfg, acfg, prob, init: State, t1, dt0) -> Solution:
    """ivpsolve.solve_adaptive_save_every_step (vdp.py:77-79): native loop, keeps all steps."""
    ast = adaptive_init(init, dt0)
    sols = [init]
    while ast.step_from.t < t1:
        ast = rejection_loop(cfg, acfg, prob, ast, t1)
        if ast.step_from.t < t1:
            sols.append(ast.step_from)
    if ast.step_from.t > t1:
        ast, sol = interpolate(cfg, prob, ast, float(t1))
    else:
        ast, sol = _right_corner(cfg, prob, ast)
    sols.append(sol)
    return Solution(
        np.asarray([s.t for s in sols]),
        np.stack([s.mean[0] for s in sols]),
        sols,
        np.asarray([s.num_steps for s in sols]),
        ast.num_attempts,
    )


def solve_fixed_grid(cfg, prob, init: State, grid) -> Solution:
    """ivpsolve.solve_fixed_grid (vdp.py:90): the pure predict/correct loop, no controller."""
    grid = np.asarray(grid, dtype=np.float64)
    state = init
    sols = [init]
    for dt in np.diff(grid):
        _, state = step(cfg, prob, state, float(dt))
        sols.append(state)
    return Solution(grid, np.stack([s.mean[0] for s in sols]), sols,
                    np.asarray([s.num_steps for s in sols]), len(grid) - 1)


# --------------------------------------------------------------------------------------------
# Backward marginalisation over stored conditionals (stats.markov_marginals, ivpsolvers.py:80-81)
# --------------------------------------------------------------------------------------------


def markov_marginals_reverse(cfg, sol: Solution):
    """Smoothed means at save_at[:-1] from the terminal marginal + per-checkpoint conditionals.

    sol.states[i].cond (i >= 1) is the backward model  x(save_at[i-1]) | x(save_at[i]).
    Returns means of shape (len(save_at), n, d), ordered like save_at.
    """
    dense = cfg.model == "dense"
    states = sol.states
    mean = states[-1].mean
    chol = states[-1].chol
    out = [mean]
    for s in reversed(states[1:]):
        c = s.cond
        if dense:
            mean = (c.G @ mean.reshape(-1) + c.xi).reshape(mean.shape)
        else:
            mean = c.G @ mean + c.xi
        chol = qr_r(np.concatenate([(c.G @ chol).T, c.Lam.T], axis=0)).T
        out.append(mean)
    return np.stack(out[::-1])


def markov_marginal_chols_reverse(cfg, sol: Solution):
    """Cholesky factors (left factors, cov = C C^T) of the smoothing marginals at every stored point, ordered in time."""
    states = sol.states
    chol = states[-1].chol
    out = [chol]
    for s in reversed(states[1:]):
        c = s.cond
        chol = qr_r(np.concatenate([(c.G @ chol).T, c.Lam.T], axis=0)).T
        out.append(chol)
    return out[::-1]


def smoother_solution(cfg_fixedpoint, acfg, prob, init: State, t1, dt0) -> Solution:
    """strategy_smoother + solve_adaptive_save_every_step (/root/reference/src/odecheckpts/ivpsolvers.py:112,133-142;
    experiments/4_brusselator/run.py:102-117) restated through the fixed-point smoother this oracle pins on the goldens:
    the accepted grid does not depend on the strategy (the error estimate is the filter's), and a fixed-point solve whose
    save_at IS that grid stores exactly one backward transition per accepted step, which is what the smoother stores."""
    cfg_filter = dataclasses.replace(cfg_fixedpoint, strategy="filter")
    grid = solve_adaptive_save_every_step(cfg_filter, acfg, prob, init, t1, dt0).t
    return solve_adaptive_save_at(cfg_fixedpoint, acfg, prob, init, grid, dt0)


def offgrid_marginal_means(cfg_fixedpoint, acfg, prob, init: State, grid, ts, dt0):
    """stats.offgrid_marginals_searchsorted (ivpsolvers.py:118,144): smoothing means at `ts` given the accepted grid, as a
    fixed-point solve over the union of the grid and the query points (an interpolated checkpoint IS the off-grid marginal)."""
    ts = np.asarray(ts, dtype=np.float64)
    both = np.union1d(np.asarray(grid, dtype=np.float64), ts)
    sol = solve_adaptive_save_at(cfg_fixedpoint, acfg, prob, init, both, dt0)
    means = markov_marginals_reverse(cfg_fixedpoint, sol)
    return means[np.searchsorted(both, ts)]


def markov_sample_reverse(cfg, sol: Solution, eps):
    """stats.markov_sample(key, markov_seq, reverse=True) (experiments/5_vs_interpolation/measure.py:69-75) for ONE joint
    sample with base samples eps[(len(t), n, d)] (isotropic model): x_T = m_T + C_T e_T, x_i = G x_{i+1} + xi + Lam e_i, with
    the triangular factors normalised to a non-negative diagonal (the sign convention of a factor is not part of the model)."""
    assert cfg.model == "isotropic"

    def pos(f):
        sg = np.where(np.diag(f) < 0, -1.0, 1.0)
        return f * sg[None, :]

    states = sol.states
    x = states[-1].mean + pos(states[-1].chol) @ eps[-1]
    out = [x]
    for i in range(len(states) - 1, 0, -1):
        c = states[i].cond
        x = c.G @ x + c.xi + pos(c.Lam) @ eps[i - 1]
        out.append(x)
    return np.stack(out[::-1])


# --------------------------------------------------------------------------------------------
# Taylor-mode initialisation (taylor.odejet_padded_scan / odejet_unroll) without autodiff:
# truncated power-series arithmetic on the normalised coefficients c_k = u^(k)(t0) / k!.
# --------------------------------------------------------------------------------------------


class Series:
    """Truncated Taylor series with vector-valued coefficients: coeffs[k] has shape (d,)."""

    def __init__(self, coeffs):
        self.c = [np.asarray(x, dtype=np.float64) for x in coeffs]

    @property
    def order(self):
        return len(self.c)

    def _bin(self, other):
        if not isinstance(other, Series):
            other = Series([np.asarray(other, dtype=np.float64)] + [0.0 * self.c[0]] * (self.order - 1))
        return other

    def __add__(self, other):
        other = self._bin(other)
        return Series([a + b for a, b in zip(self.c, other.c)])

    __radd__ = __add__

    def __neg__(self):
        return Series([-a for a in self.c])

    def __sub__(self, other):
        return self + (-self._bin(other))

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        other = self._bin(other)
        m = min(self.order, other.order)
        return Series([sum(self.c[j] * other.c[k - j] for j in range(k + 1)) for k in range(m)])

    __rmul__ = __mul__

    def __truediv__(self, other):
        other = self._bin(other)
        m = min(self.order, other.order)
        out = []
        for k in range(m):
            acc = self.c[k] - sum(out[j] * other.c[k - j] for j in range(k))
            out.append(acc / other.c[0])
        return Series(out)

    def __pow__(self, r):
        if isinstance(r, int) and r >= 0:
            out = Series([np.ones_like(self.c[0])] + [0.0 * self.c[0]] * (self.order - 1))
            for _ in range(r):
                out = out * self
            return out
        # real power via the ODE  a' b = r a b'  with a = b^r  (needs b_0 != 0)
        a = [self.c[0] ** r]
        for k in range(1, self.order):
            acc = sum((r * (k - j) - j) * self.c[k - j] * a[j] for j in range(k))
            a.append(acc / (k * self.c[0]))
        return Series(a)

    def map_linear(self, fn):
        return Series([fn(x) for x in self.c])


def taylor_coefficients(f_series: Callable, u0s, num: int):
    """Derivatives u, u', ..., u^(k-1+num) at t0 for u^(k) = f(u, ..., u^(k-1)).

    `f_series(*ys)` evaluates the vector field on `Series` arguments. `u0s` holds the k initial
    derivative values. Returns UN-normalised derivatives, as probdiffeq's odejet routines do
    (ivpsolvers.py:65-67, vdp.py:69, run.py:64-66).
    """
    k = len(u0s)
    derivs = [np.atleast_1d(np.asarray(u, dtype=np.float64)) for u in u0s]
    for _ in range(num):
        m = len(derivs)
        # series of u^(i), i < k, truncated to the known order
        ys = []
        for i in range(k):
            ys.append(Series([derivs[i + j] / math.factorial(j) for j in range(m - k + 1)]))
        fx = f_series(*ys)
        j = m - k  # newest coefficient of u^(k) series
        derivs.append(fx.c[j] * math.factorial(j))
    return derivs
