"""Solution routines for (batches of) initial value problems: the reference's L3 entry point.

`solve` has the reference's signature and return structure
(/root/reference/src/odecheckpts/ivpsolvers.py:14-91): it assembles isotropic IWP(nu) + EKF0 +
fixed-point smoother (+ dynamic / no calibration) + PI control, returns a closure
`solve_(u0: tuple, p, output_scale=1.0) -> (qoi[len(save_at), d], {"solution", "u0_solve"})`, and
raises ValueError for unknown methods / calibrations / non-tuple u0. The stepping, checkpoint
interpolation and backward marginalisation run in the HIP kernels behind probdiffeq_hip.
Extensions: u0 entries may have shape (B, d) and atol / rtol shape (B,) to solve B IVPs at once.
"""

from __future__ import annotations

import warnings

import numpy as np
from probdiffeq_hip import ivpsolve, ivpsolvers, stats, taylor
from probdiffeq_hip.impl import impl


def solve(method: str, vf, u0_like, /, save_at, *, dt0, atol, rtol, ode_order=1, calibrate="dynamic"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # the reference passes u0[0] and indexes it again (ivpsolvers.py:33), which selects ode_shape=();
        # for an isotropic model only the trailing dimension matters, so accept either
        like = np.asarray(u0_like[0] if isinstance(u0_like, tuple) else u0_like)
        impl.select("isotropic", ode_shape=like.shape[-1:] if like.ndim else ())

    if method[:3] != "ts0" or not method[-1].isdigit():
        raise ValueError(f"unknown method {method!r}: expected 'ts0-<num_derivatives>'")
    num_derivatives = int(method[-1])
    correction = ivpsolvers.correction_ts0(ode_order=ode_order)
    ibm = ivpsolvers.prior_ibm(num_derivatives=num_derivatives)
    strategy = ivpsolvers.strategy_fixedpoint(ibm, correction)
    if calibrate == "dynamic":
        solver = ivpsolvers.solver_dynamic(strategy)
    elif calibrate == "none":
        solver = ivpsolvers.solver(strategy)
    else:
        raise ValueError(f"unknown calibration {calibrate!r}")
    control = ivpsolve.control_proportional_integral()
    asolver = ivpsolve.adaptive(solver, atol=atol, rtol=rtol, control=control)
    save_at = np.asarray(save_at, dtype=np.float64)

    def solve_(u0: tuple, p, output_scale=1.0):
        if not isinstance(u0, tuple):
            raise ValueError("Tuple expected.")
        vf_bound = vf.bind(p)
        tcoeffs = taylor.odejet_padded_scan(vf_bound, u0, num=num_derivatives + 1 - ode_order)
        init = solver.initial_condition(tcoeffs, output_scale=output_scale)
        sol = ivpsolve.solve_adaptive_save_at(vf_bound, init, save_at=save_at, dt0=dt0, adaptive_solver=asolver)

        markov_seq_posterior = stats.markov_select_terminal(sol.posterior)
        margs_posterior = stats.markov_marginals(markov_seq_posterior, reverse=True)
        mean = np.concatenate([margs_posterior.mean, markov_seq_posterior.terminal[:, None]], axis=1)
        qoi = mean[:, :, 0, :]
        aux = {"solution": sol, "u0_solve": sol.u}
        return (qoi if init.batched else qoi[0]), aux

    return solve_


def solve_via_interpolate(method: str, vf, u0_like, /, save_at, *, dt0, atol, rtol):
    """The reference's "textbook" baseline (ivpsolvers.py:94-148): isotropic IWP(nu) + EKF0 with strategy_smoother and dynamic
    calibration, adaptive steps over [save_at[0] - 1e-6, save_at[-1] + 1e-6] with EVERY accepted step stored, then the posterior
    evaluated at `save_at` between the steps. Same signature and return structure as `solve`; one IVP per call (the number of
    stored steps differs between trajectories)."""
    small_value = 1e-6
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        like = np.asarray(u0_like[0] if isinstance(u0_like, tuple) else u0_like)
        impl.select("isotropic", ode_shape=like.shape[-1:] if like.ndim else ())
    if method[:3] != "ts0" or not method[-1].isdigit():
        raise ValueError(f"unknown method {method!r}: expected 'ts0-<num_derivatives>'")
    num_derivatives = int(method[-1])
    strategy = ivpsolvers.strategy_smoother(ivpsolvers.prior_ibm(num_derivatives=num_derivatives), ivpsolvers.correction_ts0())
    solver = ivpsolvers.solver_dynamic(strategy)
    asolver = ivpsolve.adaptive(solver, atol=atol, rtol=rtol, control=ivpsolve.control_proportional_integral())
    save_at = np.asarray(save_at, dtype=np.float64)

    def solve_(u0: tuple, p, output_scale=1.0):
        if not isinstance(u0, tuple):
            raise ValueError("Tuple expected.")
        vf_bound = vf.bind(p)
        tcoeffs = taylor.odejet_padded_scan(vf_bound, u0, num=num_derivatives)
        init = solver.initial_condition(tcoeffs, output_scale=output_scale)
        if init.batched:
            raise ValueError("solve_via_interpolate stores every accepted step: one initial value problem per call")
        # small perturbation so that all save_at values are in the interior of the domain (ivpsolvers.py:135-139)
        sol = ivpsolve.solve_adaptive_save_every_step(vf_bound, init, t0=save_at[0] - small_value, t1=save_at[-1] + small_value,
                                                      dt0=dt0, adaptive_solver=asolver)
        dense, _ = stats.offgrid_marginals_searchsorted(ts=save_at, solution=sol, solver=solver)
        return dense, {"solution": sol, "u0_solve": sol.u}

    return solve_


def solve_diffrax(method: str, vf, _u0_like, /, save_at, *, dt0, atol, rtol, ode_order=1):
    raise NotImplementedError("diffrax Runge-Kutta competitors are outside the hot-path scope (SURVEY.md section 2, row 3)")


def asolve_scipy(method: str, vf, /, time_span, *, atol, rtol):
    """Plot-quality truth from a scipy integrator (the reference's `asolve_scipy`, ivpsolvers.py:196-210: same signature, same
    return value). Returns `solve_(u0: tuple, p) -> (t[K], y[K, d])` on scipy's own adaptive grid; host only, no GPU involved."""
    import scipy.integrate

    def solve_(u0: tuple, p):
        if not isinstance(u0, tuple):
            raise ValueError("Tuple expected.")
        (y0,) = u0
        res = scipy.integrate.solve_ivp(lambda t, y: np.asarray(vf(y, t=t, p=p)), time_span, np.asarray(y0, dtype=np.float64),
                                        method=method, atol=atol, rtol=rtol)
        return res.t, res.y.T

    return solve_

