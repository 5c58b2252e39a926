"""Solver construction: prior, correction, strategy, calibration (probdiffeq.ivpsolvers surface).

Reference call sites: src/odecheckpts/ivpsolvers.py:35-50, experiments/4_brusselator/run.py:54-61,
experiments/1_van_der_pol/vdp.py:62-71. The objects are plain configuration records; the arithmetic
they select runs in the HIP kernels (csrc/pnode_iso.hip, csrc/pnode_dense.hip).
"""

from __future__ import annotations

import dataclasses

import numpy as np

from .impl import impl


@dataclasses.dataclass(frozen=True)
class Prior:
    num_derivatives: int


@dataclasses.dataclass(frozen=True)
class Correction:
    kind: str  # "ts0" | "ts1"
    ode_order: int


@dataclasses.dataclass(frozen=True)
class Strategy:
    kind: str  # "filter" | "fixedpoint" | "smoother"
    prior: Prior
    correction: Correction
    model: str  # "isotropic" | "dense"
    ode_shape: tuple

    @property
    def is_suitable_for_save_at(self):
        return self.kind in ("filter", "fixedpoint")

    @property
    def is_suitable_for_save_every_step(self):
        return self.kind in ("filter", "smoother")


@dataclasses.dataclass(frozen=True)
class InitialCondition:
    tcoeffs: np.ndarray  # (B, n, d)
    output_scale: np.ndarray  # (B,)
    batched: bool


@dataclasses.dataclass(frozen=True)
class Solver:
    strategy: Strategy
    calibration: str  # "dynamic" | "none"

    @property
    def error_contraction_rate(self):
        return self.strategy.prior.num_derivatives + 1

    def initial_condition(self, tcoeffs, output_scale=1.0):
        """Exact derivatives at t0 (from `taylor.odejet_*`), zero covariance. Arrays may carry a batch axis."""
        arrs = [np.asarray(c, dtype=np.float64) for c in tcoeffs]
        n = self.strategy.prior.num_derivatives + 1
        if len(arrs) != n:
            raise ValueError(f"expected {n} Taylor coefficients, got {len(arrs)}")
        arrs = [a[None] if a.ndim == 0 else a for a in arrs]
        batched = arrs[0].ndim == 2
        if not batched:
            arrs = [a[None, :] for a in arrs]
        tc = np.stack(arrs, axis=1)  # (B, n, d)
        scale = np.broadcast_to(np.asarray(output_scale, dtype=np.float64), (tc.shape[0],)).copy()
        return InitialCondition(np.ascontiguousarray(tc), scale, batched)


def prior_ibm(num_derivatives: int = 4) -> Prior:
    num_derivatives = int(num_derivatives)
    if not 1 <= num_derivatives <= 8:
        raise ValueError("num_derivatives must be in 1..8")
    return Prior(num_derivatives)


def correction_ts0(ode_order: int = 1) -> Correction:
    return Correction("ts0", int(ode_order))


def correction_ts1(ode_order: int = 1) -> Correction:
    return Correction("ts1", int(ode_order))


def _strategy(kind, prior, correction):
    model, ode_shape = impl.current()
    if correction.kind == "ts1" and model != "dense":
        raise ValueError("correction_ts1 needs impl.select('dense', ...)")
    if correction.ode_order > prior.num_derivatives:
        raise ValueError("ode_order must not exceed num_derivatives")
    return Strategy(kind, prior, correction, model, ode_shape)


def strategy_filter(prior: Prior, correction: Correction) -> Strategy:
    return _strategy("filter", prior, correction)


def strategy_fixedpoint(prior: Prior, correction: Correction) -> Strategy:
    return _strategy("fixedpoint", prior, correction)


def strategy_smoother(prior: Prior, correction: Correction) -> Strategy:
    """Stores one backward transition per accepted step (O(#steps) memory): the 'textbook' baseline the reference
    compares against (ivpsolvers.py:112, run.py:103). Isotropic model, through ivpsolve.solve_adaptive_save_every_step;
    stats.offgrid_marginals_searchsorted evaluates the posterior between the steps."""
    return _strategy("smoother", prior, correction)


def solver(strategy: Strategy) -> Solver:
    return Solver(strategy, "none")


def solver_dynamic(strategy: Strategy) -> Solver:
    return Solver(strategy, "dynamic")
