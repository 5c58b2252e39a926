"""Initial value problems of the reference zoo that the hot path covers.

Counterparts of /root/reference/src/odecheckpts/ivps.py: brusselator (:124-156), van_der_pol (:159-167),
rigid_body (:20-29), pleiades_1st / pleiades_2nd (:44-99), logistic (:8-17), three_body_restricted (:32-41; the
reference's experiment 5). Each returns what the reference's function returns, with `vf` a device-backed VectorField that
is also callable on NumPy arrays as `vf(*u, t=..., p=...)`. `neural_ode_mlp` (a training loop's model)
is outside the hot path and raises.
"""

from __future__ import annotations

import numpy as np
from probdiffeq_hip import vector_fields as V


def logistic():
    vf = V.logistic(1.0, 1.0)
    return vf, (np.asarray([0.1]),), (0.0, 10.0), (1.0, 1.0)


def rigid_body(*, time_span=(0.0, 10.0)):
    vf = V.rigid_body()
    return vf, (np.asarray([1.0, 0.0, 0.9]),), tuple(time_span), (-2.0, 1.25, -0.5)


def _pleiades_initial():
    u0 = np.asarray([3.0, 3.0, -1.0, -3.00, 2.0, -2.00, 2.0, 3.0, -3.0, 2.0, 0.00, 0.0, -4.00, 4.0])
    du0 = np.asarray([0.0, 0.0, 0.0, 0.00, 0.0, 1.75, -1.5, 0.0, 0.0, 0.0, -1.25, 1.0, 0.00, 0.0])
    return u0, du0


def pleiades_2nd():
    return V.pleiades_order2(), _pleiades_initial(), (0.0, 3.0)


def pleiades_1st():
    u0, du0 = _pleiades_initial()
    return V.pleiades_order1(), (np.concatenate([u0, du0]),), (0.0, 3.0), ()


def brusselator(N, t0=0.0, tmax=10.0):
    vf = V.brusselator(int(N))
    x0 = np.linspace(0, 1, num=int(N))
    y0 = np.concatenate([np.sin(2 * np.pi * x0) + 1, 3.0 * np.ones(int(N))])
    return vf, (y0,), (t0, tmax), ()


def van_der_pol(mu=10.0**3):
    return V.van_der_pol_order2(mu), (np.asarray([2.0]), np.asarray([0.0])), (0.0, 6.3)


def van_der_pol_first_order(mu=1.0):
    """First-order form used by the reference's neural-ODE script (experiments/old/7_node/node.py:23-37)."""
    return V.van_der_pol_order1(mu), (np.asarray([2.0, 0.0]),), (0.0, 6.3), ()


def three_body_restricted():
    """Arenstorf orbit of the restricted three-body problem, second-order form (reference ivps.py:32-41, which takes the defaults
    of diffeqzoo.ivps.three_body_restricted: the constants of Hairer, Norsett & Wanner; pinned by the reference's committed
    experiments/5_vs_interpolation/data_solution.npy, tests/test_oracle_golden.py). Returns (vf, (u0, du0), time_span)."""
    u0 = np.asarray([0.994, 0.0])
    du0 = np.asarray([0.0, -2.00158510637908252240537862224])
    return V.three_body_restricted(), (u0, du0), (0.0, 17.0652165601579625588917206249)


def neural_ode_mlp(*, layer_sizes):
    raise NotImplementedError("neural_ode_mlp is outside the benchmark configurations (SURVEY.md section 2, row 5)")
