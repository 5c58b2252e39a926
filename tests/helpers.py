"""Shared test helpers: oracle problems built from the product's VectorField descriptors."""

import numpy as np
import pn_oracle as po
from probdiffeq_hip import taylor
from probdiffeq_hip import vector_fields as V


def oracle_problem(vf: V.VectorField) -> po.Problem:
    """f from the NumPy face of the vector field; Jacobian by complex-step (exact to rounding)."""
    k, d = vf.ode_order, vf.d

    def f(y, t):
        return np.asarray(vf(*[y[i] for i in range(k)]), dtype=np.float64)

    def jac(y, t):
        J = np.zeros((d, k, d))
        h = 1e-30
        for i in range(k):
            for b in range(d):
                yc = [np.array(y[j], dtype=np.complex128) for j in range(k)]
                yc[i][b] += 1j * h
                J[:, i, b] = np.imag(vf(*yc)) / h
        return J

    return po.Problem(f, jac, k, d)


def brusselator_y0(N, phase=0.0):
    x = np.linspace(0, 1, N)
    return np.concatenate([1.0 + np.sin(2 * np.pi * x + phase), 3.0 * np.ones(N)])


def tcoeffs(vf, u0s, nu):
    return taylor.odejet_padded_scan(vf, u0s, num=nu + 1 - vf.ode_order)


def oracle_state_from(cfg, means, chol, t, sigma, cond=None):
    return po.State(float(t), np.array(means), np.array(chol), float(sigma), 0, cond)


def scipy_truth(vf, y0, t_eval, method="DOP853"):
    """Independent Runge-Kutta solution of a first-order problem at `t_eval` (tolerances 1e-12): the stand-in for the
    diffrax solvers the reference's test compares with (tests/test_ivpsolvers.py:11-24)."""
    import scipy.integrate

    t_eval = np.asarray(t_eval, dtype=np.float64)
    sol = scipy.integrate.solve_ivp(lambda t, y: np.asarray(vf(y, t=t)), (t_eval[0], t_eval[-1]), np.asarray(y0, dtype=np.float64),
                                    method=method, rtol=1e-12, atol=1e-12, t_eval=t_eval)
    return sol.y.T
