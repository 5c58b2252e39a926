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
