"""Taylor-mode initialisation without autodiff (replaces probdiffeq.taylor.odejet_*).

Reference call sites: taylor.odejet_padded_scan (src/odecheckpts/ivpsolvers.py:65-67,
experiments/1_van_der_pol/vdp.py:69) and taylor.odejet_unroll (experiments/4_brusselator/run.py:64-66).
Both return the exact derivatives u, u', ..., of the ODE solution at t0. Here they are computed by
truncated power-series arithmetic on the normalised coefficients c_k = u^(k)(t0)/k!; every built-in
vector field is a rational/power expression, so +, -, *, /, ** on `Series` suffice. Arrays may carry
leading batch axes: coefficient arrays have shape (..., d). One-off, host-side (not on the hot path).
"""

from __future__ import annotations

import math

import numpy as np


class Series:
    """Truncated Taylor series; coeffs[k] is an array of shape (..., d)."""

    def __init__(self, coeffs):
        self.c = [np.asarray(x, dtype=np.float64) for x in coeffs]

    @property
    def order(self):
        return len(self.c)

    def _lift(self, other):
        if isinstance(other, Series):
            return other
        zero = np.zeros_like(self.c[0])
        return Series([zero + np.asarray(other, dtype=np.float64)] + [zero] * (self.order - 1))

    def __add__(self, other):
        other = self._lift(other)
        return Series([a + b for a, b in zip(self.c, other.c)])

    __radd__ = __add__

    def __neg__(self):
        return Series([-a for a in self.c])

    def __sub__(self, other):
        return self + (-self._lift(other))

    def __rsub__(self, other):
        return (-self) + other

    def __mul__(self, other):
        other = self._lift(other)
        m = min(self.order, other.order)
        return Series([sum(self.c[j] * other.c[k - j] for j in range(k + 1)) for k in range(m)])

    __rmul__ = __mul__

    def __truediv__(self, other):
        other = self._lift(other)
        m = min(self.order, other.order)
        out = []
        for k in range(m):
            acc = self.c[k] - sum(out[j] * other.c[k - j] for j in range(k))
            out.append(acc / other.c[0])
        return Series(out)

    def __rtruediv__(self, other):
        return self._lift(other) / self

    def __pow__(self, r):
        if isinstance(r, int) and r >= 0:
            out = self._lift(1.0)
            for _ in range(r):
                out = out * self
            return out
        # a = b^r obeys  a' b = r a b'  (needs b_0 != 0)
        a = [self.c[0] ** r]
        for k in range(1, self.order):
            acc = sum((r * (k - j) - j) * self.c[k - j] * a[j] for j in range(k))
            a.append(acc / (k * self.c[0]))
        return Series(a)

    def map_linear(self, fn):
        return Series([fn(x) for x in self.c])

    def __getitem__(self, idx):
        """Component selection, applied to every coefficient (so user vector fields can slice their state)."""
        return Series([x[idx] for x in self.c])


def concatenate(parts, axis=-1):
    """np.concatenate that also accepts truncated series (for vector fields written once for arrays and Series)."""
    if isinstance(parts[0], Series):
        return Series([np.concatenate([q.c[k] for q in parts], axis=axis) for k in range(parts[0].order)])
    return np.concatenate(parts, axis=axis)


def _derivatives(vf, u0s, num):
    k = len(u0s)
    derivs = [np.asarray(u, dtype=np.float64) for u in u0s]
    derivs = [u if u.ndim >= 1 else u[None] for u in derivs]
    evaluate = vf.series if hasattr(vf, "series") else vf
    for _ in range(num):
        m = len(derivs)
        ys = [Series([derivs[i + j] / math.factorial(j) for j in range(m - k + 1)]) for i in range(k)]
        fx = evaluate(*ys)
        j = m - k
        derivs.append(fx.c[j] * math.factorial(j))
    return derivs


def odejet_padded_scan(vf, inits, /, num: int):
    """[u, u', ..., u^(k-1+num)] at t0 for u^(k) = vf(u, ..., u^(k-1)); `inits` holds the k initial values."""
    return _derivatives(vf, list(inits), num)


def odejet_unroll(vf, inits, /, num: int):
    """Same result as `odejet_padded_scan` (the reference's two routines differ only in how JAX traces them)."""
    return _derivatives(vf, list(inits), num)
