"""Vector-field descriptors: the host-side face of the device functions in csrc/pnode_vf.hpp.

The reference passes JAX callables `vf(*y, t, p)` (src/odecheckpts/ivps.py) that probdiffeq traces and
differentiates. On MI355X the right-hand side and its analytic Jacobian are device code selected by
`vf_id`; a `VectorField` carries that id plus parameters and is ALSO callable on NumPy arrays with the
reference's signature, so that host-side checks (scipy truth solves, Taylor initialisation) use the
same object.
"""

from __future__ import annotations

import dataclasses

import numpy as np

VF_BRUSSELATOR = 0
VF_VDP_ORDER2 = 1
VF_VDP_ORDER1 = 2
VF_RIGID_BODY = 3
VF_PLEIADES_ORDER2 = 4
VF_PLEIADES_ORDER1 = 5
VF_LOGISTIC = 6
VF_LINEAR = 7
VF_THREE_BODY = 8
VF_USER = 100


@dataclasses.dataclass(frozen=True)
class VectorField:
    """u^(k) = f(u, ..., u^(k-1)); `params` may be a tuple (shared) or a (B, P) array (per trajectory)."""

    vf_id: int
    d: int
    ode_order: int
    params: tuple = ()
    name: str = ""

    def bind(self, p):
        """Return a copy with parameters `p` (None / () keeps the current ones)."""
        if p is None or (hasattr(p, "__len__") and len(p) == 0):
            return self
        arr = np.asarray(p, dtype=np.float64)
        prm = tuple(map(float, arr.reshape(-1))) if arr.ndim <= 1 else arr
        return dataclasses.replace(self, params=prm)

    def param_array(self):
        if isinstance(self.params, np.ndarray):
            return self.params
        return np.asarray(self.params, dtype=np.float64)

    # --- NumPy / Series evaluation (works on anything supporting + - * / **) ---------------------
    def __call__(self, *ys, t=0.0, p=None):
        prm = self.bind(p).params if p is not None else self.params
        if isinstance(prm, np.ndarray) and prm.ndim == 2:
            raise ValueError("host evaluation needs shared parameters")
        return _EVAL[self.vf_id](self, prm, *ys)

    def series(self, *ys):
        """Evaluate on truncated Taylor series (probdiffeq_hip.taylor.Series)."""
        return _EVAL[self.vf_id](self, self.params, *ys)


def _split(y, i0, i1):
    if hasattr(y, "map_linear"):
        return y.map_linear(lambda a: a[..., i0:i1])
    return y[..., i0:i1]


def _concat(parts):
    if hasattr(parts[0], "map_linear"):
        return type(parts[0])([np.concatenate([q.c[k] for q in parts], axis=-1) for k in range(parts[0].order)])
    return np.concatenate(parts, axis=-1)


def _laplace(w, pad):
    def lap(a, padv):
        left = np.concatenate([np.full(a.shape[:-1] + (1,), padv), a[..., :-1]], axis=-1)
        right = np.concatenate([a[..., 1:], np.full(a.shape[:-1] + (1,), padv)], axis=-1)
        return left - 2.0 * a + right

    if hasattr(w, "map_linear"):
        return type(w)([lap(c, pad if k == 0 else 0.0) for k, c in enumerate(w.c)])
    return lap(w, pad)


def _brusselator(vf, prm, y):
    n = vf.d // 2
    alpha = prm[0] if len(prm) else 1.0 / 50.0
    c = alpha * (n + 1) ** 2
    u, v = _split(y, 0, n), _split(y, n, 2 * n)
    uuv = u * u * v
    return _concat([1.0 + uuv - 4.0 * u + c * _laplace(u, 1.0), 3.0 * u - uuv + c * _laplace(v, 3.0)])


def _vdp2(vf, prm, y, dy):
    return prm[0] * (dy * (1.0 - y * y) - y)


def _vdp1(vf, prm, y):
    x, v = _split(y, 0, 1), _split(y, 1, 2)
    return _concat([v + 0.0 * x, prm[0] * (1.0 - x * x) * v - x])


def _rigid(vf, prm, y):
    a, b, c = prm
    y1, y2, y3 = _split(y, 0, 1), _split(y, 1, 2), _split(y, 2, 3)
    return _concat([a * y2 * y3, b * y1 * y3, c * y1 * y2])


def _pleiades_acc(pos):
    xs = [_split(pos, i, i + 1) for i in range(7)]
    ys = [_split(pos, 7 + i, 8 + i) for i in range(7)]
    ax, ay = [], []
    for i in range(7):
        sx, sy = None, None
        for j in range(7):
            if j == i:
                continue
            dx, dy = xs[j] - xs[i], ys[j] - ys[i]
            r3 = (dx * dx + dy * dy) ** 1.5
            tx, ty = (j + 1.0) * dx / r3, (j + 1.0) * dy / r3
            sx = tx if sx is None else sx + tx
            sy = ty if sy is None else sy + ty
        ax.append(sx)
        ay.append(sy)
    return _concat(ax + ay)


def _pleiades2(vf, prm, y, dy):
    return _pleiades_acc(y)


def _pleiades1(vf, prm, y):
    return _concat([_split(y, 14, 28), _pleiades_acc(_split(y, 0, 14))])


def _logistic(vf, prm, y):
    return prm[0] * y * (1.0 - prm[1] * y)


def _linear(vf, prm, y):
    return prm[0] * y


def _three_body(vf, prm, y, dy):
    """Restricted three-body problem in the rotating frame (Arenstorf orbit), second-order form; prm = (standardised moon mass,)."""
    mu = prm[0]
    mp = 1.0 - mu
    y0, y1 = _split(y, 0, 1), _split(y, 1, 2)
    v0, v1 = _split(dy, 0, 1), _split(dy, 1, 2)
    d1 = ((y0 + mu) * (y0 + mu) + y1 * y1) ** 1.5
    d2 = ((y0 - mp) * (y0 - mp) + y1 * y1) ** 1.5
    return _concat([y0 + 2.0 * v1 - mp * (y0 + mu) / d1 - mu * (y0 - mp) / d2, y1 - 2.0 * v0 - mp * y1 / d1 - mu * y1 / d2])


_EVAL = {
    VF_BRUSSELATOR: _brusselator,
    VF_VDP_ORDER2: _vdp2,
    VF_VDP_ORDER1: _vdp1,
    VF_RIGID_BODY: _rigid,
    VF_PLEIADES_ORDER2: _pleiades2,
    VF_PLEIADES_ORDER1: _pleiades1,
    VF_LOGISTIC: _logistic,
    VF_LINEAR: _linear,
    VF_THREE_BODY: _three_body,
}


def brusselator(N, alpha=1.0 / 50.0):
    return VectorField(VF_BRUSSELATOR, 2 * N, 1, (float(alpha),), "brusselator")


def van_der_pol_order2(mu=1e3):
    return VectorField(VF_VDP_ORDER2, 1, 2, (float(mu),), "van_der_pol")


def van_der_pol_order1(mu=1.0):
    return VectorField(VF_VDP_ORDER1, 2, 1, (float(mu),), "van_der_pol_1st")


def rigid_body(a=-2.0, b=1.25, c=-0.5):
    return VectorField(VF_RIGID_BODY, 3, 1, (float(a), float(b), float(c)), "rigid_body")


def pleiades_order2():
    return VectorField(VF_PLEIADES_ORDER2, 14, 2, (), "pleiades")


def pleiades_order1():
    return VectorField(VF_PLEIADES_ORDER1, 28, 1, (), "pleiades_1st")


def logistic(a=1.0, b=1.0, d=1):
    return VectorField(VF_LOGISTIC, d, 1, (float(a), float(b)), "logistic")


def linear(lam, d=1):
    return VectorField(VF_LINEAR, d, 1, (float(lam),), "linear")


def three_body_restricted(standardised_moon_mass=0.012277471):
    """The IVP of the reference's experiment 5 (src/odecheckpts/ivps.py:32-41 -> diffeqzoo.ivps.three_body_restricted; d = 2, order 2)."""
    return VectorField(VF_THREE_BODY, 2, 2, (float(standardised_moon_mass),), "three_body_restricted")


@dataclasses.dataclass(frozen=True)
class UserVectorField:
    """Escape hatch for ODEs that are not built into the device code (PNODE_VF_USER).

    `fun(*ys, t)` is an ordinary NumPy function with the reference's calling convention; it must broadcast over a
    leading batch axis (ys[i] has shape (B, d)) and, for the Taylor initialisation, consist of + - * / ** so that it
    also runs on `taylor.Series`. The HOST evaluates it for the whole batch before every attempted step (one extra
    device round trip per attempt), so this path trades speed for generality; `jac(*ys, t) -> (B, d, k*d)` is used
    with correction_ts1 (default: complex-step differentiation of `fun`, exact to rounding for analytic `fun`).
    """

    fun: object
    d: int
    ode_order: int = 1
    jac: object = None
    name: str = "user"
    vf_id: int = VF_USER
    params: tuple = ()

    def bind(self, p):
        if p is None or (hasattr(p, "__len__") and len(p) == 0):
            return self
        return dataclasses.replace(self, fun=_Bound(self.fun, p))

    def param_array(self):
        return np.zeros(0)

    def __call__(self, *ys, t=0.0, p=None):
        f = self.bind(p).fun if p is not None else self.fun
        return f(*ys, t=t)

    def series(self, *ys):
        return self.fun(*ys, t=0.0)

    def jacobian(self, ys, t):
        """(B, d, k*d) Jacobian at ys = [(B, d)] * k."""
        if self.jac is not None:
            return np.asarray(self.jac(*ys, t=t), dtype=np.float64)
        k, d = self.ode_order, self.d
        B = ys[0].shape[0]
        out = np.zeros((B, d, k * d))
        h = 1e-30
        for i in range(k):
            for c in range(d):
                yc = [np.array(y, dtype=np.complex128) for y in ys]
                yc[i][:, c] += 1j * h
                out[:, :, i * d + c] = np.imag(self.fun(*yc, t=t)) / h
        return out


@dataclasses.dataclass(frozen=True)
class _Bound:
    fun: object
    p: object

    def __call__(self, *ys, t=0.0):
        return self.fun(*ys, t=t, p=self.p)


def user(fun, d, ode_order=1, jac=None, name="user"):
    return UserVectorField(fun, int(d), int(ode_order), jac, name)
