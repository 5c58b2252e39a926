"""State-space model selection: `impl.select("isotropic" | "dense", ode_shape=(d,))`.

Mirrors probdiffeq.impl.impl.select as called at src/odecheckpts/ivpsolvers.py:32-33,
experiments/4_brusselator/run.py:51 and experiments/1_van_der_pol/vdp.py:61. The reference keeps a
global, mutable selection; here the selection is only a DEFAULT that `ivpsolvers.strategy_*` snapshots
into the strategy object, so solvers built earlier are unaffected by a later `select`.
"""

from __future__ import annotations

import warnings


class _Impl:
    def __init__(self):
        self.name = None
        self.ode_shape = None

    def select(self, name: str, *, ode_shape):
        if name not in ("isotropic", "dense"):
            raise ValueError(f"unknown state-space model {name!r} (expected 'isotropic' or 'dense')")
        ode_shape = tuple(int(s) for s in ode_shape)
        if len(ode_shape) > 1:
            raise ValueError("ode_shape must be () or (d,)")
        if self.name is not None and (self.name, self.ode_shape) != (name, ode_shape):
            warnings.warn("An implementation has already been selected; overriding the default.", stacklevel=2)
        self.name, self.ode_shape = name, ode_shape

    def current(self):
        if self.name is None:
            raise ValueError("No state-space model selected: call impl.select(name, ode_shape=...) first.")
        return self.name, self.ode_shape


impl = _Impl()
