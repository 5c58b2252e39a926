"""probdiffeq_hip: the probdiffeq solver API surface used by pnkraemer/code-adaptive-prob-ode-solvers,
re-implemented for AMD MI355X (gfx950). Python host code drives hand-written HIP kernels through the
C-ABI of include/pnode.h via ctypes; batches of independent IVPs are the unit of work.

    from probdiffeq_hip import ivpsolve, ivpsolvers, stats, taylor
    from probdiffeq_hip.impl import impl

mirrors `from probdiffeq import ...` / `from probdiffeq.impl import impl` of the reference
(/root/reference/src/odecheckpts/ivpsolvers.py:10-11).
"""

from . import impl as _impl_module  # noqa: F401
from . import ivpsolve, ivpsolvers, stats, taylor, vector_fields  # noqa: F401

__all__ = ["ivpsolve", "ivpsolvers", "stats", "taylor", "vector_fields"]
