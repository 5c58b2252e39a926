"""odecheckpts: the reference's thin solver/problem layer (L3), on top of probdiffeq_hip.

Same entry points as /root/reference/src/odecheckpts (ivps.*, ivpsolvers.solve / asolve_scipy); every
array argument may carry a leading batch axis, which is how independent IVPs / tolerance sweeps are
handed to the GPU.
"""

from . import ivps, ivpsolvers  # noqa: F401
