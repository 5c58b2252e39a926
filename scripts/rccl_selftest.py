#!/usr/bin/env python3
"""World-size-1 check of the ctypes RCCL binding (probdiffeq_hip/sharding.py) on one GPU: signatures, struct passing, buffers."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "code-adaptive-prob-ode-solvers_amd"))
from probdiffeq_hip import sharding
c = sharding.RcclGather(0, 1, 0)
a = np.arange(12.0).reshape(3, 4)
out = c.all_gather(a)
assert np.array_equal(out, a), out
b = np.arange(10, dtype=np.int64)
assert np.array_equal(c.all_gather(b), b)
c.close()
print("rccl ctypes all_gather (world 1): ok")
