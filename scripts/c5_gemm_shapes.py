#!/usr/bin/env python3
"""Diagnostic build (make -C csrc timing): one config-5-sized filter step with the shapes of its batched GEMMs logged by the host
(stderr of libpnode_hip_timing.so), to be paired with the rocprofv3 kernel trace of the same run (same order)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
from probdiffeq_hip import _backend
_backend.LIB_PATH = os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd", "csrc", "libpnode_hip_timing.so")
from helpers import brusselator_y0, tcoeffs
from probdiffeq_hip import vector_fields as V

N, B, nu = int(sys.argv[1]), int(sys.argv[2]), 3
strategy = sys.argv[3] if len(sys.argv) > 3 else "filter"
vf = V.brusselator(N)
y0 = np.stack([brusselator_y0(N, 0.1 * b) for b in range(B)])
tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy=strategy, calibration="dynamic",
                    vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
bt.set_initial(tc, np.zeros(B), np.ones(B))
bt.set_tolerances(1e-6, 1e-6)
err, one = np.zeros(B), np.ones(B, np.uint8)
bt.attempt_step(np.full(B, 0.005), one, err)
bt.commit(one)
print("=== second step", file=sys.stderr, flush=True)
bt.attempt_step(np.full(B, 0.005), one, err)
bt.commit(one)
print("=== end", file=sys.stderr, flush=True)
bt.close()
