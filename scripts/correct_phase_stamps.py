#!/usr/bin/env python3
"""In-kernel cycle stamps of dense_correct_mfma_kernel (wavefront 0 of a workgroup) from the DIAGNOSTIC library
(make -C csrc timing).   python scripts/correct_phase_stamps.py [B]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from probdiffeq_hip import _backend
_backend.LIB_PATH = os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd", "csrc", "libpnode_hip_timing.so")
from helpers import brusselator_y0, tcoeffs
from probdiffeq_hip import vector_fields as V

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N, nu = 32, 3
vf = V.brusselator(N)
y0 = np.stack([brusselator_y0(N, 0.1 * b) for b in range(B)])
tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy="filter",
                    calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
bt.set_initial(tc, np.zeros(B), np.ones(B))
bt.set_tolerances(1e-6, 1e-6)
err, one = np.zeros(B), np.ones(B, np.uint8)
lib = bt._lib
lib.pnode_debug_read_scratch.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
for it in range(3):
    bt.attempt_step(np.full(B, 0.005), one, err)
    bt.commit(one)
names = ["stage J, T11", "F1 (MFMA)", "F -> LDS/global", "S = F F^T", "S -> LDS", "Cholesky", "Dinv, w, TRSM", "W -> LDS, strips"]
tot = np.zeros(8)
pick = list(range(0, B, max(1, B // 16)))
for b in pick:
    out = (C.c_uint64 * 8)()
    assert lib.pnode_debug_read_scratch(bt._h, b, 64, 8, out) == 0
    tot += np.array(list(out), dtype=np.float64) / len(pick)
print(f"B={B}: dense_correct_mfma_kernel, cycles of wavefront 0 by phase, mean of {len(pick)} workgroups")
for n, x in zip(names, tot):
    print(f"  {n:>18s} {x:10.0f}")
print(f"  {'total':>18s} {tot.sum():10.0f}")
