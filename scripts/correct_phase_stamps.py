#!/usr/bin/env python3
"""In-kernel cycle stamps of dense_gain_mfma_kernel (every wavefront of a workgroup) from the DIAGNOSTIC library
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
names = ["form F", "barrier (z)", "sweep", "barrier A", "publish + forward", "barrier B", "apply", "-"]
tot = np.zeros((4, 8))
pick = list(range(0, B, max(1, B // 16)))
for b in pick:
    out = (C.c_uint64 * 32)()
    assert lib.pnode_debug_read_scratch(bt._h, b, 64, 32, out) == 0
    tot += np.array(list(out), dtype=np.float64).reshape(4, 8) / len(pick)
print(f"B={B}: dense_gain_mfma_kernel, cycles per wavefront (rows) by phase (columns), mean of {len(pick)} workgroups")
print("wave " + " ".join(f"{n:>18s}" for n in names) + "      total")
for w in range(4):
    print(f"{w:4d} " + " ".join(f"{x:18.0f}" for x in tot[w]) + f" {tot[w].sum():10.0f}")
