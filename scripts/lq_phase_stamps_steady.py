#!/usr/bin/env python3
"""Steady-state cycle stamps of the persistent one-launch LQ: the stamps accumulate over the trajectories a workgroup takes, so the
difference between two consecutive trajectories of ONE workgroup (b and b + number of CUs) is one trajectory in steady state.
Needs the diagnostic library (make -C csrc timing).   python scripts/lq_phase_stamps_steady.py [B] [CUs]"""
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
CUS = int(sys.argv[2]) if len(sys.argv) > 2 else 256
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


def read(b):
    out = (C.c_uint64 * 64)()
    assert lib.pnode_debug_read_scratch(bt._h, b, 0, 64, out) == 0
    return np.array(list(out), dtype=np.float64).reshape(8, 8)


names = ["load", "wait panel", "apply", "sweep", "wait slot", "publish", "store", "-"]
tot = np.zeros((8, 8))
pairs = [(w + 2 * CUS, w + 3 * CUS) for w in (0, CUS // 2, CUS - 1)]
for lo, hi in pairs:
    tot += (read(hi) - read(lo)) / len(pairs)
print(f"B={B}: steady state, cycles per wavefront (rows) by phase (columns), one trajectory (4th minus 3rd of a workgroup), mean of {len(pairs)} workgroups")
print("wave " + " ".join(f"{n:>15s}" for n in names) + "           total")
for w in range(8):
    print(f"{w:4d} " + " ".join(f"{x:15.0f}" for x in tot[w]) + f" {tot[w].sum():15.0f}")
