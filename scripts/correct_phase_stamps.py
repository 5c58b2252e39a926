#!/usr/bin/env python3
"""Cycle stamps of the measurement-update kernel of the dense MFMA path (dense_correct_mfma_kernel): per wavefront, when the strip
of F was formed, the LQ of F (the chain of d pivots) done, the panels permuted, the strips rotated, the tail done -- for the
workgroups of one CU slot (b and b + 256 run on the same CU at the same time when two workgroups fit).
Needs the diagnostic library (make -C csrc timing).   python scripts/correct_phase_stamps.py [B]"""
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


def read(b):
    out = (C.c_uint64 * 64)()
    assert lib.pnode_debug_read_scratch(bt._h, b, 128, 64, out) == 0
    return np.array(list(out), dtype=np.float64).reshape(8, 8)[:4, :7]


names = ["form F", "LQ of F", "permute", "strips", "barrier", "tail"]
t0 = min(read(b)[:, 0].min() for b in range(B))
print(f"B={B}: cycles; start = entry of the workgroup relative to the first workgroup of the launch; then the phases of wavefront 0..3")
for b in (17, 17 + 256, 17 + 512, 17 + 768, 100, 100 + 256):
    if b >= B:
        continue
    w = read(b)
    print(f"workgroup {b:4d}: start {w[:, 0].min() - t0:9.0f}  end {w[:, 6].max() - t0:9.0f}")
    for wv in range(4):
        print("   wave %d " % wv + " ".join(f"{n}: {w[wv, i + 1] - w[wv, i]:7.0f}" for i, n in enumerate(names)))
