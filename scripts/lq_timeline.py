#!/usr/bin/env python3
"""Timeline of the persistent one-launch LQ: absolute cycle stamps of every panel of two consecutive trajectories of one workgroup
(arrival at the sweep, end of the sweep, slot free, publication). Needs the diagnostic library (make -C csrc timing).
  python scripts/lq_timeline.py [B] [CUs] [workgroup]"""
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
WG = int(sys.argv[3]) if len(sys.argv) > 3 else 17
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
    out = (C.c_uint64 * 128)()
    assert lib.pnode_debug_read_scratch(bt._h, b, 0, 128, out) == 0
    return np.array(list(out), dtype=np.float64)[64:].reshape(8, 2, 4)  # [wavefront][its pass][arrive, swept, slot free, published]


events = []
for t in range(B // CUS):
    w = read(WG + t * CUS)
    for wv in range(8):
        gi, j = wv >> 2, wv & 3
        for q in range(2):
            events.append((w[wv, q, 0], t, gi + 2 * q, j, w[wv, q]))
events.sort(key=lambda e: (e[1], e[2], e[3]))
t0 = min(e[0] for e in events)
print(f"B={B}, workgroup {WG}: cycles since the first event; one line per panel (trajectory, pass, panel)")
print("traj pass panel     arrive      swept  slot free  published   (published - previous panel's)")
prev = None
for _, t, P, j, w in events:
    gap = "" if prev is None else f"{w[3] - prev:10.0f}"
    print(f"{t:4d} {P:4d} {j:5d} " + " ".join(f"{x - t0:10.0f}" for x in w) + "   " + gap)
    prev = w[3]
