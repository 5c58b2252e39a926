#!/usr/bin/env python3
"""One longer randomised comparison of the one-launch LQ (persistent workgroups, default) with the two-launch LQ
(pnode_set_generic_path(h, 3)): B trajectories (several per workgroup), random step sizes, activity masks and accept/reject
decisions over many attempts, at D = 128, 192, 256. Prints the largest deviations; exits non-zero beyond rounding level.
  python scripts/soak_lq.py [B] [attempts]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from probdiffeq_hip import _backend, vector_fields as V
from helpers import brusselator_y0, tcoeffs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ATT = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = False
for N, nu in ((16, 3), (16, 5), (32, 3)):
    rng = np.random.default_rng(1000 * N + nu)
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in rng.uniform(0.0, 2.0, B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)

    def make(path):
        bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy="filter",
                            calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
        if path:
            bt.set_generic_path(path)
        bt.set_initial(tc, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
        return bt

    one, two = make(0), make(3)
    e1, e2 = np.zeros(B), np.zeros(B)
    worst_e = worst_m = 0.0
    for it in range(ATT):
        dts = 0.004 * rng.uniform(0.5, 2.0, B)
        act = (rng.uniform(size=B) < 0.8).astype(np.uint8)
        acc = (act & (rng.uniform(size=B) < 0.85)).astype(np.uint8)
        one.attempt_step(dts, act, e1)
        two.attempt_step(dts, act, e2)
        live = act.astype(bool)
        if not np.all(np.isfinite(e1[live])):
            print(f"N={N} nu={nu} attempt {it}: non-finite error estimate"); bad = True; break
        worst_e = max(worst_e, float(np.max(np.abs(e1[live] - e2[live]) / e2[live])))
        one.commit(acc); two.commit(acc)
        m1, m2 = one.read_mean(0), two.read_mean(0)
        worst_m = max(worst_m, float(np.max(np.abs(m1[:, 0] - m2[:, 0])) / np.max(np.abs(m2[:, 0]))))
    print(f"N={N} nu={nu} D={(nu + 1) * vf.d} B={B} attempts={ATT}: max rel dev of the error estimates {worst_e:.2e}, of the means {worst_m:.2e}", flush=True)
    bad = bad or worst_e > 1e-6 or worst_m > 1e-9
    one.close(); two.close()
sys.exit(1 if bad else 0)
