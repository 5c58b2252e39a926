#!/usr/bin/env python3
"""Soak of the dense model over shapes (run on the GPU box): Brusselator with d = 2N (N from 1 .. 40: the fused one-wavefront kernel for D <= 32, the
register-resident MFMA kernels for d = 32, 64, the large-D kernels, the generic kernels for every other shape), stiff Van der Pol (d = 1) and the rigid body
(d = 3); nu = 1 .. 4, EKF1, filter and fixed-point smoother. Per configuration: the device-resident controller against the host controller on the default
path (identical accept / reject sequences), and the default path against the generic kernels (different arithmetic: step counts within 2 %, outputs to 1e-6).
Prints one line per configuration and a summary; exit code 1 on any mismatch."""
import os
import sys
import time
import warnings

import numpy as np

warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
from helpers import brusselator_y0, tcoeffs  # noqa: E402
from probdiffeq_hip import _backend, ivpsolve, ivpsolvers  # noqa: E402
from probdiffeq_hip import vector_fields as V  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    bad = 0
    t0 = time.time()
    fixed = [("bruss", 16, 3, "filter"), ("bruss", 32, 3, "filter"), ("bruss", 32, 3, "fixedpoint"), ("bruss", 16, 4, "fixedpoint"), ("vdp", 0, 4, "filter"),
             ("rigid", 0, 3, "fixedpoint"), ("bruss", 40, 1, "filter"), ("bruss", 48, 1, "fixedpoint")]
    for it in range(count):
        if it < len(fixed):
            kind, N, nu, strategy = fixed[it]
        else:
            kind = str(rng.choice(["bruss", "bruss", "bruss", "vdp", "rigid"]))
            N = int(rng.choice([rng.integers(1, 5), rng.integers(5, 17), 16, 32, rng.integers(17, 41)]))
            nu = int(rng.integers(1, 5))
            strategy = "fixedpoint" if rng.random() < 0.4 else "filter"
        B = 2
        if kind == "bruss":
            vf = V.brusselator(N)
            u0s = [np.stack([brusselator_y0(N, 0.3 + 0.4 * b) for b in range(B)])]
            t1, dt0 = (0.4 if N <= 16 else 0.15), 0.005
        elif kind == "vdp":
            vf = V.van_der_pol_order2(50.0)
            u0s = [np.tile([[2.0]], (B, 1)), np.tile([[0.0]], (B, 1))]
            t1, dt0 = 0.5, 0.01
        else:
            vf = V.rigid_body()
            u0s = [np.tile([[1.0, 0.0, 0.9]], (B, 1))]
            t1, dt0 = 2.0, 0.1
        if vf.ode_order > nu:
            continue
        D = (nu + 1) * vf.d
        if D > 320 and not (D % 64 == 0 and vf.d % 32 == 0):
            continue  # (generic kernels only: minutes per solve)
        impl.select("dense", ode_shape=(vf.d,))
        strat = {"filter": ivpsolvers.strategy_filter, "fixedpoint": ivpsolvers.strategy_fixedpoint}[strategy]
        solver = ivpsolvers.solver_dynamic(strat(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts1(ode_order=vf.ode_order)))
        tol = np.array([1e-4, 1e-6])
        init = solver.initial_condition(tcoeffs(vf, u0s, nu), 1.0)
        asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol)
        save_at = np.linspace(0.0, t1, 4)
        run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=dt0, adaptive_solver=asolver, device_controller=dc)
        _backend.dense_path = 0
        host = run(False)
        status = []
        try:
            dev = run(True)
            same = np.array_equal(dev.num_steps, host.num_steps)
            rel = float(np.max(np.abs(dev.u - host.u)) / max(np.max(np.abs(host.u)), 1e-300))
            ok = same and rel < 1e-9 and np.all(np.isfinite(dev.u))
            status.append(f"device:{'ok' if ok else 'MISMATCH'}({rel:.1e})")
            bad += not ok
        except (ValueError, RuntimeError) as e:
            status.append(f"device:refused({str(e)[:40]})")
        if D <= 128:
            _backend.dense_path = 1
            try:
                gen = run(False)
                close = np.all(np.abs(gen.num_steps[:, -1] - host.num_steps[:, -1]) <= np.maximum(1, 0.02 * host.num_steps[:, -1]))
                rel = float(np.max(np.abs(gen.u - host.u)) / max(np.max(np.abs(host.u)), 1e-300))
                ok = close and rel < 1e-6 and np.all(np.isfinite(gen.u))
                status.append(f"generic:{'ok' if ok else 'MISMATCH'}({rel:.1e}, steps {gen.num_steps[:, -1].tolist()})")
                bad += not ok
            finally:
                _backend.dense_path = 0
        print(f"{kind:5s} d={vf.d:3d} nu={nu} D={D:3d} {strategy:10s} steps={host.num_steps[:, -1].tolist()} finite={bool(np.all(np.isfinite(host.u)))} " + " ".join(status), flush=True)
        bad += not np.all(np.isfinite(host.u))
    print(f"{count} configurations, {bad} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
