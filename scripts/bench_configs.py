#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations C1-C4 on one GPU, whole adaptive solves.

Prints one JSON line per configuration: accepted steps/s (sum over the batch / wall time incl. the host
controller), attempted steps/s, batched attempts, and a truth check against scipy where cheap.
Not the headline bench (that is bench.py); this is the coverage table for DESIGN.md / profiles/.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

from odecheckpts import ivps  # noqa: E402
from probdiffeq_hip import ivpsolve, ivpsolvers, taylor  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402


def run(name, vf, u0s, t0, t1, *, model, corr, strategy, nu, atol, rtol, dt0, num_save=5, truth=None, device_controller=False):
    import warnings

    warnings.simplefilter("ignore")
    d = vf.d
    impl.select(model, ode_shape=(d,))
    correction = (ivpsolvers.correction_ts1 if corr == "ts1" else ivpsolvers.correction_ts0)(ode_order=vf.ode_order)
    strat = (ivpsolvers.strategy_fixedpoint if strategy == "fixedpoint" else ivpsolvers.strategy_filter)(
        ivpsolvers.prior_ibm(nu), correction)
    solver = ivpsolvers.solver_dynamic(strat)
    tcs = taylor.odejet_padded_scan(vf, u0s, num=nu + 1 - vf.ode_order)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(t0, t1, num_save)
    asolver = ivpsolve.adaptive(solver, atol=atol, rtol=rtol)
    # warm-up (first launch compiles nothing but pages code in)
    ivpsolve.solve_adaptive_save_at(vf, init, save_at=np.linspace(t0, t0 + 1e-3 * (t1 - t0), 2), dt0=dt0, adaptive_solver=asolver,
                                    device_controller=device_controller)  # (the same kernels as the timed solve)
    tic = time.perf_counter()
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=dt0, adaptive_solver=asolver,
                                          device_controller=device_controller)
    el = time.perf_counter() - tic
    acc = int(np.sum(sol.num_steps[..., -1]))
    att = int(np.sum(sol.num_attempts))
    B = init.tcoeffs.shape[0]
    out = {"config": name, "batch": B, "d": d, "nu": nu, "model": model, "correction": corr, "strategy": strategy,
           "controller": ("device (" + device_controller + ")" if isinstance(device_controller, str) else "device") if device_controller else "host",
           "accepted_steps": acc, "attempted_steps": att, "seconds": round(el, 4),
           "accepted_steps_per_s": acc / el, "finite": bool(np.all(np.isfinite(sol.u)))}
    if truth is not None:
        out["max_abs_err_vs_truth_at_t1"] = float(np.max(np.abs(np.atleast_2d(sol.u)[..., -1, :] - truth)))
    print(json.dumps(out), flush=True)


def _device_memory():
    """(free, total) bytes of the current device through the HIP runtime (ctypes)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return free.value, total.value


def config5(B=64, t1=2.0, tol=1e-4, num_save=5, strategies=("filter", "fixedpoint")):
    """C5: Brusselator N=128 (d=256), nu=3, dense EKF1 (D=1024), fixed-memory checkpoint smoothing; 64 trajectories per GPU
    (512 over 8). Phase-shifted initial profiles as in bench.problem_batch. `c5full` = as specified: t in [0, 10], 200 save_at
    points (experiments/4_brusselator/run.py:119), with the device memory in use printed before the line."""
    N = 128
    vf, u0, (t0, _), _ = ivps.brusselator(N)
    x = np.linspace(0.0, 1.0, N)
    ph = 2.0 * np.pi * np.arange(B) / B
    y0 = np.concatenate([1.0 + np.sin(2 * np.pi * x[None, :] + ph[:, None]), 3.0 * np.ones((B, N))], axis=1)
    for strategy in strategies:
        run(f"C5 brusselator N=128 dense EKF1 nu=3 {strategy}, t1={t1}, {num_save} save_at points, tol {tol}", vf, [y0], t0, t1,
            model="dense", corr="ts1", strategy=strategy, nu=3, atol=tol, rtol=tol, dt0=0.01, num_save=num_save)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "c5":
        config5()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "dense":
        import bench

        vf, tcs = bench.problem_batch(1024)
        for dev in (False, True):
            run("C4 brusselator dense EKF1 nu=3", vf, [tcs[0]], 0.0, 10.0, model="dense", corr="ts1", strategy="filter", nu=3,
                atol=1e-6, rtol=1e-6, dt0=0.01, device_controller=dev)
        config5()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c4fp":
        # the headline shape with fixed-memory checkpoint smoothing (dense fixed-point smoother, D = 256: extra-row LQ and
        # backward-transition merge on the large-D kernels), both controllers
        import bench

        vf, tcs = bench.problem_batch(1024)
        for dev in (False, True):
            run("C4 brusselator dense EKF1 nu=3 fixedpoint, 5 save_at points", vf, [tcs[0]], 0.0, 10.0, model="dense", corr="ts1",
                strategy="fixedpoint", nu=3, atol=1e-6, rtol=1e-6, dt0=0.01, device_controller=dev)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "c5full":
        D, K, B = 1024, 199, 64
        plan = {"per_trajectory_MB": {"checkpoint transitions ck_G + ck_Lam (K x 2 x D^2 x 8 B)": 2 * K * D * D * 8 / 2**20,
                                      "5 state slots (factor + running G, Lam): 15 x D^2 x 8 B": 15 * D * D * 8 / 2**20,
                                      "LQ scratch M, R (2 x 2D x D x 8 B)": 4 * D * D * 8 / 2**20},
                "batch": B}
        plan["total_GB"] = B * sum(plan["per_trajectory_MB"].values()) / 1024
        print(json.dumps({"config5_memory_plan": plan}), flush=True)
        config5(B=B, t1=10.0, tol=1e-4, num_save=200, strategies=("fixedpoint",))
        free, total = _device_memory()
        print(json.dumps({"device_memory_after_solve_GB": {"free": free / 2**30, "total": total / 2**30}}), flush=True)
        return
    from scipy.integrate import solve_ivp

    # C1: Van der Pol mu=1, d=2, nu=3, EKF0 isotropic, single trajectory
    vf, u0, (t0, t1), _ = ivps.van_der_pol_first_order(1.0)
    tr = solve_ivp(lambda t, y: vf(y), (t0, t1), u0[0], rtol=1e-12, atol=1e-12, method="DOP853").y[:, -1]
    run("C1 vdp mu=1 (single)", vf, [u0[0]], t0, t1, model="isotropic", corr="ts0", strategy="filter", nu=3,
        atol=1e-6, rtol=1e-6, dt0=0.1, truth=tr)
    # C2: stiff Van der Pol mu=1e3 (2nd order, d=1), nu=4, EKF1 dense, tolerance sweep B=256
    vf, u0, (t0, t1) = ivps.van_der_pol(1e3)
    B = 256
    tols = 10.0 ** np.linspace(-2, -8, B)
    run("C2 vdp mu=1e3 EKF1 sweep", vf, [np.tile(u0[0], (B, 1)), np.tile(u0[1], (B, 1))], t0, t1, model="dense", corr="ts1",
        strategy="filter", nu=4, atol=tols, rtol=tols, dt0=0.01)
    # C3a: rigid body d=3, t in [0, 50], nu = 2..5 isotropic fixedpoint, 1024 tolerances each
    vf, u0, (t0, t1), _ = ivps.rigid_body(time_span=(0.0, 50.0))
    tr = solve_ivp(lambda t, y: vf(y), (t0, t1), u0[0], rtol=1e-13, atol=1e-13, method="LSODA").y[:, -1]
    B = 1024  # BASELINE config 3: batch = 1024 on one GPU
    rt = 10.0 ** np.linspace(-1, -7, B)
    for nu in (2, 3, 4, 5):
        run(f"C3 rigid body nu={nu} sweep", vf, [np.tile(u0[0], (B, 1))], t0, t1, model="isotropic", corr="ts0",
            strategy="fixedpoint", nu=nu, atol=1e-3 * rt, rtol=rt, dt0=0.1)
    # C3b: Pleiades 2nd order d=14
    vf, u0, (t0, t1) = ivps.pleiades_2nd()
    for nu in (3, 5):
        run(f"C3 pleiades nu={nu} sweep", vf, [np.tile(u0[0], (B, 1)), np.tile(u0[1], (B, 1))], t0, t1, model="isotropic",
            corr="ts0", strategy="fixedpoint", nu=nu, atol=1e-3 * rt, rtol=rt, dt0=0.01)
    # C4: Brusselator N=32, B=1024: isotropic EKF0 (the model the reference uses) and dense EKF1 (headline)
    import bench

    vf, tcs = bench.problem_batch(1024)
    y0 = tcs[0]
    run("C4 brusselator iso EKF0 nu=3", vf, [y0], 0.0, 10.0, model="isotropic", corr="ts0", strategy="filter", nu=3,
        atol=1e-6, rtol=1e-6, dt0=0.01)
    run("C4 brusselator dense EKF1 nu=3", vf, [y0], 0.0, 10.0, model="dense", corr="ts1", strategy="filter", nu=3,
        atol=1e-6, rtol=1e-6, dt0=0.01)
    # opt-in device-resident controller on the latency-bound isotropic configurations
    run("C4 brusselator iso EKF0 nu=3", vf, [y0], 0.0, 10.0, model="isotropic", corr="ts0", strategy="filter", nu=3,
        atol=1e-6, rtol=1e-6, dt0=0.01, device_controller=True)
    run("C4 brusselator iso EKF0 nu=4 fixedpoint 200 ckpts", vf, [y0], 0.0, 10.0, model="isotropic", corr="ts0",
        strategy="fixedpoint", nu=4, atol=1e-8, rtol=1e-8, dt0=0.01, num_save=200)
    run("C4 brusselator iso EKF0 nu=4 fixedpoint 200 ckpts", vf, [y0], 0.0, 10.0, model="isotropic", corr="ts0",
        strategy="fixedpoint", nu=4, atol=1e-8, rtol=1e-8, dt0=0.01, num_save=200, device_controller=True)
    vf, u0, (t0, t1), _ = ivps.rigid_body(time_span=(0.0, 50.0))
    # (tolerance sweeps are bound by their slowest members: both kernels of the smoother's device controller)
    for dc in (True, "pipelined"):
        run("C3 rigid body nu=3 sweep", vf, [np.tile(u0[0], (B, 1))], t0, t1, model="isotropic", corr="ts0",
            strategy="fixedpoint", nu=3, atol=1e-3 * rt, rtol=rt, dt0=0.1, device_controller=dc)
    vf, u0, (t0, t1) = ivps.pleiades_2nd()
    for dc in (True, "pipelined"):
        run("C3 pleiades nu=3 sweep", vf, [np.tile(u0[0], (B, 1)), np.tile(u0[1], (B, 1))], t0, t1, model="isotropic",
            corr="ts0", strategy="fixedpoint", nu=3, atol=1e-3 * rt, rtol=rt, dt0=0.01, device_controller=dc)
    vf, u0, (t0, t1), _ = ivps.van_der_pol_first_order(1.0)
    run("C1 vdp mu=1 (single)", vf, [u0[0]], t0, t1, model="isotropic", corr="ts0", strategy="filter", nu=3,
        atol=1e-6, rtol=1e-6, dt0=0.1, device_controller=True)
    vf, u0, (t0, t1) = ivps.van_der_pol(1e3)
    run("C2 vdp mu=1e3 EKF1 sweep", vf, [np.tile(u0[0], (256, 1)), np.tile(u0[1], (256, 1))], t0, t1, model="dense", corr="ts1",
        strategy="filter", nu=4, atol=tols, rtol=tols, dt0=0.01, device_controller=True)
    config5()


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main()
