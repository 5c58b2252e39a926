#!/usr/bin/env python3
"""The reference's experiment 4 (experiments/4_brusselator/run.py:45-151) on the MI355X path.

Brusselator N = 2 ... 512, isotropic IWP(4) + EKF0 + fixed-point smoother + dynamic calibration, tol 1e-8,
dt0 0.01, 200 checkpoints. Writes the same result-dict schema as the reference
({"N", "runtime", "memory", "ts", "ys", "num_steps"}; run.py:23-38,142-151) to
experiments/data_checkpoint.npy and prints, per N, the accepted-step count next to the reference's committed
one (tests/golden/brusselator.npz) and the run time next to the reference's (JAX/CPU, compile time included).
The "textbook" half of the experiment (run.py:82-116: terminal-value solve for the step count and the memory a
store-every-step smoother would need, then that smoother itself while it fits) goes to experiments/data_textbook.npy
with the reference's keys; the smoother runs while the steps fit the device-side store (ivpsolve.SMOOTHER_MAX_STEPS)
and the reference's own 4000 MB gate.

  python experiments/brusselator_checkpoints.py [--max-n 512] [--controller host|device|pipelined] [--nosave] [--outdir DIR]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd"))

from odecheckpts import ivps  # noqa: E402
from probdiffeq_hip import ivpsolve, ivpsolvers, taylor  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-n", type=int, default=512)
    ap.add_argument("--controller", default="pipelined", choices=["host", "device", "pipelined"],
                    help="pipelined: device-resident controller with the backward transitions beside the filter (d <= 64; the one-wavefront "
                         "device controller beyond)")
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--outdir", default=os.path.dirname(os.path.abspath(__file__)))
    args = ap.parse_args(argv)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "brusselator.npz"))
    results = {"N": [], "runtime": [], "memory": [], "ts": [], "ys": [], "num_steps": []}
    textbook = {"N": [], "runtime": [], "memory": [], "ts": [], "ys": [], "num_steps": []}
    print(f"{'N':>5s} {'steps':>9s} {'reference':>9s} {'runtime s':>10s} {'ref runtime s':>13s} {'max |ys - ref|':>15s}")
    for i, N in enumerate(int(n) for n in golden["N"]):
        if N > args.max_n:
            break
        vf, u0, (t0, t1), params = ivps.brusselator(N=N)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            impl.select("isotropic", ode_shape=(2 * N,))
        num, tol = 4, 1e-8
        ctrl = ivpsolve.control_proportional_integral()
        ibm = ivpsolvers.prior_ibm(num_derivatives=num)
        ts0 = ivpsolvers.correction_ts0(ode_order=1)
        strategy = ivpsolvers.strategy_fixedpoint(ibm, ts0)
        solver = ivpsolvers.solver_dynamic(strategy)
        adaptive_solver = ivpsolve.adaptive(solver, atol=tol, rtol=tol, control=ctrl)
        tcoeffs = taylor.odejet_unroll(vf, u0, num=num)
        init = solver.initial_condition(tcoeffs, 1.0)
        save_at = np.linspace(t0, t1, num=200)
        size_init = 3 * 8 * ((num + 1) * 2 * N + 3 * (num + 1) ** 2 + (num + 1) * 2 * N)  # three state copies (run.py:68-70)
        # ---- textbook half (run.py:82-116)
        terminal = ivpsolve.solve_adaptive_terminal_values(vf, init, t0=t0, t1=t1, dt0=0.01, adaptive_solver=adaptive_solver)
        nsteps_terminal = int(np.amax(terminal.num_steps))
        total_memory = nsteps_terminal * size_init / 1024**2
        textbook["N"].append(N)
        textbook["memory"].append(total_memory)
        textbook["num_steps"].append(nsteps_terminal)
        if total_memory < 4000 and nsteps_terminal < ivpsolve.SMOOTHER_MAX_STEPS:
            solver_ = ivpsolvers.solver_dynamic(ivpsolvers.strategy_smoother(ibm, ts0))
            adaptive_solver_ = ivpsolve.adaptive(solver_, atol=tol, rtol=tol, control=ctrl)
            count0 = time.perf_counter()
            ivpsolve.solve_adaptive_save_every_step(vf, solver_.initial_condition(tcoeffs, 1.0), t0=t0, t1=t1, dt0=0.01,
                                                    adaptive_solver=adaptive_solver_)
            textbook["runtime"].append(time.perf_counter() - count0)
        # ---- checkpoint half (run.py:118-138)
        count0 = time.perf_counter()
        solution = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=adaptive_solver,
                                                   device_controller={"host": False, "device": True, "pipelined": "pipelined"}[args.controller])
        count1 = time.perf_counter() - count0
        steps = int(np.amax(solution.num_steps))
        results["N"].append(N)
        results["runtime"].append(count1)
        results["memory"].append(len(save_at) * size_init / 1024**2)
        results["ts"].append(solution.t)
        results["ys"].append(solution.u)
        results["num_steps"].append(steps)
        key = f"ys_N{N}"
        dev = f"{np.max(np.abs(solution.u - golden[key])):15.2e}" if key in golden else f"{'(not committed)':>15s}"
        print(f"{N:5d} {steps:9d} {int(golden['num_steps_checkpoint'][i]):9d} {count1:10.3f} {golden['runtime_checkpoint'][i]:13.3f} {dev}",
              flush=True)
    if not args.nosave:
        os.makedirs(args.outdir, exist_ok=True)
        np.save(os.path.join(args.outdir, "data_checkpoint.npy"), results, allow_pickle=True)
        np.save(os.path.join(args.outdir, "data_textbook.npy"), textbook, allow_pickle=True)


if __name__ == "__main__":
    main()
