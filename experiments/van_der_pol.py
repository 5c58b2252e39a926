#!/usr/bin/env python3
"""The reference's experiment 1 (experiments/1_van_der_pol/vdp.py:45-132) on the MI355X path: stiff Van der Pol (mu = 1e3, second-order
form, d = 1), IWP(4) prior, dense EKF1, dynamic calibration.

  1. baseline: adaptive solve at tol 1e-3 that stores every accepted step (its grid is the work an adaptive solver needs);
  2. the same grid replayed by the fixed-step solver ("adaptive"), a fixed grid with the same number of points
     ("fixed, inaccurate": diverges) and one fine enough for the smallest adaptive step ("fixed, accurate"), each timed.

Writes the reference's eight files (<outdir>/vdp_baseline_grid.npy, vdp_baseline_solution.npy, vdp_grid_{adaptive, fixed_accurate,
fixed_inaccurate}.npy, vdp_runtime_{...}.npy) and prints grid sizes and run times next to the reference's committed ones
(tests/golden/vdp.npz). The reference asserts NaNs on the coarse grid; this path overflows to 1e62 without a NaN (numerics):
the script reports which it was.

  python experiments/van_der_pol.py [--nosave] [--outdir DIR]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd"))

from probdiffeq_hip import ivpsolve, ivpsolvers, taylor  # noqa: E402
from probdiffeq_hip import vector_fields as V  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402


def timed(fun):
    fun()  # warm-up (code paging), as the reference's jit_and_time runs the function twice
    tic = time.perf_counter()
    out = fun()
    return time.perf_counter() - tic, out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--outdir", default=os.path.dirname(os.path.abspath(__file__)))
    args = ap.parse_args(argv)
    t0, t1, num = 0.0, 6.3, 4
    vf = V.van_der_pol_order2(1e3)
    impl.select("dense", ode_shape=(1,))
    strategy = ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(num_derivatives=num), ivpsolvers.correction_ts1(ode_order=2))
    solver = ivpsolvers.solver_dynamic(strategy)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [np.array([2.0]), np.array([0.0])], num=num - 1), 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-3, rtol=1e-3, control=ivpsolve.control_proportional_integral())
    base = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=t0, t1=t1, dt0=0.01, adaptive_solver=asolver)
    steps = np.diff(base.t)
    required = int((t1 - t0) / np.amin(steps))
    grids = {"adaptive": base.t, "fixed_inaccurate": np.linspace(t0, t1, len(base.t)), "fixed_accurate": np.linspace(t0, t1, required)}
    runs = {name: timed(lambda g=g: ivpsolve.solve_fixed_grid(vf, init, grid=g, solver=solver)) for name, g in grids.items()}
    coarse = runs["fixed_inaccurate"][1].u
    how = "NaN" if np.any(np.isnan(coarse)) else ("non-finite" if not np.all(np.isfinite(coarse)) else f"finite, max |u| = {np.max(np.abs(coarse)):.1e}")
    assert (not np.all(np.isfinite(coarse))) or np.max(np.abs(coarse)) > 1e6, "the coarse equidistant grid is supposed to diverge"
    try:
        ref = np.load(os.path.join(ROOT, "tests", "golden", "vdp.npz"))
        ref_n = f" (reference {len(ref['grid'])})"
    except OSError:
        ref_n = ""
    print(f"baseline: {len(base.t)} grid points{ref_n}, smallest step {np.amin(steps):.2e} -> {required} equidistant steps for the same resolution")
    print(f"coarse equidistant grid: {how}")
    for name, (rt, sol) in runs.items():
        print(f"  {name:17s} {len(sol.t):9,d} points  {rt:8.4f} s")
    if not args.nosave:
        os.makedirs(args.outdir, exist_ok=True)
        pre = os.path.join(args.outdir, "vdp")
        np.save(pre + "_baseline_grid.npy", base.t)
        np.save(pre + "_baseline_solution.npy", base.u)
        for name, (rt, sol) in runs.items():
            np.save(f"{pre}_grid_{name}.npy", sol.t)
            np.save(f"{pre}_runtime_{name}.npy", np.asarray(rt))
        print("saved to", args.outdir)
    return base, runs


if __name__ == "__main__":
    main()
