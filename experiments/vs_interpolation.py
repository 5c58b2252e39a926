#!/usr/bin/env python3
"""The reference's experiment 5 (experiments/5_vs_interpolation/measure.py:27-232) on the MI355X path.

Restricted three-body problem (Arenstorf orbit, second-order form), isotropic IWP(4) + EKF0, the uncalibrated solver.
Two ways to K joint posterior samples at 50 save_at points, for K = 5, 50, 500 and tol = 1e-4, 1e-7, 1e-10:

  "ATS (ours)": strategy_fixedpoint + solve_adaptive_save_at, then stats.markov_sample over the 49 stored transitions;
  "AS":         strategy_smoother + solve_adaptive_save_every_step for the grid, solve_fixed_grid on (that grid U save_at), then
                stats.markov_sample over EVERY step and a selection of the save_at entries.

Writes the reference's result schema -- {i: {"No. Samples", "Tolerance", "No. steps", "Time (s): AS", "Time (s): ATS (ours)"}}
to experiments/data_results_vs_interpolation.npy and the baseline solution (nu = 3 filter at every accepted step, tol 1e-7) to
experiments/data_solution_vs_interpolation.npy -- and prints "No. steps" next to the reference's committed 448 / 2,570 / 14,469.

  python experiments/vs_interpolation.py [--samples 5 50 500] [--nosave] [--outdir DIR]
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
from probdiffeq_hip import ivpsolve, ivpsolvers, stats, taylor  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402

TOLS = ((1e-4, "$10^{-4}$"), (1e-7, "$10^{-7}$"), (1e-10, "$10^{-10}$"))


def build(vf, inits, t0, kind, num_derivs):
    prior = ivpsolvers.prior_ibm(num_derivatives=num_derivs)
    strategy = getattr(ivpsolvers, f"strategy_{kind}")(prior, ivpsolvers.correction_ts0(ode_order=2))
    solver = ivpsolvers.solver(strategy)
    tcoeffs = taylor.odejet_padded_scan(vf, inits, num=num_derivs + 1 - 2)
    return solver, solver.initial_condition(tcoeffs, 1.0)


def sample_mean(solution, num_samples):
    """Mean over the joint samples at the stored points, terminal point appended (measure.py:69-75)."""
    seq = stats.markov_select_terminal(solution.posterior)
    (qoi, _), (init, _) = stats.markov_sample(1, seq, shape=(num_samples,), reverse=True)
    return np.concatenate([qoi, init[..., None, :]], axis=-2).mean(axis=0)


def checkpoint_runner(vf, inits, tspan, num_samples):
    solver, init = build(vf, inits, tspan[0], "fixedpoint", 4)

    def run(tol, save_at):
        asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol, control=ivpsolve.control_proportional_integral())
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller="pipelined")
        return sample_mean(sol, num_samples), len(save_at)

    return run


def textbook_runner(vf, inits, tspan, num_samples):
    solver, init = build(vf, inits, tspan[0], "smoother", 4)

    def adaptive_grid(tol, save_at):
        asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol, control=ivpsolve.control_proportional_integral())
        sol = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=save_at[0], t1=save_at[-1], dt0=0.01, adaptive_solver=asolver)
        return sol.t

    def run(grid, save_at):
        sol = ivpsolve.solve_fixed_grid(vf, init, grid=grid, solver=solver)
        means = sample_mean(sol, num_samples)
        return means[np.searchsorted(grid, save_at)], len(grid) - 1

    return adaptive_grid, run


def best_of(fun, runs=3):
    out = []
    for _ in range(runs):
        tic = time.perf_counter()
        fun()
        out.append(time.perf_counter() - tic)
    return min(out)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, nargs="+", default=[5, 50, 500])
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--outdir", default=os.path.dirname(os.path.abspath(__file__)))
    args = ap.parse_args(argv)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "three_body.npz"))
    want = dict(zip((float(t) for t in golden["smoother_tols"]), (int(n) for n in golden["smoother_num_steps"])))
    vf, inits, tspan = ivps.three_body_restricted()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        impl.select("isotropic", ode_shape=(2,))
    # baseline: nu = 3 filter at every accepted step (measure.py:211-232)
    solver, init = build(vf, inits, tspan[0], "filter", 3)
    asolver = ivpsolve.adaptive(solver, atol=1e-7, rtol=1e-7, control=ivpsolve.control_proportional_integral())
    baseline = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=tspan[0], t1=tspan[1], dt0=0.01, adaptive_solver=asolver)
    print(f"baseline: {len(baseline.t) - 1} accepted steps (reference {len(golden['baseline_solution']) - 1}), "
          f"max |u - reference| {np.max(np.abs(baseline.u - golden['baseline_solution'])):.1e}")
    save_at = np.linspace(tspan[0], tspan[1])  # 50 points (measure.py:186)
    results, i = {}, 1
    for n in args.samples:
        ckpt = checkpoint_runner(vf, inits, tspan, n)
        grid_of, text = textbook_runner(vf, inits, tspan, n)
        for tol, label in TOLS:
            grid = grid_of(tol, save_at)
            nsteps = len(grid) - 1
            full = np.union1d(grid, save_at)
            t_as = best_of(lambda: text(full, save_at))
            t_ats = best_of(lambda: ckpt(tol, save_at))
            results[i] = {"No. Samples": f"{n}", "Tolerance": label, "No. steps": f"{nsteps:,}", "Time (s): AS": f"{t_as:.3f}",
                          "Time (s): ATS (ours)": f"{t_ats:.3f}"}
            print(f"K={n:4d} tol={tol:.0e}: steps {nsteps:6,d} (reference {want[tol]:6,d})  AS {t_as:.3f} s  ATS {t_ats:.3f} s")
            i += 1
    if not args.nosave:
        os.makedirs(args.outdir, exist_ok=True)
        np.save(os.path.join(args.outdir, "data_results_vs_interpolation.npy"), results, allow_pickle=True)
        np.save(os.path.join(args.outdir, "data_solution_vs_interpolation.npy"), baseline.u)
        print("saved to", args.outdir)
    return results, baseline


if __name__ == "__main__":
    main()
