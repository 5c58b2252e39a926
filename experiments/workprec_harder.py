#!/usr/bin/env python3
"""The reference's experiment 3 (experiments/3_workprec_harder/run_harder.py:17-122) on the MI355X path: work-precision of the
probabilistic solvers on the Pleiades problem in its second-order form (d = 14, `ode_order=2`), TS0(3) / TS0(5) / TS0(8) through
`odecheckpts.solve` (fixed-point smoother, 50 checkpoints), tolerances 0.1**arange(start, stop) with rtol = 10 tol,
atol = 1e-3 rtol, dt0 = 0.1 (run_harder.py:42-60).

Writes the reference's result files with the reference's schema (run_harder.py:111-115, 176-183): `data_results.npy`, a pickled
dict {label: {"list_of_args", "length_of_longest_vector", "work_min", "work_mean", "work_std", "precision"}}, plus
`data_ts.npy`, `data_ys.npy`, `data_checkpoints.npy`, so that the reference's plot_harder.py reads them unchanged. The Runge-Kutta
competitors (diffrax) are out of scope; the reference solution is scipy DOP853 at 1e-13 on the first-order form instead of
Dopri5 at 1e-15. An extra label per order, "Prob(nu) (one batch) via probdiffeq_hip", times the whole tolerance sweep as ONE
batched solve with the device-resident controller (work = time of the batch / number of tolerances).

  python experiments/workprec_harder.py [--start 3 --stop 10 --repeats 3 --nosave --outdir DIR]
"""
import argparse
import os
import statistics
import sys
import timeit

import numpy as np
import scipy.integrate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "code-adaptive-prob-ode-solvers_amd"))

from odecheckpts import ivps, ivpsolvers  # noqa: E402
from probdiffeq_hip import ivpsolve, stats, taylor  # noqa: E402
from probdiffeq_hip import ivpsolvers as pn_solvers  # noqa: E402
from probdiffeq_hip.impl import impl  # noqa: E402

KEYS = ("list_of_args", "length_of_longest_vector", "work_min", "work_mean", "work_std", "precision")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--start", type=int, default=3)
    ap.add_argument("--stop", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--outdir", default=os.path.dirname(os.path.abspath(__file__)))
    args = ap.parse_args(argv)

    vf1, u01, tspan, params = ivps.pleiades_1st()
    ts, ys = ivpsolvers.asolve_scipy("LSODA", vf1, tspan, atol=1e-13, rtol=1e-13)(u01, params)
    vf, u0, tspan = ivps.pleiades_2nd()  # (the reference unpacks a fourth value, `params`, that its own ivps.py:40,98 does not return)
    params = ()
    tols = 0.1 ** np.arange(args.start, args.stop, step=1.0)
    xs = np.linspace(ts.min(), ts.max(), num=50)
    dt0 = 0.1
    reference = scipy.integrate.solve_ivp(lambda t, y: vf1(y, t=t, p=params), tspan, u01[0], method="DOP853", atol=1e-13, rtol=1e-13,
                                          t_eval=xs).y.T[:, :14]

    def precision(received):
        err = np.abs(reference - np.asarray(received))
        return np.linalg.norm(err) / np.sqrt(err.size)

    def timer(fun):
        fun()
        return list(timeit.repeat(fun, number=1, repeat=args.repeats))

    def alg_ts0(n):
        def fun(tol):
            rtol = 10.0 * tol
            return ivpsolvers.solve(f"ts0-{n}", vf, u0[0], save_at=xs, dt0=dt0, atol=1e-3 * rtol, rtol=rtol, ode_order=2)(u0, params)
        return fun

    results = {}
    for n in (3, 5, 8):
        label = f"Prob({n}) via probdiffeq"
        rec = {k: [] for k in KEYS[1:]}
        for tol in tols:
            sol, aux = alg_ts0(n)(tol)
            times = timer(lambda: alg_ts0(n)(tol)[0])  # noqa: B023
            rec["length_of_longest_vector"].append(len(aux["u0_solve"]))
            rec["precision"].append(precision(sol))
            rec["work_min"].append(min(times))
            rec["work_mean"].append(statistics.mean(times))
            rec["work_std"].append(statistics.stdev(times) if len(times) > 1 else 0.0)
        results[label] = {k: np.asarray(v) for k, v in rec.items()}
        results[label]["list_of_args"] = np.asarray(tols)
        print(label, "precision", np.array2string(results[label]["precision"], precision=3), "work_min",
              np.array2string(results[label]["work_min"], precision=4), flush=True)
    # the sweep as one batch, controller on the device
    k = len(tols)
    for n in (3, 5, 8):
        impl.select("isotropic", ode_shape=(14,))
        solver = pn_solvers.solver_dynamic(pn_solvers.strategy_fixedpoint(pn_solvers.prior_ibm(n), pn_solvers.correction_ts0(ode_order=2)))
        rt = 10.0 * tols
        u0b = tuple(np.broadcast_to(u, (k, 14)).copy() for u in u0)
        vfb = vf.bind(params)
        init = solver.initial_condition(taylor.odejet_padded_scan(vfb, u0b, num=n - 1), output_scale=1.0)
        asolver = ivpsolve.adaptive(solver, atol=1e-3 * rt, rtol=rt)

        def fun():
            sol = ivpsolve.solve_adaptive_save_at(vfb, init, save_at=xs, dt0=dt0, adaptive_solver=asolver, device_controller=True)  # noqa: B023
            post = stats.markov_select_terminal(sol.posterior)
            margs = stats.markov_marginals(post, reverse=True)
            return np.concatenate([margs.mean, post.terminal[:, None]], axis=1)[:, :, 0, :]

        sol = fun()
        times = timer(fun)
        results[f"Prob({n}) (one batch) via probdiffeq_hip"] = {
            "list_of_args": tols, "length_of_longest_vector": np.full(k, len(xs)),
            "work_min": np.full(k, min(times) / k), "work_mean": np.full(k, statistics.mean(times) / k),
            "work_std": np.full(k, (statistics.stdev(times) if len(times) > 1 else 0.0) / k),
            "precision": np.asarray([precision(sol[i]) for i in range(k)])}
        print(f"Prob({n}) (one batch)", "precision", np.array2string(results[f"Prob({n}) (one batch) via probdiffeq_hip"]["precision"], precision=3),
              "work per tolerance", f"{min(times) / k:.4g}", flush=True)
    if args.nosave:
        print("Skipped saving.")
    else:
        os.makedirs(args.outdir, exist_ok=True)
        np.save(os.path.join(args.outdir, "data_results.npy"), results, allow_pickle=True)
        np.save(os.path.join(args.outdir, "data_ts.npy"), ts)
        np.save(os.path.join(args.outdir, "data_ys.npy"), ys)
        np.save(os.path.join(args.outdir, "data_checkpoints.npy"), xs)
        print("Saving successful.")
    return results


if __name__ == "__main__":
    main()
