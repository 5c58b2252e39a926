#!/usr/bin/env python3
"""The reference's experiment 2 (experiments/2_workprec_simple/run_simple.py:17-217) on the MI355X path: work-precision of the
probabilistic solvers on the rigid-body problem, TS0(2) / TS0(4) through `odecheckpts.solve` ("jit loop" labels: fixed-point
smoother, five checkpoints) and through `odecheckpts.solve_via_interpolate` ("jit step" labels: store-every-step smoother +
off-grid marginals), tolerances 0.1**arange(start, stop) with rtol = 100 tol, atol = 1e-3 rtol (run_simple.py:40-42).

Writes the reference's result files with the reference's schema (run_simple.py:128-136, 206-215): `data_results.npy`, a pickled
dict {label: {"list_of_args", "length_of_longest_vector", "work_min", "work_mean", "work_std", "precision"}}, plus
`data_ts.npy`, `data_ys.npy`, `data_checkpoints.npy`, so that the reference's plot_simple.py reads them unchanged. The
Runge-Kutta competitors (diffrax) are out of scope; the reference solution is scipy DOP853 at 1e-13 instead of Dopri8 at 1e-15.
An extra label, "TS0(2) (one batch) via probdiffeq_hip", times the whole tolerance sweep as ONE batched solve (work = time of
the batch / number of tolerances): the sweep is the batch axis on the GPU.

  python experiments/workprec_simple.py [--start 3 --stop 10 --repeats 3 --nosave --outdir DIR]
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

KEYS = ("list_of_args", "length_of_longest_vector", "work_min", "work_mean", "work_std", "precision")


def workprec(fun, *, precision_fun, timeit_fun):
    """param -> (solution, aux)  becomes  params -> work-precision record (run_simple.py:181-215)."""

    def parameter_list_to_workprecision(list_of_args):
        rec = {k: [] for k in KEYS[1:]}
        for arg in list_of_args:
            sol, aux = fun(arg)
            times = timeit_fun(lambda: fun(arg)[0])  # noqa: B023
            rec["length_of_longest_vector"].append(len(aux["u0_solve"]))
            rec["precision"].append(precision_fun(sol))
            rec["work_min"].append(min(times))
            rec["work_mean"].append(statistics.mean(times))
            rec["work_std"].append(statistics.stdev(times) if len(times) > 1 else 0.0)
        out = {k: np.asarray(v) for k, v in rec.items()}
        out["list_of_args"] = np.asarray(list_of_args)
        return out

    return parameter_list_to_workprecision


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--start", type=int, default=3)
    ap.add_argument("--stop", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--nosave", action="store_true")
    ap.add_argument("--outdir", default=os.path.dirname(os.path.abspath(__file__)))
    args = ap.parse_args(argv)

    vf, u0, tspan, params = ivps.rigid_body(time_span=(0.0, 50.0))
    # plot-quality truth as the reference takes it (run_simple.py:22-24 -> ivpsolvers.asolve_scipy)
    ts, ys = ivpsolvers.asolve_scipy("LSODA", vf, tspan, atol=1e-13, rtol=1e-13)(tuple(u0[:1]), params)
    n0, n1 = args.start, args.stop
    n1_short = n0 + (1 + n1 - n0) * 2 // 3
    tols = 0.1 ** np.arange(n0, n1, step=1.0)
    tols_short = 0.1 ** np.arange(n0, n1_short, step=1.0)
    xs = np.linspace(ts.min(), ts.max(), num=5)
    dt0 = ts.max() - ts.min()
    reference = scipy.integrate.solve_ivp(lambda t, y: vf(y, t=t, p=params), tspan, u0[0], method="DOP853", atol=1e-13, rtol=1e-13,
                                          t_eval=xs).y.T

    def precision(received):
        err = np.abs(reference - np.asarray(received))
        return np.linalg.norm(err) / np.sqrt(err.size)

    def timer(fun):
        fun()
        return list(timeit.repeat(fun, number=1, repeat=args.repeats))

    def loop(method):
        def fun(tol):
            rtol = 100.0 * tol
            return ivpsolvers.solve(method, vf, u0[0], save_at=xs, dt0=dt0, atol=1e-3 * rtol, rtol=rtol)(u0, params)
        return fun

    def step(method):
        def fun(tol):
            rtol = 100.0 * (1e-3 if tol < 1e-8 else tol)
            return ivpsolvers.solve_via_interpolate(method, vf, u0[0], save_at=xs, dt0=dt0, atol=1e-3 * rtol, rtol=rtol)(u0, params)
        return fun

    algorithms = {
        "TS0(2) (jit step) via probdiffeq": (tols_short, step("ts0-2")),
        "TS0(4) (jit step) via probdiffeq": (tols_short, step("ts0-4")),
        "TS0(2) (jit loop) via probdiffeq": (tols, loop("ts0-2")),
        "TS0(4) (jit loop) via probdiffeq": (tols, loop("ts0-4")),
    }
    results = {}
    for label, (tt, algo) in algorithms.items():
        results[label] = workprec(algo, precision_fun=precision, timeit_fun=timer)(tt)
        print(label, "precision", np.array2string(results[label]["precision"], precision=3), "work_min",
              np.array2string(results[label]["work_min"], precision=4), flush=True)
    # the sweep as one batch
    for method in ("ts0-2", "ts0-4"):
        rt = 100.0 * tols
        u0b = (np.broadcast_to(u0[0], (len(tols), 3)).copy(),)
        fun = lambda: ivpsolvers.solve(method, vf, u0[0], save_at=xs, dt0=dt0, atol=1e-3 * rt, rtol=rt)(u0b, params)  # noqa: E731,B023
        sol, aux = fun()
        times = timer(lambda: fun()[0])
        k = len(tols)
        results[f"TS0({method[-1]}) (one batch) via probdiffeq_hip"] = {
            "list_of_args": tols, "length_of_longest_vector": np.full(k, len(xs)),
            "work_min": np.full(k, min(times) / k), "work_mean": np.full(k, statistics.mean(times) / k),
            "work_std": np.full(k, (statistics.stdev(times) if len(times) > 1 else 0.0) / k),
            "precision": np.asarray([precision(sol[i]) for i in range(k)])}
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
