"""Generate the golden fixtures under tests/golden/ from the reference's committed result files.

Run in the BUILD container only (needs /root/reference); the fixtures it writes are plain
float64/int64 arrays (np.savez, no pickle) and are what travels to the GPU box.

Sources (all are outputs of the real JAX/probdiffeq path, committed in the reference):
  experiments/1_van_der_pol/vdp_baseline_grid.npy, vdp_baseline_solution.npy  (vdp.py:74-80,113-116)
  experiments/1_van_der_pol/vdp_grid_fixed_inaccurate.npy                        (vdp.py:94-97)
  experiments/1_van_der_pol/vdp_runtime_*.npy                                    (vdp.py:28-43)
  experiments/4_brusselator/data_checkpoint.npy  ys/ts/num_steps/runtime         (run.py:119-151)
  experiments/4_brusselator/data_textbook.npy    num_steps                       (run.py:81-99)
  experiments/2_workprec_simple/data_ts.npy, data_ys.npy (rigid body truth)      (run_simple.py:22-24)
  experiments/3_workprec_harder/data_ts.npy, data_ys.npy (Pleiades truth)        (run_harder.py:22-24)
  experiments/{2,3}_*/data_results.npy  precision / work arrays                  (run_simple.py:208-215)
  experiments/5_vs_interpolation/data_solution.npy  filter solution at every accepted step, restricted three-body problem
                                                    (measure.py:155-163,211-232); data_results.npy "No. steps" (measure.py:177-199)

The dict-valued .npy files hold pickled jax.Array objects. They are read with a WHITELIST
unpickler: only numpy globals plus a stub for jax._src.array._reconstruct_array are allowed.
"""

import importlib
import os
import pickle

import numpy as np

REF = "/root/reference/experiments"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Whitelist(pickle.Unpickler):
    def find_class(self, module, name):
        if module.split(".")[0] == "numpy":
            return getattr(importlib.import_module(module), name)
        if (module, name) == ("jax._src.array", "_reconstruct_array"):

            def rebuild(fun, args, arr_state, aval_state):
                a = fun(*args)
                a.__setstate__(arr_state)
                return a

            return rebuild
        raise pickle.UnpicklingError(f"global {module}.{name} is not whitelisted")


def load_object_npy(path):
    with open(path, "rb") as f:
        version = np.lib.format.read_magic(f)
        if version == (1, 0):
            _, _, dtype = np.lib.format.read_array_header_1_0(f)
        else:
            _, _, dtype = np.lib.format.read_array_header_2_0(f)
        assert dtype.hasobject, path
        obj = _Whitelist(f).load()
    if isinstance(obj, np.ndarray) and obj.shape == ():
        obj = obj.item()
    return obj


def f64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


def main():
    # --- Van der Pol (dense EKF1, nu=4, filter, dynamic) -------------------------------------
    d1 = f"{REF}/1_van_der_pol"
    np.savez(
        f"{OUT}/vdp.npz",
        grid=f64(np.load(f"{d1}/vdp_baseline_grid.npy")),
        solution=f64(np.load(f"{d1}/vdp_baseline_solution.npy")),
        grid_adaptive=f64(np.load(f"{d1}/vdp_grid_adaptive.npy")),
        grid_fixed_inaccurate=f64(np.load(f"{d1}/vdp_grid_fixed_inaccurate.npy")),
        runtime_adaptive=f64(np.load(f"{d1}/vdp_runtime_adaptive.npy")),
        runtime_fixed_accurate=f64(np.load(f"{d1}/vdp_runtime_fixed_accurate.npy")),
        runtime_fixed_inaccurate=f64(np.load(f"{d1}/vdp_runtime_fixed_inaccurate.npy")),
    )

    # --- Brusselator (isotropic TS0, nu=4, fixedpoint, dynamic, tol 1e-8) --------------------
    d4 = f"{REF}/4_brusselator"
    ck = load_object_npy(f"{d4}/data_checkpoint.npy")
    tb = load_object_npy(f"{d4}/data_textbook.npy")
    out = {
        "N": np.asarray(ck["N"], dtype=np.int64),
        "num_steps_checkpoint": np.asarray([int(x) for x in ck["num_steps"]], dtype=np.int64),
        "num_steps_terminal": np.asarray([int(x) for x in tb["num_steps"]], dtype=np.int64),
        "runtime_checkpoint": f64([float(x) for x in ck["runtime"]]),
        "runtime_textbook": f64([float(x) for x in tb["runtime"]]),
        "memory_checkpoint_mb": f64([float(x) for x in ck["memory"]]),
        "memory_textbook_mb": f64([float(x) for x in tb["memory"]]),
        "ts": f64(ck["ts"][0]),
    }
    for n, ts, ys in zip(ck["N"], ck["ts"], ck["ys"]):
        assert np.array_equal(f64(ts), out["ts"])
        if n <= 128:  # 200 x 2N doubles; N=128 -> 410 KB. Larger N: step counts only.
            out[f"ys_N{n}"] = f64(ys)
    np.savez_compressed(f"{OUT}/brusselator.npz", **out)

    # --- Rigid body / Pleiades: truth solutions + work-precision curves ----------------------
    for tag, sub in (("rigid_body", "2_workprec_simple"), ("pleiades", "3_workprec_harder")):
        dd = f"{REF}/{sub}"
        ts = f64(np.load(f"{dd}/data_ts.npy"))
        ys = f64(np.load(f"{dd}/data_ys.npy"))
        # keep every 8th point of the dense truth (plot-quality grid) + the end point
        idx = np.unique(np.concatenate([np.arange(0, len(ts), 8), [len(ts) - 1]]))
        res = load_object_npy(f"{dd}/data_results.npy")
        flat = {"ts": ts[idx], "ys": ys[idx]}
        cps = f"{dd}/data_checkpoints.npy"
        if os.path.exists(cps):
            flat["checkpoints"] = f64(np.load(cps))
        for method, vals in res.items():
            for key, arr in vals.items():
                flat[f"{method}__{key}"] = f64(arr)
        np.savez_compressed(f"{OUT}/{tag}.npz", **flat)
        print(tag, {k: v.shape for k, v in flat.items()})

    # --- Restricted three-body problem (isotropic TS0, ode_order 2, uncalibrated solver) -----
    d5 = f"{REF}/5_vs_interpolation"
    res5 = load_object_npy(f"{d5}/data_results.npy")
    steps = {}
    for row in res5.values():  # the nine rows repeat three tolerances
        tol = {"$10^{-4}$": 1e-4, "$10^{-7}$": 1e-7, "$10^{-10}$": 1e-10}[row["Tolerance"]]
        steps[tol] = int(row["No. steps"].replace(",", ""))
    np.savez_compressed(
        f"{OUT}/three_body.npz",
        baseline_solution=f64(np.load(f"{d5}/data_solution.npy")),  # nu = 3, filter, tol 1e-7, every accepted step (measure.py:163)
        smoother_tols=f64(sorted(steps, reverse=True)),
        smoother_num_steps=np.asarray([steps[t] for t in sorted(steps, reverse=True)], dtype=np.int64),  # nu = 4 (measure.py:177-181)
    )

    for name in ("vdp.npz", "brusselator.npz", "rigid_body.npz", "pleiades.npz", "three_body.npz"):
        print(name, os.path.getsize(f"{OUT}/{name}") // 1024, "KB")


if __name__ == "__main__":
    main()
