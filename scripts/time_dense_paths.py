#!/usr/bin/env python3
"""Step time of the dense kernels by path and size (Brusselator N, nu = 3): A/B tool behind the numbers in DESIGN.md §6.

  python scripts/time_dense_paths.py N B PATH [filter|fixedpoint] [--phases] [--nu=3]

PATH: 0 default dispatch, 1 generic kernels, 2 large-D kernels, 3 MFMA kernels with the two-launch LQ, 4 large-D kernels with the 64 x 64
GEMM tile only (pnode_set_generic_path).
--phases prints the begin | LQ input | LQ | measurement-update split from HIP events (filter only).
Run under `rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 scripts/time_dense_paths.py ...` and feed OUT to
scripts/parse_kernel_trace.py for the per-kernel table.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, sub))

import numpy as np  # noqa: E402
from helpers import brusselator_y0, tcoeffs  # noqa: E402
from probdiffeq_hip import _backend  # noqa: E402
from probdiffeq_hip import vector_fields as V  # noqa: E402


def run(N, B, path, strategy="filter", phases=False, steps=4, nu=3):
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, 0.1 * b) for b in range(B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy=strategy,
                        calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
    if path:
        bt.set_generic_path(path)
    bt.set_initial(tc, np.zeros(B), np.ones(B))
    bt.set_tolerances(1e-6, 1e-6)
    err, one = np.zeros(B), np.ones(B, np.uint8)
    bt.attempt_step(np.full(B, 0.005), one, err)
    bt.commit(one)
    if phases:
        bt.set_phase_timing(True)
    ph = np.zeros(4)
    t0 = time.perf_counter()
    for _ in range(steps):
        bt.attempt_step(np.full(B, 0.005), one, err)
        bt.commit(one)
        if phases:
            ph += np.array(bt.last_phase_ms())
    dt = (time.perf_counter() - t0) / steps
    line = f"N={N} D={(nu + 1) * 2 * N} B={B} path={path} {strategy}: {dt * 1e3:.2f} ms/step err0={err[0]:.6e}"
    if phases:
        line += f" phases(begin, lq_input, lq, correct)={np.round(ph / steps, 3)}"
    if strategy == "fixedpoint":
        tq = np.full(B, 0.005 * (steps + 1) - 0.002)
        t0 = time.perf_counter()
        bt.interpolate(tq, np.zeros(B, np.int32))
        line += f" interpolate={1e3 * (time.perf_counter() - t0):.2f} ms"
    print(line, flush=True)
    bt.close()


if __name__ == "__main__":
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    nu = next((int(x.split("=")[1]) for x in sys.argv if x.startswith("--nu=")), 3)
    run(int(a[0]), int(a[1]), int(a[2]), a[3] if len(a) > 3 else "filter", phases="--phases" in sys.argv, nu=nu)
