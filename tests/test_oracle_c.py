"""The C restatement (oracle/pn_oracle_c.c, the OpenMP CPU baseline) against the NumPy oracle."""

import ctypes as C
import os

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import vector_fields as V

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libpn_oracle_c.so"))
    dp = C.POINTER(C.c_double)
    lib.pnoc_step_brusselator.restype = C.c_double
    lib.pnoc_step_brusselator.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, dp, dp, dp]
    lib.pnoc_bench_brusselator.restype = C.c_double
    lib.pnoc_bench_brusselator.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_double, C.c_double, C.c_double, C.c_int,
                                           C.POINTER(C.c_long), C.POINTER(C.c_long)]
    return lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _openblas_path():
    import glob

    import scipy

    hits = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so"))
    return hits[0].encode() if hits else None


@pytest.mark.parametrize("backend", ["loops", "lapack"])
def test_c_step_matches_numpy_oracle(backend):
    lib = _lib()
    lib.pnoc_use_lapack.argtypes = [C.c_char_p]
    if backend == "lapack":
        path = _openblas_path()
        if path is None or lib.pnoc_use_lapack(path) != 1:
            pytest.skip("no OpenBLAS with LAPACK found next to scipy")
    else:
        lib.pnoc_use_lapack(None)
    N, nu = 6, 3
    vf = V.brusselator(N)
    d, D = 2 * N, (nu + 1) * 2 * N
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    prob = oracle_problem(vf)
    st = po.initial_state(cfg, tcoeffs(vf, [brusselator_y0(N, 0.2)], nu), 0.0)
    mean = np.ascontiguousarray(st.mean.reshape(-1))
    L = np.zeros((D, D))
    for dt in (0.01, 0.02, 0.015):
        e_ref, st = po.step(cfg, prob, st, dt)
        mo, Lo, err = np.empty(D), np.empty((D, D)), np.empty(d)
        sigma = lib.pnoc_step_brusselator(N, nu, _p(mean), _p(L), dt, _p(mo), _p(Lo), _p(err))
        assert abs(sigma - st.output_scale) <= 1e-8 * st.output_scale
        assert np.allclose(err, e_ref, rtol=1e-7, atol=0)
        assert np.max(np.abs(mo[:d] - st.mean[0])) < 1e-12
        cov, cov_ref = Lo @ Lo.T, st.chol @ st.chol.T
        assert np.max(np.abs(cov - cov_ref)) < 1e-7 * np.max(np.abs(cov_ref))
        mean, L = mo, Lo
    lib.pnoc_use_lapack(None)


def test_c_bench_loop_runs_and_counts():
    lib = _lib()
    N, nu, B = 4, 3, 3
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.5, 1.0)])
    tc = np.ascontiguousarray(np.stack(tcoeffs(vf, [y0], nu), axis=1))
    acc, att = C.c_long(), C.c_long()
    el = lib.pnoc_bench_brusselator(N, nu, B, _p(tc), 1e-6, 0.01, 0.2, 2, C.byref(acc), C.byref(att))
    assert 0.15 < el < 5.0 and att.value >= acc.value > 10
