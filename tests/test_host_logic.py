"""CPU-side tests: host logic, API surface and error behaviour, C-ABI exports, multi-process gather."""

import ctypes
import os
import re

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import _backend, ivpsolve, ivpsolvers, sharding, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pnode.h")).read()
    declared = sorted(set(re.findall(r"\b(pnode_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(os.path.abspath(_backend.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/pnode.h but not exported"
    assert set(_backend.SYMBOLS) <= set(declared)
    assert _backend.load_library().pnode_version() >= 100


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must fail loudly (RuntimeError), never fall back."""
    lib = _backend.load_library()
    if lib.pnode_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        _backend.Batch(batch=1, d=3, nu=2, ode_order=1, model="isotropic", correction="ts0", strategy="filter",
                       calibration="dynamic", vf_id=V.VF_RIGID_BODY, vf_params=np.array([-2.0, 1.25, -0.5]))


def test_argument_validation_happens_before_the_device():
    lib = _backend.load_library()
    cfg = _backend.PnodeConfig(batch=1, d=3, nu=9, ode_order=1, model=0, correction=0, strategy=0, calibration=1,
                               vf_id=3, num_vf_params=3, vf_params_per_trajectory=0, num_checkpoints=0, device=0)
    h = ctypes.c_void_p()
    assert lib.pnode_create(ctypes.byref(cfg), ctypes.byref(h)) == -1  # PNODE_EINVAL
    assert b"nu" in lib.pnode_last_error()
    assert lib.pnode_create(None, ctypes.byref(h)) == -1


@pytest.mark.parametrize("name", ["brusselator", "vdp2", "vdp1", "rigid", "pleiades2", "pleiades1", "logistic"])
def test_taylor_coefficients_against_oracle_and_finite_differences(name):
    vf, u0s = {
        "brusselator": (V.brusselator(6), [brusselator_y0(6, 0.3)]),
        "vdp2": (V.van_der_pol_order2(5.0), [np.array([2.0]), np.array([0.1])]),
        "vdp1": (V.van_der_pol_order1(2.0), [np.array([2.0, 0.1])]),
        "rigid": (V.rigid_body(), [np.array([1.0, 0.0, 0.9])]),
        "pleiades2": (V.pleiades_order2(), [np.array([3.0, 3.0, -1.0, -3.0, 2.0, -2.0, 2.0, 3.0, -3.0, 2.0, 0.0, 0.0, -4.0, 4.0]),
                                            np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.75, -1.5, 0.0, 0.0, 0.0, -1.25, 1.0, 0.0, 0.0])]),
        "pleiades1": (V.pleiades_order1(), [np.concatenate([np.array([3.0, 3.0, -1.0, -3.0, 2.0, -2.0, 2.0, 3.0, -3.0, 2.0, 0.0, 0.0, -4.0, 4.0]),
                                                            np.array([0.0, 0.0, 0.0, 0.0, 0.0, 1.75, -1.5, 0.0, 0.0, 0.0, -1.25, 1.0, 0.0, 0.0])])]),
        "logistic": (V.logistic(1.0, 1.0), [np.array([0.1])]),
    }[name]
    k = vf.ode_order
    tc = taylor.odejet_padded_scan(vf, u0s, num=4)
    assert len(tc) == k + 4
    # first computed coefficient is f itself
    assert np.allclose(tc[k], vf(*u0s), rtol=1e-14, atol=0)
    # oracle's independent series implementation
    ref = po.taylor_coefficients(vf.series, u0s, 4)
    for a, b in zip(tc, ref):
        assert np.allclose(a, b, rtol=1e-13, atol=1e-13)
    # integrate the ODE a short time with scipy and compare with the Taylor polynomial
    from scipy.integrate import solve_ivp

    h = 1e-2

    def rhs(t, y):
        ys = np.split(y, k)
        return np.concatenate(ys[1:] + [np.asarray(vf(*ys))])

    sol = solve_ivp(rhs, (0, h), np.concatenate(u0s), rtol=1e-13, atol=1e-13, method="DOP853").y[:, -1]
    import math

    poly = sum(tc[j] * h**j / math.factorial(j) for j in range(len(tc)))
    err_full = np.max(np.abs(poly - sol[: vf.d]))
    err_short = np.max(np.abs(poly - tc[-1] * h ** (len(tc) - 1) / math.factorial(len(tc) - 1) - sol[: vf.d]))
    scale = max(1.0, np.max(np.abs(tc[-1])))
    assert err_full < 50.0 * scale * h ** len(tc)
    # the highest coefficient matters: dropping it must hurt (unless it is itself negligible)
    assert err_short > 3.0 * err_full or np.max(np.abs(tc[-1])) * h ** (len(tc) - 1) < 1e-12


def test_batched_taylor_matches_unbatched():
    vf = V.brusselator(5)
    y = np.stack([brusselator_y0(5, ph) for ph in (0.0, 0.5, 1.0)])
    batched = taylor.odejet_unroll(vf, [y], num=3)
    for b in range(3):
        single = taylor.odejet_unroll(vf, [y[b]], num=3)
        for cb, cs in zip(batched, single):
            assert np.array_equal(cb[b], cs)


def test_pi_controller_matches_oracle():
    ctrl = ivpsolve.control_proportional_integral()
    ref = po.PIControl()
    rng = np.random.default_rng(0)
    dt = rng.uniform(1e-3, 1e-1, 64)
    prev = rng.uniform(0.1, 1.0, 64)
    err = np.concatenate([rng.uniform(1e-6, 3.0, 60), [1.0, 0.0, 1e12, 1e-300]])
    ndt, nprev = ctrl.apply(dt, prev, err, 5)
    for i in range(64):
        if err[i] == 0.0:
            assert ndt[i] == 10.0 * dt[i]
            continue
        d_ref, p_ref = ref.apply(dt[i], prev[i], err[i], 5)
        assert np.isclose(ndt[i], d_ref, rtol=1e-15) and nprev[i] == p_ref
    # factor limits
    assert np.all(ndt <= 10.0 * dt + 1e-18) and np.all(ndt >= 0.2 * dt - 1e-18)


def test_solver_construction_and_errors():
    impl.select("isotropic", ode_shape=(4,))
    prior = ivpsolvers.prior_ibm(num_derivatives=3)
    with pytest.raises(ValueError):
        ivpsolvers.strategy_filter(prior, ivpsolvers.correction_ts1())  # ts1 needs dense
    with pytest.raises(ValueError):
        ivpsolvers.prior_ibm(num_derivatives=12)
    with pytest.raises(ValueError):
        impl.select("blockdiag", ode_shape=(4,))
    strat = ivpsolvers.strategy_fixedpoint(prior, ivpsolvers.correction_ts0(ode_order=1))
    solver = ivpsolvers.solver_dynamic(strat)
    assert solver.error_contraction_rate == 4 and solver.calibration == "dynamic"
    assert ivpsolvers.solver(strat).calibration == "none"
    init = solver.initial_condition([np.ones(4)] * 4, output_scale=2.0)
    assert init.tcoeffs.shape == (1, 4, 4) and not init.batched and init.output_scale[0] == 2.0
    init_b = solver.initial_condition([np.ones((3, 4))] * 4)
    assert init_b.tcoeffs.shape == (3, 4, 4) and init_b.batched
    with pytest.raises(ValueError):
        solver.initial_condition([np.ones(4)] * 3)
    # a later impl.select must not change an existing strategy
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        impl.select("dense", ode_shape=(4,))
    assert strat.model == "isotropic"
    with pytest.raises(TypeError):
        ivpsolve.solve_fixed_grid(lambda y, t: y, init, grid=np.linspace(0, 1, 3), solver=solver)
    # strategy_smoother on a fixed grid exists for the isotropic model (the reference's use, measure.py:129-140); a grid that is
    # not increasing is refused before a device is touched
    smoother = ivpsolvers.solver_dynamic(ivpsolvers.strategy_smoother(prior, ivpsolvers.correction_ts0()))
    with pytest.raises(ValueError):
        ivpsolve.solve_fixed_grid(V.linear(-1.0, d=4), smoother.initial_condition([np.ones(4)] * 4),
                                  grid=np.array([0.0, 0.5, 0.5]), solver=smoother)
    with pytest.raises(ValueError):
        ivpsolve.adaptive(solver, norm_ord=1)


def test_odecheckpts_solve_surface():
    from odecheckpts import ivps
    from odecheckpts import ivpsolvers as l3

    vf, u0, tspan, args = ivps.logistic()
    save_at = np.linspace(*tspan, num=5)
    with pytest.raises(ValueError):
        l3.solve("ts1-2", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3)
    with pytest.raises(ValueError):
        l3.solve("ts0-2", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3, calibrate="mle")
    fn = l3.solve("ts0-2", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3)
    with pytest.raises(ValueError):
        fn(u0[0], args)  # tuple expected
    with pytest.raises(ValueError):
        l3.solve_via_interpolate("ts1-2", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3)
    fn2 = l3.solve_via_interpolate("ts0-2", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3)
    with pytest.raises(ValueError):
        fn2(u0[0], args)  # tuple expected
    with pytest.raises(NotImplementedError):
        l3.solve_diffrax("tsit5", vf, u0[0], save_at, dt0=0.1, atol=1e-3, rtol=1e-3)
    for name in ("rigid_body", "pleiades_1st", "pleiades_2nd", "van_der_pol", "logistic"):
        out = getattr(ivps, name)()
        assert out[0].d == np.asarray(out[1][0]).shape[-1]
    vf_b, (y0,), (t0, t1), _ = ivps.brusselator(8)
    assert vf_b.d == 16 and y0.shape == (16,) and (t0, t1) == (0.0, 10.0)


def test_sharding_helpers():
    for total, world in ((1024, 8), (10, 3), (5, 8)):
        covered = []
        for rank in range(world):
            lo, hi = sharding.shard_bounds(total, world, rank)
            covered += list(range(lo, hi))
        assert covered == list(range(total))
        inter = np.sort(np.concatenate([sharding.interleaved_indices(total, world, r) for r in range(world)]))
        assert np.array_equal(inter, np.arange(total))
    assert sharding.gather_results(np.arange(3.0)) is not None  # no communicator: identity


def _sharded_worker(rank, world, port, out_dir):
    """One rank of the sharded path with the device replaced by its checker (no GPU here): Taylor initialisation of the
    local shard (product host code), the oracle's adaptive solve of every local trajectory, then the result gather."""
    import pn_oracle as po
    import torch.distributed as dist
    from helpers import oracle_problem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = 6
    lo, hi = sharding.shard_bounds(total, world, rank)
    u, steps = _solve_shard(lo, hi, total)
    comm = sharding.GlooGather()
    full_u = sharding.gather_results(u, comm)
    full_steps = sharding.gather_results(steps, comm)
    # the TCP hand-over of the ncclUniqueId that the RCCL gather uses (the 128 bytes themselves need a GPU to mean anything)
    blob = bytes(range(128)) if rank == 0 else None
    got = sharding._tcp_exchange(blob, rank, world, "127.0.0.1", port + 29)
    assert got == bytes(range(128))
    if rank == 0:
        np.save(os.path.join(out_dir, "u.npy"), full_u)
        np.save(os.path.join(out_dir, "steps.npy"), full_steps)
    dist.barrier()
    dist.destroy_process_group()


def _solve_shard(lo, hi, total):
    import pn_oracle as po
    from helpers import oracle_problem

    vf = V.brusselator(4)
    y0 = np.stack([brusselator_y0(4, 2 * np.pi * b / total) for b in range(lo, hi)])
    tcs = taylor.odejet_unroll(vf, [y0], num=3)
    cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy="filter", calibration="dynamic")
    save_at = np.linspace(0.0, 0.5, 4)
    us, steps = [], []
    for b in range(hi - lo):
        sol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-5, 1e-5), oracle_problem(vf),
                                        po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.01)
        us.append(sol.u)
        steps.append(np.asarray(sol.num_steps, dtype=np.int64))
    return np.stack(us), np.stack(steps)


def test_two_rank_sharded_solve_and_gather_equals_single_process(tmp_path):
    """The N>1 path on CPU (gloo, world_size 2): shard_bounds -> per-rank solve of the shard -> all_gather of the
    per-trajectory results == the unsharded run, bit for bit (no exchange during integration)."""
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    u_ref, steps_ref = _solve_shard(0, 6, 6)
    assert np.array_equal(np.load(tmp_path / "u.npy"), u_ref)
    assert np.array_equal(np.load(tmp_path / "steps.npy"), steps_ref)


def test_asolve_scipy_reproduces_the_reference_truth_bit_for_bit(golden):
    """odecheckpts.ivpsolvers.asolve_scipy (reference ivpsolvers.py:196-210) as run_simple.py:22-24 calls it: LSODA at 1e-13 on the
    rigid body. The reference's committed data_ts / data_ys (subsampled in the fixture) are reproduced exactly."""
    from odecheckpts import ivps, ivpsolvers

    vf, u0, tspan, params = ivps.rigid_body(time_span=(0.0, 50.0))
    ts, ys = ivpsolvers.asolve_scipy("LSODA", vf, tspan, atol=1e-13, rtol=1e-13)(tuple(u0[:1]), params)
    g = golden["rigid_body"]
    idx = np.searchsorted(ts, g["ts"])
    assert ts.shape == (1992,) and ys.shape == (1992, 3)
    assert np.array_equal(ts[idx], g["ts"]) and np.array_equal(ys[idx], g["ys"])
    with pytest.raises(ValueError, match="Tuple expected"):
        ivpsolvers.asolve_scipy("LSODA", vf, tspan, atol=1e-3, rtol=1e-3)(u0[0], params)

