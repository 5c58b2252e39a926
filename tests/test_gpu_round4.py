"""Round-4 GPU cases for the isotropic model in the lane layout (csrc/pnode_lanes.hpp): every device controller against the host
controller over the ODE dimensions that fill a wavefront differently, the orders the reference's experiment 3 uses (nu = 8), and
the shapes the one-wavefront smoother controller of round 3 could not serve."""

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import ivpsolve, ivpsolvers, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _solver(nu, strategy, ode_order=1, dynamic=True):
    strat = {"filter": ivpsolvers.strategy_filter, "fixedpoint": ivpsolvers.strategy_fixedpoint}[strategy]
    s = strat(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts0(ode_order=ode_order))
    return ivpsolvers.solver_dynamic(s) if dynamic else ivpsolvers.solver(s)


@pytest.mark.parametrize("N", [24, 31, 32, 33])
@pytest.mark.parametrize("strategy", ["filter", "fixedpoint"])
def test_device_controllers_equal_the_host_controller_when_the_wavefront_is_full(hip_lib, N, strategy):
    """d = 48, 62, 64 (one ODE dimension per lane, the state of a trajectory in registers over a whole launch) and d = 66 (loops
    over the dimensions), tolerances down to 1e-7: identical accept / reject sequences with every controller. The copies of the
    n x n factor in the four 16-lane rows of a wavefront must stay identical over thousands of attempts; a sum that differed in
    the last bit between the halves of a wavefront (a fused multiply-add in the first stage of the butterfly) once made them drift
    apart, visible only from d = 49 on and only at tight tolerances."""
    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(vf.d,))
    solver = _solver(3, strategy)
    tol = np.array([1e-4, 1e-6, 1e-7])
    y0 = np.stack([brusselator_y0(N, 0.8)] * 3)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [y0], num=3), 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol)
    save_at = np.linspace(0.0, 0.1, 3)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
    host = run(False)
    assert np.all(np.isfinite(host.u)) and host.num_steps[:, -1].min() > 20
    for dc in (True, "pipelined") if strategy == "fixedpoint" and vf.d <= 64 else (True,):
        dev = run(dc)
        assert np.array_equal(dev.num_steps, host.num_steps) and np.array_equal(dev.num_attempts, host.num_attempts), dc
        assert relmax(dev.u, host.u) < 1e-10, dc


def _oracle_fp(vf, tcs, b, nu, atol, rtol, save_at, dt0):
    cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(atol, rtol), oracle_problem(vf), po.initial_state(cfg, [c[b] for c in tcs], 0.0),
                                     save_at, dt0)
    return osol, po.markov_marginals_reverse(cfg, osol)[:, 0]


@pytest.mark.parametrize("nu", [5, 8])
def test_pleiades_high_order_fixedpoint_smoother_vs_oracle(hip_lib, nu):
    """The orders of the reference's experiment 3 (experiments/3_workprec_harder/run_harder.py:77: alg_ts0(3), (5), (8)) on its
    problem: Pleiades in second-order form (d = 14, ode_order = 2), isotropic EKF0 fixed-point smoother, tolerances as the script
    forms them (rtol = 10 tol, atol = 1e-3 rtol). nu = 8 is n = 9, the largest instantiation of every isotropic kernel: host
    controller, one-wavefront and pipelined device controllers against the oracle (identical accept / reject sequences, smoothed
    means to 1e-8 relative), and the two device controllers -- different kernels for every nu again -- against each other."""
    from odecheckpts import ivps

    vf, u0, _ = ivps.pleiades_2nd()
    impl.select("isotropic", ode_shape=(14,))
    solver = _solver(nu, "fixedpoint", ode_order=2)
    tol = np.array([1e-3, 1e-5])
    rt = 10.0 * tol
    inits = [np.tile(u0[0], (2, 1)), np.tile(u0[1], (2, 1))]
    tcs = taylor.odejet_padded_scan(vf, inits, num=nu - 1)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, 0.6, 5)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * rt, rtol=rt)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver, device_controller=dc)
    host, one, pipe = run(False), run("one-wavefront"), run("pipelined")
    for b in range(2):
        osol, sm = _oracle_fp(vf, tcs, b, nu, 1e-3 * rt[b], rt[b], save_at, 0.1)
        assert osol.num_steps[-1] >= 4
        for name, sol in (("host", host), ("one-wavefront", one), ("pipelined", pipe)):
            assert np.array_equal(sol.num_steps[b], osol.num_steps), (name, b, sol.num_steps[b], osol.num_steps)
            assert relmax(sol.u[b], sm) < 1e-8, (name, b)
    assert np.array_equal(one.num_attempts, host.num_attempts) and np.array_equal(pipe.num_attempts, host.num_attempts)
    assert relmax(pipe.u, one.u) < 1e-11


def test_config3_pleiades_sweep_full_batch(hip_lib):
    """BASELINE config 3, second problem, at its batch size: Pleiades (experiments/3), isotropic EKF0 fixed-point smoother, 1024
    tolerances in one batch, nu = 3 with the host controller and nu = 8 with the device-resident one. Sampled members against the
    oracle, the sweep against a tight scipy solution, and composition invariance (members solved alone give the bits of the batch)."""
    from odecheckpts import ivps
    from scipy.integrate import solve_ivp

    vf, u0, _ = ivps.pleiades_2nd()
    vf1, u01, _, _ = ivps.pleiades_1st()
    B = 1024
    rt = 10.0 ** np.linspace(-1.0, -6.0, B)
    t1 = 0.75
    truth = solve_ivp(lambda t, y: vf1(y), (0.0, t1), u01[0], rtol=1e-13, atol=1e-13, method="DOP853").y[:14, -1]
    save_at = np.linspace(0.0, t1, 4)
    impl.select("isotropic", ode_shape=(14,))
    for nu, dc in ((3, False), (8, True)):
        solver = _solver(nu, "fixedpoint", ode_order=2)
        tcs = taylor.odejet_padded_scan(vf, [np.tile(u0[0], (B, 1)), np.tile(u0[1], (B, 1))], num=nu - 1)
        asolve = lambda idx: ivpsolve.solve_adaptive_save_at(  # noqa: E731
            vf, solver.initial_condition([c[idx] for c in tcs], 1.0), save_at=save_at, dt0=0.1,
            adaptive_solver=ivpsolve.adaptive(solver, atol=1e-3 * rt[idx], rtol=rt[idx]), device_controller=dc)
        sol = asolve(np.arange(B))
        assert np.all(np.isfinite(sol.u))
        err = np.sqrt(np.mean((sol.u[:, -1] - truth) ** 2, axis=1))
        assert np.median(err[-64:]) < 1e-2 * np.median(err[:64]) and np.median(err[-64:]) < 1e-5, (nu, np.median(err[:64]), np.median(err[-64:]))
        steps = sol.num_steps[:, -1]
        assert np.median(steps[-64:]) > 1.2 * np.median(steps[:64])
        for b in (0, 400, 800):
            osol, sm = _oracle_fp(vf, tcs, b, nu, 1e-3 * rt[b], rt[b], save_at, 0.1)
            assert np.array_equal(sol.num_steps[b], osol.num_steps), (nu, b, sol.num_steps[b], osol.num_steps)
            assert relmax(sol.u[b], sm) < 1e-8, (nu, b)
        idx = np.array([5, 512, 1000])
        sub = asolve(idx)
        assert np.array_equal(sub.num_steps, sol.num_steps[idx]) and np.array_equal(sub.u, sol.u[idx])


def test_experiment_3_result_files_have_the_reference_schema(hip_lib, golden, tmp_path):
    """SURVEY 8(f)-4, experiment 3: experiments/workprec_harder.py writes what the reference's run_harder.py writes
    (run_harder.py:111-115, 176-183) and its precision curves land on the reference's committed ones (`data_results.npy`,
    "Prob(3) / Prob(5) / Prob(8) via probdiffeq"; grade B in SURVEY 8(c): same algorithm and tolerances, different arithmetic --
    within half an order of magnitude per tolerance, the first four tolerances here)."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("workprec_harder", os.path.join(root, "experiments", "workprec_harder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(["--start", "3", "--stop", "7", "--repeats", "2", "--outdir", str(tmp_path)])
    res = np.load(tmp_path / "data_results.npy", allow_pickle=True).item()
    g = golden["pleiades"]
    for nu in (3, 5, 8):
        label = f"Prob({nu}) via probdiffeq"
        rec = res[label]
        assert set(rec.keys()) == {"list_of_args", "length_of_longest_vector", "work_min", "work_mean", "work_std", "precision"}
        assert all(len(rec[key]) == 4 for key in rec)
        assert np.allclose(rec["list_of_args"], g[f"{label}__list_of_args"][:4])
        assert np.all(rec["length_of_longest_vector"] == 50)
        ratio = rec["precision"] / g[f"{label}__precision"][:4]
        assert np.all((ratio > 10 ** -0.5) & (ratio < 10 ** 0.5)), (label, rec["precision"], g[f"{label}__precision"][:4])
        batch = res[f"Prob({nu}) (one batch) via probdiffeq_hip"]
        assert np.allclose(batch["precision"], rec["precision"], rtol=1e-6)  # the sweep as one batch: the same solves
    assert np.allclose(np.load(tmp_path / "data_checkpoints.npy"), g["checkpoints"])
    ts, ys = np.load(tmp_path / "data_ts.npy"), np.load(tmp_path / "data_ys.npy")
    assert ys.shape == (len(ts), 28) and ts[0] == 0.0 and ts[-1] == 3.0


@pytest.mark.parametrize("strategy", ["filter", "fixedpoint"])
@pytest.mark.parametrize("N", [8, 40])
def test_checkpoints_inside_the_controller_kernel(hip_lib, N, strategy):
    """Isotropic model, device-resident controller, nu <= 5: the launch walks over the save_at points itself (interpolation and right
    corner inside the controller kernel, csrc/pnode_iso.hip: iso_walk_checkpoints) -- the host's loop of
    /root/reference/src/odecheckpts/ivpsolvers.py:71-77 moved onto the device. 60 save_at points at loose tolerances, so that one
    accepted step passes several of them; trajectories of different speed in one batch; an iteration budget of 7 attempts per launch,
    so that the solve is spread over many launches that resume mid-interval: accepted-step counts AT EVERY CHECKPOINT, attempts and
    (smoothed) means as with the host controller. d = 16 runs the register form, d = 80 the loops over the ODE dimensions."""
    from probdiffeq_hip import _backend

    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(vf.d,))
    solver = _solver(3, strategy)
    tol = np.array([1e-2, 1e-4, 1e-6])
    y0 = np.stack([brusselator_y0(N, 0.3 * b) for b in range(3)])
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [y0], num=3), 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol)
    save_at = np.linspace(0.0, 1.5, 60)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
    host, dev = run(False), run(True)
    if N == 8:
        assert np.any(np.diff(host.num_steps[0]) == 0)  # several checkpoints of the loosest member fall into one step (N = 40: stiffer, 20 steps each)
    assert np.array_equal(dev.num_steps, host.num_steps) and np.array_equal(dev.num_attempts, host.num_attempts)
    assert relmax(dev.u, host.u) < 1e-10
    # the same through the C-ABI with a budget of 7 attempts per launch
    batch = ivpsolve._make_batch(vf, init, solver, len(save_at) - 1)
    try:
        batch.set_initial(init.tcoeffs, np.zeros(3), init.output_scale)
        batch.set_tolerances(1e-3 * tol, tol)
        steps, attempts, failed = batch.solve_adaptive_save_at(save_at, np.full(3, 0.01), asolver.control, asolver.numerical_zero,
                                                               max_attempts=7, device_controller=True)
        assert not failed.any()
        assert np.array_equal(steps, host.num_steps) and np.array_equal(attempts, host.num_attempts)
    finally:
        batch.close()


@pytest.mark.parametrize("nu", [3, 5])
def test_stored_backward_transitions_vs_oracle(hip_lib, nu):
    """pnode_read_checkpoint_transitions: the transitions x(c_{k-1}) | x(c_k) ~ N(G x + xi, Lam^T Lam) the fixed-point smoother
    stores -- probdiffeq's `posterior.conditional`, /root/reference/src/odecheckpts/ivpsolvers.py:80 -- entry by entry against the
    oracle's (restricted three-body problem in second-order form, isotropic EKF0)."""
    from odecheckpts import ivps

    vf, (u0, du0), (t0, _) = ivps.three_body_restricted()
    impl.select("isotropic", ode_shape=(vf.d,))
    solver = _solver(nu, "fixedpoint", ode_order=2)
    tcs = taylor.odejet_padded_scan(vf, [u0[None], du0[None]], num=nu - 1)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(t0, t0 + 1.5, 7)
    tol = 1e-5
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol))
    osol, sm = _oracle_fp(vf, tcs, 0, nu, 1e-3 * tol, tol, save_at, 0.01)
    assert np.array_equal(sol.num_steps[0], osol.num_steps) and osol.num_steps[-1] > 30
    assert relmax(sol.u[0], sm) < 1e-9
    G, xi, Lam = sol.posterior.batch.read_checkpoint_transitions()
    K = len(save_at) - 1  # store entry c: x(save_at[c]) | x(save_at[c + 1])
    assert G.shape == (1, K, nu + 1, nu + 1) and xi.shape == (1, K, nu + 1, vf.d)
    # the first transition leaves a state without uncertainty: G = 0, xi = the initial mean
    assert np.all(G[0, 0] == 0.0) and np.allclose(xi[0, 0], np.stack([c[0] for c in tcs]), rtol=1e-14, atol=0)
    # (entries of G span 20 orders of magnitude in these coordinates and the gain is a solve with a factor whose condition number grows
    # with nu: two correct implementations differ by up to 1e-5 of the largest entry at nu = 5, 1e-9 at nu = 3 -- the smoothed means above,
    # which is what the transitions are for, agree to 1e-9 at both orders; this loop is about the layout of what is read back)
    eps = {3: 1e-7, 5: 1e-4}[nu]
    for c in range(K):
        oc = osol.states[c + 1].cond
        assert relmax(G[0, c], oc.G) < eps, c
        assert relmax(xi[0, c], oc.xi) < eps, c
        assert np.all(np.diag(Lam[0, c]) >= 0.0) and np.all(np.tril(Lam[0, c], -1) == 0.0)
        assert relmax(Lam[0, c].T @ Lam[0, c], oc.Lam @ oc.Lam.T) < 10 * eps, c


def test_large_path_gemm_tiles_give_identical_bits(hip_lib):
    """pnode_set_generic_path(h, 2 | 4): the large-D kernels with the 128 x 128 double-buffered GEMM tile where the shape allows
    (long inner dimension, at least 256 tiles in the launch) against the 64 x 64 tile everywhere -- the same sums in the same order,
    element by element: error estimates, means and factors are equal bit for bit after several attempts (D = 512, 128
    trajectories: the products with K = 512 take the large tile). Same kernels as BASELINE config 5's, at a size the suite can afford."""
    from probdiffeq_hip import _backend

    N, nu, B = 64, 3, 128
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, 0.05 * b) for b in range(B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    out = []
    for path in (2, 4):
        bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy="filter", calibration="dynamic",
                            vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
        try:
            bt.set_generic_path(path)
            bt.set_initial(tc, np.zeros(B), np.ones(B))
            bt.set_tolerances(1e-6, 1e-6)
            err, one = np.zeros(B), np.ones(B, np.uint8)
            errs = []
            for dt in (0.004, 0.006, 0.005):
                bt.attempt_step(np.full(B, dt) * (1.0 + 0.001 * np.arange(B)), one, err)
                bt.commit(one)
                errs.append(err.copy())
            out.append((np.stack(errs), bt.read_mean(0), bt.read_factor(0)))
        finally:
            bt.close()
    (e2, m2, f2), (e4, m4, f4) = out
    assert np.all(np.isfinite(e2)) and np.all(np.isfinite(f2)) and np.max(np.abs(f2)) > 0
    assert np.array_equal(e2, e4) and np.array_equal(m2, m4) and np.array_equal(f2, f4)


@pytest.mark.parametrize("strategy", ["filter", "fixedpoint"])
@pytest.mark.parametrize("N", [65, 70, 96, 100])
def test_four_wavefront_kernels_when_the_dimensions_do_not_fill_the_team(hip_lib, N, strategy):
    """d = 130, 140, 192, 200 ODE dimensions on the kernels that give a trajectory four wavefronts (d > 128): the last wavefronts have few
    or no dimensions of their own. Lanes beyond d repeat dimension d - 1 -- harmless inside one wavefront, a race across wavefronts (a whole
    wavefront of repeats redid another wavefront's update of m[d - 1] and of the smoother's offset at its own time: NaNs with the host
    controller at d = 140, found by scripts/soak_iso_shapes.py; such a wavefront now skips the loop bodies). Host and device controller
    against the oracle: identical decisions, smoothed means to 1e-8."""
    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(vf.d,))
    nu, tol = 2, 1e-3
    solver = _solver(nu, strategy)
    y0 = np.stack([brusselator_y0(N, 0.3), brusselator_y0(N, 0.7)])
    tcs = taylor.odejet_padded_scan(vf, [y0], num=nu)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, 0.12, 4)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol)
    cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy=strategy, calibration="dynamic")
    refs = []
    for b in range(2):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-3 * tol, tol), oracle_problem(vf), po.initial_state(cfg, [c[b] for c in tcs], 0.0),
                                         save_at, 0.01)
        refs.append((osol.num_steps, po.markov_marginals_reverse(cfg, osol)[:, 0] if strategy == "fixedpoint" else osol.u))
    for dc in (False, True):
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
        for b in range(2):
            assert np.array_equal(sol.num_steps[b], refs[b][0]) and refs[b][0][-1] > 20, (dc, b, sol.num_steps[b], refs[b][0])
            assert relmax(sol.u[b], refs[b][1]) < 1e-8, (dc, b)


def test_shape_soak_of_the_isotropic_kernels(hip_lib):
    """scripts/soak_iso_shapes.py at a fixed seed, 36 configurations (random d = 2 .. 140, nu = 1 .. 8, filter / fixed-point smoother, three tolerances): every
    device-resident controller takes the host controller's decisions exactly and ends finite. (Outputs apart by more than 1e-9 with identical decisions are
    counted separately -- nu >= 6 and the stiff large-N Brusselator amplify rounding differences between two kernels, profiles/r04_iso_shape_soak.txt -- and
    must stay rare.)"""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("soak_iso_shapes", os.path.join(root, "scripts", "soak_iso_shapes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hard, soft = mod.main(seed=3, count=36)
    assert hard == 0 and soft <= 3, (hard, soft)
