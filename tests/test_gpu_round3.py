"""Round-3 GPU cases: the rank-(D - d) state factor of the MFMA path, path switches on a reused handle, the hand-shake
time-out of the one-launch LQ as a device error, and the longer randomised comparison of the two LQ kernels (it was a script)."""

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import _backend, ivpsolve, ivpsolvers, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _batch(vf, nu, B, path=0):
    bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy="filter",
                        calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
    if path:
        bt.set_generic_path(path)
    return bt


@pytest.mark.parametrize("N,nu", [(16, 3), (16, 5), (32, 3)])
def test_mfma_state_factor_has_rank_D_minus_d(hip_lib, N, nu):
    """The measurement update of the MFMA path rotates the first 2d columns of L^- by the reflectors of F = H L^- and keeps the
    D - d columns that carry covariance (the observation is noise-free: the other d are known exactly). What the read-back hands
    out is R = [Ls | 0]^T: d zero rows, R^T R equal to the oracle's covariance, H Sigma H^T = 0 to rounding."""
    vf = V.brusselator(N)
    d, D = vf.d, (nu + 1) * vf.d
    B = 3
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.4, 1.3)])
    tc = tcoeffs(vf, [y0], nu)
    bt = _batch(vf, nu, B)
    bt.set_initial(np.stack(tc, axis=1), np.zeros(B), np.ones(B))
    bt.set_tolerances(1e-5, 1e-5)
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    prob = oracle_problem(vf)
    sts = [po.initial_state(cfg, [c[b] for c in tc], 0.0) for b in range(B)]
    err = np.zeros(B)
    for dt in (0.01, 0.02, 0.015):  # (the steps of test_dense_mfma_path_matches_oracle_and_generic)
        bt.attempt_step(np.full(B, dt), np.ones(B, np.uint8), err)
        bt.commit(np.ones(B, np.uint8))
        sts = [po.step(cfg, prob, st, dt)[1] for st in sts]
        R = bt.read_factor(0)
        for b in range(B):
            assert np.all(R[b][D - d:] == 0.0)  # the columns the factor does not have
            cov = R[b].T @ R[b]
            ref = sts[b].chol @ sts[b].chol.T
            assert relmax(cov, ref) < 1e-6  # (sigma_hat carries 1e-7 relative: cancellation in z)
            # observed combination E1 - J E0 of the state: zero posterior variance
            assert np.linalg.matrix_rank(R[b], tol=1e-9 * np.abs(R[b]).max()) <= D - d
    bt.close()


def test_reused_handle_may_switch_path_after_set_initial(hip_lib):
    """A handle that has stepped on the generic kernels (row-major factor in all three slots), is re-initialised and switched to
    the MFMA path (strip order, D - d columns) must give the results of a fresh handle: pnode_set_initial clears every slot."""
    N, nu, B = 16, 3, 4
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in np.linspace(0.0, 1.5, B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    dts = (0.004, 0.006, 0.009, 0.012)

    def run(bt):
        err = np.zeros(B)
        errs = []
        for dt in dts:
            bt.attempt_step(np.full(B, dt), np.ones(B, np.uint8), err)
            bt.commit(np.ones(B, np.uint8))
            errs.append(err.copy())
        R = bt.read_factor(0)
        return np.array(errs), bt.read_mean(0), np.einsum("bji,bjk->bik", R, R)

    reused = _batch(vf, nu, B, path=1)
    reused.set_initial(tc, np.zeros(B), np.ones(B))
    reused.set_tolerances(1e-5, 1e-5)
    run(reused)  # every slot now holds a row-major factor
    reused.set_initial(tc, np.zeros(B), np.ones(B))
    reused.set_generic_path(0)
    got = run(reused)
    reused.close()
    fresh = _batch(vf, nu, B)
    fresh.set_initial(tc, np.zeros(B), np.ones(B))
    fresh.set_tolerances(1e-5, 1e-5)
    want = run(fresh)
    fresh.close()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_lq_handshake_timeout_is_a_device_error_not_a_result(hip_lib):
    """Every wait of the one-launch LQ is a bounded spin. With the bound set to one poll round the first real wait fails: the
    step must come back as an error that names the hand-shake (PNODE_EDEVICE -> RuntimeError), not as NaN error norms that the
    controller would read as a failed ODE. After re-initialisation the handle works again, bit for bit like a fresh one."""
    N, nu, B = 32, 3, 6
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in np.linspace(0.0, 1.0, B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    bt = _batch(vf, nu, B)
    bt.set_initial(tc, np.zeros(B), np.ones(B))
    bt.set_tolerances(1e-5, 1e-5)
    bt.debug_set_lq_spin_limit(1)
    err = np.zeros(B)
    with pytest.raises(RuntimeError, match="hand-shake timed out"):
        bt.attempt_step(np.full(B, 0.005), np.ones(B, np.uint8), err)
    with pytest.raises(RuntimeError):  # the state is undefined until set_initial
        bt.attempt_step(np.full(B, 0.005), np.ones(B, np.uint8), err)
    bt.debug_set_lq_spin_limit(0)
    bt.set_initial(tc, np.zeros(B), np.ones(B))
    bt.attempt_step(np.full(B, 0.005), np.ones(B, np.uint8), err)
    fresh = _batch(vf, nu, B)
    fresh.set_initial(tc, np.zeros(B), np.ones(B))
    fresh.set_tolerances(1e-5, 1e-5)
    err2 = np.zeros(B)
    fresh.attempt_step(np.full(B, 0.005), np.ones(B, np.uint8), err2)
    assert np.all(np.isfinite(err)) and np.array_equal(err, err2)
    bt.close()
    fresh.close()


def test_one_launch_lq_soak_against_two_launch_lq(hip_lib):
    """The headline shape at the headline batch: D = 256, 1024 trajectories (four per persistent workgroup), 20 attempts with
    random step sizes, activity masks and accept / reject decisions; the one-launch LQ against the two-launch LQ. Guards the
    flag protocol of the persistent workgroups (trajectories skipped inside a workgroup's sequence, slot reuse across
    trajectories) under the load it runs at."""
    N, nu, B, ATT = 32, 3, 1024, 20
    rng = np.random.default_rng(1000 * N + nu)
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in rng.uniform(0.0, 2.0, B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    one, two = _batch(vf, nu, B), _batch(vf, nu, B, path=3)
    for bt in (one, two):
        bt.set_initial(tc, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
    e1, e2 = np.zeros(B), np.zeros(B)
    worst_e = worst_m = 0.0
    for it in range(ATT):
        dts = 0.004 * rng.uniform(0.5, 2.0, B)
        act = (rng.uniform(size=B) < 0.8).astype(np.uint8)
        acc = (act & (rng.uniform(size=B) < 0.85)).astype(np.uint8)
        one.attempt_step(dts, act, e1)
        two.attempt_step(dts, act, e2)
        live = act.astype(bool)
        assert np.all(np.isfinite(e1[live])), it
        worst_e = max(worst_e, float(np.max(np.abs(e1[live] - e2[live]) / e2[live])))
        one.commit(acc)
        two.commit(acc)
        m1, m2 = one.read_mean(0), two.read_mean(0)
        worst_m = max(worst_m, float(np.max(np.abs(m1[:, 0] - m2[:, 0])) / np.max(np.abs(m2[:, 0]))))
    one.close()
    two.close()
    assert worst_e < 1e-6 and worst_m < 1e-9, (worst_e, worst_m)


@pytest.mark.parametrize("N", [16, 32])
def test_nan_member_on_the_mfma_path_stays_nan_and_leaves_its_neighbours_alone(hip_lib, N):
    """A trajectory whose state is NaN must come back with a NaN error norm (the controller then stops it, like the reference,
    SURVEY section 5: NaN is an expected result, not trapped) from the register-resident kernels too -- the sweep has no
    reflection flag and guards its reciprocal square root with 1e-200, neither may swallow a NaN -- and the other members of the
    batch must get the bits they get without it."""
    nu, B = 3, 3
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.4, 1.3)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    bad = tc.copy()
    bad[1, 0, 3] = np.nan
    out = []
    for init in (tc, bad):
        bt = _batch(vf, nu, B)
        bt.set_initial(init, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
        err = np.zeros(B)
        errs = []
        for dt in (0.01, 0.02, 0.015):
            bt.attempt_step(np.full(B, dt), np.ones(B, np.uint8), err)
            bt.commit(np.ones(B, np.uint8))
            errs.append(err.copy())
        out.append((np.array(errs), bt.read_mean(0)))
        bt.close()
    (e_ok, m_ok), (e_bad, m_bad) = out
    assert np.all(np.isnan(e_bad[:, 1])) and np.all(np.isnan(m_bad[1, 0]))
    for b in (0, 2):
        assert np.array_equal(e_ok[:, b], e_bad[:, b]) and np.array_equal(m_ok[b], m_bad[b])


def _three_body(nu, strategy="filter", model="isotropic", corr="ts0"):
    from odecheckpts import ivps
    from probdiffeq_hip import ivpsolve, ivpsolvers, taylor
    from probdiffeq_hip.impl import impl

    vf, (u0, du0), (t0, t1) = ivps.three_body_restricted()
    impl.select(model, ode_shape=(2,))
    correction = (ivpsolvers.correction_ts0 if corr == "ts0" else ivpsolvers.correction_ts1)(ode_order=2)
    strat = getattr(ivpsolvers, f"strategy_{strategy}")(ivpsolvers.prior_ibm(num_derivatives=nu), correction)
    solver = ivpsolvers.solver(strat)  # uncalibrated, as experiment 5 builds it (measure.py:47, 100, 215)
    tcs = taylor.odejet_padded_scan(vf, [u0, du0], num=nu + 1 - 2)
    return vf, solver, solver.initial_condition(tcs, 1.0), t0, t1, ivpsolve


def test_three_body_baseline_solution_of_experiment_5_through_api(hip_lib, golden):
    """The reference's experiment-5 baseline (measure.py:155-163, 211-232; committed as data_solution.npy: the filter's mean at every
    accepted step of the Arenstorf orbit) through the public API on the GPU: the same 7 540 accepted steps, every value within 1e-9."""
    ref = golden["three_body"]["baseline_solution"]
    vf, solver, init, t0, t1, ivpsolve = _three_body(3)
    asolver = ivpsolve.adaptive(solver, atol=1e-7, rtol=1e-7, control=ivpsolve.control_proportional_integral())
    sol = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=t0, t1=t1, dt0=0.01, adaptive_solver=asolver)
    assert sol.u.shape == ref.shape == (7541, 2)
    assert np.max(np.abs(sol.u - ref)) < 1e-9


@pytest.mark.parametrize("k,strategy", [(0, "smoother"), (1, "smoother"), (2, "filter")])
def test_three_body_step_counts_of_experiment_5_through_api(hip_lib, golden, k, strategy):
    """'No. steps' of experiment 5 (measure.py:177-199: nu = 4, the AS runner = strategy_smoother + solve_adaptive_save_every_step):
    448 / 2 570 / 14 469 accepted steps at tol 1e-4 / 1e-7 / 1e-10, bit-exact (the tightest through the filter: the decisions are
    the smoother's, without 14 469 stored transitions)."""
    g = golden["three_body"]
    tol, want = float(g["smoother_tols"][k]), int(g["smoother_num_steps"][k])
    vf, solver, init, t0, t1, ivpsolve = _three_body(4, strategy)
    asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol, control=ivpsolve.control_proportional_integral())
    sol = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=t0, t1=t1, dt0=0.01, adaptive_solver=asolver)
    assert len(sol.t) - 1 == want and sol.t[-1] == t1


def test_three_body_analytic_jacobian_on_the_device(hip_lib):
    """correction_ts1 on the restricted three-body problem (dense model, D = 8: the fused one-wavefront kernel and the generic
    kernels): single steps against the oracle, whose Jacobian is a complex-step derivative of the NumPy vector field."""
    from odecheckpts import ivps

    vf, (u0, du0), _ = ivps.three_body_restricted()
    nu, B = 3, 1
    tc = tcoeffs(vf, [u0[None], du0[None]], nu)
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    prob = oracle_problem(vf)
    for generic in (False, True):
        bt = _backend.Batch(batch=B, d=2, nu=nu, ode_order=2, model="dense", correction="ts1", strategy="filter", calibration="dynamic",
                            vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
        if generic:
            bt.set_generic_path(True)
        bt.set_initial(np.stack(tc, axis=1), np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-6, 1e-6)
        st = po.initial_state(cfg, [c[0] for c in tc], 0.0)
        err = np.zeros(B)
        for dt in (1e-3, 2e-3, 2e-3):
            bt.attempt_step(np.full(B, dt), np.ones(B, np.uint8), err)
            bt.commit(np.ones(B, np.uint8))
            e, st = po.step(cfg, prob, st, dt)
            en = po.normalise_error(po.AdaptiveConfig(1e-6, 1e-6), e, st.mean[0], st.mean[0])
            assert abs(err[0] - en) <= 1e-7 * en
            assert relmax(bt.read_mean(0)[0], np.stack(st.mean)) < 1e-9
        bt.close()


def test_experiment_5_script_reproduces_the_reference_step_counts_and_schema(hip_lib, golden):
    """experiments/vs_interpolation.py = the reference's experiment 5 (measure.py:153-209) through the public API: fixed-point
    solver + save_at + markov_sample against smoother + fixed grid (the union of its adaptive grid and save_at) + markov_sample.
    The result dict has the reference's keys, 'No. steps' equals the reference's committed 448 / 2,570 / 14,469, and the two routes
    to the sample mean agree at the save_at points."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vs_interpolation", os.path.join(root, "experiments", "vs_interpolation.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    results, baseline = mod.main(["--samples", "5", "--nosave"])
    g = golden["three_body"]
    assert np.max(np.abs(baseline.u - g["baseline_solution"])) < 1e-9
    assert [results[i]["No. steps"] for i in (1, 2, 3)] == [f"{int(n):,}" for n in g["smoother_num_steps"]]
    assert set(results[1]) == {"No. Samples", "Tolerance", "No. steps", "Time (s): AS", "Time (s): ATS (ours)"}
    # the two routes sample the same posterior family: their sample means at the save_at points agree to the solvers' accuracy
    from odecheckpts import ivps

    vf, inits, tspan = ivps.three_body_restricted()
    save_at = np.linspace(tspan[0], tspan[1])
    ats, _ = mod.checkpoint_runner(vf, inits, tspan, 200)(1e-7, save_at)
    grid_of, text = mod.textbook_runner(vf, inits, tspan, 200)
    as_, _ = text(np.union1d(grid_of(1e-7, save_at), save_at), save_at)
    assert ats.shape == as_.shape == (50, 2)
    assert np.max(np.abs(ats - as_)) < 1e-3


@pytest.mark.parametrize("N", [8, 32])
def test_device_resident_controller_of_the_dense_fixedpoint_smoother(hip_lib, N):
    """Round-2 verdict, item 5 (second half): the device-resident controller of the multi-kernel dense paths also drives
    strategy_fixedpoint (ivpsolvers.py:43, run.py:59) -- the backward-transition kernels that follow an attempt take their
    arguments from the StepArg the controller kernel wrote, the checkpoint hand-over (interpolation, reset of the running
    transition) stays with the host between the launches. N = 8: generic kernels (D = 32 is stepped by the fused small kernel
    only for the filter), N = 32: D = 256, the extra-row LQ on the large-D kernels. Identical decisions, smoothed means within
    1e-10 of the host controller's, and pnode_advance_device takes a fixed-point handle."""
    vf = V.brusselator(N)
    B = 4
    y0 = np.stack([brusselator_y0(N, 0.7 * b) for b in range(B)])
    impl.select("dense", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts1()))
    tol = 10.0 ** -np.linspace(4, 5.5, B)
    asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [y0], num=3), 1.0)
    save_at = np.linspace(0.0, 0.3, 4)
    host = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver)
    dev = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=True)
    assert np.array_equal(dev.num_steps, host.num_steps)
    assert np.array_equal(dev.num_attempts, host.num_attempts)
    assert host.num_steps[:, -1].min() >= 4 and len(set(host.num_steps[:, -1])) > 1
    assert np.all(np.isfinite(host.u))
    assert relmax(dev.u, host.u) < 1e-10
    batch = ivpsolve._make_batch(vf, init, solver, 1)
    batch.set_initial(init.tcoeffs, np.zeros(B), init.output_scale)
    batch.set_tolerances(tol, tol)
    dt, lp = np.full(B, 0.01), np.zeros(B)
    status, attempts = batch.advance_device(np.full(B, np.inf), dt, lp, asolver.control, asolver.numerical_zero, 3)
    assert np.all(status == 0) and np.all(attempts == 3)
    batch.close()


def test_two_device_controllers_of_the_isotropic_fixedpoint_smoother_agree(hip_lib):
    """The device-resident controller of the isotropic fixed-point smoother (ivpsolvers.py:43) exists as two kernels: one wavefront
    per trajectory (iso_advance_kernel) and one workgroup per trajectory with the backward transitions of the accepted steps on
    three wavefronts beside the filter (iso_advance_fp_kernel; on request). Same expressions
    in the same order: identical decisions bit for bit -- also the host controller's -- and smoothed means equal to rounding (the
    multiply-adds of the transitions are contracted differently in the two kernels) (Brusselator N = 8, nu = 4 as in experiment 4, a tolerance sweep so that the trajectories differ)."""
    N, B = 8, 6
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, 0.5 * b) for b in range(B)])
    impl.select("isotropic", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0()))
    tol = 10.0 ** -np.linspace(3, 7, B)
    asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [y0], num=4), 1.0)
    save_at = np.linspace(0.0, 2.0, 9)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
    host, pipe, one, auto = run(False), run("pipelined"), run("one-wavefront"), run(True)
    assert host.num_steps[:, -1].min() >= 10 and len(set(host.num_steps[:, -1])) > 1
    for dev in (pipe, one, auto):
        assert np.array_equal(dev.num_steps, host.num_steps)
        assert np.array_equal(dev.num_attempts, host.num_attempts)
        assert relmax(dev.u, host.u) < 1e-11
    assert relmax(pipe.u, one.u) < 1e-12 and np.array_equal(auto.u, one.u)  # (True = one wavefront per trajectory)
    with pytest.raises(ValueError):
        run("something else")


def test_pipelined_smoother_controller_with_a_diverging_member(hip_lib):
    """Failure path of the pipelined device controller (iso_advance_fp_kernel): a member whose ODE blows up in finite time
    (u' = u (1 + 5 u), blow-up at t = log(6/5)) beside two healthy ones on ragged save_at grids. The three controllers -- host,
    one wavefront per trajectory, pipelined -- report the same step counts, flag the same member, and leave the neighbours'
    smoothed means identical (the reference propagates NaNs, vdp.py:107-108)."""
    vfb = V.VectorField(V.VF_LOGISTIC, 1, 1, np.array([[1.0, 1.0], [1.0, -5.0], [1.0, 1.0]]), "logistic")
    impl.select("isotropic", ode_shape=(1,))
    u0 = np.array([[0.1], [1.0], [0.3]])
    bb = np.array([[1.0], [-5.0], [1.0]])
    tcs = [u0, u0 * (1 - bb * u0)]
    tcs.append(tcs[1] * (1 - 2 * bb * u0))
    fp = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(2), ivpsolvers.correction_ts0()))
    init = fp.initial_condition(tcs, 1.0)
    save_at = np.stack([np.linspace(0.0, 2.0, 4), np.linspace(0.0, 1.0, 4), np.linspace(0.0, 3.0, 4)])
    asolver = ivpsolve.adaptive(fp, atol=1e-8, rtol=1e-6)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vfb, init, save_at=save_at, dt0=0.05, adaptive_solver=asolver, device_controller=dc)
    host, one, pipe = run(False), run("one-wavefront"), run("pipelined")
    ex = lambda u, t: u * np.exp(t) / (1.0 + u * (np.exp(t) - 1.0))  # noqa: E731
    for sol in (host, one, pipe):
        assert abs(sol.u[0, -1, 0] - ex(0.1, 2.0)) < 1e-4 and abs(sol.u[2, -1, 0] - ex(0.3, 3.0)) < 1e-4
        assert not np.all(np.isfinite(sol.u[1])) or np.max(np.abs(sol.u[1])) > 1e3
    assert np.array_equal(one.num_steps, pipe.num_steps) and np.array_equal(one.num_steps[[0, 2]], host.num_steps[[0, 2]])
    assert relmax(one.u[[0, 2]], pipe.u[[0, 2]]) < 1e-12
    assert np.array_equal(np.isnan(one.u), np.isnan(pipe.u))
    assert relmax(one.u[[0, 2]], host.u[[0, 2]]) < 1e-11


def test_pipelined_smoother_handshake_timeout_is_a_device_error(hip_lib):
    """The hand-shakes of the pipelined smoother controller (ring of accepted steps, composition chain, in LDS) are bounded spins
    like the LQ's: with the bound set to one poll the smoother wavefronts give up before the filter wavefront has handed over its
    first step, and the solve reports PNODE_EDEVICE -- never a quietly failed trajectory."""
    N = 8
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, 0.3)])
    tc = np.stack(tcoeffs(vf, [y0], 4), axis=1)
    bt = _backend.Batch(batch=1, d=vf.d, nu=4, ode_order=1, model="isotropic", correction="ts0", strategy="fixedpoint",
                        calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=3)
    bt.set_initial(tc, np.zeros(1), np.ones(1))
    bt.set_tolerances(1e-6, 1e-6)
    bt.set_smoother_pipeline(1)
    bt.debug_set_lq_spin_limit(1)
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0()))
    control = ivpsolve.adaptive(solver, atol=1e-6, rtol=1e-6).control
    with pytest.raises(RuntimeError, match="pipelined smoother"):
        bt.solve_adaptive_save_at(np.linspace(0.0, 1.0, 4), np.full(1, 0.01), control, 1e-12, device_controller=True)
    bt.close()


def test_experiment_1_script_writes_the_references_files(hip_lib, golden, tmp_path):
    """experiments/van_der_pol.py = the reference's experiment 1 (experiments/1_van_der_pol/vdp.py:45-132): the eight result files
    with the reference's names, the baseline grid within the statistical band of the committed one (test_oracle_golden explains
    why it is not the same grid), the replay of the adaptive grid and the fine equidistant grid finite, the coarse one diverged."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("van_der_pol", os.path.join(root, "experiments", "van_der_pol.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    base, runs = mod.main(["--outdir", str(tmp_path)])
    names = ["baseline_grid", "baseline_solution", "grid_adaptive", "grid_fixed_accurate", "grid_fixed_inaccurate", "runtime_adaptive",
             "runtime_fixed_accurate", "runtime_fixed_inaccurate"]
    for nme in names:
        assert os.path.exists(os.path.join(str(tmp_path), f"vdp_{nme}.npy")), nme
    ref = golden["vdp"]
    assert abs(len(base.t) - len(ref["grid"])) <= 0.03 * len(ref["grid"]) and base.t[-1] == 6.3
    assert np.max(np.abs(base.u[:25, 0] - ref["solution"][:25, 0])) < 1e-12
    assert np.all(np.isfinite(runs["adaptive"][1].u)) and np.all(np.isfinite(runs["fixed_accurate"][1].u))
    assert abs(runs["fixed_accurate"][1].u[-1, 0] - runs["adaptive"][1].u[-1, 0]) < 1e-2 * abs(runs["adaptive"][1].u[-1, 0])
    assert np.load(os.path.join(str(tmp_path), "vdp_grid_fixed_accurate.npy")).shape[0] > 100 * len(base.t)


@pytest.mark.parametrize("case", ["rigid_nu2", "rigid_nu5", "pleiades2_nu3", "bruss32_nu3", "logistic_nu4"])
def test_pipelined_smoother_controller_over_shapes(hip_lib, case):
    """The pipelined device controller against the one-wavefront one and the host controller over the shapes the reference's
    experiments use (d = 1, 3, 14, 64; ode order 1 and 2; nu = 2 .. 5): identical decisions, smoothed means equal to rounding."""
    from odecheckpts import ivps

    if case.startswith("rigid"):
        vf, u0, (t0, t1), _ = ivps.rigid_body(time_span=(0.0, 5.0))
        inits, nu = [np.tile(u0[0], (3, 1))], int(case[-1])
    elif case.startswith("pleiades"):
        vf, u0, (t0, t1) = ivps.pleiades_2nd()
        t1, inits, nu = 0.5, [np.tile(u0[0], (3, 1)), np.tile(u0[1], (3, 1))], 3
    elif case.startswith("bruss"):
        vf = V.brusselator(32)
        t0, t1, inits, nu = 0.0, 0.5, [np.stack([brusselator_y0(32, 0.4 * b) for b in range(3)])], 3
    else:
        vf = V.logistic(1.0, 1.0)
        t0, t1, inits, nu = 0.0, 2.0, [np.array([[0.1], [0.2], [0.3]])], 4
    impl.select("isotropic", ode_shape=(vf.d,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts0(ode_order=vf.ode_order)))
    tol = np.array([1e-3, 1e-5, 1e-7])
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, inits, num=nu + 1 - vf.ode_order), 1.0)
    save_at = np.linspace(t0, t1, 6)
    run = lambda dc: ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
    host, one, pipe = run(False), run("one-wavefront"), run("pipelined")
    assert np.all(np.isfinite(host.u)) and host.num_steps[:, -1].min() >= 3
    for dev in (one, pipe):
        assert np.array_equal(dev.num_steps, host.num_steps) and np.array_equal(dev.num_attempts, host.num_attempts)
        assert relmax(dev.u, host.u) < 1e-10
    assert relmax(pipe.u, one.u) < 1e-12
