"""Round-2 GPU parity cases: terminal values against the reference's golden step counts, BASELINE config 1 as a whole
solve, shard invariance of the headline batch."""

import numpy as np
import pn_oracle as po
import pytest
import scipy.integrate
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import ivpsolve, ivpsolvers, sharding, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("N", [16, 32])
def test_terminal_values_golden_step_counts(hip_lib, golden, N):
    """solve_adaptive_terminal_values as the reference calls it (experiments/4_brusselator/run.py:82-99): isotropic
    nu=4 EKF0, fixed-point strategy, dynamic calibration, tol 1e-8, dt0 0.01, t in [0, 10]. The accepted-step count is
    pinned by the reference's committed `data_textbook.npy` (`num_steps`), the terminal value by the oracle."""
    g = golden["brusselator"]
    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0(ode_order=1)))
    asolver = ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-8, control=ivpsolve.control_proportional_integral())
    tcs = taylor.odejet_unroll(vf, [brusselator_y0(N)], num=4)
    init = solver.initial_condition(tcs, 1.0)
    sol = ivpsolve.solve_adaptive_terminal_values(vf, init, t0=0.0, t1=10.0, dt0=0.01, adaptive_solver=asolver)
    i = list(g["N"]).index(N)
    assert int(sol.num_steps) == int(g["num_steps_terminal"][i])
    assert sol.u.shape == (2 * N,) and float(sol.t) == 10.0
    # the checkpointed solve of the same configuration ends in the same state (golden `ys[-1]`)
    assert np.max(np.abs(sol.u - g[f"ys_N{N}"][-1])) < 1e-9
    cfg = po.SolverConfig(nu=4, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-8, 1e-8), oracle_problem(vf), po.initial_state(cfg, tcs, 0.0),
                                     np.array([0.0, 10.0]), 0.01)
    assert int(osol.num_steps[-1]) == int(sol.num_steps)
    assert relmax(sol.u, osol.u[-1]) < 1e-9


def test_config1_van_der_pol_mu1_whole_solve(hip_lib):
    """BASELINE config 1 (/root/reference/experiments/old/7_node/node.py:23-37,88-91): Van der Pol mu = 1 in first-order
    form (d = 2), y0 = (2, 0), t in [0, 6.3]; nu = 3 IWP prior, EKF0, isotropic, single trajectory, tol 1e-6, dt0 0.1.
    Whole adaptive solve against the oracle (identical decisions, means 1e-8) and against scipy (reference-test style)."""
    vf = V.van_der_pol_order1(1.0)
    impl.select("isotropic", ode_shape=(2,))
    save_at = np.linspace(0.0, 6.3, 22)
    tcs = tcoeffs(vf, [np.array([2.0, 0.0])], 3)
    truth = scipy.integrate.solve_ivp(lambda t, y: vf(y), (0.0, 6.3), [2.0, 0.0], method="DOP853", rtol=1e-12, atol=1e-12,
                                      t_eval=save_at).y.T
    for strat, ostrat in (("filter", "filter"), ("fixedpoint", "fixedpoint")):
        strategy = getattr(ivpsolvers, f"strategy_{strat}")(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts0())
        solver = ivpsolvers.solver_dynamic(strategy)
        asolver = ivpsolve.adaptive(solver, atol=1e-6, rtol=1e-6)
        init = solver.initial_condition(tcs, 1.0)
        cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy=ostrat, calibration="dynamic")
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-6, 1e-6), oracle_problem(vf), po.initial_state(cfg, tcs, 0.0),
                                         save_at, 0.1)
        for dev in (False, True):
            sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver, device_controller=dev)
            assert np.array_equal(sol.num_steps, osol.num_steps), (strat, dev)
            ref = po.markov_marginals_reverse(cfg, osol)[:, 0] if strat == "fixedpoint" else osol.u
            assert relmax(sol.u, ref) < 1e-8, (strat, dev)
            assert np.allclose(sol.u, truth, atol=1e-3, rtol=1e-3), (strat, dev)  # sqrt(tol), tests/test_ivpsolvers.py:52
    # terminal values / every step / fixed grid drivers on the same problem
    term = ivpsolve.solve_adaptive_terminal_values(vf, init, t0=0.0, t1=6.3, dt0=0.1, adaptive_solver=asolver)
    assert int(term.num_steps) == int(osol.num_steps[-1])
    fsolver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts0()))
    every = ivpsolve.solve_adaptive_save_every_step(vf, fsolver.initial_condition(tcs, 1.0), t0=0.0, t1=6.3, dt0=0.1,
                                                    adaptive_solver=ivpsolve.adaptive(fsolver, atol=1e-6, rtol=1e-6))
    assert every.t[-1] == 6.3 and len(every.t) >= int(osol.num_steps[-1])
    fixed = ivpsolve.solve_fixed_grid(vf, fsolver.initial_condition(tcs, 1.0), grid=every.t, solver=fsolver)
    assert relmax(fixed.u[:-1], every.u[:-1]) < 1e-9  # the replay of the accepted grid reproduces the adaptive run


def test_headline_batch_is_shard_invariant(hip_lib):
    """the 1024-trajectory headline batch (Brusselator N=32, nu=3, dense EKF1) solved as ONE batch and as
    8 shards of 128 (`sharding.shard_bounds`, the split bench.py --scaling strong uses at 8 GPUs), one after the other on
    one GPU: accepted steps, attempts, times and means bit-identical."""
    N, NU, TOL, DT0, B, W = 32, 3, 1e-6, 0.01, 1024, 8
    vf = V.brusselator(N)
    x = np.linspace(0.0, 1.0, N)
    ph = 2.0 * np.pi * np.arange(B) / B
    y0 = np.concatenate([1.0 + np.sin(2 * np.pi * x[None, :] + ph[:, None]), 3.0 * np.ones((B, N))], axis=1)
    impl.select("dense", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(NU), ivpsolvers.correction_ts1()))
    asolver = ivpsolve.adaptive(solver, atol=TOL, rtol=TOL)
    save_at = np.array([0.0, 0.02, 0.05])

    def run(ys):
        init = solver.initial_condition(taylor.odejet_padded_scan(vf, [ys], num=NU), 1.0)
        return ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=DT0, adaptive_solver=asolver)

    whole = run(y0)
    for rank in range(W):
        lo, hi = sharding.shard_bounds(B, W, rank)
        assert hi - lo == 128
        part = run(y0[lo:hi])
        assert np.array_equal(part.num_steps, whole.num_steps[lo:hi]), rank
        assert np.array_equal(part.num_attempts, whole.num_attempts[lo:hi]), rank
        assert np.array_equal(part.u, whole.u[lo:hi]), rank
        assert np.array_equal(part.output_scale, whole.output_scale[lo:hi]), rank


@pytest.mark.parametrize("N", [8, 32])
def test_device_resident_controller_of_the_dense_model(hip_lib, N):
    """SURVEY 8(f)-1 for the headline shape: dense EKF1 nu=3 on the Brusselator (N=32: d=64, D=256, MFMA kernels; N=8: generic
    kernels) with the accept/reject decision and the PI law in a controller kernel between the attempts
    (`device_controller=True`, pnode_advance_device) against the host controller: identical decisions (accepted steps and
    attempts per checkpoint), identical step sizes up to the last bits of log/exp, means within 1e-10."""
    vf = V.brusselator(N)
    B = 6
    y0 = np.stack([brusselator_y0(N, 0.7 * b) for b in range(B)])
    impl.select("dense", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts1()))
    tol = 10.0 ** -np.linspace(4, 6, B)  # a tolerance sweep: the trajectories need different numbers of attempts
    asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol)
    init = solver.initial_condition(taylor.odejet_padded_scan(vf, [y0], num=3), 1.0)
    save_at = np.linspace(0.0, 0.4, 5)
    host = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver)
    dev = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=True)
    assert np.array_equal(dev.num_steps, host.num_steps)
    assert np.array_equal(dev.num_attempts, host.num_attempts)
    assert host.num_steps[:, -1].min() >= 5 and len(set(host.num_steps[:, -1])) > 1
    assert relmax(dev.u, host.u) < 1e-10
    # the building block itself: 7 rounds without a host synchronisation advance every trajectory by 7 attempts
    batch = ivpsolve._make_batch(vf, init, solver, 1)
    batch.set_initial(init.tcoeffs, np.zeros(B), init.output_scale)
    batch.set_tolerances(tol, tol)
    dt, lp = np.full(B, 0.01), np.zeros(B)
    status, attempts = batch.advance_device(np.full(B, np.inf), dt, lp, asolver.control, asolver.numerical_zero, 7)
    assert np.all(status == 0) and np.all(attempts == 7)
    t, steps = batch.read_time()
    assert np.all(steps >= 1) and np.all(steps <= 7) and np.all(t > 0)
    batch.close()


def test_overflow_turns_into_nan_like_the_oracle(hip_lib, golden):
    """The reference expects NaNs from a solve that diverges (`assert jnp.any(jnp.isnan(...))`, experiments/1_van_der_pol/
    vdp.py:107-108): a NaN is a result, not an error, and must come out of the kernels as a NaN. Deterministic instance: the
    logistic-type ODE y' = y^2 - y from y0 = 2 blows up at t = ln 2; on a fixed grid past that time the EKF0 state overflows
    within a few steps in ANY implementation. The HIP path follows the oracle while the numbers are finite and is NaN afterwards,
    entry for entry with the oracle's NaNs from the second non-finite step on.
    (The reference's own instance, stiff Van der Pol on its too-coarse equidistant grid, is chaotic on the way to the overflow:
    HIP path and oracle agree to 1e-7 for 360 steps (|u| up to 1e9), separate at step ~400 and then follow different exploding
    orbits -- the oracle's crosses 1e75, where S = F F^T overflows, at step 573, the HIP path's peaks at 1e62
    (scripts/diag_vdp_coarse.py). test_gpu_parity.py::test_vdp_golden_prefix_through_api therefore asserts divergence there,
    and the NaN contract is asserted here.)"""
    vf = V.logistic(-1.0, 1.0)
    impl.select("isotropic", ode_shape=(1,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts0()))
    tcs = tcoeffs(vf, [np.array([2.0])], 3)
    grid = np.linspace(0.0, 3.0, 61)
    sol = ivpsolve.solve_fixed_grid(vf, solver.initial_condition(tcs, 1.0), grid=grid, solver=solver)
    cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy="filter", calibration="dynamic")
    with np.errstate(all="ignore"):
        osol = po.solve_fixed_grid(cfg, oracle_problem(vf), po.initial_state(cfg, tcs, 0.0), grid)
    u, ou = sol.u[:, 0], osol.u[:, 0]
    first = int(np.argmax(~np.isfinite(ou)))
    assert 15 < first < 30
    tame = np.abs(ou[:first]) < 1e100
    assert relmax(u[:first][tame], ou[:first][tame]) < 1e-8
    assert np.all(np.isnan(u[first + 1:])) and np.all(np.isnan(ou[first + 1:]))
    assert np.any(np.isnan(u))  # the reference's assertion


@pytest.mark.parametrize("case", ["small_vdp2", "generic_rigid", "mfma_bruss16", "large_bruss16", "iso_bruss8"])
def test_checkpoint_covariance_factors_of_the_filter(hip_lib, case, monkeypatch):
    """solution.posterior of the reference carries a Cholesky factor per save_at point (ivpsolvers.py:71-81, run.py:136-138):
    the filter's covariance at every checkpoint (interpolated ones: a time update without a measurement; corner ones: the
    state itself, whose factor is not triangular: the measurement update is a projector) against the oracle, R^T R to 1e-7, on every dense device path (the MFMA path keeps its state factor in strip
    order and de-tiles it for the store) and for the isotropic model."""
    from probdiffeq_hip import _backend

    model, corr = "dense", "ts1"
    if case == "small_vdp2":
        vf, nu, u0s, tol, save_at, dt0 = V.van_der_pol_order2(5.0), 4, [np.array([2.0]), np.array([0.0])], 1e-4, np.linspace(0, 2.0, 5), 0.01
    elif case == "generic_rigid":
        vf, nu, u0s, tol, save_at, dt0 = V.rigid_body(), 3, [np.array([1.0, 0.0, 0.9])], 1e-4, np.linspace(0, 4.0, 5), 0.1
        monkeypatch.setattr(_backend, "dense_path", 1)
    elif case == "iso_bruss8":
        model, corr = "isotropic", "ts0"
        vf, nu, u0s, tol, save_at, dt0 = V.brusselator(8), 3, [brusselator_y0(8)], 1e-5, np.linspace(0, 0.5, 4), 0.01
    else:
        vf, nu, u0s, tol, save_at, dt0 = V.brusselator(16), 3, [brusselator_y0(16)], 1e-4, np.linspace(0, 0.3, 4), 0.01
        if case.startswith("large"):
            monkeypatch.setattr(_backend, "dense_path", 2)
    impl.select(model, ode_shape=(vf.d,))
    correction = getattr(ivpsolvers, f"correction_{corr}")(ode_order=vf.ode_order)
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(nu), correction))
    tcs = tcoeffs(vf, u0s, nu)
    # a save_at point that coincides with an accepted step exercises the corner branch: take it from a first solve
    first = ivpsolve.solve_adaptive_save_every_step(vf, solver.initial_condition(tcs, 1.0), t0=save_at[0], t1=save_at[-1], dt0=dt0,
                                                    adaptive_solver=ivpsolve.adaptive(solver, atol=tol, rtol=tol))
    save_at = np.sort(np.append(save_at, first.t[3]))
    sol = ivpsolve.solve_adaptive_save_at(vf, solver.initial_condition(tcs, 1.0), save_at=save_at, dt0=dt0, checkpoint_factors=True,
                                          adaptive_solver=ivpsolve.adaptive(solver, atol=tol, rtol=tol))
    cfg = po.SolverConfig(nu=nu, model=model, correction=corr, strategy="filter", calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tol, tol), oracle_problem(vf), po.initial_state(cfg, tcs, 0.0), save_at, dt0)
    assert np.array_equal(sol.num_steps, osol.num_steps)
    R = sol.posterior.checkpoint_factors()
    m = (nu + 1) * (vf.d if model == "dense" else 1)
    assert R.shape == (len(save_at) - 1, m, m)
    for i in range(1, len(save_at)):
        cov_ref = osol.states[i].chol @ osol.states[i].chol.T
        cov = R[i - 1].T @ R[i - 1]
        # (Van der Pol: z = m'' - f cancels large terms, the factors carry ~1e-6 of rounding noise after a few dozen steps)
        assert np.max(np.abs(cov - cov_ref)) <= (1e-5 if case == "small_vdp2" else 1e-7) * np.max(np.abs(cov_ref)), (case, i)


@pytest.mark.gpu
@pytest.mark.parametrize("N,nu", [(16, 3), (16, 5), (32, 3)])
def test_persistent_workgroups_with_inactive_trajectories_in_their_sequence(hip_lib, N, nu):
    """The one-launch LQ runs one persistent workgroup per CU; with more trajectories than CUs a workgroup takes several, the passes
    of consecutive trajectories overlap (group A is in the next trajectory while group B finishes the current one) and inactive
    trajectories are skipped inside the sequence. 700 trajectories (2-3 per workgroup), random activity masks and random
    accept/reject decisions over three attempts, at D = 128, 192 (an odd number of 64-row blocks: group A owns two consecutive
    passes across the trajectory boundary) and 256: every error estimate and every mean must agree with the two-launch kernels
    (`pnode_set_generic_path(h, 3)`), which know nothing of workgroup sequences."""
    from probdiffeq_hip import _backend, vector_fields as V
    from helpers import brusselator_y0, tcoeffs

    rng = np.random.default_rng(7)
    B = 700
    vf = V.brusselator(N)
    y0 = np.stack([brusselator_y0(N, ph) for ph in rng.uniform(0.0, 2.0, B)])
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)

    def make(path):
        bt = _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=1, model="dense", correction="ts1", strategy="filter",
                            calibration="dynamic", vf_id=vf.vf_id, vf_params=vf.param_array(), num_checkpoints=2)
        if path:
            bt.set_generic_path(path)
        bt.set_initial(tc, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
        return bt

    one, two = make(0), make(3)
    e1, e2 = np.zeros(B), np.zeros(B)
    for it in range(3):
        dts = 0.01 * rng.uniform(0.5, 2.0, B)
        act = (rng.uniform(size=B) < 0.7).astype(np.uint8)
        acc = (act & (rng.uniform(size=B) < 0.8)).astype(np.uint8)
        one.attempt_step(dts, act, e1)
        two.attempt_step(dts, act, e2)
        live = act.astype(bool)
        assert np.all(np.isfinite(e1[live]))
        assert np.max(np.abs(e1[live] - e2[live]) / e2[live]) < 1e-7, it
        one.commit(acc)
        two.commit(acc)
        m1, m2 = one.read_mean(0), two.read_mean(0)
        assert np.max(np.abs(m1[:, 0] - m2[:, 0])) <= 1e-10 * np.max(np.abs(m2[:, 0])), it
        assert np.max(np.abs(m1[:, 1] - m2[:, 1])) <= 1e-8 * np.max(np.abs(m2[:, 1])), it
    R1, R2 = one.read_factor(0), two.read_factor(0)
    for b in (0, 255, 256, 511, 512, B - 1):
        c1, c2 = R1[b].T @ R1[b], R2[b].T @ R2[b]
        assert np.max(np.abs(c1 - c2)) <= 1e-8 * np.max(np.abs(c2)), b
    one.close()
    two.close()
