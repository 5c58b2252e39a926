"""Full-size (BASELINE configuration) checks through size-independent properties."""

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import ivpsolve, ivpsolvers, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

pytestmark = pytest.mark.gpu

N, NU, TOL, DT0 = 32, 3, 1e-6, 0.01


def _phase_batch(B, idx=None):
    x = np.linspace(0.0, 1.0, N)
    b = np.arange(B) if idx is None else np.asarray(idx)
    phases = 2.0 * np.pi * b / B
    return np.concatenate([1.0 + np.sin(2 * np.pi * x[None, :] + phases[:, None]), 3.0 * np.ones((len(b), N))], axis=1)


def _dense_solver():
    impl.select("dense", ode_shape=(2 * N,))
    return ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(NU), ivpsolvers.correction_ts1()))


def _run_attempts(vf, y0, num_attempts):
    solver = _dense_solver()
    tcs = taylor.odejet_padded_scan(vf, [y0], num=NU)
    init = solver.initial_condition(tcs, 1.0)
    B = y0.shape[0]
    batch = ivpsolve._make_batch(vf, init, solver, 1)
    batch.set_initial(init.tcoeffs, np.zeros(B), init.output_scale)
    batch.set_tolerances(TOL, TOL)
    loop = ivpsolve._AdaptiveLoop(batch, ivpsolve.adaptive(solver, atol=TOL, rtol=TOL), np.zeros(B), DT0)
    for _ in range(num_attempts):
        loop.attempt(np.ones(B, dtype=bool))
    return batch, loop, tcs


def test_headline_batch_is_composition_invariant_and_matches_oracle(hip_lib):
    """C4 (B = 1024 dense EKF1): a trajectory's result must not depend on what else is in the batch
    (bit-identical in a batch of 1024 and in a batch of 5), and must match the oracle."""
    vf = V.brusselator(N)
    B = 1024
    pick = [0, 1, 333, 777, 1023]
    big, loop_big, tcs_big = _run_attempts(vf, _phase_batch(B), 12)
    small, loop_small, _ = _run_attempts(vf, _phase_batch(B, pick), 12)
    m_big, m_small = big.read_mean(0), small.read_mean(0)
    assert np.array_equal(loop_big.steps[pick], loop_small.steps)
    assert np.array_equal(loop_big.t[pick], loop_small.t)
    assert np.array_equal(m_big[pick], m_small)
    assert np.array_equal(big.read_output_scale(0)[pick], small.read_output_scale(0))
    assert loop_big.steps.min() >= 4  # the initial transient rejects a few attempts
    # oracle on three of them, replaying the same controller
    cfg = po.SolverConfig(nu=NU, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    acfg = po.AdaptiveConfig(TOL, TOL)
    prob = oracle_problem(vf)
    for b in (0, 333, 1023):
        ast = po.adaptive_init(po.initial_state(cfg, [c[b] for c in tcs_big], 0.0), DT0)
        attempts = 0
        while attempts < 12:
            before = ast.num_attempts
            ast = po.rejection_loop(cfg, acfg, prob, ast, np.inf)
            attempts = ast.num_attempts
            if attempts > 12:
                break
        # compare at the last state both have reached: replay exactly loop_big.steps[b] accepted steps
        ast = po.adaptive_init(po.initial_state(cfg, [c[b] for c in tcs_big], 0.0), DT0)
        for _ in range(int(loop_big.steps[b])):
            ast = po.rejection_loop(cfg, acfg, prob, ast, np.inf)
        assert abs(ast.step_from.t - loop_big.t[b]) <= 1e-9 * loop_big.t[b]  # dt inherits the ~1e-8 noise of the error norm
        ref = ast.step_from.mean
        assert np.max(np.abs(m_big[b, 0] - ref[0])) / np.max(np.abs(ref[0])) < 1e-10
        assert np.max(np.abs(m_big[b, 1] - ref[1])) / np.max(np.abs(ref[1])) < 1e-8


def test_noise_free_measurement_property_full_batch(hip_lib):
    """After a correction the innovation covariance must vanish: H Sigma H^T = (H L)(H L)^T ~ 0 relative to
    the predicted one, and Sigma = R^T R stays symmetric PSD with rank <= D - d (checked on a sample)."""
    vf = V.brusselator(N)
    B = 1024
    batch, loop, _ = _run_attempts(vf, _phase_batch(B), 6)
    R = batch.read_factor(0)  # (B, D, D) with Sigma = R^T R
    means = batch.read_mean(0)
    d, D = 2 * N, (NU + 1) * 2 * N
    prob = oracle_problem(vf)
    for b in (0, 511, 1023):
        L = R[b].T
        J = prob.jac(means[b][:1], 0.0)[:, 0, :]
        H = np.zeros((d, D))
        H[:, d:2 * d] = np.eye(d)
        H[:, :d] = -J
        HL = H @ L
        # the corrected mean sits on the ODE manifold of the linearisation, the factor carries no mass along H
        assert np.linalg.norm(HL) < 1e-6 * np.linalg.norm(H) * np.linalg.norm(L)
        sv = np.linalg.svd(L, compute_uv=False)
        assert np.sum(sv > 1e-9 * sv[0]) <= D - d + 2


def test_isotropic_full_batch_contains_the_reference_trajectory(hip_lib, golden):
    """1024 isotropic EKF0 fixed-point solves of the reference's experiment-4 configuration in one batch
    (phase family); trajectory 0 IS the reference IVP: accepted steps bit-exact, checkpoint means 1e-9."""
    g = golden["brusselator"]
    vf = V.brusselator(N)
    B = 1024
    impl.select("isotropic", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0()))
    y0 = _phase_batch(B)
    assert np.max(np.abs(y0[0] - brusselator_y0(N))) == 0.0
    tcs = taylor.odejet_unroll(vf, [y0], num=4)
    sol = ivpsolve.solve_adaptive_save_at(vf, solver.initial_condition(tcs, 1.0), save_at=g["ts"], dt0=0.01,
                                          adaptive_solver=ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-8))
    i = list(g["N"]).index(N)
    assert int(sol.num_steps[0, -1]) == int(g["num_steps_checkpoint"][i])
    assert np.max(np.abs(sol.u[0] - g[f"ys_N{N}"])) < 1e-9
    assert sol.u.shape == (B, 200, 2 * N) and np.all(np.isfinite(sol.u))
    # phase symmetry of the problem family is NOT exact (Dirichlet pads), but every trajectory must stay bounded
    assert np.max(np.abs(sol.u)) < 10.0
    # step counts vary smoothly over the family
    assert sol.num_steps[:, -1].min() > 5000 and sol.num_steps[:, -1].max() < 30000


def test_brusselator_n128_three_quarters_of_a_million_decisions(hip_lib, golden):
    """Reference experiment 4 at N = 128 (d = 256, the ODE size of BASELINE config 5), device-resident controller:
    190 024 accepted steps, bit-exact with the committed JAX result; smoothed means within 1e-9."""
    g = golden["brusselator"]
    Nn = 128
    vf = V.brusselator(Nn)
    impl.select("isotropic", ode_shape=(2 * Nn,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0()))
    tcs = taylor.odejet_unroll(vf, [brusselator_y0(Nn)], num=4)
    sol = ivpsolve.solve_adaptive_save_at(vf, solver.initial_condition(tcs, 1.0), save_at=g["ts"], dt0=0.01,
                                          adaptive_solver=ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-8), device_controller=True)
    i = list(g["N"]).index(Nn)
    assert int(np.max(sol.num_steps)) == int(g["num_steps_checkpoint"][i]) == 190024
    assert np.max(np.abs(sol.u - g["ys_N128"])) < 1e-9


def test_config5_size_dense_smoother_on_the_large_path(hip_lib):
    """BASELINE config 5 at full state size: Brusselator N = 128 (d = 256), nu = 3, dense EKF1 (D = 1024) with
    fixed-memory checkpoint smoothing. A short adaptive solve with two checkpoints against the oracle: identical
    accept/reject decisions, one-step-smoothed checkpoints and fully smoothed means. Batch of 2 (config 5 runs 64 per
    GPU; the kernels are batch-strided, so the extra trajectories only add runtime)."""
    Nl, nu, tol = 128, 3, 1e-3
    vf = V.brusselator(Nl)
    impl.select("dense", ode_shape=(2 * Nl,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts1()))
    y0 = np.stack([brusselator_y0(Nl, 0.0), brusselator_y0(Nl, 0.8)])
    tcs = tcoeffs(vf, [y0], nu)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.array([0.0, 0.004, 0.008])
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.001, adaptive_solver=ivpsolve.adaptive(solver, atol=tol, rtol=tol))
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="fixedpoint", calibration="dynamic")
    for b in range(2):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tol, tol), oracle_problem(vf),
                                         po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.001)
        assert np.array_equal(sol.num_steps[b], osol.num_steps), b
        sm = po.markov_marginals_reverse(cfg, osol)
        assert np.max(np.abs(sol.u[b] - sm[:, 0])) / np.max(np.abs(sm[:, 0])) < 1e-8, b


def test_config3_work_precision_sweep_full_batch(hip_lib):
    """BASELINE config 3 at its batch size: rigid body (experiments/2), isotropic EKF0 fixed-point smoother, 1024
    tolerances in one batch for nu = 3 and nu = 5. Sampled members against the oracle (identical decisions), the whole
    sweep against a tight scipy solution (error follows the tolerance), and composition invariance (a member solved
    alone gives bit-identical results)."""
    from scipy.integrate import solve_ivp

    vf = V.rigid_body()
    B = 1024
    rt = 10.0 ** np.linspace(-1.0, -7.0, B)
    u0 = np.array([1.0, 0.0, 0.9])
    t1 = 10.0
    truth = solve_ivp(lambda t, y: vf(y), (0.0, t1), u0, rtol=1e-13, atol=1e-13, method="LSODA").y[:, -1]
    save_at = np.linspace(0.0, t1, 5)
    impl.select("isotropic", ode_shape=(3,))
    for nu in (3, 5):
        solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts0()))
        tcs = tcoeffs(vf, [np.tile(u0, (B, 1))], nu)
        init = solver.initial_condition(tcs, 1.0)
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-3 * rt, rtol=rt))
        assert np.all(np.isfinite(sol.u))
        err = np.sqrt(np.mean((sol.u[:, -1] - truth) ** 2, axis=1))
        # work-precision behaviour: two decades of tolerance buy at least one decade of error, tight end is accurate
        assert np.median(err[-64:]) < 1e-2 * np.median(err[:64]) and np.median(err[-64:]) < 1e-5
        steps = sol.num_steps[:, -1]
        assert np.median(steps[-64:]) > 1.2 * np.median(steps[:64])
        cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
        for b in (0, 333, 700, 1023):
            osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-3 * rt[b], rt[b]), oracle_problem(vf),
                                             po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.1)
            sm = po.markov_marginals_reverse(cfg, osol)
            if b < 512:   # loose and moderate tolerances: identical decisions
                assert np.array_equal(sol.num_steps[b], osol.num_steps), (nu, b)
                assert np.max(np.abs(sol.u[b] - sm[:, 0])) < 1e-8, (nu, b)
            else:         # tight tolerances: z = m' - f cancels to rounding level and single accept/reject decisions flip
                assert np.all(np.abs(sol.num_steps[b] - osol.num_steps) <= 0.03 * osol.num_steps + 2), (nu, b)
                assert np.max(np.abs(sol.u[b] - sm[:, 0])) < 100 * rt[b], (nu, b)
        idx = [5, 512, 1000]
        sub = ivpsolve.solve_adaptive_save_at(vf, solver.initial_condition([c[idx] for c in tcs], 1.0), save_at=save_at, dt0=0.1,
                                              adaptive_solver=ivpsolve.adaptive(solver, atol=1e-3 * rt[idx], rtol=rt[idx]))
        assert np.array_equal(sub.num_steps, sol.num_steps[idx]) and np.array_equal(sub.u, sol.u[idx])


def test_config2_stiff_tolerance_sweep_full_batch(hip_lib):
    """BASELINE config 2 at its batch size: stiff Van der Pol (mu = 1e3), nu = 4, dense EKF1, 256 tolerances in one batch
    (fused one-wavefront kernel), t in [0, 2] (through the first stiff transition). The sweep stays finite, work grows
    as the tolerance tightens, the terminal value converges to a tight Radau solution, and a member solved alone is
    bit-identical (composition invariance)."""
    from scipy.integrate import solve_ivp

    mu = 1e3
    vf = V.van_der_pol_order2(mu)
    B = 256
    tols = 10.0 ** np.linspace(-2.0, -7.0, B)
    t1 = 2.0
    truth = solve_ivp(lambda t, y: [y[1], mu * (y[1] * (1 - y[0] ** 2) - y[0])], (0.0, t1), [2.0, 0.0], rtol=1e-12, atol=1e-12,
                      method="Radau").y[0, -1]
    impl.select("dense", ode_shape=(1,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts1(ode_order=2)))
    u0s = [np.full((B, 1), 2.0), np.zeros((B, 1))]
    tcs = tcoeffs(vf, u0s, 4)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, t1, 3)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=ivpsolve.adaptive(solver, atol=tols, rtol=tols))
    assert np.all(np.isfinite(sol.u))
    err = np.abs(sol.u[:, -1, 0] - truth)
    assert np.median(err[-32:]) < 1e-2 * np.median(err[:32]) and np.median(err[-32:]) < 1e-5
    steps = sol.num_steps[:, -1]
    assert np.median(steps[-32:]) > 5 * np.median(steps[:32])
    idx = [3, 128, 250]
    sub = ivpsolve.solve_adaptive_save_at(vf, solver.initial_condition([c[idx] for c in tcs], 1.0), save_at=save_at, dt0=0.01,
                                          adaptive_solver=ivpsolve.adaptive(solver, atol=tols[idx], rtol=tols[idx]))
    assert np.array_equal(sub.num_steps, sol.num_steps[idx]) and np.array_equal(sub.u, sol.u[idx])


def test_config5_as_specified_batch_64_with_200_save_at_points(hip_lib):
    """BASELINE config 5 as specified for one GPU: Brusselator N = 128 (d = 256), nu = 3, dense EKF1 (D = 1024), fixed-memory
    checkpoint smoothing, 64 trajectories, save_at = 200 points (/root/reference/experiments/4_brusselator/run.py:119) -- the 199
    stored backward transitions are what makes the run 208 GB of device memory -- over a SHORT time span (the full span is
    scripts/bench_configs.py c5full, 35 s). The oracle cannot follow at this size in test time (its checks at D = 1024 are
    test_config5_size_dense_smoother_on_the_large_path); here: finite, every checkpoint stored, the smoothed means against the
    filter's at the same points and against scipy, and composition invariance -- two members solved alone give the bits of the batch."""
    from scipy.integrate import solve_ivp

    Nl, nu, tol, B, K = 128, 3, 1e-3, 64, 200
    vf = V.brusselator(Nl)
    impl.select("dense", ode_shape=(2 * Nl,))
    x = np.linspace(0.0, 1.0, Nl)
    ph = 2.0 * np.pi * np.arange(B) / B
    y0 = np.concatenate([1.0 + np.sin(2 * np.pi * x[None, :] + ph[:, None]), 3.0 * np.ones((B, Nl))], axis=1)
    save_at = np.linspace(0.0, 0.02, K)
    sols = {}
    for strategy in ("fixedpoint", "filter"):
        strat = ivpsolvers.strategy_fixedpoint if strategy == "fixedpoint" else ivpsolvers.strategy_filter
        solver = ivpsolvers.solver_dynamic(strat(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts1()))
        run = lambda idx, sv=solver: ivpsolve.solve_adaptive_save_at(  # noqa: E731
            vf, sv.initial_condition(tcoeffs(vf, [y0[idx]], nu), 1.0), save_at=save_at, dt0=0.001,
            adaptive_solver=ivpsolve.adaptive(sv, atol=tol, rtol=tol))
        sols[strategy] = run(np.arange(B)) if strategy == "fixedpoint" else run(np.array([0, 63]))
        if strategy == "fixedpoint":
            sub = run(np.array([0, 63]))
    fp = sols["fixedpoint"]
    assert fp.u.shape == (B, K, 2 * Nl) and np.all(np.isfinite(fp.u))
    assert np.all(np.diff(fp.num_steps, axis=1) >= 0) and fp.num_steps[:, -1].min() >= 3
    assert np.array_equal(sub.num_steps, fp.num_steps[[0, 63]]) and np.array_equal(sub.u, fp.u[[0, 63]])
    fl = sols["filter"]
    assert np.array_equal(fl.num_steps, fp.num_steps[[0, 63]])  # the smoother takes the filter's steps
    assert np.max(np.abs(fl.u - fp.u[[0, 63]])) < 50 * tol  # smoothed against filtered means: both inside the tolerance band
    truth = solve_ivp(lambda t, y: np.asarray(vf(y)), (0.0, save_at[-1]), y0[0], rtol=1e-10, atol=1e-10, method="LSODA").y[:, -1]
    assert np.max(np.abs(fp.u[0, -1] - truth)) < 50 * tol
