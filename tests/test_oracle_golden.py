"""The CPU oracle (oracle/pn_oracle.py) against the reference's committed outputs (tests/golden/)."""

import math

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, tcoeffs
from probdiffeq_hip import vector_fields as V


def test_iwp_constants():
    a, q = po.iwp_system_matrices(3)
    assert np.array_equal(a[0], [1, 3, 3, 1]) and np.array_equal(a[3], [0, 0, 0, 1])
    assert np.allclose(q @ q.T, [[1 / (7 - i - j) for j in range(4)] for i in range(4)])
    p, pinv = po.iwp_preconditioner(0.1, 3)
    assert np.allclose(p * pinv, 1.0) and np.isclose(p[3], 0.1**0.5) and np.isclose(p[0], 0.1**3.5 / 6)


def _vdp_oracle_run(u0_scale=1.0):
    vf = V.van_der_pol_order2(1e3)
    cfg = po.SolverConfig(nu=4, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    tc = tcoeffs(vf, [np.array([2.0 * u0_scale]), np.array([0.0])], 4)
    return po.solve_adaptive_save_every_step(cfg, po.AdaptiveConfig(1e-3, 1e-3), oracle_problem(vf), po.initial_state(cfg, tc, 0.0), 6.3,
                                             0.01)


def test_vdp_grid_prefix_and_statistics(golden):
    """Dense EKF1, nu=4, filter, dynamic, tol 1e-3 on stiff Van der Pol (reference vdp.py:52-80).

    The accepted grid is a fingerprint of every constant in the error estimate and the PI law, the rejection branch included
    (attempts 2, 3 and 5 of the run are rejected). The relative deviation of the oracle's grid from the reference's committed
    one stays below 1e-13 for the first 7 accepted points, 1e-12 for the first 26, 1e-6 for the first 212 and 1e-4 for the
    first 423: it grows like the separation of two neighbouring solutions of a chaotic map, not like a wrong constant (which
    shows within three steps, see reference_state in AdaptiveConfig). Beyond that the grid is compared statistically, see
    test_vdp_step_count_lies_inside_the_rounding_ensemble."""
    g = golden["vdp"]
    sol = _vdp_oracle_run()
    grid, ref = g["grid"], g["solution"][:, 0]
    rel = np.abs(sol.t[1:425] - grid[1:425]) / grid[1:425]
    assert np.max(rel[:7]) < 1e-13
    assert np.max(rel[:26]) < 1e-12
    assert np.max(rel[:212]) < 1e-6
    assert np.max(rel[:423]) < 1e-4
    assert np.max(np.abs(sol.u[:100, 0] - ref[:100])) < 1e-7
    assert sol.t[-1] == 6.3 and grid[-1] == 6.3
    # solution on the slow manifold (|du/dt| moderate): interpolate ours onto the golden grid
    mine = np.interp(grid, sol.t, sol.u[:, 0])
    slow = np.abs(np.gradient(ref, grid)) < 5.0
    assert slow.sum() > 500
    assert np.max(np.abs(mine[slow] - ref[slow])) < 2e-2


def test_vdp_step_count_lies_inside_the_rounding_ensemble(golden):
    """The reference took 2 911 accepted steps (vdp_baseline_grid.npy), the oracle 2 933. Twelve oracle runs whose initial value
    differs by k ulp (k = -6 .. 5) -- perturbations of the size of one rounding error -- spread over 2 912 .. 2 951 steps (48
    runs: mean 2930.3, sigma 10.2, minimum 2912, two of them <= 2912; residual noise of 1e-15 .. 1e-13 injected into f does not
    move the mean): the step count is a property of the rounding path, and the reference's count sits at the lower edge of the
    oracle's own ensemble (about the 5th percentile), not outside it. Asserted here: the reference's count within 3 sigma of the
    ensemble mean and no further than 10 steps below the ensemble minimum, the ensemble itself tight (sigma < 1 %)."""
    n_ref = len(golden["vdp"]["grid"]) - 1
    counts = np.array([len(_vdp_oracle_run(1.0 + k * 2.2e-16).t) - 1 for k in range(-6, 6)])
    mean, sigma = counts.mean(), counts.std()
    assert sigma < 0.01 * mean
    assert abs(n_ref - mean) < 3.0 * sigma + 1.0, (n_ref, mean, sigma, counts)
    assert n_ref >= counts.min() - 10


# relative deviation allowed up to accepted point k (measured: 9e-15, 1.1e-13, 1.4e-11, 8.3e-9, 1.4e-6, 1.1e-3 at the stiff jump, t = 5.87)
VDP_REPLAY_PROFILE = ((60, 1e-13), (100, 1e-12), (200, 2e-10), (800, 1e-7), (2400, 2e-5), (2912, 2e-2))
VDP_REPLAY_PROFILE_ORACLE = ((60, 5e-14), (100, 5e-13), (200, 1e-10), (800, 5e-8), (2400, 5e-6), (2912, 5e-3))


def test_vdp_fixed_grid_replay_pins_dense_ekf1_over_the_whole_reference_run(golden):
    """The reference replays its own accepted grid with solve_fixed_grid (vdp.py:88-91): no controller, hence no chaotic grid
    selection. The oracle on that grid against vdp_baseline_solution.npy over ALL 2 912 points pins the dense predict / correct
    arithmetic over the whole run (the adaptive tests above only pin a prefix). The deviation grows with the stiffness of the
    problem (largest at the jump of the relaxation oscillation, t = 5.87), not with a wrong constant."""
    g = golden["vdp"]
    vf = V.van_der_pol_order2(1e3)
    cfg = po.SolverConfig(nu=4, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    tc = tcoeffs(vf, [np.array([2.0]), np.array([0.0])], 4)
    sol = po.solve_fixed_grid(cfg, oracle_problem(vf), po.initial_state(cfg, tc, 0.0), g["grid"])
    ref = g["solution"][:, 0]
    assert sol.u.shape[0] == len(ref) == 2912
    rel = np.abs(sol.u[:, 0] - ref) / np.abs(ref)
    for k, tol in VDP_REPLAY_PROFILE_ORACLE:
        assert np.max(rel[:k]) < tol, (k, np.max(rel[:k]), tol)


@pytest.mark.parametrize("N", [4, 16, 32, 64])
def test_brusselator_checkpoint_solution(golden, N):
    """Isotropic EKF0 nu=4, fixed-point smoother, dynamic calibration, tol 1e-8, 200 checkpoints
    (reference experiments/4_brusselator/run.py:48-66,119-138): accepted-step count and smoothed means."""
    g = golden["brusselator"]
    vf = V.brusselator(N)
    prob = oracle_problem(vf)
    cfg = po.SolverConfig(nu=4, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    tc = tcoeffs(vf, [brusselator_y0(N)], 4)
    sol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-8, 1e-8), prob, po.initial_state(cfg, tc, 0.0), g["ts"], 0.01)
    i = list(g["N"]).index(N)
    assert sol.num_steps[-1] == g["num_steps_checkpoint"][i] == g["num_steps_terminal"][i]
    smoothed = po.markov_marginals_reverse(cfg, sol)[:, 0]
    ys = g[f"ys_N{N}"]
    assert np.array_equal(ys[0], brusselator_y0(N)) or np.max(np.abs(ys[0] - brusselator_y0(N))) < 1e-15
    # N=4 relaxes to a steady state and its backward pass is ill-conditioned: 1e-9 there
    tol = 1e-9 if N == 4 else 1e-10
    assert np.max(np.abs(smoothed - ys)) < tol


def test_brusselator_n8_is_decision_sensitive(golden):
    """N=8 sits on a rounding-sensitive accept/reject decision (an attempt with error norm 0.9989):
    the reference counts 882 steps, this restatement 883. Pin that the difference is exactly that."""
    g = golden["brusselator"]
    vf = V.brusselator(8)
    cfg = po.SolverConfig(nu=4, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    tc = tcoeffs(vf, [brusselator_y0(8)], 4)
    sol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-8, 1e-8), oracle_problem(vf), po.initial_state(cfg, tc, 0.0), g["ts"], 0.01)
    assert abs(int(sol.num_steps[-1]) - int(g["num_steps_checkpoint"][2])) <= 1
    smoothed = po.markov_marginals_reverse(cfg, sol)[:, 0]
    assert np.max(np.abs(smoothed - g["ys_N8"])) < 5e-9


def test_rigid_body_and_pleiades_definitions(golden):
    """The IVP definitions (diffeqzoo defaults / ivps.py:66-99) against the reference's truth solutions."""
    from scipy.integrate import solve_ivp

    g = golden["rigid_body"]
    vf = V.rigid_body()
    s = solve_ivp(lambda t, y: vf(y), (0, 50), [1.0, 0.0, 0.9], method="LSODA", atol=1e-13, rtol=1e-13, t_eval=g["ts"])
    assert np.max(np.abs(s.y.T - g["ys"])) < 1e-8
    g = golden["pleiades"]
    vf = V.pleiades_order1()
    y0 = g["ys"][0]
    s = solve_ivp(lambda t, y: vf(y), (0, 3), y0, method="DOP853", atol=1e-12, rtol=1e-12, t_eval=g["ts"])
    assert np.max(np.abs(s.y.T - g["ys"])) < 1e-6


def test_config1_van_der_pol_mu1_oracle_against_scipy():
    """BASELINE config 1 (experiments/old/7_node/node.py:23-37,88-91): Van der Pol mu=1 (d=2), nu=3, EKF0, isotropic,
    tol 1e-6 on the oracle, against an independent Runge-Kutta truth at sqrt(tol) like tests/test_ivpsolvers.py:52."""
    import scipy.integrate
    from helpers import oracle_problem, tcoeffs
    from probdiffeq_hip import vector_fields as V

    vf = V.van_der_pol_order1(1.0)
    save_at = np.linspace(0.0, 6.3, 22)
    tcs = tcoeffs(vf, [np.array([2.0, 0.0])], 3)
    truth = scipy.integrate.solve_ivp(lambda t, y: vf(y), (0.0, 6.3), [2.0, 0.0], method="DOP853", rtol=1e-12, atol=1e-12,
                                      t_eval=save_at).y.T
    for strategy in ("filter", "fixedpoint"):
        cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy=strategy, calibration="dynamic")
        sol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-6, 1e-6), oracle_problem(vf), po.initial_state(cfg, tcs, 0.0),
                                        save_at, 0.1)
        u = po.markov_marginals_reverse(cfg, sol)[:, 0] if strategy == "fixedpoint" else sol.u
        assert np.allclose(u, truth, atol=1e-3, rtol=1e-3)
        assert np.max(np.abs(u - truth)) < 1e-4  # (2e-6 .. 3e-6 in practice)
        assert 300 < int(sol.num_steps[-1]) < 600


def _three_body_setup(nu):
    from odecheckpts import ivps

    vf, (u0, du0), (t0, t1) = ivps.three_body_restricted()
    cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy="filter", calibration="none")
    tc = tcoeffs(vf, [u0, du0], nu)
    return vf, cfg, tc, t0, t1


def test_three_body_baseline_solution_of_experiment_5(golden):
    """Experiment 5's baseline (measure.py:155-163, 211-232): restricted three-body problem (Arenstorf orbit, second-order form),
    isotropic EKF0, nu = 3, strategy_filter, the UNCALIBRATED solver (ivpsolvers.solver, output scale 1), tol 1e-7, dt0 0.01,
    solve_adaptive_save_every_step. The reference committed the solution at every accepted step (data_solution.npy): the oracle
    takes exactly the same 7 540 steps and agrees to 2.2e-10 over the whole orbit. Pins ode_order = 2, the uncalibrated solver, the
    PI controller on a second problem, and the IVP's constants (diffeqzoo is not installed)."""
    ref = golden["three_body"]["baseline_solution"]
    vf, cfg, tc, t0, t1 = _three_body_setup(3)
    sol = po.solve_adaptive_save_every_step(cfg, po.AdaptiveConfig(1e-7, 1e-7), oracle_problem(vf), po.initial_state(cfg, tc, t0), t1, 0.01)
    assert sol.u.shape == ref.shape == (7541, 2)
    dev = np.max(np.abs(sol.u - ref), axis=1)
    assert np.max(dev[:1000]) < 1e-12 and np.max(dev) < 1e-9


@pytest.mark.parametrize("k", [0, 1, 2])
def test_three_body_step_counts_of_experiment_5(golden, k):
    """The 'No. steps' column of experiment 5 (measure.py:177-199: nu = 4, solve_adaptive_save_every_step at tol 1e-4, 1e-7, 1e-10,
    448 / 2 570 / 14 469 accepted steps; the smoother's forward pass takes the filter's decisions): bit-exact."""
    g = golden["three_body"]
    tol, want = float(g["smoother_tols"][k]), int(g["smoother_num_steps"][k])
    vf, cfg, tc, t0, t1 = _three_body_setup(4)
    sol = po.solve_adaptive_save_every_step(cfg, po.AdaptiveConfig(tol, tol), oracle_problem(vf), po.initial_state(cfg, tc, t0), t1, 0.01)
    assert len(sol.t) - 1 == want


def test_vdp_coarse_grid_leaves_the_fp64_range_in_exact_arithmetic(golden):
    """The reference asserts NaNs on its too-coarse equidistant grid (experiments/1_van_der_pol/vdp.py:94-97, 107-108). Is that an
    accident of its rounding path, or the algorithm? The dense EKF1 of that run (d = 1, nu = 4, dynamic calibration) restated
    in covariance form -- equivalent to the square-root form when nothing is rounded -- and run with mpmath at 40 digits over the
    first 800 points of the committed grid: the state leaves the limit cycle around step 380 and grows super-exponentially,
    |m| > 1e140 and sigma^2 > 1e290 at step 750 (30, 80 and 200 digits agree on the magnitudes, not on the digits: the orbit is
    chaotic, the blow-up is not). No fp64 evaluation of this solver on this grid can stay finite much beyond step 1200, where |u|
    itself passes 1e308; a finite result is not a legitimate rounding path. The HIP path asserts NaNs like the reference
    (test_gpu_parity.py::test_vdp_golden_prefix_through_api)."""
    mpmath = pytest.importorskip("mpmath")
    mp, mpf, matrix = mpmath.mp, mpmath.mpf, mpmath.matrix
    mp.dps = 40
    grid = golden["vdp"]["grid_fixed_inaccurate"]
    nu, n, mu = 4, 5, mpf(1000)
    A, Qb = matrix(n, n), matrix(n, n)
    for i in range(n):
        for j in range(n):
            if j >= i:
                A[i, j] = mpf(math.comb(nu - i, nu - j))
            Qb[i, j] = mpf(1) / (2 * nu + 1 - i - j)
    vf = lambda u, du: mu * (du * (1 - u * u) - u)  # noqa: E731
    tc = po.taylor_coefficients(lambda u, du: 1000.0 * (du * (1.0 - u * u) - u), [np.array([2.0]), np.array([0.0])], 3)
    m = matrix([mpf(float(c[0])) for c in tc])
    S = matrix(n, n)
    seen = {}
    for s in range(1, 801):
        dt = mpf(float(grid[s])) - mpf(float(grid[s - 1]))
        P, Pi = matrix(n, n), matrix(n, n)
        for i in range(n):
            P[i, i] = dt ** (mpf(nu - i) + mpf(1) / 2) / math.factorial(nu - i)
            Pi[i, i] = 1 / P[i, i]
        At, Qu = P * A * Pi, P * Qb * P
        me = At * m
        u, du = me[0], me[1]
        z = me[2] - vf(u, du)
        H = matrix(1, n)
        H[0, 0], H[0, 1], H[0, 2] = -mu * (-2 * u * du - 1), -mu * (1 - u * u), 1
        sig2 = z * z / (H * Qu * H.T)[0, 0]
        Sm = At * S * At.T + sig2 * Qu
        Sz = (H * Sm * H.T)[0, 0]
        K = Sm * H.T / Sz
        m, S = me - K * z, Sm - K * Sz * K.T
        if s in (250, 500, 750):
            seen[s] = (max(abs(m[i]) for i in range(n)), sig2)
    assert seen[250][0] < 1e3                              # still on the limit cycle (the oracle and the HIP path agree to 1e-7 here)
    assert seen[500][0] > mpf(10) ** 40                    # left it
    assert seen[750][0] > mpf(10) ** 140 and seen[750][1] > mpf(10) ** 290
