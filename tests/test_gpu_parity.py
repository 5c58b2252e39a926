"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp64): single-step quantities 1e-10 relative; whole adaptive solves 1e-8 relative on
posterior means (BASELINE north star), with accept/reject sequences required to be identical.
"""

import numpy as np
import pn_oracle as po
import pytest
from helpers import brusselator_y0, oracle_problem, scipy_truth, tcoeffs
from probdiffeq_hip import _backend, ivpsolve, ivpsolvers, stats
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl

pytestmark = pytest.mark.gpu


def rel(a, b, floor=1e-300):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))


def relmax(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def _iso_batch(vf, nu, strategy, calibration, B, K=4):
    return _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=vf.ode_order, model="isotropic", correction="ts0",
                          strategy=strategy, calibration=calibration, vf_id=vf.vf_id, vf_params=vf.param_array(),
                          num_checkpoints=K)


@pytest.mark.parametrize("nu", [2, 3, 4, 5])
@pytest.mark.parametrize("strategy", ["filter", "fixedpoint"])
def test_iso_steps_match_oracle(hip_lib, nu, strategy):
    """Three consecutive attempted steps (accepted unconditionally) of the fused isotropic kernel."""
    N = 16
    vf = V.brusselator(N)
    B = 5
    phases = np.linspace(0, 1, B)
    y0 = np.stack([brusselator_y0(N, ph) for ph in phases])
    tc = tcoeffs(vf, [y0], nu)
    batch = _iso_batch(vf, nu, strategy, "dynamic", B)
    batch.set_initial(np.stack(tc, axis=1), np.zeros(B), np.ones(B))
    batch.set_tolerances(1e-6, 1e-6)
    cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy=strategy, calibration="dynamic")
    acfg = po.AdaptiveConfig(1e-6, 1e-6)
    prob = oracle_problem(vf)
    states = [po.initial_state(cfg, [c[b] for c in tc], 0.0) for b in range(B)]
    err = np.zeros(B)
    for it, dt0 in enumerate([0.01, 0.004, 0.007]):
        dts = dt0 * (1.0 + 0.1 * np.arange(B))
        batch.attempt_step(dts, np.ones(B, np.uint8), err)
        batch.commit(np.ones(B, np.uint8))
        means = batch.read_mean(0)
        fac = batch.read_factor(0)
        sig = batch.read_output_scale(0)
        for b in range(B):
            e, st = po.step(cfg, prob, states[b], float(dts[b]))
            en = po.normalise_error(acfg, e, st.mean[0], states[b].mean[0])
            # z = m[1] - f(m[0]) cancels O(100)-sized terms down to the local error, so the error
            # norm / sigma carry ~1e-8 relative rounding noise; the gain feeds that noise into the
            # high derivatives (weakly observed), hence the graded tolerances.
            assert abs(err[b] - en) <= 1e-6 * en, (it, b)
            assert abs(sig[b] - st.output_scale) <= 1e-6 * st.output_scale
            for i in range(nu + 1):
                tol_i = 1e-12 if i == 0 else (1e-10 if i == 1 else 1e-6)
                assert relmax(means[b, i], st.mean[i]) < tol_i, (it, b, i)
            cov = fac[b].T @ fac[b]
            cov_ref = st.chol @ st.chol.T
            assert relmax(cov, cov_ref) < 1e-6, (it, b)
            states[b] = st
    batch.close()


def test_iso_inactive_trajectories_untouched(hip_lib):
    vf = V.brusselator(8)
    B = 3
    y0 = np.stack([brusselator_y0(8, ph) for ph in (0.0, 0.3, 0.6)])
    tc = tcoeffs(vf, [y0], 3)
    batch = _iso_batch(vf, 3, "filter", "dynamic", B)
    batch.set_initial(np.stack(tc, axis=1), np.zeros(B), np.ones(B))
    err = np.full(B, -1.0)
    batch.attempt_step(np.full(B, 0.01), np.array([1, 0, 1], np.uint8), err)
    assert err[1] == -1.0 and err[0] > 0 and err[2] > 0
    batch.commit(np.array([1, 0, 0], np.uint8))
    t, steps = batch.read_time()
    assert list(steps) == [1, 0, 0] and t[0] == 0.01 and t[1] == 0.0 and t[2] == 0.0
    means = batch.read_mean(0)
    assert np.array_equal(means[1], np.stack(tc, axis=1)[1]) and np.array_equal(means[2], np.stack(tc, axis=1)[2])
    batch.close()


@pytest.mark.parametrize("N", [16, 32])
def test_brusselator_golden_through_api(hip_lib, golden, N):
    """The reference's experiment 4 configuration end to end through the probdiffeq-style API:
    accepted-step count bit-exact, smoothed checkpoint means vs the committed JAX output."""
    g = golden["brusselator"]
    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(2 * N,))
    ibm = ivpsolvers.prior_ibm(num_derivatives=4)
    ts0 = ivpsolvers.correction_ts0(ode_order=1)
    strategy = ivpsolvers.strategy_fixedpoint(ibm, ts0)
    solver = ivpsolvers.solver_dynamic(strategy)
    ctrl = ivpsolve.control_proportional_integral()
    asolver = ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-8, control=ctrl)
    from probdiffeq_hip import taylor

    tcs = taylor.odejet_unroll(vf, [brusselator_y0(N)], num=4)
    init = solver.initial_condition(tcs, 1.0)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=g["ts"], dt0=0.01, adaptive_solver=asolver)
    i = list(g["N"]).index(N)
    assert int(np.max(sol.num_steps)) == int(g["num_steps_checkpoint"][i])
    assert sol.u.shape == (200, 2 * N)
    assert np.max(np.abs(sol.u - g[f"ys_N{N}"])) < 1e-9
    # reference post-processing (ivpsolvers.py:80-84)
    seq = stats.markov_select_terminal(sol.posterior)
    margs = stats.markov_marginals(seq, reverse=True)
    assert margs.mean.shape[1] == 199
    # ... and the oracle on the same problem
    cfg = po.SolverConfig(nu=4, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-8, 1e-8), oracle_problem(vf),
                                     po.initial_state(cfg, tcs, 0.0), g["ts"], 0.01)
    assert np.array_equal(osol.num_steps, sol.num_steps)
    assert int(sol.num_attempts) == osol.num_attempts
    sm = po.markov_marginals_reverse(cfg, osol)
    assert relmax(sol.posterior.means[0, :, 0], sm[:, 0]) < 1e-8
    # one-step-smoothed checkpoint values (before the backward pass)
    ck = sol.posterior.batch.read_checkpoint_means()[0]
    assert relmax(ck[:, 0], osol.u[1:]) < 1e-10


def test_iso_batch_tolerance_sweep_filter(hip_lib):
    """Rigid body (reference experiment 2), filter, tolerance sweep as the batch axis; each trajectory
    must equal the oracle run at that tolerance (same accept/reject sequence)."""
    vf = V.rigid_body()
    tols = 10.0 ** np.array([-3.0, -4.0, -5.0, -7.0])
    B = len(tols)
    impl.select("isotropic", ode_shape=(3,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts0()))
    u0 = np.tile(np.array([1.0, 0.0, 0.9]), (B, 1))
    tcs = tcoeffs(vf, [u0], 3)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, 10.0, 6)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tols, rtol=tols)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver)
    cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy="filter", calibration="dynamic")
    for b, tol in enumerate(tols):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-3 * tol, tol), oracle_problem(vf),
                                         po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.1)
        if tol >= 1e-5:
            assert np.array_equal(sol.num_steps[b], osol.num_steps)
            assert relmax(sol.u[b], osol.u) < 1e-9
        else:
            # atol = 1e-10: rounding noise in the weakly observed 3rd derivative (amplified by the gain)
            # reaches the 1e-6 relative level in the error norm, so individual accept/reject decisions
            # may differ between ANY two implementations; the solutions agree far below the tolerance.
            assert abs(int(sol.num_steps[b, -1]) - int(osol.num_steps[-1])) <= 0.06 * osol.num_steps[-1]
            assert np.max(np.abs(sol.u[b] - osol.u)) < 1e-2 * tol ** 0.5


# --------------------------------------------------------------------------------------------
# dense EKF1
# --------------------------------------------------------------------------------------------


def _dense_batch(vf, nu, correction, B, K=2, calibration="dynamic"):
    return _backend.Batch(batch=B, d=vf.d, nu=nu, ode_order=vf.ode_order, model="dense", correction=correction,
                          strategy="filter", calibration=calibration, vf_id=vf.vf_id, vf_params=vf.param_array(),
                          num_checkpoints=K)


@pytest.mark.parametrize("path", ["default", "generic"])
@pytest.mark.parametrize("case", ["vdp2", "brusselator8", "rigid", "vdp1_ts0"])
def test_dense_steps_match_oracle(hip_lib, case, path):
    """Single attempted steps of the dense model against the oracle. "default" takes the fused one-wavefront kernel
    for D <= 32 (vdp2, rigid, vdp1_ts0) and the MFMA kernels for brusselator8; "generic" forces the generic kernels."""
    if case == "vdp2":
        vf, nu, corr, u0s = V.van_der_pol_order2(1e3), 4, "ts1", [np.array([[2.0]]), np.array([[0.0]])]
        dts = [1e-4, 1e-4, 2e-4]
    elif case == "brusselator8":
        vf, nu, corr, u0s = V.brusselator(8), 3, "ts1", [brusselator_y0(8)[None]]
        dts = [0.01, 0.02, 0.03]
    elif case == "rigid":
        vf, nu, corr, u0s = V.rigid_body(), 2, "ts1", [np.array([[1.0, 0.0, 0.9]])]
        dts = [0.05, 0.1, 0.1]
    else:
        vf, nu, corr, u0s = V.van_der_pol_order1(1.0), 3, "ts0", [np.array([[2.0, 0.0]])]
        dts = [0.05, 0.05, 0.1]
    tc = tcoeffs(vf, u0s, nu)
    B = 1
    batch = _dense_batch(vf, nu, corr, B)
    if path == "generic":
        batch.set_generic_path(True)
    batch.set_initial(np.stack(tc, axis=1), np.zeros(B), np.ones(B))
    batch.set_tolerances(1e-4, 1e-4)
    cfg = po.SolverConfig(nu=nu, model="dense", correction=corr, strategy="filter", calibration="dynamic")
    acfg = po.AdaptiveConfig(1e-4, 1e-4)
    prob = oracle_problem(vf)
    st = po.initial_state(cfg, [c[0] for c in tc], 0.0)
    err = np.zeros(B)
    for it, dt in enumerate(dts):
        batch.attempt_step(np.full(B, dt), np.ones(B, np.uint8), err)
        batch.commit(np.ones(B, np.uint8))
        e, new = po.step(cfg, prob, st, dt)
        en = po.normalise_error(acfg, e, new.mean[0], st.mean[0])
        assert abs(err[0] - en) <= 1e-8 * en, (it, err[0], en)
        means = batch.read_mean(0)[0]
        for i in range(nu + 1):
            assert relmax(means[i], new.mean[i]) < 1e-9, (it, i)
        R = batch.read_factor(0)[0]
        assert relmax(R.T @ R, new.chol @ new.chol.T) < 1e-7, it
        assert abs(batch.read_output_scale(0)[0] - new.output_scale) <= 1e-8 * new.output_scale
        st = new
    batch.close()


@pytest.mark.parametrize("case", ["small", "mfma", "twolq", "large", "generic"])
def test_dense_inactive_and_rejected_trajectories_untouched(hip_lib, case):
    """Every dense device path: an inactive trajectory keeps its state and error slot bit for bit, a rejected attempt
    leaves STEP_FROM intact (the next attempt from the same state reproduces the same numbers), an accepted one advances."""
    if case == "small":
        vf, nu, y0 = V.rigid_body(), 3, np.array([[1.0, 0.0, 0.9], [0.9, 0.1, 0.8], [1.1, -0.1, 1.0]])
    else:
        N = {"mfma": 16, "twolq": 16, "large": 48, "generic": 12}[case]
        vf, nu, y0 = V.brusselator(N), 3, np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.3, 0.6)])
    B = 3
    tc = np.stack(tcoeffs(vf, [y0], nu), axis=1)
    batch = _dense_batch(vf, nu, "ts1", B)
    if case == "twolq":
        batch.set_generic_path(3)
    batch.set_initial(tc, np.zeros(B), np.ones(B))
    batch.set_tolerances(1e-5, 1e-5)
    err = np.full(B, -1.0)
    batch.attempt_step(np.full(B, 0.01), np.array([1, 0, 1], np.uint8), err)
    assert err[1] == -1.0 and err[0] > 0 and err[2] > 0
    e_first = err.copy()
    batch.commit(np.array([1, 0, 0], np.uint8))   # trajectory 2: rejected
    t, steps = batch.read_time()
    assert list(steps) == [1, 0, 0] and t[0] == 0.01 and t[1] == 0.0 and t[2] == 0.0
    means = batch.read_mean(0)
    assert np.array_equal(means[1], tc[1]) and np.array_equal(means[2], tc[2])
    assert not np.array_equal(means[0], tc[0])
    assert np.all(batch.read_factor(0)[1] == 0.0) and np.all(batch.read_factor(0)[2] == 0.0)
    # the rejected trajectory repeats the attempt from the untouched state: identical error estimate
    batch.attempt_step(np.full(B, 0.01), np.array([0, 0, 1], np.uint8), err)
    assert err[2] == e_first[2]
    batch.close()


LQ_PATHS = {"one_launch": 0, "two_launch": 3}  # pnode_set_generic_path values that select the LQ kernel of the MFMA path


@pytest.mark.parametrize("lq", list(LQ_PATHS))
@pytest.mark.parametrize("N,nu", [(16, 3), (32, 3), (16, 5), (32, 4)])
def test_dense_mfma_path_matches_oracle_and_generic(hip_lib, N, nu, lq):
    """The MFMA-tiled LQ / measurement-update kernels (d % 16 == 0) against the oracle and against the
    generic HIP kernels, over several consecutive steps with growing covariance. Both time-update kernels of the MFMA path:
    the one-launch LQ with overlapping 64-row passes and register-resident panel factorisation (csrc/pnode_dense_lq4.hip,
    the default at D = 128, 192, 256) and the two-launch LQ with 32-row passes (csrc/pnode_dense_mfma.hip)."""
    vf = V.brusselator(N)
    B = 3
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.4, 1.1)])
    tc = tcoeffs(vf, [y0], nu)
    tc_arr = np.stack(tc, axis=1)
    fast = _dense_batch(vf, nu, "ts1", B)
    if LQ_PATHS[lq]:
        fast.set_generic_path(LQ_PATHS[lq])
    slow = _dense_batch(vf, nu, "ts1", B)
    slow.set_generic_path(True)
    for bt in (fast, slow):
        bt.set_initial(tc_arr, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    acfg = po.AdaptiveConfig(1e-5, 1e-5)
    prob = oracle_problem(vf)
    states = [po.initial_state(cfg, [c[b] for c in tc], 0.0) for b in range(B)]
    ef, es = np.zeros(B), np.zeros(B)
    for it, dt0 in enumerate([0.01, 0.02, 0.015, 0.03]):
        dts = dt0 * (1.0 + 0.2 * np.arange(B))
        fast.attempt_step(dts, np.ones(B, np.uint8), ef)
        slow.attempt_step(dts, np.ones(B, np.uint8), es)
        fast.commit(np.ones(B, np.uint8))
        slow.commit(np.ones(B, np.uint8))
        mf, ms = fast.read_mean(0), slow.read_mean(0)
        Rf, Rs = fast.read_factor(0), slow.read_factor(0)
        for b in range(B):
            e, new = po.step(cfg, prob, states[b], float(dts[b]))
            en = po.normalise_error(acfg, e, new.mean[0], states[b].mean[0])
            assert abs(ef[b] - en) <= 1e-7 * en, (it, b, ef[b], en)
            assert abs(ef[b] - es[b]) <= 1e-7 * en
            for i in range(nu + 1):
                tol_i = 1e-11 if i == 0 else (1e-9 if i == 1 else 1e-6)
                assert relmax(mf[b, i], new.mean[i]) < tol_i, (it, b, i)
                assert relmax(mf[b, i], ms[b, i]) < tol_i, (it, b, i)
            cov_ref = new.chol @ new.chol.T
            assert relmax(Rf[b].T @ Rf[b], cov_ref) < 1e-7, (it, b)
            assert relmax(Rs[b].T @ Rs[b], cov_ref) < 1e-7, (it, b)
            states[b] = new
    fast.close()
    slow.close()


@pytest.mark.parametrize("N,nu", [(16, 3), (32, 3), (64, 3), (16, 5), (48, 2), (96, 2), (112, 3)])
def test_dense_large_path_matches_oracle_and_mfma_path(hip_lib, N, nu):
    """The HBM-tiled large-D kernels (batched MFMA GEMMs, blocked LQ / Cholesky / triangular solve; the path BASELINE
    config 5 takes at D = 1024) forced at sizes the oracle steps through quickly: several steps with per-trajectory
    step sizes, a rejected step (destination slots drift apart) and an inactive trajectory, against the oracle and the
    default path. (16, 5): D = 192 = 128 + 64, a shorter last outer block; (48, 2): d = 96, D = 288 is not a multiple of
    64 -> not supported there, the generic kernels are compared instead; (96, 2): D = 576 and (112, 3): D = 896 are the sizes
    whose K does not split into whole 256-column slices -- the split-K of the in-block G GEMM must fall back to 3 x 192
    resp. one slice (round-1 advisor finding).)"""
    vf = V.brusselator(N)
    B = 3
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.4, 1.1)])
    tc = tcoeffs(vf, [y0], nu)
    tc_arr = np.stack(tc, axis=1)
    large = _dense_batch(vf, nu, "ts1", B)
    if ((nu + 1) * vf.d) % 64 == 0:
        large.set_generic_path(2)
    else:
        with pytest.raises(ValueError):
            large.set_generic_path(2)
    ref = _dense_batch(vf, nu, "ts1", B)
    for bt in (large, ref):
        bt.set_initial(tc_arr, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
    cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    acfg = po.AdaptiveConfig(1e-5, 1e-5)
    prob = oracle_problem(vf)
    states = [po.initial_state(cfg, [c[b] for c in tc], 0.0) for b in range(B)]
    el, er = np.zeros(B), np.zeros(B)
    plan = [(0.01, [1, 1, 1], [1, 1, 1]), (0.02, [1, 1, 1], [1, 0, 1]), (0.015, [1, 1, 0], [1, 1, 0]), (0.03, [1, 1, 1], [1, 1, 1])]
    for it, (dt0, active, accept) in enumerate(plan):
        # (N >= 96: the diffusion term is N^2-stiff; with the full step sizes the EKF1 steps overshoot to |u| ~ 10^3 and the
        # comparison measures the conditioning of an unstable trajectory)
        dts = dt0 * (1.0 + 0.2 * np.arange(B)) * (0.25 if N >= 96 else 1.0)
        act = np.array(active, np.uint8)
        acc = np.array(accept, np.uint8)
        large.attempt_step(dts, act, el)
        ref.attempt_step(dts, act, er)
        large.commit(acc)
        ref.commit(acc)
        ml, mr = large.read_mean(0), ref.read_mean(0)
        Rl = large.read_factor(0)
        for b in range(B):
            if not active[b]:
                continue
            e, new = po.step(cfg, prob, states[b], float(dts[b]))
            en = po.normalise_error(acfg, e, new.mean[0], states[b].mean[0])
            assert abs(el[b] - en) <= 1e-7 * en, (it, b, el[b], en)
            assert abs(el[b] - er[b]) <= 1e-7 * en
            if not accept[b]:
                continue
            for i in range(nu + 1):
                tol_i = 1e-11 if i == 0 else (1e-9 if i == 1 else (1e-6 if nu <= 4 else 1e-4))
                if N >= 96 and i == 0:
                    tol_i = 2e-10
                assert relmax(ml[b, i], new.mean[i]) < tol_i, (it, b, i)
                assert relmax(ml[b, i], mr[b, i]) < tol_i, (it, b, i)
            # nu = 7: the covariance spans 20 orders of magnitude; the comparison is conditioning-limited
            assert relmax(Rl[b].T @ Rl[b], new.chol @ new.chol.T) < (1e-7 if nu <= 4 else 1e-5), (it, b)
            states[b] = new
    large.close()
    ref.close()


def test_dense_maximum_size_generic_path(hip_lib):
    """BASELINE config 5 size (Brusselator N=128: d=256, D=1024), N=64, nu=4 (D=640) and N=48, nu=3 (d=96, D=384): shapes
    beyond the register-resident kernels, on the large-D path by default; N=24, nu=3 (d=48: generic kernels). Two steps
    against the oracle."""
    for N, nu in ((128, 3), (64, 4), (48, 3), (24, 3)):
        vf = V.brusselator(N)
        tc = tcoeffs(vf, [brusselator_y0(N)[None]], nu)
        batch = _dense_batch(vf, nu, "ts1", 1)
        batch.set_initial(np.stack(tc, axis=1), np.zeros(1), np.ones(1))
        batch.set_tolerances(1e-5, 1e-5)
        cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
        acfg = po.AdaptiveConfig(1e-5, 1e-5)
        st = po.initial_state(cfg, [c[0] for c in tc], 0.0)
        prob = oracle_problem(vf)
        err = np.zeros(1)
        for dt in (0.005, 0.01):
            batch.attempt_step(np.full(1, dt), np.ones(1, np.uint8), err)
            batch.commit(np.ones(1, np.uint8))
            e, st = po.step(cfg, prob, st, dt)
            en = po.normalise_error(acfg, e, st.mean[0], st.mean[0])
            assert abs(err[0] - en) <= 1e-8 * en
            assert relmax(batch.read_mean(0)[0, 0], st.mean[0]) < 1e-12
        batch.close()


def test_vdp_golden_prefix_through_api(hip_lib, golden, monkeypatch):
    """Reference experiment 1 (vdp.py:52-80): stiff Van der Pol, dense EKF1 nu=4, tol 1e-3. The first
    25 accepted grid points agree with the committed JAX grid to rounding (see test_oracle_golden for
    why the remainder is only statistically comparable)."""
    g = golden["vdp"]
    vf = V.van_der_pol_order2(1e3)
    impl.select("dense", ode_shape=(1,))
    ibm = ivpsolvers.prior_ibm(num_derivatives=4)
    strategy = ivpsolvers.strategy_filter(ibm, ivpsolvers.correction_ts1(ode_order=2))
    solver = ivpsolvers.solver_dynamic(strategy)
    from probdiffeq_hip import taylor

    tcs = taylor.odejet_padded_scan(vf, [np.array([2.0]), np.array([0.0])], num=3)
    init = solver.initial_condition(tcs, 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-3, rtol=1e-3, control=ivpsolve.control_proportional_integral())
    sol = ivpsolve.solve_adaptive_save_every_step(vf, init, t0=0.0, t1=6.3, dt0=0.01, adaptive_solver=asolver)
    grid = g["grid"]
    assert np.max(np.abs(sol.t[:25] - grid[:25])) < 1e-13
    assert np.max(np.abs(sol.u[:25, 0] - g["solution"][:25, 0])) < 1e-12
    assert abs(len(sol.t) - len(grid)) <= 0.03 * len(grid)
    assert sol.t[-1] == 6.3
    # fixed-grid replay of the reference's accepted grid (vdp.py:88-91): pure predict/correct loop
    # over ALL 2 912 points with the deviation profile of the oracle's replay (test_oracle_golden.VDP_REPLAY_PROFILE, one
    # decade of head room for the different rounding path), on the fused one-wavefront kernel and on the generic kernels
    from test_oracle_golden import VDP_REPLAY_PROFILE

    ref = g["solution"][:, 0]
    fixed = ivpsolve.solve_fixed_grid(vf, init, grid=grid, solver=solver)
    assert fixed.u.shape == (len(grid), 1)
    generic = _dense_batch(vf, 4, "ts1", 1)
    generic.set_generic_path(True)
    generic.set_initial(init.tcoeffs, np.zeros(1), init.output_scale)
    u_generic = generic.solve_fixed_grid(grid)[0]
    generic.close()
    for name, u in (("fused", fixed.u[:, 0]), ("generic", u_generic[:, 0])):
        dev = np.abs(u - ref) / np.abs(ref)
        for k, tol in VDP_REPLAY_PROFILE:
            assert np.max(dev[:k]) < tol, (name, k, np.max(dev[:k]), tol)
    # coarse equidistant grid: the reference asserts that NaNs appear (vdp.py:94-97,107-108) -- asserted here as the reference
    # asserts it. (Until round 4 the HIP path stayed finite: the innovation variance F F^T overflowed at |u| ~ 1e54, gain =
    # finite / inf = 0 switched the measurement update off, and the solution crept on to 1e61. In exact arithmetic the solver on
    # this grid leaves the fp64 range, tests/test_oracle_golden.py::test_vdp_coarse_grid_leaves_the_fp64_range_in_exact_arithmetic;
    # an overflowed sum of squares is NaN now, pnode_common.hpp::overflow_to_nan.)
    for generic in (False, True):
        monkeypatch.setattr(_backend, "dense_path", 1 if generic else 0)
        coarse = ivpsolve.solve_fixed_grid(vf, init, grid=g["grid_fixed_inaccurate"], solver=solver)
        assert not np.all(np.isfinite(coarse.u)), generic
        first_bad = int(np.argmax(~np.isfinite(coarse.u[:, 0])))
        assert 380 < first_bad < 1300, (generic, first_bad)  # after the orbit has left the limit cycle, before |u| itself can overflow


def test_dense_brusselator_adaptive_vs_oracle(hip_lib):
    """Dense EKF1 nu=3 on Brusselator N=8 and N=32 (the headline configuration at test size): whole
    adaptive solve against the oracle, identical decisions, means within 1e-8 relative."""
    for N, tol, t1 in ((4, 1e-5, 2.0), (32, 1e-4, 1.0)):
        vf = V.brusselator(N)
        impl.select("dense", ode_shape=(2 * N,))
        solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts1()))
        B = 2
        y0 = np.stack([brusselator_y0(N, 0.0), brusselator_y0(N, 0.7)])
        tcs = tcoeffs(vf, [y0], 3)
        init = solver.initial_condition(tcs, 1.0)
        save_at = np.linspace(0.0, t1, 5)
        asolver = ivpsolve.adaptive(solver, atol=tol, rtol=tol)
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver)
        cfg = po.SolverConfig(nu=3, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
        for b in range(B):
            osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tol, tol), oracle_problem(vf),
                                             po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.01)
            assert np.array_equal(sol.num_steps[b], osol.num_steps), (N, b)
            assert relmax(sol.u[b], osol.u) < 1e-8, (N, b)


def test_bad_configuration_raises(hip_lib):
    with pytest.raises(ValueError):
        _backend.Batch(batch=1, d=3, nu=3, ode_order=1, model="isotropic", correction="ts1", strategy="filter",
                       calibration="dynamic", vf_id=V.VF_RIGID_BODY, vf_params=np.ones(3))
    with pytest.raises(ValueError):
        _backend.Batch(batch=1, d=5, nu=3, ode_order=1, model="isotropic", correction="ts0", strategy="filter",
                       calibration="dynamic", vf_id=V.VF_BRUSSELATOR, vf_params=None)


@pytest.mark.parametrize("method", ["ts0-2", "ts0-4"])
def test_odecheckpts_solve_like_the_reference_test(hip_lib, method):
    """Port of the reference's only test (tests/test_ivpsolvers.py:31-52): the probabilistic solver agrees
    with an independent solver on the logistic ODE within sqrt(tol); here the independent solver is scipy
    (diffrax is not available), and `aux["u0_solve"]` must be present."""
    from odecheckpts import ivps
    from odecheckpts import ivpsolvers as l3

    vf, u0, time_span, args = ivps.logistic()
    dt0, atol, rtol = 0.1, 1e-3, 1e-3
    save_at = np.linspace(*time_span, num=5)
    solve1 = l3.solve(method, vf, u0[0], save_at, dt0=dt0, atol=atol, rtol=rtol)
    solution1, aux1 = solve1(u0, args)
    assert "u0_solve" in aux1 and "solution" in aux1
    truth = scipy_truth(vf.bind(args), u0[0], save_at)
    assert solution1.shape == truth.shape == (5, 1)
    assert np.allclose(solution1, truth, atol=np.sqrt(atol), rtol=np.sqrt(rtol))
    # batched call: three initial values at once, each equal to its own single solve
    u0b = (np.array([[0.1], [0.2], [0.05]]),)
    solb, _ = solve1(u0b, args)
    assert solb.shape == (3, 5, 1)
    assert np.allclose(solb[0], solution1, rtol=1e-12, atol=1e-14)


def test_pleiades_second_order_isotropic(hip_lib, golden):
    """Reference experiment 3 configuration (run_harder.py:42-60): Pleiades, ode_order=2, nu=3, fixedpoint;
    RMSE against the committed truth solution at the final time is at the level the reference reports."""
    from odecheckpts import ivps
    from odecheckpts import ivpsolvers as l3

    vf, u0, (t0, t1) = ivps.pleiades_2nd()
    g = golden["pleiades"]
    save_at = np.linspace(t0, t1, 6)
    tol = 1e-6
    fn = l3.solve("ts0-3", vf, u0[0], save_at, dt0=0.01, atol=1e-3 * tol, rtol=tol, ode_order=2)
    sol, aux = fn(u0, ())
    truth_end = g["ys"][-1][:14]
    assert np.sqrt(np.mean((sol[-1] - truth_end) ** 2)) < 1e-4
    cfg = po.SolverConfig(nu=3, model="isotropic", correction="ts0", strategy="fixedpoint", calibration="dynamic")
    tcs = tcoeffs(vf, list(u0), 3)
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-3 * tol, tol), oracle_problem(vf),
                                     po.initial_state(cfg, tcs, 0.0), save_at, 0.01)
    sm = po.markov_marginals_reverse(cfg, osol)[:, 0]
    assert abs(int(aux["solution"].num_steps[-1]) - int(osol.num_steps[-1])) <= 0.05 * osol.num_steps[-1]
    assert np.max(np.abs(sol - sm)) < 1e-6


def test_device_resident_controller_matches_host_controller(hip_lib, golden):
    """the accept/reject loop + PI law inside the kernel (one wavefront per trajectory) must
    reproduce the host-controlled solve: reference Brusselator configuration (bit-exact step count vs the golden
    vector) and a rigid-body tolerance sweep with per-trajectory save_at grids."""
    from probdiffeq_hip import taylor

    g = golden["brusselator"]
    N = 16
    vf = V.brusselator(N)
    impl.select("isotropic", ode_shape=(2 * N,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts0()))
    y0 = np.stack([brusselator_y0(N, ph) for ph in (0.0, 0.9)])
    init = solver.initial_condition(taylor.odejet_unroll(vf, [y0], num=4), 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-8)
    host = ivpsolve.solve_adaptive_save_at(vf, init, save_at=g["ts"], dt0=0.01, adaptive_solver=asolver)
    dev = ivpsolve.solve_adaptive_save_at(vf, init, save_at=g["ts"], dt0=0.01, adaptive_solver=asolver, device_controller=True)
    i = list(g["N"]).index(N)
    assert int(dev.num_steps[0, -1]) == int(g["num_steps_checkpoint"][i])
    assert np.array_equal(dev.num_steps, host.num_steps)
    assert np.array_equal(dev.num_attempts, host.num_attempts)
    assert np.max(np.abs(dev.u - host.u)) < 1e-11
    assert np.max(np.abs(dev.u[0] - g[f"ys_N{N}"])) < 1e-9
    # filter, tolerance sweep, per-trajectory checkpoints
    vf = V.rigid_body()
    tols = 10.0 ** np.array([-3.0, -4.0, -5.0, -6.0])
    B = len(tols)
    impl.select("isotropic", ode_shape=(3,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts0()))
    u0 = np.tile(np.array([1.0, 0.0, 0.9]), (B, 1))
    init = solver.initial_condition(tcoeffs(vf, [u0], 3), 1.0)
    save_at = np.stack([np.linspace(0.0, 5.0 + b, 7) for b in range(B)])
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tols, rtol=tols)
    host = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver)
    dev = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver, device_controller=True)
    assert np.array_equal(dev.num_steps, host.num_steps)
    assert np.max(np.abs(dev.u - host.u)) < 1e-10
    # and the Python/NumPy driver agrees with the native host driver
    py = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=asolver, python_controller=True)
    assert np.array_equal(py.num_steps, host.num_steps)
    assert np.max(np.abs(py.u - host.u)) < 1e-10


def test_small_dense_fused_kernel_and_device_controller(hip_lib):
    """Dense EKF1 with (nu+1) d <= 32 (BASELINE config 2's shape: stiff Van der Pol, d=1, nu=4, tolerance sweep) runs as
    one fused one-wavefront kernel per attempt. Whole adaptive solves: against the oracle, against the generic kernels,
    and with the controller resident on the device."""
    vf = V.van_der_pol_order2(1e3)
    tols = 10.0 ** np.array([-2.0, -3.0, -4.0, -5.0])
    B = len(tols)
    impl.select("dense", ode_shape=(1,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(4), ivpsolvers.correction_ts1(ode_order=2)))
    u0s = [np.full((B, 1), 2.0), np.zeros((B, 1))]
    tcs = tcoeffs(vf, u0s, 4)
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.linspace(0.0, 1.0, 5)
    asolver = ivpsolve.adaptive(solver, atol=tols, rtol=tols)
    host = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver)
    dev = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=True)
    # identical decisions until the stiff transition near t = 0.8; inside it the accept/reject sequence is chaotic in the
    # last bits (device vs host libm log/exp in the controller), as between the oracle and the committed JAX grid
    assert np.array_equal(dev.num_steps[:, :4], host.num_steps[:, :4])
    assert np.all(np.abs(dev.num_steps[:, -1] - host.num_steps[:, -1]) <= 0.05 * host.num_steps[:, -1])
    assert np.max(np.abs(dev.u[:, :4] - host.u[:, :4])) < 1e-10
    assert np.all(np.abs(dev.u[:, -1, 0] - host.u[:, -1, 0]) < 20 * tols)
    cfg = po.SolverConfig(nu=4, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
    for b in range(B):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tols[b], tols[b]), oracle_problem(vf),
                                         po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.01)
        # (inside the stiff transition single accept / reject decisions flip with the last bits: the oracle's own ensemble under
        # perturbations of one ulp spreads by about 1 %, test_oracle_golden.py; r04's Householder time update sits at 2.3 % for tol 1e-4)
        assert abs(int(host.num_steps[b, -1]) - int(osol.num_steps[-1])) <= max(2, 0.03 * osol.num_steps[-1]), b
        assert relmax(host.u[b], osol.u) < 1e-5 * max(1.0, tols[b] / 1e-5), b
    # non-stiff, d = 3, nu = 2 (D = 9) and d = 8, nu = 3 (D = 32, the largest fused size): identical decisions
    for vf2, nu, u0 in ((V.rigid_body(), 2, np.array([[1.0, 0.0, 0.9]])), (V.brusselator(4), 3, brusselator_y0(4)[None])):
        impl.select("dense", ode_shape=(vf2.d,))
        solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts1()))
        tcs = tcoeffs(vf2, [u0], nu)
        init = solver.initial_condition(tcs, 1.0)
        sa = np.linspace(0.0, 2.0, 4)
        asolver = ivpsolve.adaptive(solver, atol=1e-5, rtol=1e-5)
        host = ivpsolve.solve_adaptive_save_at(vf2, init, save_at=sa, dt0=0.01, adaptive_solver=asolver)
        dev = ivpsolve.solve_adaptive_save_at(vf2, init, save_at=sa, dt0=0.01, adaptive_solver=asolver, device_controller=True)
        cfg = po.SolverConfig(nu=nu, model="dense", correction="ts1", strategy="filter", calibration="dynamic")
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-5, 1e-5), oracle_problem(vf2),
                                         po.initial_state(cfg, [c[0] for c in tcs], 0.0), sa, 0.01)
        assert np.array_equal(host.num_steps[0], osol.num_steps), vf2.d
        assert np.array_equal(dev.num_steps, host.num_steps), vf2.d
        assert relmax(host.u[0], osol.u) < 1e-8 and np.max(np.abs(dev.u - host.u)) < 1e-11, vf2.d


@pytest.mark.parametrize("case", ["brusselator4", "rigid", "vdp2", "brusselator16_large", "brusselator16_generic"])
def test_dense_fixedpoint_smoother_vs_oracle(hip_lib, case, monkeypatch):
    """strategy_fixedpoint with the dense model (BASELINE config 5's smoother, at test size): adaptive solve with
    checkpoints; accepted steps, one-step-smoothed checkpoint values and fully smoothed means against the oracle.
    "brusselator16_large" (d = 32, D = 128) forces the large-D kernels config 5 runs on at D = 1024."""
    if case == "brusselator4":
        vf, nu, corr, u0s, tol, save_at, dt0 = V.brusselator(4), 3, "ts1", [brusselator_y0(4)], 1e-5, np.linspace(0, 1.0, 6), 0.01
    elif case in ("brusselator16_large", "brusselator16_generic"):
        # d = 32, D = 128: the large-D kernels are the default for the smoother; "generic" pins the reference kernels
        monkeypatch.setattr(_backend, "dense_path", 2 if case.endswith("large") else 1)
        vf, nu, corr, u0s, tol, save_at, dt0 = V.brusselator(16), 3, "ts1", [brusselator_y0(16)], 1e-4, np.linspace(0, 0.5, 4), 0.01
    elif case == "rigid":
        vf, nu, corr, u0s, tol, save_at, dt0 = V.rigid_body(), 2, "ts1", [np.array([1.0, 0.0, 0.9])], 1e-4, np.linspace(0, 4.0, 5), 0.1
    else:
        vf, nu, corr, u0s, tol, save_at, dt0 = (V.van_der_pol_order2(5.0), 4, "ts1", [np.array([2.0]), np.array([0.0])], 1e-4,
                                               np.linspace(0, 2.0, 5), 0.01)
    impl.select("dense", ode_shape=(vf.d,))
    correction = ivpsolvers.correction_ts1(ode_order=vf.ode_order)
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(nu), correction))
    tcs = tcoeffs(vf, u0s, nu)
    init = solver.initial_condition(tcs, 1.0)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=dt0, adaptive_solver=ivpsolve.adaptive(solver, atol=tol, rtol=tol))
    cfg = po.SolverConfig(nu=nu, model="dense", correction=corr, strategy="fixedpoint", calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tol, tol), oracle_problem(vf), po.initial_state(cfg, tcs, 0.0), save_at, dt0)
    assert np.array_equal(sol.num_steps, osol.num_steps)
    ck = sol.posterior.batch.read_checkpoint_means()[0]
    assert relmax(ck[:, 0], osol.u[1:]) < 1e-9           # one-step-smoothed interpolants
    sm = po.markov_marginals_reverse(cfg, osol)
    assert relmax(sol.u, sm[:, 0]) < 1e-8                 # backward pass over the stored transitions
    assert relmax(sol.posterior.means[0, :, 1], sm[:, 1]) < 1e-6
    # the stored transitions themselves (probdiffeq's posterior.conditional): pnode_read_checkpoint_transitions, dense layout
    G, xi, Lam = sol.posterior.batch.read_checkpoint_transitions()
    D = (nu + 1) * vf.d
    K = len(save_at) - 1  # store entry c: x(save_at[c]) | x(save_at[c + 1])
    assert G.shape == (1, K, D, D) and xi.shape == (1, K, D)
    for c in range(K):
        oc = osol.states[c + 1].cond
        assert relmax(G[0, c], oc.G) < 1e-6, c
        assert relmax(xi[0, c], oc.xi) < 1e-6, c
        assert relmax(Lam[0, c] @ Lam[0, c].T, oc.Lam @ oc.Lam.T) < 1e-5, c


@pytest.mark.parametrize("model", ["isotropic", "dense"])
def test_uncalibrated_solver_matches_oracle(hip_lib, model):
    """ivpsolvers.solver(strategy) (calibrate="none", ivpsolvers.py:47-48): fixed output scale in the covariance,
    locally calibrated error estimate; adaptive solve against the oracle, two output scales in one batch."""
    vf = V.rigid_body()
    scales = np.array([1.0, 25.0])
    B = 2
    impl.select(model, ode_shape=(3,))
    corr = ivpsolvers.correction_ts0() if model == "isotropic" else ivpsolvers.correction_ts1()
    strat = ivpsolvers.strategy_fixedpoint if model == "isotropic" else ivpsolvers.strategy_filter
    solver = ivpsolvers.solver(strat(ivpsolvers.prior_ibm(3), corr))
    u0 = np.tile(np.array([1.0, 0.0, 0.9]), (B, 1))
    tcs = tcoeffs(vf, [u0], 3)
    init = solver.initial_condition(tcs, output_scale=scales)
    save_at = np.linspace(0.0, 5.0, 6)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.1, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-6, rtol=1e-4))
    assert np.allclose(sol.output_scale, scales)
    cfg = po.SolverConfig(nu=3, model=model, correction="ts0" if model == "isotropic" else "ts1",
                          strategy="fixedpoint" if model == "isotropic" else "filter", calibration="none")
    for b in range(B):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-6, 1e-4), oracle_problem(vf),
                                         po.initial_state(cfg, [c[b] for c in tcs], 0.0, scales[b]), save_at, 0.1)
        ref = po.markov_marginals_reverse(cfg, osol)[:, 0] if model == "isotropic" else osol.u
        assert np.array_equal(sol.num_steps[b], osol.num_steps)
        assert relmax(sol.u[b], ref) < 1e-9
    # the covariance scales with output_scale^2 while the means of the filter do not depend on it
    R = sol.posterior.batch.read_factor(0)
    cov0, cov1 = R[0].T @ R[0], R[1].T @ R[1]
    if model == "dense":
        assert relmax(cov1, cov0 * (scales[1] / scales[0]) ** 2) < 1e-6


@pytest.mark.parametrize("path", [0, 2])
def test_uncalibrated_dense_on_the_mfma_and_large_paths(hip_lib, path, monkeypatch):
    """calibrate="none" with the dense model at Brusselator N=16 (d=32, D=128): the register-resident MFMA kernels
    (default) and the large-D kernels (forced), adaptive solve with the fixed output scale against the oracle."""
    if path:
        monkeypatch.setattr(_backend, "dense_path", path)
    N = 16
    vf = V.brusselator(N)
    impl.select("dense", ode_shape=(2 * N,))
    solver = ivpsolvers.solver(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), ivpsolvers.correction_ts1()))
    scales = np.array([1.0, 9.0])
    y0 = np.stack([brusselator_y0(N, 0.0), brusselator_y0(N, 0.5)])
    tcs = tcoeffs(vf, [y0], 3)
    init = solver.initial_condition(tcs, output_scale=scales)
    save_at = np.linspace(0.0, 0.5, 3)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-5, rtol=1e-5))
    cfg = po.SolverConfig(nu=3, model="dense", correction="ts1", strategy="filter", calibration="none")
    for b in range(2):
        osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-5, 1e-5), oracle_problem(vf),
                                         po.initial_state(cfg, [c[b] for c in tcs], 0.0, scales[b]), save_at, 0.01)
        assert np.array_equal(sol.num_steps[b], osol.num_steps), b
        assert relmax(sol.u[b], osol.u) < 1e-9, b
    assert np.allclose(sol.output_scale, scales)


def test_edge_cases_single_checkpoint_ragged_save_at_and_nan_propagation(hip_lib):
    """B = 1 with one checkpoint; per-trajectory (ragged) save_at grids; a trajectory whose vector field blows up
    must come back as NaN/flagged without disturbing its batch neighbours (the reference propagates NaNs,
    vdp.py:107-108, it never traps them)."""
    vf = V.logistic(1.0, 1.0)
    impl.select("isotropic", ode_shape=(1,))
    solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(2), ivpsolvers.correction_ts0()))
    init = solver.initial_condition(tcoeffs(vf, [np.array([0.1])], 2), 1.0)
    sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=np.array([0.0, 1.0]), dt0=0.1, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-6, rtol=1e-6))
    exact = 0.1 * np.exp(1.0) / (1.0 + 0.1 * (np.exp(1.0) - 1.0))
    assert sol.u.shape == (2, 1) and abs(sol.u[-1, 0] - exact) < 1e-4
    # ragged grids + a diverging member: u' = u (1 - b u) with b < 0 blows up in finite time
    vfb = V.VectorField(V.VF_LOGISTIC, 1, 1, np.array([[1.0, 1.0], [1.0, -5.0], [1.0, 1.0]]), "logistic")
    u0 = np.array([[0.1], [1.0], [0.3]])
    tcs = [u0, u0 * (1 - np.array([[1.0], [-5.0], [1.0]]) * u0)]
    tcs.append(tcs[1] * (1 - 2 * np.array([[1.0], [-5.0], [1.0]]) * u0))
    init = solver.initial_condition(tcs, 1.0)
    save_at = np.stack([np.linspace(0.0, 2.0, 4), np.linspace(0.0, 1.0, 4), np.linspace(0.0, 3.0, 4)])
    sol = ivpsolve.solve_adaptive_save_at(vfb, init, save_at=save_at, dt0=0.05, adaptive_solver=ivpsolve.adaptive(solver, atol=1e-8, rtol=1e-6))
    ex = lambda u, t: u * np.exp(t) / (1.0 + u * (np.exp(t) - 1.0))  # noqa: E731
    assert abs(sol.u[0, -1, 0] - ex(0.1, 2.0)) < 1e-4 and abs(sol.u[2, -1, 0] - ex(0.3, 3.0)) < 1e-4
    # the diverging member (blow-up at t = log(6/5) ~ 0.18) never reaches its later checkpoints
    assert not np.all(np.isfinite(sol.u[1])) or sol.num_steps[1, -1] == 0 or np.max(np.abs(sol.u[1])) > 1e3
    # the same batch with the fixed-point smoother: the member that fails before its first checkpoint has no backward pass to
    # run; it reports its initial value at t0 (not the zero-initialised output buffer) and NaN afterwards
    fp = ivpsolvers.solver_dynamic(ivpsolvers.strategy_fixedpoint(ivpsolvers.prior_ibm(2), ivpsolvers.correction_ts0()))
    sol = ivpsolve.solve_adaptive_save_at(vfb, fp.initial_condition(tcs, 1.0), save_at=save_at, dt0=0.05,
                                          adaptive_solver=ivpsolve.adaptive(fp, atol=1e-8, rtol=1e-6))
    assert abs(sol.u[0, -1, 0] - ex(0.1, 2.0)) < 1e-4 and abs(sol.u[2, -1, 0] - ex(0.3, 3.0)) < 1e-4
    if sol.num_steps[1, 1] == 0:  # (failed inside the first interval, as the blow-up time says it must)
        assert sol.u[1, 0, 0] == 1.0 and np.all(np.isnan(sol.u[1, 1:]))
    else:
        assert not np.all(np.isfinite(sol.u[1])) or np.max(np.abs(sol.u[1])) > 1e3


def test_user_vector_field_escape_hatch(hip_lib):
    """A vector field that is NOT built into the device code (host-evaluated, PNODE_VF_USER): Lotka-Volterra, isotropic
    EKF0 and dense EKF1, against the oracle driven by the same NumPy function and against scipy."""
    from scipy.integrate import solve_ivp

    def lotka(y, *, t, p=(1.5, 1.0, 3.0, 1.0)):
        a, bb, c, dd = p
        u, v = y[..., 0:1], y[..., 1:2]
        from probdiffeq_hip import taylor

        return taylor.concatenate([a * u - bb * u * v, dd * u * v - c * v])

    vf = V.user(lotka, d=2, ode_order=1, name="lotka_volterra")
    u0 = np.array([[1.0, 1.0], [1.5, 0.7]])
    save_at = np.linspace(0.0, 2.0, 5)
    truth = np.stack([solve_ivp(lambda t, y: lotka(y, t=t), (0, 2.0), u, rtol=1e-12, atol=1e-12, t_eval=save_at).y.T for u in u0])
    for model, corr, tol in (("isotropic", "ts0", 1e-6), ("dense", "ts1", 1e-6)):
        impl.select(model, ode_shape=(2,))
        correction = ivpsolvers.correction_ts0() if corr == "ts0" else ivpsolvers.correction_ts1()
        solver = ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3), correction))
        tcs = tcoeffs(vf, [u0], 3)
        init = solver.initial_condition(tcs, 1.0)
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.05, adaptive_solver=ivpsolve.adaptive(solver, atol=tol, rtol=tol))
        assert np.max(np.abs(sol.u - truth)) < 1e-4
        cfg = po.SolverConfig(nu=3, model=model, correction=corr, strategy="filter", calibration="dynamic")

        def f1(y, t):
            return lotka(y[0], t=t)

        def j1(y, t):
            J = np.zeros((2, 1, 2))
            h = 1e-30
            for c in range(2):
                yc = np.array(y[0], dtype=np.complex128)
                yc[c] += 1j * h
                J[:, 0, c] = np.imag(lotka(yc, t=t)) / h
            return J

        prob = po.Problem(f1, j1, 1, 2)
        for b in range(2):
            osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(tol, tol), prob, po.initial_state(cfg, [c[b] for c in tcs], 0.0), save_at, 0.05)
            # f is evaluated at the host's copy of the extrapolated mean (1 ulp away from the kernel's): allow a
            # single accept/reject decision to differ
            assert np.max(np.abs(sol.num_steps[b] - osol.num_steps)) <= 2
            assert relmax(sol.u[b], osol.u) < 1e-6


@pytest.mark.parametrize("case", ["small_rigid", "small_vdp2", "mfma16", "mfma32", "twolq16", "twolq32", "large16", "large48"])
def test_differential_fuzz_fast_paths_against_generic_kernels(hip_lib, case):
    """Seeded differential run of every fast dense path against the generic kernels: 25 attempts with random step sizes,
    random activity masks and random accept/reject decisions (slots drift apart, rejected states are re-used). The two
    handles must agree on every error estimate, mean and covariance throughout."""
    rng = np.random.default_rng(20260104)
    B = 6
    if case == "small_rigid":
        vf, nu, u0s, dt_scale = V.rigid_body(), 3, [np.array([1.0, 0.0, 0.9]) + 0.05 * rng.standard_normal((B, 3))], 0.05
    elif case == "small_vdp2":
        vf, nu, u0s, dt_scale = (V.van_der_pol_order2(20.0), 4, [2.0 + 0.1 * rng.standard_normal((B, 1)), 0.1 * rng.standard_normal((B, 1))],
                                 0.005)
    else:
        N = int(case[-2:])
        vf, nu, dt_scale = V.brusselator(N), 3, 0.01
        u0s = [np.stack([brusselator_y0(N, ph) for ph in rng.uniform(0, 2, B)])]
    tc = np.stack(tcoeffs(vf, u0s, nu), axis=1)
    fast = _dense_batch(vf, nu, "ts1", B)
    if case.startswith("large"):
        fast.set_generic_path(2)
    if case.startswith("twolq"):
        fast.set_generic_path(3)  # MFMA kernels with the two-launch LQ
    slow = _dense_batch(vf, nu, "ts1", B)
    slow.set_generic_path(1)
    for bt in (fast, slow):
        bt.set_initial(tc, np.zeros(B), np.ones(B))
        bt.set_tolerances(1e-5, 1e-5)
    ef, es = np.zeros(B), np.zeros(B)
    for it in range(25):
        dts = dt_scale * rng.uniform(0.3, 2.0, B)
        act = (rng.uniform(size=B) < 0.8).astype(np.uint8)
        if it == 0:
            act[:] = 1
        acc = (act.astype(bool) & (rng.uniform(size=B) < 0.7)).astype(np.uint8)
        fast.attempt_step(dts, act, ef)
        slow.attempt_step(dts, act, es)
        m = act.astype(bool)
        assert np.all(np.abs(ef[m] - es[m]) <= 1e-6 * np.abs(es[m])), (it, ef, es)
        fast.commit(acc)
        slow.commit(acc)
        mf, ms = fast.read_mean(0), slow.read_mean(0)
        Rf, Rs = fast.read_factor(0), slow.read_factor(0)
        for b in range(B):
            assert relmax(mf[b, 0], ms[b, 0]) < 1e-10, (it, b)
            assert relmax(mf[b, 1], ms[b, 1]) < 1e-8, (it, b)
            cf, cs = Rf[b].T @ Rf[b], Rs[b].T @ Rs[b]
            assert np.max(np.abs(cf - cs)) <= 1e-6 * max(np.max(np.abs(cs)), 1e-300), (it, b)
    fast.close()
    slow.close()
