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
