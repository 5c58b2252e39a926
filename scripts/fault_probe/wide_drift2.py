import os, sys, warnings
import numpy as np
warnings.simplefilter("ignore")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("code-adaptive-prob-ode-solvers_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
from helpers import brusselator_y0, oracle_problem
import pn_oracle as po
from probdiffeq_hip import ivpsolve, ivpsolvers, taylor
from probdiffeq_hip import vector_fields as V
from probdiffeq_hip.impl import impl
np.set_printoptions(linewidth=220, precision=3)
N, nu = int(sys.argv[1]) if len(sys.argv) > 1 else 69, 2
vf = V.brusselator(N)
impl.select("isotropic", ode_shape=(vf.d,))
tol = np.array([1e-5])
y0 = np.stack([brusselator_y0(N, 0.7)])
save_at = np.linspace(0.0, 0.12, 4)
tcs = taylor.odejet_padded_scan(vf, [y0], num=nu)
for strategy in ("filter", "fixedpoint"):
    strat = {"filter": ivpsolvers.strategy_filter, "fixedpoint": ivpsolvers.strategy_fixedpoint}[strategy]
    solver = ivpsolvers.solver_dynamic(strat(ivpsolvers.prior_ibm(nu), ivpsolvers.correction_ts0()))
    init = solver.initial_condition(tcs, 1.0)
    asolver = ivpsolve.adaptive(solver, atol=1e-3 * tol, rtol=tol)
    cfg = po.SolverConfig(nu=nu, model="isotropic", correction="ts0", strategy=strategy, calibration="dynamic")
    osol = po.solve_adaptive_save_at(cfg, po.AdaptiveConfig(1e-3 * tol[0], tol[0]), oracle_problem(vf), po.initial_state(cfg, [c[0] for c in tcs], 0.0), save_at, 0.01)
    for dc in (False, True):
        sol = ivpsolve.solve_adaptive_save_at(vf, init, save_at=save_at, dt0=0.01, adaptive_solver=asolver, device_controller=dc)
        ck = sol.posterior.batch.read_checkpoint_means()[0]            # (K, n, d): stored (one-step-smoothed / filter) means at the checkpoints
        om = np.stack([osol.states[i].mean for i in range(1, len(save_at))])
        diff = np.abs(ck - om)
        print(strategy, "dc", dc, "steps", sol.num_steps[0], list(osol.num_steps))
        print("   stored checkpoint means vs oracle, max over dims, rows (u, u', u'') x checkpoints:\n", np.max(diff, axis=2).T)
        c = len(save_at) - 2
        j = int(np.argmax(diff[c, 0]))
        print("   last checkpoint: worst dim", j, "hip", ck[c, :, j], "oracle", om[c, :, j], " #dims with |diff u| > 1e-10:", int(np.sum(diff[c, 0] > 1e-10)), np.nonzero(diff[c, 0] > 1e-10)[0][:20])
