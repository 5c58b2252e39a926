import sys, os, ctypes as C
sys.path[:0]=['.','code-adaptive-prob-ode-solvers_amd','tests','oracle']
os.environ['PNODE_ABLATE']='128'
import numpy as np
import bench
from probdiffeq_hip import _backend, ivpsolve, ivpsolvers
from probdiffeq_hip.impl import impl
for B in (256, 1024):
    vf,tcs=bench.problem_batch(B)
    impl.select('dense',ode_shape=(64,))
    solver=ivpsolvers.solver_dynamic(ivpsolvers.strategy_filter(ivpsolvers.prior_ibm(3),ivpsolvers.correction_ts1()))
    init=solver.initial_condition(tcs,1.0)
    batch=ivpsolve._make_batch(vf,init,solver,1,device=0)
    batch.set_initial(init.tcoeffs,np.zeros(B),init.output_scale); batch.set_tolerances(1e-6,1e-6)
    err=np.zeros(B)
    lib=batch._lib
    lib.pnode_debug_read_scratch.argtypes=[C.c_void_p,C.c_int32,C.c_int32,C.c_int32,C.POINTER(C.c_uint64)]
    prev=None
    for i in range(14):
        batch.attempt_step(np.full(B,0.01),np.ones(B,np.uint8),err); batch.commit(np.ones(B,np.uint8))
        out=(C.c_uint64*8)()
        lib.pnode_debug_read_scratch(batch._h,3,600,8,out)
        cur=np.array(list(out)[:5],dtype=np.float64)
        if i>=12: print('B',B,'attempt',i,'sweep phase cycles (one WG, per launch set: read+fma, reduce, scalars, update+publish, barrier):', cur.astype(int), 'sum', int(cur.sum()))
