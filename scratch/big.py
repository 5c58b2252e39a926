import sys, time
sys.path[:0]=['.','code-adaptive-prob-ode-solvers_amd','tests','oracle']
import numpy as np, pn_oracle as po
from helpers import *
from probdiffeq_hip import _backend
for N,nu in [(128,3),(64,4)]:
    vf=V.brusselator(N); B=1
    tc=tcoeffs(vf,[brusselator_y0(N)[None]],nu)
    batch=_backend.Batch(batch=B,d=vf.d,nu=nu,ode_order=1,model='dense',correction='ts1',strategy='filter',calibration='dynamic',vf_id=vf.vf_id,vf_params=vf.param_array(),num_checkpoints=1)
    batch.set_initial(np.stack(tc,axis=1),np.zeros(B),np.ones(B)); batch.set_tolerances(1e-5,1e-5)
    cfg=po.SolverConfig(nu=nu,model='dense',correction='ts1',strategy='filter',calibration='dynamic')
    st=po.initial_state(cfg,[c[0] for c in tc],0.0); prob=oracle_problem(vf); err=np.zeros(B)
    for dt in (0.005,0.01):
        t0=time.time(); batch.attempt_step(np.full(B,dt),np.ones(B,np.uint8),err); batch.commit(np.ones(B,np.uint8)); tg=time.time()-t0
        e,st=po.step(cfg,prob,st,dt); en=po.normalise_error(po.AdaptiveConfig(1e-5,1e-5),e,st.mean[0],st.mean[0])
        m=batch.read_mean(0)[0]
        print(N,nu,'D',(nu+1)*2*N,'gpu step s',round(tg,3),'err rel dev',abs(err[0]-en)/en,'u dev',np.max(np.abs(m[0]-st.mean[0]))/np.max(np.abs(st.mean[0])))
