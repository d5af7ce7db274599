import numpy as np, time, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT','/root/repo'))
from sleqp_amd import synth, _lib
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'tests'))
from plan_emul import Plan
lib=_lib.load()
J = synth.banded_jacobian(100000, 50000, 20, 200, 0)
N, cp, ri, vals = synth.kkt_lower_from_jacobian(J)[:4]
for i in range(3):
    t=time.perf_counter(); P=Plan(lib,N,cp,ri,vals); print("plan", round(time.perf_counter()-t,4), P.t_order, P.t_symbolic, P.t_total, flush=True)
