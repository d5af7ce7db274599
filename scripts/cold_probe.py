"""Cold hipfact_set_matrix (analysis + upload of the plan + first factorisation) with the phase ticks of HIPFACT_TIMING."""
import os, sys, time
os.environ["HIPFACT_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
w = HipFact(device=0); w.set_matrix(SleqpMat(N, N, cp, ri, vx)); w.solve(b); del w   # device / code objects warm
for sv in (1, 0):
    f = HipFact(device=0)
    f.set_option("superset_vtable", sv)
    t0 = time.perf_counter()
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    t1 = time.perf_counter()
    f.solve(b); f.solution_raw(0, 1)
    t2 = time.perf_counter()
    print(f"superset_vtable={sv}: cold set_matrix {1e3*(t1-t0):.1f} ms (analysis_s {1e3*f.info('analysis_s'):.1f}), first solve {1e3*(t2-t1):.1f} ms", file=sys.stderr)
    del f
