"""Where the vtable solve goes: hipfact_solve_sparse (dense-as-sparse rhs) and hipfact_solution(0, n) timed apart."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat, SleqpVec
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
n = J.shape[1]
f = HipFact(device=0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
rhs = SleqpVec.from_raw(b)
for _ in range(5):
    f.solve(rhs); f.solution_raw(0, n)
ts, tq = [], []
for _ in range(50):
    t0 = time.perf_counter(); f.solve(rhs); t1 = time.perf_counter(); f.solution_raw(0, n); t2 = time.perf_counter()
    ts.append(t1 - t0); tq.append(t2 - t1)
print(f"solve_sparse (nnz {rhs.nnz}): median {1e3*np.median(ts):.3f} ms   solution(0, n): median {1e3*np.median(tq):.3f} ms")
bd = np.ascontiguousarray(b)
ts, tq = [], []
for _ in range(50):
    t0 = time.perf_counter(); f.solve(bd); t1 = time.perf_counter(); f.solution_raw(0, n); t2 = time.perf_counter()
    ts.append(t1 - t0); tq.append(t2 - t1)
print(f"solve_dense: median {1e3*np.median(ts):.3f} ms   solution(0, n): median {1e3*np.median(tq):.3f} ms")
