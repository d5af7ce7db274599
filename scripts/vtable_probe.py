"""Changed working set through the plain hipfact_set_matrix (row dictionary): time per call against the number of host threads."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from bench import make_problem
from sleqp_amd import synth
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
m, n = J.shape
rng = np.random.default_rng(5)
mats = []
for it in range(8):
    ci = np.full(m, -1, dtype=np.int32)
    keep = np.ones(m, dtype=bool)
    if it: keep[rng.choice(m, m // 100, replace=False)] = False
    vi = np.full(n, -1, dtype=np.int32)
    ci[keep] = np.arange(int(keep.sum()))
    mats.append(oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci))
f = HipFact(device=0)
for (NN, kc, kr, kd) in mats[:2]:
    f.set_matrix(SleqpMat(NN, NN, kc, kr, kd))
ts = []
for rep in range(3):
    for (NN, kc, kr, kd) in mats:
        t0 = time.perf_counter()
        f.set_matrix(SleqpMat(NN, NN, kc, kr, kd))
        ts.append(time.perf_counter() - t0)
print(f"threads {os.environ.get('HIPFACT_VTABLE_THREADS', '8')}: median {1e3*np.median(ts):.2f} ms  min {1e3*min(ts):.2f}  analyses {f.info('analyses')}")
