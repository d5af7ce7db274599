import os, sys
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from sleqp_amd import synth
from sleqp_amd.fact import HipFact, SpMat, StandardAugJac
from sleqp_amd.sparse import SleqpMat
from test_gpu_parity import _ws
from util import rel_err
n, m = 1500, 700
J, dcols = synth.with_dense_columns(synth.banded_jacobian(n, m, 10, 80, 17), 3, 5)
rng = np.random.default_rng(9)
vi, ci, W = _ws(n, m, rng, 1.0, 0.0)
B = sp.random(n, n, density=3.0 / n, random_state=3)
HL = sp.tril(B @ B.T + 0.5 * sp.eye(n), format="csc"); HL.sort_indices()
g = rng.standard_normal(n)
N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
want, its_ref = oracle.OracleFact(N, kc, kr, kd).steihaug(n, HL.indptr, HL.indices, HL.data, g, trust_radius=30.0, stat_tol=1e-6)
for opts in ({}, {"xupd_fused": 0}, {"refine_check_every": 1}, {"dense_mode": 2}, {"dense_mode": 0}, {"rhs_fused": 0}):
    f = HipFact(device=0)
    for k, v in opts.items(): f.set_option(k, v)
    aug = StandardAugJac(n, f); aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    H = SpMat(f, SleqpMat.from_scipy(HL))
    for method in (0, 1):
        step, dual, its = f.tr_solve(H, g, 30.0, method=method, stat_tol=1e-6, max_iter=200)
        print(opts, "method", method, "its", its, "ref", its_ref, "err", rel_err(step, want), "norm", np.linalg.norm(step), "late", f.info("late_columns"), "AWs", np.abs(J @ step).max(), flush=True)
    H.free(); f.free()
