"""Golden vectors (tests/golden/kkt_cases.npz, made by tests/golden/make_golden.py).

CPU: the oracle reproduces its committed outputs (guards the checker itself).
GPU: the HIP path, driven through the C ABI, reproduces them: K assembly
bit-exact (integer/byte work), solves within REL_TOL (fp64).
"""
import numpy as np
import pytest

import oracle
from util import REL_TOL, ZERO_EPS, golden_cases, rel_err

CASES = golden_cases()
IDS = [c.name for c in CASES]


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_oracle_reproduces_golden(c):
    N, kc, kr, kd = oracle.fill_aug_jac(c.n, c.m, c.jp, c.ji, c.jx, c.var_index, c.cons_index, lower_only=True)
    assert N == c.N and np.array_equal(kc, c.K_cols) and np.array_equal(kr, c.K_rows) and np.array_equal(kd, c.K_data)
    _, fc, fr, fd = oracle.fill_aug_jac(c.n, c.m, c.jp, c.ji, c.jx, c.var_index, c.cons_index, lower_only=False)
    assert np.array_equal(fc, c.Kfull_cols) and np.array_equal(fr, c.Kfull_rows) and np.array_equal(fd, c.Kfull_data)
    f = oracle.OracleFact(N, kc, kr, kd)
    f.solve_dense(c.rhs_dense)
    assert np.array_equal(f.raw_solution(), c.sol_dense)
    i, d = f.project_nullspace(c.n, c.g_idx, c.g_dat, ZERO_EPS)
    assert np.array_equal(i, c.proj_idx) and np.array_equal(d, c.proj_dat)
    i, d = f.solve_lsq(c.n, c.g_idx, c.g_dat, ZERO_EPS)
    assert np.array_equal(i, c.lsq_idx) and np.array_equal(d, c.lsq_dat)
    i, d = f.solve_min_norm(c.n, c.b_idx, c.b_dat, ZERO_EPS)
    assert np.array_equal(i, c.mn_idx) and np.array_equal(d, c.mn_dat)
    assert np.array_equal(oracle.mat_mult_vec(c.m, c.n, c.jp, c.ji, c.jx, c.x_idx, c.x_dat), c.Jx)


def _dense(dim, idx, dat):
    out = np.zeros(dim)
    out[idx] = dat
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
@pytest.mark.parametrize("device_assembly", [True, False], ids=["asm_device", "asm_host"])
def test_device_reproduces_golden(c, device_assembly):
    from sleqp_amd.fact import HipFact, SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    fact = HipFact()
    aug = StandardAugJac(c.n, fact, zero_eps=ZERO_EPS, device_assembly=device_assembly)
    J = SleqpMat(c.m, c.n, c.jp, c.ji, c.jx)
    aug.set_iterate(J, c.var_index, c.cons_index)
    K = aug.K
    # fill_aug_jac parity: bit-exact CSC arrays
    assert K.num_cols == c.N
    assert np.array_equal(K.cols, c.K_cols) and np.array_equal(K.rows, c.K_rows) and np.array_equal(K.data, c.K_data)
    W = c.N - c.n
    # dense solve
    fact.solve(c.rhs_dense)
    assert rel_err(fact.solution_raw(0, c.N), c.sol_dense) <= REL_TOL
    # the three AugJac flavours
    g = SleqpVec(c.n, c.g_idx, c.g_dat)
    proj = aug.project_nullspace(g)
    assert rel_err(proj.to_raw(), _dense(c.n, c.proj_idx, c.proj_dat)) <= REL_TOL
    lsq = aug.solve_lsq(g)
    assert rel_err(lsq.to_raw(), _dense(W, c.lsq_idx, c.lsq_dat)) <= REL_TOL
    mn = aug.solve_min_norm(SleqpVec(W, c.b_idx, c.b_dat))
    assert rel_err(mn.to_raw(), _dense(c.n, c.mn_idx, c.mn_dat)) <= REL_TOL
    # SpMV
    if c.m > 0:
        S = SpMat(fact, J)
        assert rel_err(S.mult_vec(SleqpVec(c.n, c.x_idx, c.x_dat)), c.Jx) <= 1e-14
        jt = S.mult_vec_trans(SleqpVec(c.m, c.y_idx, c.y_dat), eps=1e-10)
        assert rel_err(jt.to_raw(), _dense(c.n, c.JTy_idx, c.JTy_dat)) <= 1e-14
        S.free()
    fact.free()
