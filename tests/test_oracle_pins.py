"""Pins the CPU oracle (oracle/kkt_oracle.c) against the known answers of the
reference's own tests and against the real LAPACK inside scipy.

Reference tests cited relative to chrhansk/sleqp v1.0.2 src/test/.
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

import oracle
from sleqp_amd import synth

TOL = 1e-8  # tolerance the reference tests use (e.g. constrained_newton_test.c:257)


def test_spmv_known_answer():
    # sparse/sleqp_sparse_matrix_test.c:12-56: [[1,0,2],[0,2,3]] * (2,4,3) = (8,17)
    cols, rows, data = [0, 1, 2, 4], [0, 1, 0, 1], [1.0, 2.0, 2.0, 3.0]
    y = oracle.mat_mult_vec(2, 3, cols, rows, data, [0, 1, 2], [2.0, 4.0, 3.0])
    assert abs(y[0] - 8.0) < TOL and abs(y[1] - 17.0) < TOL
    # transposed product with the same data: M^T (1, 1) = (1, 2, 5)
    idx, val = oracle.mat_mult_vec_trans(2, 3, cols, rows, data, [0, 1], [1.0, 1.0], 0.0)
    assert idx.tolist() == [0, 1, 2] and np.allclose(val, [1.0, 2.0, 5.0])
    # eps filter (mat.c:357: pushed only if !is_zero(sum, eps))
    idx, val = oracle.mat_mult_vec_trans(2, 3, cols, rows, data, [0, 1], [1.0, 1.0], 1.5)
    assert idx.tolist() == [1, 2]


def test_vec_marshal():
    raw = oracle.vec_to_raw(5, [1, 3], [2.0, -4.0])
    assert raw.tolist() == [0.0, 2.0, 0.0, -4.0, 0.0]
    idx, val = oracle.vec_set_from_raw([0.0, 1e-21, -3.0, 1e-19], 1e-20)
    assert idx.tolist() == [2, 3] and val.tolist() == [-3.0, 1e-19]


def _newton_fixture():
    # constrained_newton_test.c:48-202: min x0^2 + x1^2, c = x1 >= 2 active at x = (1, 2)
    n, m = 2, 1
    jp, ji, jx = [0, 0, 1], [0], [1.0]  # cons_jac: entry (0, 1) = 1 (:108)
    var_index, cons_index = [-1, -1], [0]
    return n, m, jp, ji, jx, var_index, cons_index


def test_fill_aug_jac_newton_fixture():
    n, m, jp, ji, jx, vi, ci = _newton_fixture()
    N, cols, rows, data = oracle.fill_aug_jac(n, m, jp, ji, jx, vi, ci, lower_only=True)
    assert N == 3
    assert cols.tolist() == [0, 1, 3, 3]  # one empty trailing column (standard_aug_jac.c:225-231)
    assert rows.tolist() == [0, 1, 2] and data.tolist() == [1.0, 1.0, 1.0]
    # full variant through add_upper (standard_aug_jac.c:34-104)
    N, cols, rows, data = oracle.fill_aug_jac(n, m, jp, ji, jx, vi, ci, lower_only=False)
    assert cols.tolist() == [0, 1, 3, 4] and rows.tolist() == [0, 1, 2, 1]


def test_constrained_newton_step_known_answer():
    # constrained_newton_test.c:204-275: Newton step = (-1, 0), tol 1e-8
    n, m, jp, ji, jx, vi, ci = _newton_fixture()
    N, cols, rows, data = oracle.fill_aug_jac(n, m, jp, ji, jx, vi, ci)
    f = oracle.OracleFact(N, cols, rows, data)
    grad = np.array([2.0, 4.0])  # obj_grad at (1, 2), :83-99
    hc, hr, hx = [0, 1, 2], [0, 1], [2.0, 2.0]  # hess_prod = 2 * direction, :118-129
    step, its = f.steihaug(n, hc, hr, hx, grad, trust_radius=10.0)
    assert np.allclose(step, [-1.0, 0.0], atol=TOL)
    # the projection used inside: P (2, 4) = (2, 0)
    idx, val = f.project_nullspace(n, [0, 1], grad)
    assert idx.tolist() == [0] and abs(val[0] - 2.0) < TOL


def test_unconstrained_newton_steps_known_answer():
    # unconstrained_newton_test.c:67-205: quadfunc at x = (1, 2), empty working set
    n = 2
    N, cols, rows, data = oracle.fill_aug_jac(n, 0, [0, 0, 0], [], [], [-1, -1], [])
    assert N == 2 and cols.tolist() == [0, 1, 2]
    f = oracle.OracleFact(N, cols, rows, data)
    grad = np.array([2.0, 4.0])
    hc, hr, hx = [0, 1, 2], [0, 1], [2.0, 2.0]
    step, _ = f.steihaug(n, hc, hr, hx, grad, trust_radius=10.0)
    assert np.allclose(step, [-1.0, -2.0], atol=TOL)  # :82-83,121
    step, _ = f.steihaug(n, hc, hr, hx, grad, trust_radius=1.0)
    assert np.allclose(step, [-0.44721359549995793, -0.89442719099991586], atol=TOL)  # :154-155


def test_dual_estimation_known_answer():
    # dual_estimation_test.c:15-103 with quadfunc_fixture.c:113-131: x = (1, 2) sits on both
    # lower bounds, obj_grad = (2, 4); solve_lsq(-grad) (dual_estimation_lsq.c:41-45) gives
    # vars_dual = (-2, -4), tol 1e-8
    n = 2
    N, cols, rows, data = oracle.fill_aug_jac(n, 0, [0, 0, 0], [], [], [0, 1], [])
    assert N == 4 and cols.tolist() == [0, 2, 4, 4, 4] and rows.tolist() == [0, 2, 1, 3]
    f = oracle.OracleFact(N, cols, rows, data)
    idx, val = f.solve_lsq(n, [0, 1], [-2.0, -4.0])
    assert idx.tolist() == [0, 1] and np.allclose(val, [-2.0, -4.0], atol=TOL)


def test_survey_reference_run():
    # SURVEY.md §8c "verified link+run" of the reference's fact_lapack.c in this container:
    # n = 2, A = [1 2], projection of (3, 1) = (2, -1) exactly
    N, cols, rows, data = oracle.fill_aug_jac(2, 1, [0, 1, 2], [0, 0], [1.0, 2.0], [-1, -1], [0])
    f = oracle.OracleFact(N, cols, rows, data)
    idx, val = f.project_nullspace(2, [0, 1], [3.0, 1.0])
    assert idx.tolist() == [0, 1] and np.allclose(val, [2.0, -1.0], atol=1e-14)


def _hs71_jac(x):
    # constrained_fixture.c:91-120
    jx = np.array([x[1] * x[2] * x[3], 2 * x[0], x[0] * x[2] * x[3], 2 * x[1], x[0] * x[1] * x[3], 2 * x[2],
                   x[0] * x[1] * x[2], 2 * x[3]])
    return [0, 2, 4, 6, 8], [0, 1, 0, 1, 0, 1, 0, 1], jx


def test_hs71_stationarity_at_reference_optimum():
    # constrained_fixture.c:270-273 optimum (tol 1e-6 in constrained_test.c:94-100); working set at
    # the optimum: x0 at its lower bound, both constraints active.  Stationarity
    # (constrained_test.c:41-82) <=> the null-space projection of grad f vanishes.
    x = np.array([1.0, 4.742999, 3.821151, 1.379408])
    jp, ji, jx = _hs71_jac(x)
    N, cols, rows, data = oracle.fill_aug_jac(4, 2, jp, ji, jx, [0, -1, -1, -1], [1, 2])
    assert N == 7
    f = oracle.OracleFact(N, cols, rows, data)
    s = x[0] + x[1] + x[2]
    grad = np.array([s * x[3] + x[0] * x[3], x[0] * x[3], x[0] * x[3] + 1, s * x[0]])  # :60-72
    idx, val = f.project_nullspace(4, [0, 1, 2, 3], grad, zero_eps=0.0)
    proj = oracle.vec_to_raw(4, idx, val)
    assert np.abs(proj).max() < 1e-4
    # the LSQ multipliers (dual_estimation_lsq.c:41-45) close the stationarity equation
    idx, val = f.solve_lsq(4, [0, 1, 2, 3], -grad, zero_eps=0.0)
    duals = oracle.vec_to_raw(3, idx, val)
    A_W = np.vstack([[1.0, 0, 0, 0], np.array(jx)[0::2], np.array(jx)[1::2]])
    assert np.abs(grad + A_W.T @ duals).max() < 1e-4


def test_lu_against_real_lapack():
    # the reference calls dgetrf_/dgetrs_ (fact_lapack.c:108-146); scipy ships the real LAPACK
    rng = np.random.default_rng(3)
    for n, m, kind in [(30, 12, "u"), (120, 60, "b"), (400, 200, "u")]:
        J = synth.banded_jacobian(n, m, 8, 60, 5) if kind == "b" else synth.uniform_jacobian(n, m, 4, 5)
        vi, ci, W = synth.working_set_all_rows(n, m, 0.1, 5)
        N, cols, rows, data = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        K = synth.kkt_full_matrix(N, cols, rows, data).toarray()
        b = rng.standard_normal(N)
        f = oracle.OracleFact(N, cols, rows, data)
        f.solve_dense(b)
        x_oracle = f.raw_solution()
        lu, piv = sla.lu_factor(K)  # LAPACK dgetrf
        x_lapack = sla.lu_solve((lu, piv), b)  # LAPACK dgetrs
        assert np.abs(x_oracle - x_lapack).max() <= 1e-11 * max(1.0, np.abs(x_lapack).max())
        assert np.abs(K @ x_oracle - b).max() < 1e-10


def test_fill_aug_jac_against_vectorised_generator():
    for seed in range(3):
        J = synth.uniform_jacobian(50, 30, 4, seed)
        rng = np.random.default_rng(seed)
        # ragged working set: some constraints inactive, some bounds active
        ci = np.full(30, -1, dtype=np.int32)
        vi = np.full(50, -1, dtype=np.int32)
        av = np.sort(rng.choice(50, 7, replace=False))
        vi[av] = np.arange(7)
        ac = np.sort(rng.choice(30, 12, replace=False))
        ci[ac] = 7 + np.arange(12)
        N, cols, rows, data = oracle.fill_aug_jac(50, 30, J.indptr, J.indices, J.data, vi, ci)
        N2, c2, r2, d2 = synth.kkt_lower_from_jacobian(J, vi, ci)
        assert N == N2 and np.array_equal(cols, c2) and np.array_equal(rows, r2) and np.array_equal(data, d2)


def test_sparse_ldl_baseline_matches_lu():
    J = synth.banded_jacobian(300, 150, 10, 80, 2)
    N, cols, rows, data = synth.kkt_lower_from_jacobian(J)
    b = np.random.default_rng(0).standard_normal(N)
    f = oracle.OracleFact(N, cols, rows, data)
    f.solve_dense(b)
    ldl = oracle.OracleLdl(N, cols, rows, data)  # natural order: all x before all y
    x = ldl.solve(b)
    assert np.abs(x - f.raw_solution()).max() < 1e-10
    # with a permutation of the y block
    perm = np.concatenate([np.arange(300), 300 + np.random.default_rng(1).permutation(150)]).astype(np.int32)
    x2 = oracle.OracleLdl(N, cols, rows, data, perm).solve(b)
    assert np.abs(x2 - f.raw_solution()).max() < 1e-10


def test_hess_prod_lower():
    H = sp.random(20, 20, density=0.2, random_state=1)
    H = (H + H.T).tocsc()
    L = sp.tril(H, format="csc")
    L.sort_indices()
    d = np.random.default_rng(0).standard_normal(20)
    y = oracle.hess_prod_lower(20, L.indptr, L.indices, L.data, d)
    assert np.allclose(y, H @ d)


def test_sparse_ldl_numeric_refactor_matches_full_factor():
    """oracle_ldl_refactor (symbolic reused, the `numeric_only` CPU baseline figure) gives the factor of a fresh
    oracle_ldl_factor on the new values: same solve, bit for bit."""
    import scipy.sparse as sp

    from sleqp_amd import synth

    J = synth.banded_jacobian(300, 140, 8, 60, 3)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(N)
    F = oracle.OracleLdl(N, cp, ri, vx)
    vx2 = vx.copy()
    off = ri != np.repeat(np.arange(N), np.diff(cp))
    vx2[off] *= 1.0 + 0.3 * rng.standard_normal(int(off.sum()))
    F.refactor(vx2)
    G = oracle.OracleLdl(N, cp, ri, vx2)
    assert np.array_equal(F.solve(b), G.solve(b))
    K = synth.kkt_full_matrix(N, cp, ri, vx2)
    assert np.abs(K @ F.solve(b) - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
