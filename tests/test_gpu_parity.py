"""GPU parity tests: the HIP path, called through the C ABI, against the CPU
oracle on the same seeded inputs (sizes the oracle finishes in seconds), the
reference tests' known answers, edge cases, and size-independent properties at
BASELINE.json's full sizes.

Tolerances (BASELINE.md "Parity"): solutions within 1e-9 relative of the
LAPACK-restating oracle; scaled residual <= 1e-12 after refinement; integer /
byte work (K assembly) bit-exact.
"""
import numpy as np
import pytest
import scipy.sparse as sp

import oracle
from sleqp_amd import synth
from util import REL_TOL, RESID_TOL, ZERO_EPS, rel_err, scaled_residual

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fact():
    from sleqp_amd.fact import HipFact

    f = HipFact(device=0)
    yield f
    f.free()


def _problem(n, m, kind, frac, seed):
    J = synth.banded_jacobian(n, m, min(12, n), min(80, n), seed) if kind == "b" else synth.uniform_jacobian(n, m, min(4, n), seed)
    vi, ci, W = synth.working_set_all_rows(n, m, frac, seed)
    return J, vi, ci, W


@pytest.mark.parametrize("n,m,kind,frac", [(2, 1, "u", 0.0), (7, 3, "u", 0.3), (64, 64, "u", 0.0), (300, 150, "b", 0.1),
                                            (1000, 500, "u", 0.0), (1500, 700, "b", 0.05)])
@pytest.mark.parametrize("refine", [0, 1])
def test_factor_solve_vs_oracle(fact, n, m, kind, frac, refine):
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    J, vi, ci, W = _problem(n, m, kind, frac, 3)
    if n == m:  # square working set: add a dominant diagonal so that A_W is well conditioned
        J = sp.csc_matrix(J + 4 * sp.eye(n))
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    ref = oracle.OracleFact(N, kc, kr, kd)
    fact.set_option("refine_steps", refine)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("saddle") == 1.0
    rng = np.random.default_rng(5)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    # dense rhs
    b = rng.standard_normal(N)
    ref.solve_dense(b)
    fact.solve(b)
    z = fact.solution_raw(0, N)
    assert rel_err(z, ref.raw_solution()) <= REL_TOL
    if refine:
        assert scaled_residual(K, z, b) <= RESID_TOL
    # sparse rhs (SleqpVec), several solution() ranges per solve
    idx = np.sort(rng.choice(N, max(1, N // 3), replace=False)).astype(np.int32)
    val = rng.standard_normal(idx.size)
    ref.solve_sparse(idx, val)
    fact.solve(SleqpVec(N, idx, val))
    want = ref.raw_solution()
    assert rel_err(fact.solution_raw(0, n), want[:n]) <= REL_TOL
    assert rel_err(fact.solution_raw(n, N), want[n:]) <= REL_TOL
    # packed output identical in structure to sleqp_vec_set_from_raw on the oracle's buffer
    sv = fact.solution(0, N, ZERO_EPS)
    oi, od = ref.solution(0, N, ZERO_EPS)
    assert rel_err(sv.to_raw(), oracle.vec_to_raw(N, oi, od)) <= REL_TOL


def _judge(fact, N, kc, kr, kd, b, tag):
    """One system against the LAPACK-restating oracle, no escape hatch: flat 1e-9 wherever the oracle
    itself is that accurate (judged by an extended-precision reference); where partial-pivoting LU has
    already lost more than that, the device must be at least as close to the reference as the oracle
    (x4 slack) - or, like the oracle on exactly dependent rows, report the matrix as singular."""
    from sleqp_amd import HipfactError
    from sleqp_amd.sparse import SleqpMat
    from util import reference_solution

    K = synth.kkt_full_matrix(N, kc, kr, kd)
    try:
        ref = oracle.OracleFact(N, kc, kr, kd)
    except ZeroDivisionError:
        with pytest.raises(HipfactError) as e:
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            fact.solve(b)
            fact.solution_raw(0, N)
        assert e.value.code == -3, tag
        return "singular"
    ref.solve_dense(b)
    zo = ref.raw_solution()
    truth = reference_solution(K.toarray(), b, ref)
    err_o = rel_err(zo, truth)
    try:
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        z = fact.solution_raw(0, N)
    except HipfactError as e:
        # only legitimate when the system is numerically singular: the oracle's own answer is then garbage
        assert e.code == -3 and err_o > 1e-6, (tag, err_o, str(e))
        return "singular"
    err_d = rel_err(z, truth)
    assert err_d <= max(REL_TOL, 4.0 * err_o), (tag, err_d, err_o)
    if err_o <= 0.1 * REL_TOL:
        assert rel_err(z, zo) <= REL_TOL, (tag, rel_err(z, zo))
    assert scaled_residual(K, z, b) <= 1e-11, tag
    return "ok"


# expected outcome of every draw of the sweep below ("S": the working set is numerically rank deficient - both the
# device and the oracle must say so, _judge checks it): pinned per seed, not a count
SWEEP_OUTCOMES = ["singular" if c == "S" else "ok" for c in "ooooSoooooooSoooooooSooo"]


def test_random_sweep_vs_oracle(fact):
    """Seeded sweep over shapes, densities, active bounds and kernel configurations (per-level
    launches / single-launch top of the tree, pull / scatter extend-add): every solution against
    the dense LAPACK-restating oracle, including the ill-conditioned and singular draws."""
    rng = np.random.default_rng(2024)
    outcomes = []
    for trial in range(24):
        n = int(rng.integers(20, 600))
        m = int(rng.integers(1, max(2, n // 2)))
        kind = "b" if trial % 2 else "u"
        frac = float(rng.choice([0.0, 0.05, 0.3]))
        J, vi, ci, W = _problem(n, m, kind, frac, 100 + trial)
        if W > n:  # more working-set rows than variables cannot have full row rank (pub_working_set.h:42-44)
            ci = ci.copy()
            ci[n - int((vi >= 0).sum()):] = -1
            W = int((vi >= 0).sum() + (ci >= 0).sum())
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        fact.set_option("factor_top_max", [100, 0, 6][trial % 3])
        fact.set_option("pull_max_children", [4, 0][(trial // 3) % 2])
        b = rng.standard_normal(N)
        outcomes.append(_judge(fact, N, kc, kr, kd, b, (trial, n, m, kind, frac)))
    assert outcomes == SWEEP_OUTCOMES, "".join("o" if o == "ok" else "S" for o in outcomes)


def test_graded_conditioning_vs_oracle(fact):
    """Row scalings 1e0..1e8, nearly parallel rows (1e-2..1e-5), column scalings and mixtures with
    active bounds (what fact_ma57.c:444-507,743-763 handles by scaling + threshold
    pivoting).  The constrained pivot order with exact row equilibration and refinement on K itself
    stays within the flat tolerance of the oracle wherever the oracle is itself accurate, and beats
    it elsewhere."""
    from util import graded_family

    seen = 0
    for name, J, vi, ci in graded_family():
        m, n = J.shape
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        b = np.random.default_rng(1).standard_normal(N)
        assert _judge(fact, N, kc, kr, kd, b, name) == "ok", name
        seen += 1
        # the three AugJac right-hand-side shapes as well (standard_aug_jac.c:306-435)
        g = np.zeros(N)
        g[:n] = np.random.default_rng(2).standard_normal(n)
        assert _judge(fact, N, kc, kr, kd, g, name + "/project") == "ok"
        c = np.zeros(N)
        c[n:] = np.random.default_rng(3).standard_normal(N - n)
        assert _judge(fact, N, kc, kr, kd, c, name + "/min_norm") == "ok"
    assert seen == 24


def test_reference_known_answers_on_device(fact):
    """constrained_newton_test.c:204-275, unconstrained_newton_test.c:67-205 and
    dual_estimation_test.c:15-103 known answers, reproduced through the device backend."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    aug = StandardAugJac(2, fact)
    aug.set_iterate(SleqpMat(1, 2, [0, 0, 1], [0], [1.0]), [-1, -1], [0])
    p = aug.project_nullspace(SleqpVec(2, [0, 1], [2.0, 4.0]))
    assert p.indices.tolist() == [0] and abs(p.data[0] - 2.0) < 1e-8
    # Newton step of the strictly convex model = -P g / 2 = (-1, 0)
    assert np.allclose(-0.5 * p.to_raw(), [-1.0, 0.0], atol=1e-8)
    # empty working set: projection is the identity, step (-1, -2)
    aug.set_iterate(SleqpMat(0, 2, [0, 0, 0], [], []), [-1, -1], [])
    p = aug.project_nullspace(SleqpVec(2, [0, 1], [2.0, 4.0]))
    assert np.allclose(-0.5 * p.to_raw(), [-1.0, -2.0], atol=1e-8)
    # both bounds active: LSQ duals of -grad = (-2, -4)
    aug.set_iterate(SleqpMat(0, 2, [0, 0, 0], [], []), [0, 1], [])
    d = aug.solve_lsq(SleqpVec(2, [0, 1], [-2.0, -4.0]))
    assert d.indices.tolist() == [0, 1] and np.allclose(d.data, [-2.0, -4.0], atol=1e-8)
    # SURVEY §8c reference run: A = [1 2], P (3, 1) = (2, -1)
    aug.set_iterate(SleqpMat(1, 2, [0, 1, 2], [0, 0], [1.0, 2.0]), [-1, -1], [0])
    p = aug.project_nullspace(SleqpVec(2, [0, 1], [3.0, 1.0]))
    assert np.allclose(p.to_raw(), [2.0, -1.0], atol=1e-13)


def test_steihaug_cg_on_device_projection(fact):
    """The reference's projected CG (tr/steihaug_solver.c) driven by device projections gives the
    same step as the oracle's restatement driven by the LAPACK-restating projections."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 120, 50
    J, vi, ci, W = _problem(n, m, "u", 0.05, 9)
    rng = np.random.default_rng(2)
    B = sp.random(n, n, density=0.03, random_state=1)
    H = (B @ B.T + sp.eye(n)).tocsc()
    HL = sp.tril(H, format="csc")
    HL.sort_indices()
    g = rng.standard_normal(n)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    ref = oracle.OracleFact(N, kc, kr, kd)
    want, its = ref.steihaug(n, HL.indptr, HL.indices, HL.data, g, trust_radius=0.7)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)

    def P(v):
        return aug.project_nullspace(SleqpVec.from_raw(v)).to_raw()

    # steihaug_solver.c:218-496 with dense vectors
    z, r = np.zeros(n), g.copy()
    gg = P(r)
    d = -gg
    rg = r @ gg
    step = np.zeros(n)
    for it in range(100):
        if abs(rg) < 1e-16:
            step = z
            break
        Bd = H @ d
        dBd = d @ Bd
        alpha = rg / dBd
        zn = z + alpha * d
        if zn @ zn >= 0.7 ** 2:
            pd, dd, pp = z @ d, d @ d, z @ z
            step = z + (-pd + np.sqrt(pd * pd - dd * (pp - 0.49))) / dd * d
            break
        z = zn
        r = r + alpha * Bd
        gg = P(r)
        beta = 1.0 / rg
        rg = r @ gg
        beta *= rg
        d = -gg + beta * d
    assert rel_err(step, want) <= 1e-8


def test_pattern_cache_and_refactor(fact):
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, W = _problem(400, 200, "b", 0.0, 1)
    N, kc, kr, kd = oracle.fill_aug_jac(400, 200, J.indptr, J.indices, J.data, vi, ci)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("analyses") == 1
    rng = np.random.default_rng(0)
    b = rng.standard_normal(N)
    for _ in range(3):  # same pattern, new values: numeric-only refactorisation
        kd2 = kd.copy()
        off = kr != np.repeat(np.arange(N), np.diff(kc))
        kd2[off] = rng.standard_normal(int(off.sum()))
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd2))
        ref = oracle.OracleFact(N, kc, kr, kd2)
        ref.solve_dense(b)
        fact.solve(b)
        assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL
    assert fact.info("analyses") == 1 and fact.info("cache_hits") == 3
    # N and nnz change between calls (standard_aug_jac.c: working set changes)
    J2, vi2, ci2, _ = _problem(300, 100, "u", 0.1, 2)
    N2, c2, r2, d2 = oracle.fill_aug_jac(300, 100, J2.indptr, J2.indices, J2.data, vi2, ci2)
    fact.set_matrix(SleqpMat(N2, N2, c2, r2, d2))
    assert fact.info("analyses") == 2
    b2 = rng.standard_normal(N2)
    ref = oracle.OracleFact(N2, c2, r2, d2)
    ref.solve_dense(b2)
    fact.solve(b2)
    assert rel_err(fact.solution_raw(0, N2), ref.raw_solution()) <= REL_TOL


def test_error_behaviour(fact):
    from sleqp_amd import HipfactError
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    # solve before set_matrix (call protocol)
    with pytest.raises(HipfactError) as e:
        fact.solve(np.zeros(0))
    assert e.value.code == -5
    # rank-deficient working set (duplicate rows): the LAPACK-restating oracle fails ("Failed to factorize using
    # LAPACK", fact_lapack.c:117-120); MA57 - the parity target - factors such a K ("Success - rank deficient",
    # fact_ma57.c:41-42) and so does the device, by static pivoting, with a warning; a right-hand side outside the
    # range of K is reported at the solve, one inside it is solved; static_pivot = 0: HIPFACT_ESINGULAR at once
    N, kc, kr, kd = oracle.fill_aug_jac(2, 2, [0, 2, 4], [0, 1, 0, 1], [1.0, 1.0, 2.0, 2.0], [-1, -1], [0, 1])
    with pytest.raises(ZeroDivisionError):
        oracle.OracleFact(N, kc, kr, kd)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("num_perturbed") >= 1 and "rank deficient" in fact.last_warning()
    with pytest.raises(HipfactError) as e:
        fact.solve(np.array([0.0, 0.0, 1.0, 0.0]))  # x0 + 2 x1 = 1 and = 0
        fact.solution_raw(0, N)
    assert e.value.code == -3
    fact.solve(np.array([0.0, 0.0, 5.0, 5.0]))  # consistent: min-norm x = (1, 2)
    assert rel_err(fact.solution_raw(0, 2), np.array([1.0, 2.0])) <= 1e-8
    fact.set_option("static_pivot", 0)
    with pytest.raises(HipfactError) as e:
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert e.value.code == -3
    fact.set_option("static_pivot", 1)
    # malformed matrix
    with pytest.raises(HipfactError) as e:
        fact.set_matrix(SleqpMat(2, 2, [0, 1, 3], [0, 0, 1], [1.0, 1.0, 1.0]))
    assert e.value.code == -1
    # rhs of the wrong dimension
    J, vi, ci, _ = _problem(20, 8, "u", 0.0, 0)
    N, kc, kr, kd = oracle.fill_aug_jac(20, 8, J.indptr, J.indices, J.data, vi, ci)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    with pytest.raises(HipfactError):
        fact.solve(SleqpVec(N + 1, [0], [1.0]))
    with pytest.raises(HipfactError):
        fact.solution_raw(0, N)  # no solve yet
    fact.solve(SleqpVec(N, [], []))  # empty rhs
    assert np.array_equal(fact.solution_raw(0, N), np.zeros(N))
    with pytest.raises(HipfactError):
        fact.solution_raw(0, N + 1)


def test_generic_mode_on_device(fact):
    from sleqp_amd.sparse import SleqpMat

    # SPD (what the reduced AugJac hands to PSD backends) and quasi-definite inputs
    B = sp.random(600, 600, density=0.01, random_state=0, format="csc")
    M = (B @ B.T + sp.eye(600) * 3).tocsc()
    n, m = 80, 30
    A = synth.uniform_jacobian(n, m, 5, 7)
    Hq = sp.diags(np.linspace(1.0, 3.0, n)) + sp.diags(np.full(n - 1, 0.2), -1) + sp.diags(np.full(n - 1, 0.2), 1)
    Q = sp.bmat([[Hq, A.T], [A, -1e-2 * sp.eye(m)]], format="csc")
    for mat in (M, Q):
        L = sp.tril(mat, format="csc")
        L.sort_indices()
        N = mat.shape[0]
        fact.set_matrix(SleqpMat(N, N, L.indptr, L.indices, L.data))
        assert fact.info("saddle") == 0.0
        b = np.random.default_rng(1).standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        ref = oracle.OracleFact(N, L.indptr, L.indices, L.data)
        ref.solve_dense(b)
        assert rel_err(z, ref.raw_solution()) <= 1e-8
        assert scaled_residual(mat, z, b) <= 1e-12


def test_generic_mode_with_tall_fronts(fact):
    """General symmetric (SPD) input whose factor fills in densely: fronts of more than 1024 rows, i.e. row-sliced
    items in the fused solve launch, reached through the non-saddle front end (gather / scatter by the permutation
    instead of the saddle products)."""
    from sleqp_amd.sparse import SleqpMat

    A = synth.uniform_jacobian(2600, 2200, 8, 4)  # S = A A^T + I: the reduced matrix of a dense-Schur problem
    M = (A @ A.T + sp.eye(2200)).tocsc()
    L = sp.tril(M, format="csc")
    L.sort_indices()
    N = M.shape[0]
    fact.set_matrix(SleqpMat(N, N, L.indptr, L.indices, L.data))
    assert fact.info("saddle") == 0.0 and fact.info("max_r") > 1024
    assert fact.info("fused_solve") == 1 and fact.info("solve_items") > fact.info("nsuper")
    rng = np.random.default_rng(2)
    dense = M.toarray()
    for _ in range(3):
        b = rng.standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        assert rel_err(z, np.linalg.solve(dense, b)) <= 1e-9
        assert scaled_residual(M, z, b) <= 1e-12
    assert fact.info("solve_timeouts") == 0


def test_condition_estimate(fact):
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(200, 80, "u", 0.0, 0)
    N, kc, kr, kd = oracle.fill_aug_jac(200, 80, J.indptr, J.indices, J.data, vi, ci)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    c = fact.cond()
    assert np.isfinite(c) and c >= 1.0


def test_spmv_kernels(fact):
    from sleqp_amd.fact import SpMat
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    rng = np.random.default_rng(0)
    for (r, c, dens) in [(1, 1, 1.0), (50, 80, 0.05), (400, 900, 0.01), (300, 300, 0.3), (2000, 10, 0.5)]:
        M = sp.random(r, c, density=dens, random_state=1, format="csc")
        M.sort_indices()
        S = SpMat(fact, SleqpMat.from_scipy(M))
        xi = np.sort(rng.choice(c, max(1, c // 2), replace=False)).astype(np.int32)
        xd = rng.standard_normal(xi.size)
        want = oracle.mat_mult_vec(r, c, M.indptr, M.indices, M.data, xi, xd)
        assert rel_err(S.mult_vec(SleqpVec(c, xi, xd)), want) <= 1e-14
        yi = np.sort(rng.choice(r, max(1, r // 2), replace=False)).astype(np.int32)
        yd = rng.standard_normal(yi.size)
        ti, td = oracle.mat_mult_vec_trans(r, c, M.indptr, M.indices, M.data, yi, yd, 0.0)
        got = S.mult_vec_trans(SleqpVec(r, yi, yd), eps=0.0)
        assert rel_err(got.to_raw(), oracle.vec_to_raw(c, ti, td)) <= 1e-14
        S.free()
    # symmetric product from the lower triangle (mex_hess.c:85-139)
    H = sp.random(300, 300, density=0.05, random_state=2)
    H = (H + H.T + sp.eye(300)).tocsc()
    HL = sp.tril(H, format="csc")
    HL.sort_indices()
    S = SpMat(fact, SleqpMat.from_scipy(HL))
    d = rng.standard_normal(300)
    assert rel_err(S.mult_vec_sym(d), oracle.hess_prod_lower(300, HL.indptr, HL.indices, HL.data, d)) <= 1e-14
    # empty matrix
    E = SpMat(fact, SleqpMat(3, 4))
    assert np.array_equal(E.mult_vec(np.ones(4)), np.zeros(3))
    # the streaming kernel (k_spmv_stream: row blocks through LDS, the default from 4 M entries on) on ragged matrices:
    # empty rows, rows longer than a block, block boundaries at odd entries, an odd number of entries; bitwise deterministic
    fact.set_option("spmv_stream_min", 0)
    for (r, c, dens, seed) in [(1, 1, 1.0, 1), (50, 80, 0.3, 2), (3000, 2500, 0.02, 3), (700, 9000, 0.2, 4), (5000, 40, 0.6, 5)]:
        M = sp.random(r, c, density=dens, random_state=seed, format="lil")
        if r > 100:
            M[7, :] = 1.0       # a row / a column longer than a block
            M[:, 3] = 2.0
            M[11, :] = 0.0      # an empty row
        M = sp.csc_matrix(M)
        M.eliminate_zeros()
        M.sort_indices()
        S = SpMat(fact, SleqpMat.from_scipy(M))
        x, yv = rng.standard_normal(c), rng.standard_normal(r)
        assert fact.info("spmv_stream") == 1
        got = S.mult_vec(x)
        assert rel_err(got, M @ x) <= 1e-13 and np.array_equal(got, S.mult_vec(x))
        got_t = S.mult_vec_trans(SleqpVec.from_raw(yv), eps=0.0).to_raw()
        assert rel_err(got_t, M.T @ yv) <= 1e-13
        fact.set_option("spmv_stream", 0)  # ... and against the lanes-per-row kernel
        assert rel_err(S.mult_vec(x), got) <= 1e-13
        fact.set_option("spmv_stream", 1)
        S.free()


def test_assembly_bit_exact(fact):
    from sleqp_amd.sparse import SleqpMat

    rng = np.random.default_rng(4)
    for n, m in [(1, 0), (5, 3), (500, 300), (5000, 2000)]:
        J = synth.uniform_jacobian(n, m, min(4, n), 2) if m > 0 else sp.csc_matrix((0, n))
        vi = np.full(n, -1, dtype=np.int32)
        av = np.sort(rng.choice(n, n // 10, replace=False))
        vi[av] = np.arange(av.size)
        ci = np.full(m, -1, dtype=np.int32)
        ac = np.sort(rng.choice(m, (2 * m) // 3, replace=False)) if m > 0 else np.zeros(0, int)
        ci[ac] = av.size + np.arange(ac.size)
        W = av.size + ac.size
        K = fact.assemble_kkt(SleqpMat.from_scipy(J), vi, ci, W)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        assert K.num_cols == N
        assert np.array_equal(K.cols, kc) and np.array_equal(K.rows, kr) and np.array_equal(K.data, kd)


def _ws(n, m, rng, row_frac, bound_frac):
    """Working-set index maps in the reference's numbering (working_set.c:114-168)."""
    vi = np.full(n, -1, dtype=np.int32)
    av = np.sort(rng.choice(n, int(round(bound_frac * n)), replace=False))
    vi[av] = np.arange(av.size)
    ci = np.full(m, -1, dtype=np.int32)
    ac = np.sort(rng.choice(m, int(round(row_frac * m)), replace=False))
    ci[ac] = av.size + np.arange(ac.size)
    return vi, ci, int(av.size + ac.size)


@pytest.mark.parametrize("kind", ["duplicate_row", "row_equals_active_bound", "scaled_duplicate"])
@pytest.mark.parametrize("boundary", ["device_assembly", "vtable"])
def test_rank_deficient_working_sets_like_ma57(fact, kind, boundary):
    """A working set with dependent rows.  MA57 factors such a K and says "Success - rank deficient", a positive status
    that the reference's MA57_CHECK_ERROR lets pass (fact_ma57.c:41-42, 118-133): the SQP run goes on.  Here the zero
    pivot of A A^T triggers static pivoting (every pivot shifted by 1e-8, solves refined against the caller's K), the
    factorisation succeeds with a warning and `num_perturbed` says how many pivots the unshifted attempt reported.  The
    null-space projection and the min-norm solve of a consistent right-hand side are unique for dependent rows: both
    within 1e-8 of the oracle on the DEDUPLICATED working set.  A right-hand side outside the range of K has no
    solution: the solve reports it as singular (what the LAPACK-restating oracle says of K itself)."""
    from sleqp_amd import HipfactError
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 700, 300
    rng = np.random.default_rng(41)
    J0 = synth.banded_jacobian(n, m, 10, 80, 29).tocsr()
    vi = np.full(n, -1, dtype=np.int32)
    av = np.sort(rng.choice(n, 30, replace=False))
    vi[av] = np.arange(av.size)
    if kind == "duplicate_row":
        extra = J0[17]
    elif kind == "scaled_duplicate":
        extra = -3.5 * J0[211]
    else:  # a constraint row that is the unit row of a variable whose bound is active
        extra = sp.csr_matrix(([1.0], ([0], [int(av[7])])), shape=(1, n))
    J = sp.vstack([J0, extra]).tocsc()
    J.sort_indices()
    ci = (av.size + np.arange(m + 1)).astype(np.int32)
    W = av.size + m + 1
    # the same working set without the dependent row
    ci_d = ci.copy()
    ci_d[m] = -1
    Wd = W - 1
    N, kc, kr, kd = oracle.fill_aug_jac(n, m + 1, J.indptr, J.indices, J.data, vi, ci)
    Nd, kcd, krd, kdd = oracle.fill_aug_jac(n, m + 1, J.indptr, J.indices, J.data, vi, ci_d)
    sv = np.linalg.svd(synth.kkt_full_matrix(N, kc, kr, kd).toarray(), compute_uv=False)
    assert sv[-1] <= 1e-13 * sv[0] < sv[-2]  # K itself is singular, with a null space of dimension one
    if kind != "scaled_duplicate":  # (exactly equal rows: the oracle's LU meets an exact zero)
        with pytest.raises(ZeroDivisionError):
            oracle.OracleFact(N, kc, kr, kd)
    ref = oracle.OracleFact(Nd, kcd, krd, kdd)
    aug = StandardAugJac(n, fact, device_assembly=(boundary == "device_assembly"))
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    # projection onto the null space of the working set
    g = rng.standard_normal(n)
    idx, val = ref.project_nullspace(n, np.arange(n), g)
    want = oracle.vec_to_raw(n, idx, val)
    got = aug.project_nullspace(SleqpVec.from_raw(g)).to_raw()
    assert rel_err(got, want) <= 1e-8, rel_err(got, want)
    # (rows that are dependent up to rounding may leave a pivot at rounding level instead of a zero: the shift then comes
    # with the first solve that stalls on it)
    assert fact.info("num_perturbed") >= 1 and fact.info("static_pivot_shift") > 0
    assert "rank deficient" in fact.last_warning()
    # min-norm solve of a consistent right-hand side c = [x0 on the active bounds; A x0]
    x0 = rng.standard_normal(n)
    rows = J.tocsr()
    c = np.concatenate([x0[av], rows @ x0])
    c_d = c[:-1]
    idx, val = ref.solve_min_norm(n, np.arange(Wd), c_d)
    want = oracle.vec_to_raw(n, idx, val)
    got = aug.solve_min_norm(SleqpVec.from_raw(c)).to_raw()
    assert rel_err(got, want) <= 1e-8, rel_err(got, want)
    # least-squares multipliers: y is not unique, A_W^T y is (the projection's complement)
    y = aug.solve_lsq(SleqpVec.from_raw(g)).to_raw()
    AW = sp.vstack([sp.eye(n, format="csr")[av], rows]).tocsr()
    idx, val = ref.solve_lsq(n, np.arange(n), g)
    y_d = oracle.vec_to_raw(Wd, idx, val)
    AWd = AW[:Wd]
    assert rel_err(AW.T @ y, AWd.T @ y_d) <= 1e-8
    # a right-hand side outside the range of K: no solution, reported at the solve
    c_bad = c.copy()
    c_bad[-1] += 1.0
    with pytest.raises(HipfactError) as e:
        aug.solve_min_norm(SleqpVec.from_raw(c_bad))
    assert e.value.code == -3
    # the next factorisation starts unperturbed: the deduplicated working set needs no shift
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci_d)
    assert fact.info("num_perturbed") == 0 and fact.info("static_pivot_shift") == 0 and fact.last_warning() is None
    idx, val = ref.project_nullspace(n, np.arange(n), g)
    assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
    # option off: the behaviour of rounds 1 - 5
    fact.set_option("static_pivot", 0)
    with pytest.raises(HipfactError) as e:
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        aug.project_nullspace(SleqpVec.from_raw(g))
    assert e.value.code == -3
    fact.set_option("static_pivot", 1)


def test_working_set_changes_reuse_the_superset_plan(fact):
    """the device assembly analyses the structure [I J^T; J 0] of a SUPERSET of the
    working set once; rows that leave the working set become unit rows, active bounds are eliminated
    exactly, so a changed working set costs a numeric refactorisation only.  Every working set is
    checked against the oracle (fill_aug_jac + LAPACK restatement) for the three AugJac solves."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 900, 400
    J = synth.banded_jacobian(n, m, 10, 80, 31)
    rng = np.random.default_rng(12)
    aug = StandardAugJac(n, fact)
    g = rng.standard_normal(n)

    def check(vi, ci, W):
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        assert np.array_equal(aug.K.cols, kc) and np.array_equal(aug.K.rows, kr) and np.array_equal(aug.K.data, kd)
        ref = oracle.OracleFact(N, kc, kr, kd)
        idx, val = ref.project_nullspace(n, np.arange(n), g)
        assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
        idx, val = ref.solve_lsq(n, np.arange(n), g)
        assert rel_err(aug.solve_lsq(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(W, idx, val)) <= REL_TOL
        c = rng.standard_normal(W)
        idx, val = ref.solve_min_norm(n, np.arange(W), c)
        assert rel_err(aug.solve_min_norm(SleqpVec.from_raw(c)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
        # a dense right-hand side in the caller's numbering, all of the solution
        b = rng.standard_normal(N)
        ref.solve_dense(b)
        fact.solve(b)
        assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL

    vi, ci, W = _ws(n, m, rng, 1.0, 0.05)
    check(vi, ci, W)
    assert fact.info("analyses") == 1 and fact.info("maps_on") == 1 and fact.info("m_struct") == m
    for it in range(6):  # rows and bounds enter and leave: same plan, numeric refactorisation only
        vi, ci, W = _ws(n, m, rng, [0.99, 0.95, 0.8, 0.6, 0.97, 1.0][it], [0.0, 0.1, 0.02, 0.3, 0.05, 0.0][it])
        check(vi, ci, W)
        assert fact.info("analyses") == 1, it
    # a much smaller working set gets a structure of its own ...
    vi, ci, W = _ws(n, m, rng, 0.2, 0.05)
    check(vi, ci, W)
    assert fact.info("analyses") == 2 and fact.info("m_struct") < m // 2
    small = (vi, ci, W)
    # ... and both structures stay cached: going back and forth costs no analysis
    vi, ci, W = _ws(n, m, rng, 0.9, 0.0)
    check(vi, ci, W)
    check(*small)
    assert fact.info("analyses") == 2 and fact.info("plan_swaps") >= 2
    # bounds only, and the empty working set
    vi, ci, W = _ws(n, m, rng, 0.0, 0.2)
    check(vi, ci, W)
    vi, ci, W = _ws(n, m, rng, 0.0, 0.0)
    check(vi, ci, W)
    # the plain path (K's own pattern analysed) agrees
    vi, ci, W = _ws(n, m, rng, 0.7, 0.1)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    p1 = aug.project_nullspace(SleqpVec.from_raw(g)).to_raw()
    fact.set_option("assemble_superset", 0)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    assert fact.info("maps_on") == 0
    p0 = aug.project_nullspace(SleqpVec.from_raw(g)).to_raw()
    fact.set_option("assemble_superset", 1)
    assert rel_err(p1, p0) <= 1e-12


def test_vtable_set_matrix_reuses_the_superset_plan(fact):
    """The PLAIN SleqpFact boundary (standard_aug_jac.c assembles K on the host, sleqp_fact_set_matrix gets nothing
    but K): the rows of A_W are recognised by content, so a working set made of rows seen before - whatever it does
    to the pattern of K - is a numeric refactorisation; a row never seen before extends the dictionary and costs one
    analysis.  All three AugJac solves against the oracle for every working set."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 900, 400
    J = synth.banded_jacobian(n, m, 10, 80, 31)
    rng = np.random.default_rng(21)
    aug = StandardAugJac(n, fact, device_assembly=False)  # fill_aug_jac on the host, fact.set_matrix(K)
    g = rng.standard_normal(n)

    def check(vi, ci, W):
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        assert np.array_equal(aug.K.cols, kc) and np.array_equal(aug.K.rows, kr) and np.array_equal(aug.K.data, kd)
        ref = oracle.OracleFact(N, kc, kr, kd)
        idx, val = ref.project_nullspace(n, np.arange(n), g)
        assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
        idx, val = ref.solve_lsq(n, np.arange(n), g)
        assert rel_err(aug.solve_lsq(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(W, idx, val)) <= REL_TOL
        if W > 0:
            c = rng.standard_normal(W)
            idx, val = ref.solve_min_norm(n, np.arange(W), c)
            assert rel_err(aug.solve_min_norm(SleqpVec.from_raw(c)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
        b = rng.standard_normal(N)
        ref.solve_dense(b)
        fact.solve(b)
        assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL

    vi, ci, W = _ws(n, m, rng, 1.0, 0.05)  # every row of J once: the dictionary is complete
    check(vi, ci, W)
    assert fact.info("analyses") == 1 and fact.info("maps_on") == 1 and fact.info("vtable_rows") == m
    for it in range(6):  # rows and bounds enter and leave: same plan, numeric refactorisation only
        vi, ci, W = _ws(n, m, rng, [0.99, 0.95, 0.8, 0.6, 0.97, 1.0][it], [0.0, 0.1, 0.02, 0.3, 0.05, 0.0][it])
        check(vi, ci, W)
        assert fact.info("analyses") == 1, it
    same = (vi, ci, W)
    check(*same)  # the very same K again: recognised by comparison, nothing rebuilt
    assert fact.info("analyses") == 1
    # changed VALUES on an unchanged pattern
    J2 = J.copy()
    J2.data = J2.data * (1.0 + 0.1 * rng.standard_normal(J2.nnz))
    Jkeep, J = J, J2
    check(*same)
    assert fact.info("analyses") == 1
    J = Jkeep
    # a much smaller working set gets a structure of its own; both stay cached
    small = _ws(n, m, rng, 0.2, 0.05)
    check(*small)
    assert fact.info("analyses") == 2 and fact.info("m_struct") < m // 2
    check(*same)
    check(*small)
    assert fact.info("analyses") == 2
    # bounds only; the empty working set takes the plain path (no rows to recognise)
    check(*_ws(n, m, rng, 0.0, 0.2))
    # rows never seen before: the dictionary grows, one analysis, then reuse again
    fact2 = type(fact)()
    aug2 = StandardAugJac(n, fact2, device_assembly=False)
    first = _ws(n, m, rng, 0.5, 0.0)

    def check2(vi, ci, W):
        aug2.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        ref = oracle.OracleFact(N, kc, kr, kd)
        idx, val = ref.project_nullspace(n, np.arange(n), g)
        assert rel_err(aug2.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL

    check2(*first)
    assert fact2.info("analyses") == 1
    vi, ci, W = first
    ci2 = ci.copy()
    out = np.flatnonzero(ci2 < 0)[:40]  # forty rows that have never been active enter
    act = np.sort(np.concatenate([np.flatnonzero(ci2 >= 0), out]))
    ci2[:] = -1
    ci2[act] = np.arange(act.size)
    check2(vi, ci2, int(act.size))
    assert fact2.info("analyses") == 2 and fact2.info("vtable_rows") == act.size
    check2(*first)  # back to a subset of the dictionary
    ci3 = ci2.copy()
    drop = act[::7]
    keep = np.setdiff1d(act, drop)
    ci3[:] = -1
    ci3[keep] = np.arange(keep.size)
    check2(vi, ci3, int(keep.size))
    assert fact2.info("analyses") == 2
    # identical rows (twins) are told apart by their order of appearance
    Jt = sp.vstack([J.tocsr()[:50], J.tocsr()[:50]]).tocsc()
    fact3 = type(fact)()
    aug3 = StandardAugJac(n, fact3, device_assembly=False)
    vi0 = np.full(n, -1, dtype=np.int32)
    ci_t = np.full(100, -1, dtype=np.int32)
    rows_t = np.r_[np.arange(0, 50, 2), 50 + np.arange(1, 50, 2)]  # no two twins together: full row rank
    ci_t[rows_t] = np.arange(rows_t.size)
    aug3.set_iterate(SleqpMat.from_scipy(Jt), vi0, ci_t)
    N, kc, kr, kd = oracle.fill_aug_jac(n, 100, Jt.indptr, Jt.indices, Jt.data, vi0, ci_t)
    ref = oracle.OracleFact(N, kc, kr, kd)
    idx, val = ref.project_nullspace(n, np.arange(n), g)
    assert rel_err(aug3.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
    # the option switches the exact-pattern path back on
    fact3.set_option("superset_vtable", 0)
    aug3.set_iterate(SleqpMat.from_scipy(Jt), vi0, ci_t)
    assert fact3.info("maps_on") == 0
    assert rel_err(aug3.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL


def _with_dense_columns(J, k, seed, frac=1.0):
    """J plus k columns that have an entry in (a share `frac` of) every row."""
    m, n = J.shape
    rng = np.random.default_rng(seed)
    cols = np.sort(rng.choice(n, k, replace=False))
    rows = np.flatnonzero(rng.random(m) < frac) if frac < 1.0 else np.arange(m)
    D = sp.csc_matrix((rng.standard_normal(rows.size * k), (np.tile(rows, k), np.repeat(cols, rows.size))), shape=(m, n))
    Jd = (J + D).tocsc()
    Jd.sort_indices()
    return Jd, cols


@pytest.mark.parametrize("mode", [1, 2], ids=["late_elimination", "low_rank_correction"])
@pytest.mark.parametrize("vtable", [1, 0], ids=["row_dictionary", "exact_pattern"])
@pytest.mark.parametrize("k", [1, 4, 16])
def test_dense_jacobian_columns_vs_oracle(fact, k, vtable, mode):
    """the reference's backends order K itself (fact_ma57.c:314-345, AMD on K) and keep a variable that
    appears in every constraint away from the fill.  Here such columns are left out of S = A A^T: eliminated late
    inside the tree (dense_mode 1, the default: M = [A_s A_s^T  A_d; A_d^T  -I]) or applied to every solve as a
    low-rank correction (dense_mode 2, dense_cols.inc).  The plan stays as sparse as without them (plus the late
    variables' own rows of L), and the three AugJac solves agree with the oracle - with and without active bounds
    on the dense variables themselves."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 1500, 700
    J0 = synth.banded_jacobian(n, m, 10, 80, 17)
    J, dcols = _with_dense_columns(J0, k, 5, frac=1.0 if k < 16 else 0.6)
    rng = np.random.default_rng(k)
    fact.set_option("dense_mode", mode)
    fact.set_option("superset_vtable", vtable)
    aug = StandardAugJac(n, fact, device_assembly=False)
    g = rng.standard_normal(n)
    vi, ci, W = _ws(n, m, rng, 1.0, 0.0)
    N0, c0, r0, d0 = oracle.fill_aug_jac(n, m, J0.indptr, J0.indices, J0.data, vi, ci)
    base = HipFactPlanStats(N0, c0, r0, d0)
    for step in range(3):
        if step == 1:  # rows leave, ordinary bounds become active
            vi, ci, W = _ws(n, m, rng, 0.93, 0.03)
        if step == 2:  # ... and a bound on one of the dense variables themselves
            vi, ci, W = _ws(n, m, rng, 0.97, 0.0)
            vi[:] = -1
            vi[dcols[0]] = 0
            ci[ci >= 0] += 1
            W += 1
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        assert np.array_equal(aug.K.cols, kc) and np.array_equal(aug.K.rows, kr)
        if step < 2:
            assert fact.info("dense_columns" if mode == 2 else "late_columns") == k
        if step == 0:
            # as sparse as the plan without the dense columns (the bar: within 1.2x), plus - late elimination - the
            # rows the k late variables themselves have in L
            assert fact.info("nnzL") <= 1.2 * base.nnzL + (k * (m + k) if mode == 1 else 0), (fact.info("nnzL"), base.nnzL)
        ref = oracle.OracleFact(N, kc, kr, kd)
        idx, val = ref.project_nullspace(n, np.arange(n), g)
        assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL, step
        idx, val = ref.solve_lsq(n, np.arange(n), g)
        assert rel_err(aug.solve_lsq(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(W, idx, val)) <= REL_TOL, step
        c = rng.standard_normal(W)
        idx, val = ref.solve_min_norm(n, np.arange(W), c)
        assert rel_err(aug.solve_min_norm(SleqpVec.from_raw(c)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL, step
        b = rng.standard_normal(N)
        ref.solve_dense(b)
        fact.solve(b)
        assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL, step
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        assert scaled_residual(K, fact.solution_raw(0, N), b) <= 1e-12


class HipFactPlanStats:
    """nnz(L) of the plan for a matrix (host analysis only, through the plan ABI)."""

    def __init__(self, N, kc, kr, kd):
        from plan_emul import Plan
        from sleqp_amd import _lib

        p = Plan(_lib.load(), N, kc, kr, kd)
        self.nnzL = p.nnzL
        self.dense = len(p.dense_cols)


def test_dense_column_treatment_is_an_option(fact):
    """`dense_mode = 0` sends a Jacobian with dense columns down the ordinary path (S = A A^T with the cliques in it):
    same solution as late elimination (1) and the low-rank split (2), a denser factor."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 900, 400
    J0 = synth.banded_jacobian(n, m, 8, 60, 23)
    J, _ = _with_dense_columns(J0, 2, 9)
    rng = np.random.default_rng(2)
    vi, ci, W = _ws(n, m, rng, 1.0, 0.0)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = rng.standard_normal(N)
    sols, nnzL = [], []
    for mode in (1, 2, 0):
        fact.set_option("dense_mode", mode)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("late_columns") == (2 if mode == 1 else 0)
        assert fact.info("dense_columns") == (2 if mode == 2 else 0)
        fact.solve(b)
        sols.append(fact.solution_raw(0, N).copy())
        nnzL.append(fact.info("nnzL"))
    ref = oracle.OracleFact(N, kc, kr, kd)
    ref.solve_dense(b)
    z = ref.raw_solution()
    assert all(rel_err(zz, z) <= REL_TOL for zz in sols)
    assert nnzL[2] > nnzL[0] and nnzL[2] > nnzL[1]


@pytest.mark.timeout(240)
def test_dense_jacobian_columns_at_full_size():
    """BASELINE configs[3] (n = 1e5, m = 5e4, nnz(J) = 1e6) plus 1, 4 and 16 dense columns: the analysis no longer
    refuses the pattern, nnz(L) stays within 1.2x of the plan without them, and the solution agrees with the
    oracle's sparse LDL^T (which orders K itself) to 1e-8; scaled residual 1e-12."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 100000, 50000
    J0 = synth.banded_jacobian(n, m, 20, 200, 0)
    vi, ci, _ = synth.working_set_all_rows(n, m)
    N0, c0, r0, d0 = synth.kkt_lower_from_jacobian(J0, vi, ci)
    base = HipFactPlanStats(N0, c0, r0, d0).nnzL
    from sleqp_amd.fact import HipFact

    for k in (1, 4, 16):
        J, dcols = _with_dense_columns(J0, k, 3)
        N, cp, ri, vx = synth.kkt_lower_from_jacobian(J, vi, ci)
        fact = HipFact(device=0)  # (a fresh row dictionary: the rows of the previous matrix would stay in the structure)
        fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
        assert fact.info("late_columns") == k and fact.info("nnzL") <= 1.2 * base + k * (m + k)
        b = np.random.default_rng(k).standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        K = synth.kkt_full_matrix(N, cp, ri, vx)
        assert scaled_residual(K, z, b) <= 1e-12
        if k == 4:
            # the oracle factors K itself, in an order that keeps the dense variables for last (what an ordering on K
            # finds by itself: fact_ma57.c:314-345; in the natural order their columns would fill the factor)
            perm = np.r_[np.setdiff1d(np.arange(n), dcols), n + np.arange(N - n), dcols].astype(np.int32)
            zo = oracle.OracleLdl(N, cp, ri, vx, perm=perm).solve(b)
            assert rel_err(z, zo) <= 1e-8
        fact.free()


@pytest.mark.parametrize("case", ["hub_row", "two_hub_rows_and_late_columns", "seventy_columns", "sub_threshold_columns",
                                  "row_only_in_late_columns"])
@pytest.mark.parametrize("boundary", ["fact_vtable", "aug_jac"])
def test_structural_robustness_vs_oracle(fact, case, boundary):
    """VERDICT round 3, item 2 (MA57 / UMFPACK order all N columns of K, fact_ma57.c:314-345, 761-763, and
    take a budget-type constraint or a variable that sits in many constraints in their stride): a dense constraint
    row, two of them plus dense columns, more dense columns than the low-rank split of round 3 could take (70 > 64),
    columns below its threshold, a row whose only entries lie in late columns - each against the LAPACK-restating
    oracle, for the three AugJac solves and a dense right-hand side, through the plain vtable (row dictionary) and
    through the device assembly, over changing working sets."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 1500, 700
    J = synth.banded_jacobian(n, m, 10, 80, 31)
    if case == "hub_row":
        J, _ = synth.with_dense_rows(J, 1, 1)
    elif case == "two_hub_rows_and_late_columns":
        J, _ = synth.with_dense_rows(J, 2, 2)
        J, _ = synth.with_dense_columns(J, 5, 3, frac=0.8)
    elif case == "seventy_columns":
        J, _ = synth.with_dense_columns(J, 70, 4, entries=150)
    elif case == "sub_threshold_columns":  # 60 entries: below max(64, 4 sqrt(m)) of round 3, above max(48, 1.5 sqrt(m))
        J, _ = synth.with_dense_columns(J, 40, 5, entries=60)
    else:
        J, cols = synth.with_dense_columns(J, 2, 6)
        Jl = J.tolil()
        keep = np.zeros(n, dtype=bool)
        keep[cols] = True
        for r in (11, 400):
            for c in list(Jl.rows[r]):
                if not keep[c]:
                    Jl[r, c] = 0.0
        J = Jl.tocsc()
        J.eliminate_zeros()
        J.sort_indices()
    rng = np.random.default_rng(12)
    aug = StandardAugJac(n, fact, device_assembly=(boundary == "aug_jac"))
    g = rng.standard_normal(n)
    for step, (rf, bf) in enumerate(((1.0, 0.0), (0.95, 0.02), (1.0, 0.01))):
        vi, ci, W = _ws(n, m, rng, rf, bf)
        if case in ("hub_row", "two_hub_rows_and_late_columns") and step == 1:
            pass  # (the dense rows may leave the working set: the plan then holds them as unit rows)
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        if step == 0:
            if case == "hub_row":
                assert fact.info("late_rows") == 1 and fact.info("late_columns") == 0
            elif case == "two_hub_rows_and_late_columns":
                assert fact.info("late_rows") >= 2 and fact.info("late_columns") == 5
            elif case == "seventy_columns":
                assert fact.info("late_columns") == 70
            elif case == "sub_threshold_columns":
                assert fact.info("late_columns") == 40
            else:
                assert fact.info("late_columns") == 2 and fact.info("late_rows") == 2
        assert fact.info("dense_fallbacks") == 0
        ref = oracle.OracleFact(N, kc, kr, kd)
        idx, val = ref.project_nullspace(n, np.arange(n), g)
        assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL, step
        idx, val = ref.solve_lsq(n, np.arange(n), g)
        assert rel_err(aug.solve_lsq(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(W, idx, val)) <= REL_TOL, step
        c = rng.standard_normal(W)
        idx, val = ref.solve_min_norm(n, np.arange(W), c)
        assert rel_err(aug.solve_min_norm(SleqpVec.from_raw(c)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL, step
        for _ in range(3):  # (several solves: the checked first one, then the unchecked steady state)
            b = rng.standard_normal(N)
            ref.solve_dense(b)
            fact.solve(b)
            assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL, step
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        assert scaled_residual(K, fact.solution_raw(0, N), b) <= 1e-12


def test_dense_treatment_falls_back_when_the_working_set_fixes_a_rows_ordinary_variables(fact):
    """with the dense columns taken apart A_s can lose row rank where A keeps it - here a
    row whose entries outside the late column all sit on variables that the working set fixes at their bounds, so
    that its pivot in A_s A_s^T is an exact zero.  K is regular (MA57 / UMFPACK factor it): the factorisation is
    repeated once on a plan that keeps every column in S, nothing is reported, the solution matches the oracle."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 900, 400
    J, cols = synth.with_dense_columns(synth.banded_jacobian(n, m, 8, 60, 23), 1, 9)
    A = J.tocsr()
    r = 123
    others = [c for c in A.indices[A.indptr[r]:A.indptr[r + 1]] if c != cols[0]]
    assert 0 < len(others) <= 8
    for boundary in ("aug_jac", "fact_vtable"):
        aug = StandardAugJac(n, fact, device_assembly=(boundary == "aug_jac"))
        rng = np.random.default_rng(3)
        g = rng.standard_normal(n)
        # first an ordinary working set (the plan with the late column), then the bounds of the row's other variables
        for step in range(2):
            vi = np.full(n, -1, dtype=np.int32)
            if step == 1:
                vi[np.sort(others)] = np.arange(len(others))
            nb = int((vi >= 0).sum())
            ci = (nb + np.arange(m)).astype(np.int32)
            W = nb + m
            before = fact.info("dense_fallbacks")
            aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
            N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
            ref = oracle.OracleFact(N, kc, kr, kd)
            idx, val = ref.project_nullspace(n, np.arange(n), g)
            assert rel_err(aug.project_nullspace(SleqpVec.from_raw(g)).to_raw(), oracle.vec_to_raw(n, idx, val)) <= REL_TOL
            if step == 0:
                assert fact.info("late_columns") == 1 and fact.info("dense_fallbacks") == before
            else:
                assert fact.info("dense_fallbacks") == before + 1


@pytest.mark.timeout(300)
def test_structural_robustness_at_scale():
    """The same at sizes where the old ordering collapsed (one dense row turned 11 tree levels into
    394 at config-4 size; 100 dense columns at n = 2e4 took 132 s of analysis): BASELINE configs[3] plus one dense
    constraint row, and n = 2e4 / m = 1e4 with 100 dense columns and with 200 columns of 300 entries - tree depth,
    fill against SuperLU's minimum-degree ordering of K (scripts/ordering_probe.py superlu: 2.36e6 / 3.60e6), analysis
    time, scaled residual 1e-12, agreement with the oracle's sparse LDL^T (dense variables / rows ordered last) 1e-8."""
    import time

    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    def run(J, late_x=(), late_y=(), levels=None, nnz_bar=None):
        m, n = J.shape
        N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
        f = HipFact(device=0)
        t0 = time.perf_counter()
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
        t_cold = time.perf_counter() - t0
        assert f.info("analysis_s") < 1.5, f.info("analysis_s")
        if levels is not None:
            assert f.info("nlevels") <= levels, f.info("nlevels")
        if nnz_bar is not None:
            assert f.info("nnzL_true") <= nnz_bar, f.info("nnzL_true")
        b = np.random.default_rng(4).standard_normal(N)
        K = synth.kkt_full_matrix(N, cp, ri, vx)
        z = None
        for _ in range(3):
            f.solve(b)
            z = f.solution_raw(0, N)
            assert scaled_residual(K, z, b) <= 1e-12
        lx, ly = np.asarray(late_x, dtype=np.int64), np.asarray(late_y, dtype=np.int64)
        perm = np.r_[np.setdiff1d(np.arange(n), lx), n + np.setdiff1d(np.arange(m), ly), lx, n + ly].astype(np.int32)
        zo = oracle.OracleLdl(N, cp, ri, vx, perm=perm).solve(b)
        assert rel_err(z, zo) <= 1e-8
        # steady-state cost of the unit
        t0 = time.perf_counter()
        for _ in range(5):
            f.set_matrix(SleqpMat(N, N, cp, ri, vx))
            f.solve(b)
            f.solution_raw(0, n)
        dt = (time.perf_counter() - t0) / 5
        f.free()
        return dt, t_cold

    J4 = synth.banded_jacobian(100000, 50000, 20, 200, 0)
    base, _ = run(J4, levels=14)
    Jr, rows = synth.with_dense_rows(J4, 1, 1)
    hub, _ = run(Jr, late_y=rows, levels=14)
    # (the boundary unit with its PCIe hops, five repetitions on a box that other jobs share: a loose bound here, the
    # device-resident ratio - 1.07 - is measured by bench.py: structural_robustness)
    assert hub <= 2.0 * base, (hub, base)
    J2 = synth.banded_jacobian(20000, 10000, 20, 200, 0)
    Jc, cols = synth.with_dense_columns(J2, 100, 1)
    run(Jc, late_x=cols, levels=12, nnz_bar=2 * 2.36e6)
    Jc, cols = synth.with_dense_columns(J2, 200, 1, entries=300)
    run(Jc, late_x=cols, levels=12, nnz_bar=2 * 3.60e6)


def test_pattern_lru_for_set_matrix(fact):
    """Unmodified standard_aug_jac.c in front of the backend: K's pattern changes with the working
    set; patterns seen before are served from the plan LRU (no analysis, no upload, no graph capture)."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 500, 220
    J = synth.banded_jacobian(n, m, 8, 60, 4)
    rng = np.random.default_rng(3)
    fact.set_option("superset_vtable", 0)  # the exact-pattern plan cache (the row dictionary has a test of its own)
    sets = [_ws(n, m, rng, f, bf) for f, bf in ((1.0, 0.0), (0.9, 0.05), (0.5, 0.0))]
    mats = [oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci) for vi, ci, _ in sets]
    for rnd in range(3):
        for N, kc, kr, kd in mats:
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            b = rng.standard_normal(N)
            ref = oracle.OracleFact(N, kc, kr, kd)
            ref.solve_dense(b)
            fact.solve(b)
            assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL
        assert fact.info("analyses") == 3, rnd
    assert fact.info("plan_swaps") >= 6


def test_reduced_matrix_is_the_sparse_product(fact):
    """S = A_W A_W^T from the device (product lists = symbolic SpGEMM, one fixed-order sum per
    structural entry) against the matrix reduced_aug_jac.c:323-377 builds (restated in the oracle) and against
    scipy, in working-set row order - sparse where the reference stores the whole lower triangle."""
    from sleqp_amd.sparse import SleqpMat

    fact.set_option("exact_pattern", 1)  # the pattern of K itself: no working-set maps between S and the caller
    for n, m, kind, frac in [(7, 3, "u", 0.3), (300, 150, "b", 0.1), (2000, 900, "u", 0.05)]:
        J, vi, ci, W = _problem(n, m, kind, frac, 5)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        S = fact.reduced_matrix()
        if W <= 1200:
            rc_, rr_, rd_ = oracle.reduced_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci, W)
            Sref = sp.csc_matrix((rd_, rr_, rc_), shape=(W, W))
            assert abs(S - Sref).max() <= 1e-13 * max(1.0, abs(Sref).max())
        K = sp.csc_matrix((kd, kr, kc), shape=(N, N))
        A = K[n:, :n].tocsr()
        want = sp.tril(A @ A.T, format="csc")
        want.sort_indices()
        assert S.shape == (W, W) and np.all(np.diff(S.indptr) >= 1)
        for j in range(W):  # rows strictly ascending per column, lower triangle (the SleqpMat invariants, mat.c:797-804)
            rows = S.indices[S.indptr[j]:S.indptr[j + 1]]
            assert np.all(np.diff(rows) > 0) and rows[0] == j
        assert abs(S - want).max() <= 1e-13 * max(1.0, abs(want).max())
        # structural entries only: as many as the product has (cancellation aside), far fewer than the dense triangle
        assert S.nnz >= want.nnz and S.nnz <= W * (W + 1) // 2


def test_determinism(fact):
    """Same inputs -> bitwise identical solution (fixed summation order, no atomics)."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(2000, 1000, "b", 0.05, 6)
    N, kc, kr, kd = oracle.fill_aug_jac(2000, 1000, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(0).standard_normal(N)
    outs = []
    for _ in range(3):
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        outs.append(fact.solution_raw(0, N))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


@pytest.mark.parametrize("kind,n,m", [("b", 6000, 3000), ("u", 1500, 700)])
def test_pull_and_scatter_extend_add_agree_bitwise(fact, kind, n, m):
    """The two extend-add paths (separate assembly kernel / gather inside the pivot, panel and
    Schur kernels) add the children's contributions in the same order: identical bits."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(n, m, kind, 0.02, 11)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(1).standard_normal(N)
    fact.set_option("refine_steps", 0)
    # per-level launches in all three runs: the dataflow launch (pull mode only) solves its panels by block
    # substitution against the posted L11, the per-level panel kernel multiplies by inv(L11) - equal to rounding only
    # (test_single_launch_top_of_tree_factorisation_agrees); what is compared bit for bit here is the extend-add
    fact.set_option("factor_top_max", 0)
    outs = []
    for pull in (0, 4, 2):
        fact.set_option("pull_max_children", pull)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        outs.append(fact.solution_raw(0, N))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    assert scaled_residual(K, outs[0], b) <= 1e-9
    fact.set_option("factor_top_max", 128)


def _agree(a, b, tol=1e-11):
    """Two schedules of the same factorisation whose panels are formed by different (backward stable) formulas:
    L21 = P21 inv(L11)^T D^-1 as a product with the inverse, or by block substitution against L11."""
    return np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(a)))


def test_single_launch_top_of_tree_factorisation_agrees(fact):
    """Per-level pivot / panel / Schur launches and the single dataflow launch for the top levels.  Same pivot
    block code and same Schur updates; the panel workgroups of the dataflow launch follow the tiles of L11 the
    pivot workgroup posts and solve by block substitution, the per-level panel kernel multiplies by inv(L11):
    agreement to rounding (no refinement: 1e-11 of the solution), and every schedule is bitwise reproducible."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 12000, 6000
    J = synth.banded_jacobian(n, m, 20, 200, 5)  # every front of the top levels has <= 4 children
    vi, ci, _ = synth.working_set_all_rows(n, m, 0.0, 5)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(3).standard_normal(N)
    fact.set_option("refine_steps", 0)
    outs = []
    for top in (0, 40, 6):
        fact.set_option("factor_top_max", top)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert (fact.info("factor_top_level") < fact.info("nlevels")) == (top > 0)
        for _ in range(3):  # refactor through the cached graph as well
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        outs.append(fact.solution_raw(0, N))
        assert fact.info("solve_timeouts") == 0
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        assert np.array_equal(outs[-1], fact.solution_raw(0, N))  # the schedule itself is deterministic
    assert _agree(outs[0], outs[1]) and _agree(outs[0], outs[2])
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    assert scaled_residual(K, outs[0], b) <= 1e-9


def test_wide_fronts_solved_by_several_workgroups(fact):
    """Fronts with many update rows are split into a head and row slices in the single-launch
    solves (dense Schur complements, BASELINE configs[2] family): against the one-workgroup
    path and the residual."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 3000, 1500
    J = synth.uniform_jacobian(n, m, 8, 21)
    vi, ci, _ = synth.working_set_all_rows(n, m, 0.0, 21)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    b = np.random.default_rng(4).standard_normal(N)
    fact.set_option("refine_steps", 0)
    fact.set_option("solve_fused", 0)  # the head / slice workgroups belong to the two-launch kernels
    outs = []
    for wide in (0, 300):
        fact.set_option("wide_min_rows", wide)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("max_r") >= 700  # the Schur complement of this family is (nearly) dense
        for _ in range(2):
            fact.solve(b)
        outs.append(fact.solution_raw(0, N))
        assert fact.info("solve_timeouts") == 0
        assert scaled_residual(K, outs[-1], b) <= 1e-10
    assert rel_err(outs[1], outs[0]) <= 1e-10
    # in the fused launch a front of more than 1024 rows is several items (row slices with their own copies of
    # their rows of the solve panel; the backward items of the slices post partial sums that slice 0 adds up):
    # repeated solves and a refactorisation in between (the polled slots must all be back at the sentinel)
    fact.set_option("solve_fused", 1)
    for rep in range(2):
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("fused_solve") == 1
        for _ in range(3):
            fact.solve(b)
            z = fact.solution_raw(0, N)
            assert rel_err(z, outs[0]) <= 1e-10 and fact.info("solve_timeouts") == 0
            assert scaled_residual(K, z, b) <= 1e-10
        b2 = np.random.default_rng(6 + rep).standard_normal(N)
        fact.solve(b2)
        assert scaled_residual(K, fact.solution_raw(0, N), b2) <= 1e-10


def test_top_of_tree_solve_variants_agree_bitwise(fact):
    """Level-by-level solve launches, the single-launch top-of-tree kernels, and their
    panel-prefetching variant run the same arithmetic in the same order: identical bits."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 12)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(2).standard_normal(N)
    fact.set_option("refine_steps", 0)
    fact.set_option("solve_fused", 0)
    outs = []
    for top_max, prefetch in ((0, 0), (256, 0), (1024, 1)):
        fact.set_option("top_max_fronts", top_max)
        fact.set_option("top_prefetch", prefetch)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        outs.append(fact.solution_raw(0, N))
        assert (fact.info("top_level") < fact.info("nlevels")) == (top_max > 0)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    assert scaled_residual(K, outs[0], b) <= 1e-9
    # the fused forward + backward launch works on another form of the factor ([X; -L21 X] instead of
    # [X; L21]): same solution to rounding, deterministic from run to run
    fact.set_option("solve_fused", 1)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("fused_solve") == 1
    fused = []
    for _ in range(4):
        fact.solve(b)
        fused.append(fact.solution_raw(0, N))
    # (from the second solve of a factorisation on the top levels of the tree are one dense block: another summation
    # order - each form is deterministic from run to run, the two agree to rounding)
    assert fact.info("top_block_active") == (1 if fact.info("top_block_cols") > 0 else 0)
    assert np.array_equal(fused[1], fused[2]) and np.array_equal(fused[1], fused[3])
    assert rel_err(fused[0], fused[1]) <= 1e-13
    fact.set_option("top_block_after", 0)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    plain = []
    for _ in range(3):
        fact.solve(b)
        plain.append(fact.solution_raw(0, N))
    assert np.array_equal(plain[0], fused[0]) and np.array_equal(plain[0], plain[1]) and np.array_equal(plain[0], plain[2])
    assert rel_err(fused[0], outs[0]) <= 1e-11 and scaled_residual(K, fused[0], b) <= 1e-9


def test_solve_panels_built_inside_the_top_of_tree_launch(fact):
    """The solve panels S = [X; -L21 X] are built by filler items of the single-launch top-of-tree
    factorisation (role 3 of k_factor_top; the root writes its own from LDS) instead of by a launch of
    their own behind it.  Same arithmetic either way: the fused solve on them must give the same bits,
    for fronts below the launch, fronts inside it (which wait for their pivot and panel workgroups)
    and the root, over repeated factorisations (the counters they wait on are cleared per factorisation)."""
    from sleqp_amd.sparse import SleqpMat

    for kind, n, m, opts in [("b", 20000, 10000, {}), ("b", 20000, 10000, {"spanel_fold_room": 8}), ("u", 3000, 1500, {})]:
        J, vi, ci, _ = _problem(n, m, kind, 0.0, 11)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        rng = np.random.default_rng(2)
        rhs = [rng.standard_normal(N) for _ in range(2)]
        outs = {}
        for fold in (1, 0):
            fact.set_option("refine_steps", 0)
            fact.set_option("solve_fused", 1)
            fact.set_option("spanel_fold", fold)
            fact.set_option("spanel_fold_room", opts.get("spanel_fold_room", 224))
            res = []
            for rep in range(2):
                fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
                assert fact.info("fused_solve") == 1.0
                if fact.info("factor_top_count") > 0:
                    assert fact.info("spanel_folded") == float(fold)
                for b in rhs:
                    fact.solve(b)
                    res.append(fact.solution_raw(0, N))
                assert fact.info("solve_timeouts") == 0 and fact.info("dataflow_fallbacks") == 0
            outs[fold] = res
        for a, b_ in zip(outs[1], outs[0]):
            assert np.array_equal(a, b_)
        assert scaled_residual(K, outs[1][0], rhs[0]) <= 1e-9
    fact.set_option("spanel_fold", 1)
    fact.set_option("spanel_fold_room", 224)
    fact.set_option("refine_steps", 1)


def test_right_hand_side_formed_inside_the_solve_launch(fact):
    """The forward items of the single-launch solve form their own rows of t = A^ b_x - D b_y (16 lanes per
    row, the partial sums and the shuffle tree of k_rhs_saddle) instead of reading what a launch in front of
    them left: same bits, with and without working-set maps, first pass and correction passes."""
    fact.set_option("top_block_breakeven", 0)  # (bits are compared ACROSS factorisations: the top block forms at the same solve in each)
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 4)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    rng = np.random.default_rng(12)
    rhs = [rng.standard_normal(N) for _ in range(3)]
    outs = {}
    for fused in (1, 0):
        fact.set_option("rhs_fused", fused)
        fact.set_option("refine_adaptive", 0)  # every in-graph pass runs: the correction pass is exercised too
        fact.set_option("refine_steps", 1)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("fused_solve") == 1.0
        res = []
        for b in rhs:
            fact.solve(b)
            res.append(fact.solution_raw(0, N))
        outs[fused] = res
    for a, b_ in zip(outs[1], outs[0]):
        assert np.array_equal(a, b_)
    assert scaled_residual(K, outs[1][0], rhs[0]) <= 1e-9
    fact.set_option("rhs_fused", 1)
    fact.set_option("refine_adaptive", 1)


def test_general_symmetric_plan_is_not_reused_for_saddle_values(fact):
    """The saddle classification needs the structure [I A^T; A 0] AND a unit diagonal.  The same pattern with
    other diagonal values is analysed as a general symmetric matrix - and that plan must not be picked up again
    when values with a unit diagonal come back (K is indefinite; only the constrained pivot order of the saddle
    mode makes static pivots safe), nor the saddle plan for the general values."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(400, 150, "b", 0.0, 2)
    N, kc, kr, kd = oracle.fill_aug_jac(400, 150, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    b = np.random.default_rng(3).standard_normal(N)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("saddle") == 1.0
    fact.solve(b)
    z0 = fact.solution_raw(0, N)
    # diagonally dominant SPD-like values on the same pattern: trailing columns are empty, so only the x block
    spd = kd.copy()
    spd[kc[:-1][:400]] = 50.0
    fact.set_matrix(SleqpMat(N, N, kc, kr, spd))
    assert fact.info("saddle") == 0.0
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("saddle") == 1.0
    fact.solve(b)
    assert np.array_equal(fact.solution_raw(0, N), z0)
    assert scaled_residual(K, z0, b) <= 1e-9


def test_deferred_refinement_verdict(fact):
    """A solve whose graph carries no correction pass (steady state of a well-conditioned factorisation) leaves
    its verdict to a workgroup of the NEXT solve's tree launch; entry points that need it earlier launch it
    themselves, and a refactorisation flushes it first (it is judged against that factorisation's pivots).
    Same solutions, same backward errors and pass counts as with a verdict launch behind every solve, over
    solves, refactorisations with different values and checks in between."""
    fact.set_option("top_block_breakeven", 0)  # (bits are compared ACROSS factorisations: the top block forms at the same solve in each)
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 6)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    rng = np.random.default_rng(21)
    rhs = [rng.standard_normal(N) for _ in range(6)]
    scale = [1.0, 3.0, 0.25]
    diag = kc[:-1][:20000]  # position of the unit diagonal entry of every x column (scaled values keep the saddle form)
    logs = {}
    fact.set_option("refine_check_every", 1)  # every solve judged (the default checks every 8th once well-conditioned)
    for lazy in (1, 0):
        fact.set_option("decide_lazy", lazy)
        log = []
        for rep, sc in enumerate(scale):
            vals = kd * sc
            vals[diag] = 1.0
            fact.set_matrix(SleqpMat(N, N, kc, kr, vals))
            assert fact.info("saddle") == 1.0
            for i, b in enumerate(rhs):
                fact.solve(b)
                if (i + rep) % 3 == 2:  # the host asks for the verdict now and then, not after every solve
                    fact.check()
                    log.append((fact.solution_raw(0, N), fact.info("last_omega"), fact.info("last_iters"), fact.info("last_status")))
            fact.check()
            log.append((fact.solution_raw(0, N), fact.info("last_omega"), fact.info("last_iters"), fact.info("last_status")))
        logs[lazy] = log
    assert len(logs[0]) == len(logs[1])
    for (za, oa, ia, sa), (zb, ob, ib, sb) in zip(logs[1], logs[0]):
        assert np.array_equal(za, zb)
        assert oa == ob and ia == ib and sa == sb
        assert sa == 0.0 and oa <= 1e-12
    fact.set_option("decide_lazy", 1)


def test_dense_chain_above_the_bushy_part_of_the_tree(fact):
    """SURVEY 8(d) config 4b in small: 20 nonzeros per row at uniform columns - A A^T fills in completely and the
    chain of 128-column fronts over the dense trailing matrix sits ABOVE the front that joins the subtrees.  Those
    levels have thousands of Schur tiles per front: they must not join the dataflow launch (at config 4b's size that
    would be 10^8 workgroups - the launch is refused) but run pivot + panel as the small dataflow launch and the Schur
    update at three workgroups per CU; the update matrices share memory along the chain."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 16000, 8000
    J = synth.uniform_jacobian(n, m, 20, 0)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    assert fact.info("nlevels") > 40 and fact.info("fused_solve") == 1
    # the launch holds the top of the chain only (fronts of up to 1024 update rows: 8 levels of <= 128 columns)
    assert fact.info("factor_top_count") < 40000
    K = synth.kkt_full_matrix(N, cp, ri, vx)
    rng = np.random.default_rng(3)
    for _ in range(2):
        b = rng.standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        assert scaled_residual(K, z, b) <= 1e-12
    assert fact.info("solve_timeouts") == 0 and fact.info("dataflow_fallbacks") == 0
    # the chain is updated two fronts at a time (k_front_schur_pair: the second front's update matrix straight from
    # the grandchild's with both panels - half the trailing-matrix traffic); one front at a time agrees to rounding
    assert fact.info("chain_pairs") >= 10
    fact.set_option("chain_pairs", 0)
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    assert fact.info("chain_pairs") == 0
    fact.solve(b)
    z1f = fact.solution_raw(0, N)
    assert scaled_residual(K, z1f, b) <= 1e-12 and _agree(z1f, z)
    fact.set_option("chain_pairs", 1)
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    # linearity at this size: K (z1 + 2 z2) = b1 + 2 b2
    b1, b2 = rng.standard_normal(N), rng.standard_normal(N)
    fact.solve(b1)
    z1 = fact.solution_raw(0, N).copy()
    fact.solve(b2)
    z2 = fact.solution_raw(0, N).copy()
    fact.solve(b1 + 2.0 * b2)
    assert rel_err(fact.solution_raw(0, N), z1 + 2.0 * z2) <= 1e-9


def test_solve_item_order_and_slicing_threshold(fact):
    """The fused solve launch runs one item per CU, so a level with more items than CUs runs in rounds: items are
    ordered biggest front first inside a level (explicit backward order behind the forward one) and a front is cut
    into row slices once a thread would hold more than `solve_whole_max` panel entries.  Order is a permutation of
    workgroups (same bits); slicing changes the association of the backward sums (agreement to rounding); different
    numbers of x-update workgroups leave the bits alone."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 40000, 20000  # the headline configuration's band at 40 % of its size: fronts of ~350 rows x 128 columns
    J = synth.banded_jacobian(n, m, 20, 200, 5)
    vi, ci, W = synth.working_set_all_rows(n, m, 0.0, 1)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    rng = np.random.default_rng(4)
    rhs = [rng.standard_normal(N) for _ in range(3)]
    outs, items = {}, {}
    for key, (srt, whole, xb) in {"default": (1, 48, 256), "plan order": (0, 48, 256), "whole": (1, 64, 256),
                                  "sliced": (1, 32, 256), "few x blocks": (1, 48, 7)}.items():
        fact.set_option("solve_sorted", srt)
        fact.set_option("solve_whole_max", whole)
        fact.set_option("xupd_blocks", xb)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        res = []
        for b in rhs:
            fact.solve(b)
            res.append(fact.solution_raw(0, N))
        assert fact.info("solve_timeouts") == 0 and fact.info("fused_solve") == 1
        outs[key], items[key] = res, fact.info("solve_items")
    fact.set_option("solve_sorted", 1)
    fact.set_option("solve_whole_max", 48)
    fact.set_option("xupd_blocks", 256)
    assert items["sliced"] > items["default"] >= items["whole"]
    for a, b_ in zip(outs["default"], outs["plan order"]):
        assert np.array_equal(a, b_)
    for a, b_ in zip(outs["default"], outs["few x blocks"]):
        assert np.array_equal(a, b_)
    for key in ("whole", "sliced"):
        for a, b_ in zip(outs["default"], outs[key]):
            assert _agree(a, b_)
    assert scaled_residual(K, outs["sliced"][0], rhs[0]) <= 1e-12


def test_x_update_inside_the_solve_launch(fact):
    """The back substitution of the leaf columns (z_x = b~_x - A^^T y^, z_y = D y^) by the last workgroups of the
    fused solve launch, polling the posted solution copy, against the separate launch behind the tree: same lanes,
    same shuffle tree, same bits - with active bounds (working-set maps), over a sequence of right-hand sides
    (the exchange slots alternate by launch parity, now advanced by the launch's own last workgroup) and with
    correction passes (accumulating mode)."""
    fact.set_option("top_block_breakeven", 0)  # (bits are compared ACROSS factorisations: the top block forms at the same solve in each)
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 6000, 2800
    J = synth.banded_jacobian(n, m, 12, 90, 13)
    rng = np.random.default_rng(8)
    vi, ci, W = _ws(n, m, rng, 0.95, 0.05)
    rhs = [rng.standard_normal(n + W) for _ in range(7)]
    outs = {}
    for fused in (1, 0):
        fact.set_option("xupd_fused", fused)
        fact.set_option("refine_check_every", 3)
        aug = StandardAugJac(n, fact)
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        res = []
        for b in rhs:
            fact.solve(b)
            res.append(fact.solution_raw(0, n + W))
        fact.set_option("refine_adaptive", 0)  # correction passes run unconditionally: the accumulating form
        fact.set_option("refine_steps", 2)
        fact.solve(rhs[0])
        res.append(fact.solution_raw(0, n + W))
        fact.set_option("refine_steps", 1)
        fact.set_option("refine_adaptive", 1)
        assert fact.info("solve_timeouts") == 0
        outs[fused] = res
    for a, b_ in zip(outs[1], outs[0]):
        assert np.array_equal(a, b_)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    assert scaled_residual(K, outs[1][0], rhs[0]) <= 1e-12
    fact.set_option("xupd_fused", 1)
    fact.set_option("refine_check_every", 8)


def test_residual_checked_on_every_kth_solve_only(fact):
    """Once a factorisation has been judged well-conditioned, the residual b - K z is taken on every k-th solve only
    (refine_check_every, default 8; the reference's MA57 path never checks, fact_ma57.c:18): same bits as with a
    check behind every solve, the checked ones report the same backward error, and an ill-conditioned factorisation
    (correction passes in the graph) keeps checking every solve."""
    fact.set_option("top_block_breakeven", 0)  # (bits are compared ACROSS factorisations: the top block forms at the same solve in each)
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 6)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    rng = np.random.default_rng(33)
    rhs = [rng.standard_normal(N) for _ in range(20)]
    outs = {}
    for every in (1, 8):
        fact.set_option("refine_check_every", every)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        res = []
        for b in rhs:
            fact.solve(b)
            res.append(fact.solution_raw(0, N))
        outs[every] = res
        assert fact.info("last_status") == 0.0 and fact.info("last_omega") <= 1e-12
    for a, b_, rh in zip(outs[1], outs[8], rhs):
        assert np.array_equal(a, b_)
        assert scaled_residual(K, a, rh) <= 1e-12
    fact.set_option("refine_check_every", 8)
    # the interval grows while the checks keep passing (refine_check_backoff 2, refine_check_max 64): solves 1, 9, 25,
    # 57, 121, 185 of a factorisation are checked; a new factorisation starts over at 8; backoff 1 keeps the interval
    for backoff, want in ((2, 6), (1, 25)):
        fact.set_option("refine_check_backoff", backoff)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        c0 = fact.info("num_checked")
        for i in range(200):
            fact.solve(rhs[i % len(rhs)])
        last = fact.solution_raw(0, N)
        assert fact.info("num_checked") - c0 == want, (backoff, fact.info("num_checked") - c0)
        assert np.array_equal(last, outs[1][199 % len(rhs)])
        assert fact.info("refine_check_interval") == (64 if backoff == 2 else 8)
    fact.set_option("refine_check_backoff", 2)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("refine_check_interval") == 8


def test_dense_chain_levels_as_small_dataflow_launches(fact):
    """Single-front levels of a dense chain below the top-of-tree launch run their pivot and panel items as ONE
    small dataflow launch (panel workgroups following the posted pivot block) instead of two launches; the sliced
    fronts of the two-launch solves exchange posted data instead of flags.  Agreement with the per-level kernels to
    rounding (substitution against the posted L11 / product with inv(L11)), bitwise reproducible over repeated
    factorisations (the posted slots are refilled with the sentinel each time)."""
    fact.set_option("top_block_breakeven", 0)  # (bits are compared ACROSS factorisations: the top block forms at the same solve in each)
    from sleqp_amd.sparse import SleqpMat

    J = synth.uniform_jacobian(3000, 1500, 10, 9)  # dense Schur complement: a chain of single-front levels
    N, kc, kr, kd = synth.kkt_lower_from_jacobian(J)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    rng = np.random.default_rng(5)
    rhs = [rng.standard_normal(N) for _ in range(2)]
    outs = {}
    fact.set_option("factor_top_levels", 4)  # most of the chain stays below the top-of-tree launch
    for fuse in (1, 0):
        fact.set_option("chain_fuse", fuse)
        fact.set_option("refine_steps", 0)
        res = []
        for rep in range(3):
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            assert (fact.info("chain_levels_fused") >= 3) == bool(fuse)
            for b in rhs:
                fact.solve(b)
                res.append(fact.solution_raw(0, N))
            assert fact.info("solve_timeouts") == 0 and fact.info("dataflow_fallbacks") == 0
        outs[fuse] = res
    for a, b_ in zip(outs[1], outs[0]):
        assert _agree(a, b_, 1e-10)
    assert np.array_equal(outs[1][0], outs[1][2]) and np.array_equal(outs[1][1], outs[1][5])
    assert scaled_residual(K, outs[1][0], rhs[0]) <= 1e-8
    fact.set_option("chain_fuse", 1)
    fact.set_option("refine_steps", 1)
    fact.set_option("factor_top_levels", 1 << 20)


def test_device_resident_loop_of_one_solve_per_factorisation():
    """A loop `refactor_device; solve_device` with no synchronising entry point in between (bench.py's unit) never
    reaches the second-solve peek that drops the correction pass from the solve graphs; the refactorisation looks at
    the last verdict that has come back instead (`factor_hint_peek`).  Well conditioned K: the pass is dropped after the
    first verdict is back, every solution still meets the tolerance.  A graded K (row scales over eight decades) in the
    same loop on a K with nearly parallel rows: if the first pass alone does not meet a quarter of the tolerance the
    pass stays, and `check` (a synchronising entry point) finishes whatever was left."""
    import ctypes as C

    import scipy.sparse as sp

    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    fact = HipFact(device=0)  # (a handle of its own: no plan state of earlier tests)
    # device buffers straight from the HIP runtime the library is linked against (torch brings a runtime of its own,
    # which does not find the device once this one is initialised)
    hip = C.CDLL("libamdhip64.so")

    def to_device(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return p

    def to_host(p, n):
        out = np.empty(n)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes), 2) == 0
        return out

    n, m = 20000, 10000
    J0 = synth.banded_jacobian(n, m, 20, 200, 5)
    rng = np.random.default_rng(3)
    Jr = sp.csr_matrix(J0)  # nearly parallel rows: another pattern (a plan state of its own), kappa ~ 1e7
    extra = Jr[rng.choice(Jr.shape[0], 40, replace=False)].copy()
    extra.data = extra.data * (1.0 + 1e-7 * rng.standard_normal(extra.data.size))
    Jg = sp.vstack([Jr, extra]).tocsc()
    Jg.sort_indices()
    for name, J, expect_drop in (("plain", J0, True), ("parallel_1e-7", Jg, None)):
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(J)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        fact.set_option("refine_steps", 1)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        d_vals = to_device(np.ascontiguousarray(kd, dtype=np.float64))
        b = rng.standard_normal(N)
        d_rhs = to_device(b)
        d_sol = to_device(np.zeros(N))
        inline = []
        for it in range(6):
            fact.refactor_device(d_vals.value)
            inline.append(int(fact.info("refine_inline")))
            fact.solve_device(d_rhs.value, d_sol.value)
            assert hip.hipDeviceSynchronize() == 0  # (the device is done; the library has not been asked anything)
        assert inline[0] == 1, (name, inline)  # nothing has come back for this plan yet
        if expect_drop:
            assert inline[-1] == 0, (name, inline)
        passes0 = fact.info("num_passes")
        fact.check()
        x = to_host(d_sol, N)
        # (against what the host boundary gives on the same K: nearly parallel rows leave the caller's residual above
        # the tolerance of the equilibrated system the device controls)
        fact.solve(b)
        ref = scaled_residual(K, fact.solution_raw(0, N), b)
        assert scaled_residual(K, x, b) <= max(4.0 * ref, RESID_TOL), (name, scaled_residual(K, x, b), ref)
        if not expect_drop and inline[-1] == 0:
            # the hint said "well conditioned": then the device's own verdict of the last solve must say so too
            assert fact.info("num_passes") == passes0 and fact.info("last_iters") == 0, (name, inline)
        assert fact.info("solve_timeouts") == 0 and fact.info("dataflow_fallbacks") == 0
        if expect_drop:
            g = HipFact(device=0)  # without the peek the loop keeps the pass
            g.set_option("factor_hint_peek", 0)
            g.set_matrix(SleqpMat(N, N, kc, kr, kd))
            for it in range(3):
                g.refactor_device(d_vals.value)
                assert int(g.info("refine_inline")) == 1, name
                g.solve_device(d_rhs.value, d_sol.value)
                assert hip.hipDeviceSynchronize() == 0
            g.check()
            g.free()
        for p_ in (d_vals, d_rhs, d_sol):
            assert hip.hipFree(p_) == 0
    fact.free()


def test_top_block_forms_when_it_pays(fact):
    """Forming the top block costs ~0.4 ms and saves ~9 us per solve.  A fresh plan forms it at the second solve (as
    if many solves were to follow); a factorisation that follows one with few solves waits until it has seen
    `top_block_breakeven` solves itself; one that follows a long one forms it at the second solve again.  Solutions
    agree with and without the block to the residual tolerance."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 6)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    rng = np.random.default_rng(2)
    be = 48

    def solves(k):
        for _ in range(k):
            b = rng.standard_normal(N)
            fact.solve(b)
            assert scaled_residual(K, fact.solution_raw(0, N), b) <= RESID_TOL

    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    solves(3)
    assert fact.info("top_block_cols") > 0 and fact.info("top_block_active") == 1  # fresh plan
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))  # ... whose previous factorisation saw three solves
    solves(3)
    assert fact.info("top_block_active") == 0
    solves(be - 4)
    assert fact.info("top_block_active") == 0
    solves(2)
    assert fact.info("top_block_active") == 1
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))  # ... and now one that saw 49
    solves(2)
    assert fact.info("top_block_active") == 1


def test_solve_sequence_with_changing_right_hand_sides(fact):
    """The single-launch solve sweeps exchange vectors element by element through slots that the
    opposite sweep puts back to a sentinel.  A slot that was not put back would hand a value of
    the PREVIOUS solve to the next one: a sequence of different right-hand sides (mixed front
    kinds: prefetching, generic and wide fronts, levels below the launch) against the
    level-by-level path (bit for bit, except that wide fronts sum their slices in another order),
    with a refactorisation in between."""
    from sleqp_amd.sparse import SleqpMat

    cases = [("b", 20000, 10000, {}), ("u", 3000, 1500, {"wide_min_rows": 300}), ("b", 20000, 10000, {"top_max_fronts": 200})]
    for kind, n, m, opts in cases:
        J, vi, ci, _ = _problem(n, m, kind, 0.0, 5)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        rng = np.random.default_rng(8)
        rhs = [rng.standard_normal(N) * 10.0 ** rng.integers(-3, 4) for _ in range(5)]
        rhs.insert(2, np.zeros(N))
        fact.set_option("refine_steps", 0)
        fact.set_option("solve_fused", 0)
        outs = {}
        for top_max in (0, opts.get("top_max_fronts", 1024)):
            fact.set_option("top_max_fronts", top_max)
            fact.set_option("wide_min_rows", opts.get("wide_min_rows", 1024))
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            res = []
            for i, b in enumerate(rhs):
                if i == 3:
                    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))  # numeric refactorisation in the middle
                fact.solve(b)
                res.append(fact.solution_raw(0, N))
                assert fact.info("solve_timeouts") == 0
                assert scaled_residual(K, res[-1], b) <= 1e-9
            outs[top_max] = res
        a, c = outs.values()
        for x, y in zip(a, c):
            assert np.array_equal(x, y) if "wide_min_rows" not in opts else rel_err(x, y) <= 1e-10
        # the same sequence through the fused launch (three kinds of sentinel slots: update vectors, x^, ysol)
        fact.set_option("solve_fused", 1)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("fused_solve") == (1 if fact.info("max_r") <= 1024 else 0)
        for i, b in enumerate(rhs):
            if i == 3:
                fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            fact.solve(b)
            z = fact.solution_raw(0, N)
            assert fact.info("solve_timeouts") == 0
            assert rel_err(z, a[i]) <= 1e-10 and scaled_residual(K, z, b) <= 1e-9
    fact.set_option("top_max_fronts", 1024)
    fact.set_option("wide_min_rows", 1024)


def test_non_finite_right_hand_side_does_not_stall_the_sweeps(fact):
    """NaN / Inf in the right-hand side flow through the polled element exchange like any other
    value (only the all-ones bit pattern is the sentinel, and arithmetic never produces it): no
    timeout, non-finite output, and the next ordinary solve is exact again."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 3)
    N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    fact.set_option("refine_steps", 0)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    b = np.random.default_rng(6).standard_normal(N)
    for _ in range(3):  # (from the second solve of a factorisation on the top levels run as one dense block: the steady state)
        fact.solve(b)
    good = fact.solution_raw(0, N)
    sentinel_nan = np.frombuffer(np.uint64(0xFFFFFFFFFFFFFFFF).tobytes(), dtype=np.float64)[0]  # the slots' own bit pattern
    for poison in (np.nan, np.inf, -np.inf, sentinel_nan):
        bad = b.copy()
        bad[N - 7] = poison   # a constraint row: enters the tree at a leaf and reaches the root
        bad[11] = poison
        fact.solve(bad)
        out = fact.solution_raw(0, N)
        assert fact.info("solve_timeouts") == 0
        assert not np.all(np.isfinite(out))
        fact.solve(b)
        again = fact.solution_raw(0, N)
        assert fact.info("solve_timeouts") == 0
        assert np.array_equal(again, good)
    assert scaled_residual(K, good, b) <= 1e-9


@pytest.mark.parametrize("shape", ["banded", "dense_chain"])
def test_dataflow_timeout_falls_back_to_per_level_launches(fact, shape):
    """The single-launch kernels assume in-order workgroup dispatch; their waits are bounded.  A timeout (injected
    here through the test hook) switches the handle to the per-level launches for good and repeats the work:
    the caller sees a correct factorisation / solution, not an error.  (dense_chain: tall fronts, i.e. row-sliced
    solve items and chain levels as small dataflow launches before the fallback.)"""
    from sleqp_amd.sparse import SleqpMat

    if shape == "banded":
        J, vi, ci, _ = _problem(20000, 10000, "b", 0.0, 17)
        N, kc, kr, kd = oracle.fill_aug_jac(20000, 10000, J.indptr, J.indices, J.data, vi, ci)
    else:
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(synth.uniform_jacobian(3000, 1500, 10, 9))
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    b = np.random.default_rng(5).standard_normal(N)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("factor_top_level") < fact.info("nlevels") and fact.info("fused_solve") == 1
    fact.solve(b)
    good = fact.solution_raw(0, N)
    # (1) during a factorisation
    fact.set_option("debug_fake_timeout", 1)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert fact.info("no_dataflow") == 1 and fact.info("dataflow_fallbacks") == 1
    fact.solve(b)
    z = fact.solution_raw(0, N)
    assert rel_err(z, good) <= 1e-11 and scaled_residual(K, z, b) <= RESID_TOL
    # (2) during a solve, on a fresh handle
    from sleqp_amd.fact import HipFact

    f2 = HipFact(device=0)
    f2.set_matrix(SleqpMat(N, N, kc, kr, kd))
    f2.solve(b)
    f2.set_option("debug_fake_timeout", 1)
    z2 = f2.solution_raw(0, N)  # the verdict is read here: the solve is repeated through the per-level kernels
    assert f2.info("no_dataflow") == 1
    assert rel_err(z2, good) <= 1e-11 and scaled_residual(K, z2, b) <= RESID_TOL
    f2.solve(2.0 * b)
    assert rel_err(f2.solution_raw(0, N), 2.0 * good) <= 1e-11
    f2.free()
    # (3) in a device-resident loop (bench.py's unit: refactor_device + solve_device, nothing synchronises in between):
    # hipfact_check finds the timeout - it may be the unchecked factorisation's - and repeats factorisation and solve on
    # the per-level path, into the caller's device buffer (two ranks sharing one GPU have been seen to time out: the
    # bench must not die of it)
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")  # (device buffers from the HIP runtime the library is linked against)

    def to_device(a):
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes), 1) == 0
        return ptr

    def to_host(ptr, count):
        out = np.empty(count)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, C.c_size_t(out.nbytes), 2) == 0
        return out

    f3 = HipFact(device=0)
    f3.set_matrix(SleqpMat(N, N, kc, kr, kd))
    d_vals = to_device(np.ascontiguousarray(kd, dtype=np.float64))
    d_b = to_device(np.ascontiguousarray(b))
    d_z = to_device(np.zeros(N))
    for _ in range(2):
        f3.refactor_device(d_vals.value)
        f3.solve_device(d_b.value, d_z.value)
    f3.set_option("debug_fake_timeout", 1)
    f3.check()
    assert f3.info("no_dataflow") == 1 and f3.info("dataflow_fallbacks") == 1
    z3 = to_host(d_z, N)
    assert rel_err(z3, good) <= 1e-11 and scaled_residual(K, z3, b) <= RESID_TOL
    f3.refactor_device(d_vals.value)
    f3.solve_device(d_b.value, d_z.value)
    f3.check()
    assert rel_err(to_host(d_z, N), good) <= 1e-11
    f3.free()
    for ptr in (d_vals, d_b, d_z):
        hip.hipFree(ptr)
    # (4) "for good" ends: after HIPFACT_DATAFLOW_RETRY more factorisations (256 by default, doubled with every
    # fallback) the handle tries the single-launch kernels again - what ended the wait is another tenant of the GPU,
    # and that one need not stay.  The exchange slots the per-level solves used as plain memory are sentinels again.
    import os

    os.environ["HIPFACT_DATAFLOW_RETRY"] = "3"
    try:
        f4 = HipFact(device=0)
    finally:
        del os.environ["HIPFACT_DATAFLOW_RETRY"]
    f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
    f4.set_option("debug_fake_timeout", 1)
    f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert f4.info("no_dataflow") == 1 and f4.info("dataflow_rearmed") == 0
    seen = []
    for rep in range(5):
        f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
        seen.append(int(f4.info("no_dataflow")))
        for scale in (1.0, -0.5):
            f4.solve(scale * b)
            z4 = f4.solution_raw(0, N)
            assert rel_err(z4, scale * good) <= 1e-11 and scaled_residual(K, z4, scale * b) <= RESID_TOL
    assert seen[0] == 1 and seen[-1] == 0 and f4.info("dataflow_rearmed") == 1 and f4.info("dataflow_fallbacks") == 1, seen
    # (the second fallback doubles the distance)
    f4.set_option("debug_fake_timeout", 1)
    f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert f4.info("no_dataflow") == 1 and f4.info("dataflow_fallbacks") == 2
    for rep in range(4):
        f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
    assert f4.info("no_dataflow") == 1
    for rep in range(4):
        f4.set_matrix(SleqpMat(N, N, kc, kr, kd))
    f4.solve(b)
    assert f4.info("no_dataflow") == 0 and f4.info("dataflow_rearmed") == 2
    assert rel_err(f4.solution_raw(0, N), good) <= 1e-11
    f4.free()


def test_pull_with_more_children_than_one_descriptor_block(fact):
    """Fronts with more than four children (amalgamation unconstrained): the gathers walk a chain of
    descriptor blocks, in child order - identical bits to the scatter kernel."""
    from sleqp_amd.sparse import SleqpMat

    n, m = 20000, 10000
    J = synth.banded_jacobian(n, m, 12, 80, 13)
    vi, ci, _ = synth.working_set_all_rows(n, m, 0.0, 13)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(1).standard_normal(N)
    fact.set_option("refine_steps", 0)
    fact.set_option("max_children", 0)
    outs = []
    for pull in (0, 4):
        fact.set_option("pull_max_children", pull)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        fact.solve(b)
        outs.append(fact.solution_raw(0, N))
    assert np.array_equal(outs[0], outs[1])
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    assert scaled_residual(K, outs[0], b) <= 1e-9


@pytest.mark.parametrize("workload", ["banded_n1e5_m5e4", "uniform_n1e4_m5e3"])
def test_full_size_properties(fact, workload):
    """BASELINE.json configs[3] / configs[2] at full size: properties that do not need the dense oracle."""
    from bench import make_problem
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    J, N, cp, ri, vx, b = make_problem(workload, 0)
    n = J.shape[1]
    K = synth.kkt_full_matrix(N, cp, ri, vx)
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    # (1) residual of a dense solve
    fact.solve(b)
    z = fact.solution_raw(0, N)
    assert scaled_residual(K, z, b) <= RESID_TOL
    # (2) agreement with the oracle's sparse LDL^T (CPU baseline, BASELINE.md: <= 1e-8); for the
    # dense-Schur family this is a 5000^2 dense factorisation inside the simplicial code (~15 s)
    x = oracle.OracleLdl(N, cp, ri, vx).solve(b)
    assert rel_err(z, x) <= 1e-8
    # (3) linearity
    b2 = np.random.default_rng(9).standard_normal(N)
    fact.solve(b2)
    z2 = fact.solution_raw(0, N)
    fact.solve(2.0 * b - 3.0 * b2)
    assert rel_err(fact.solution_raw(0, N), 2.0 * z - 3.0 * z2) <= 1e-9
    # (4) projection: A P g = 0, P idempotent, (g - P g) in range(A^T)
    g = np.zeros(N)
    g[:n] = np.random.default_rng(3).standard_normal(n)
    fact.solve(SleqpVec.from_raw(g))
    Pg = fact.solution_raw(0, n)
    A = J.tocsr()
    assert np.abs(A @ Pg).max() <= 1e-9 * np.abs(A).sum(axis=1).max() * np.abs(g).max()
    g2 = np.zeros(N)
    g2[:n] = Pg
    fact.solve(SleqpVec.from_raw(g2))
    assert rel_err(fact.solution_raw(0, n), Pg) <= 1e-9
    # (5) min-norm solve satisfies A x = rhs
    rhs = np.zeros(N)
    rhs[n:] = np.random.default_rng(4).standard_normal(N - n)
    fact.solve(SleqpVec.from_raw(rhs))
    x = fact.solution_raw(0, n)
    assert np.abs(A @ x - rhs[n:]).max() <= 1e-9 * max(1.0, np.abs(x).max()) * np.abs(A).sum(axis=1).max()


@pytest.mark.parametrize("boundary", ["vtable", "assembly", "vtable_exact"])
def test_full_size_with_active_bounds(fact, boundary):
    """SURVEY 8(d)'s variant "10 % random active bounds" at BASELINE configs[3]'s full size.
    The reference puts the unit rows of active bounds first in every working set (working_set.c:139, 167-168;
    standard_aug_jac.c:163-185); here they are eliminated in front of the analysis on every boundary - the row
    dictionary behind the plain vtable, the device assembly, and the exact-rows path of superset_vtable = 0 - so the
    tree keeps the depth of the constraint rows' own.  Against the oracle's sparse LDL^T and the residual; then the set
    of active bounds changes by +-1 % of n per call with a single analysis."""
    from sleqp_amd.fact import StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 100000, 50000
    J = synth.banded_jacobian(n, m, 20, 200, 0)
    rng = np.random.default_rng(5)
    if boundary == "vtable_exact":
        fact.set_option("superset_vtable", 0)
    aug = StandardAugJac(n, fact, device_assembly=(boundary == "assembly"))

    def one(vi, ci, against_oracle):
        aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        if boundary == "assembly":  # the device's own fill_aug_jac, bit for bit
            assert np.array_equal(aug.K.cols, kc) and np.array_equal(aug.K.rows, kr) and np.array_equal(aug.K.data, kd)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        b = rng.standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        assert scaled_residual(K, z, b) <= RESID_TOL
        if against_oracle:
            assert rel_err(z, oracle.OracleLdl(N, kc, kr, kd).solve(b)) <= 1e-8

    vi, ci, W = synth.working_set_all_rows(n, m, 0.1, 0)
    one(vi, ci, True)
    assert fact.info("maps_on") == 1 and fact.info("nlevels") <= 12 and fact.info("analyses") == 1
    active = vi >= 0
    for it in range(3):  # 1 % of the variables leave their bounds, 1 % others hit theirs
        leave = rng.choice(np.flatnonzero(active), n // 100, replace=False)
        enter = rng.choice(np.flatnonzero(~active), n // 100, replace=False)
        active[leave] = False
        active[enter] = True
        vi = np.full(n, -1, dtype=np.int32)
        vi[active] = np.arange(int(active.sum()), dtype=np.int32)
        ci = (int(active.sum()) + np.arange(m)).astype(np.int32)
        one(vi, ci, False)
        if boundary != "vtable_exact":  # (no reuse across patterns there: every changed pattern is analysed on its own rows)
            assert fact.info("analyses") == 1, it
        assert fact.info("nlevels") <= 12
    assert fact.info("dataflow_fallbacks") == 0


def test_top_block_under_graded_conditioning(fact):
    """VERDICT round 4 (parity, soft spot ii): the single-product top block of the solve applies an explicit inverse
    Z = S_T^-1 while the residual is only taken on every k-th solve (8, 16, 32, 64 after checks that pass).  A size at
    which the block forms (n = 2e4, m = 1e4: nine levels), graded families on it - row scalings over six decades,
    nearly parallel rows, column scalings -, 150 solves per factorisation so that the interval reaches its maximum:
    EVERY solution is checked here on the host (scaled residual on the caller's K), the ones the device did not check
    included."""
    import scipy.sparse as sp

    from sleqp_amd.sparse import SleqpMat

    n, m = 20000, 10000
    J0 = synth.banded_jacobian(n, m, 20, 200, 5)
    rng = np.random.default_rng(8)

    def near_parallel(J, eps, k=40):
        Jr = sp.csr_matrix(J)
        extra = Jr[rng.choice(Jr.shape[0], k, replace=False)].copy()
        extra.data = extra.data * (1.0 + eps * rng.standard_normal(extra.data.size))
        out = sp.vstack([Jr, extra]).tocsc()
        out.sort_indices()
        return out

    d6 = np.logspace(0, 6, m)
    rng.shuffle(d6)
    dc = np.logspace(0, 3, n)
    rng.shuffle(dc)
    cases = [("rowscale_1e6", sp.csc_matrix(sp.diags(d6) @ J0)), ("parallel_1e-4", near_parallel(J0, 1e-4)),
             ("colscale_1e3", sp.csc_matrix(J0 @ sp.diags(dc)))]
    for name, J in cases:
        J.sort_indices()
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(J)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        worst, checked0 = 0.0, fact.info("num_refined")
        for k in range(150):
            b = rng.standard_normal(N)
            fact.solve(b)
            worst = max(worst, scaled_residual(K, fact.solution_raw(0, N), b))
        assert fact.info("top_block_active") == 1 and fact.info("top_block_cols") > 500, name
        # (the equilibrated backward error the device controls is <= 1e-12 / kappa-scaled; on the caller's K a graded
        # family loses what its row scales span - the bound below is what the per-front sweep gives as well)
        fact.set_option("top_block_after", 1 << 30)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        ref = 0.0
        for k in range(20):
            b = rng.standard_normal(N)
            fact.solve(b)
            ref = max(ref, scaled_residual(K, fact.solution_raw(0, N), b))
        fact.set_option("top_block_after", 2)
        assert worst <= max(4.0 * ref, RESID_TOL), (name, worst, ref)
        assert fact.info("dataflow_fallbacks") == 0


def test_wide_separators_of_a_2d_grid(fact):
    """a PDE-constrained 2-D grid (5-point Laplacian states + controls).  J J^T is a 13-point
    stencil, the separators of the dissection are wider than one front (128 columns) and become chains of fronts:
    against the oracle's sparse LDL^T and the residual, the AugJac solves against the sparse product formulas."""
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    for g in (40, 96):
        J = synth.grid2d_jacobian(g, 1)
        m, n = J.shape
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(J)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("max_w") <= 128
        b = np.random.default_rng(g).standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        assert scaled_residual(K, z, b) <= RESID_TOL
        assert rel_err(z, oracle.OracleLdl(N, kc, kr, kd).solve(b)) <= 1e-8
        gvec = np.zeros(N)
        gvec[:n] = np.random.default_rng(3).standard_normal(n)
        fact.solve(SleqpVec.from_raw(gvec))
        Pg = fact.solution_raw(0, n)
        A = J.tocsr()
        assert np.abs(A @ Pg).max() <= 1e-9 * np.abs(A).sum(axis=1).max() * np.abs(gvec).max()
        assert fact.info("dataflow_fallbacks") == 0


def test_plane_separators_of_a_3d_grid(fact):
    """VERDICT round 5, missing 3: a PDE-constrained 3-D grid (7-point Laplacian states + controls).  J J^T is a 25-point
    stencil and the separators of the dissection are planes of ~2 g^2 cells - chains of many 128-column fronts with
    thousands of update rows (dense-chain machinery: paired trailing updates, row-sliced solve items): against the
    oracle's sparse LDL^T and the residual; the projection lands in the null space of the constraint rows."""
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    for g in (12, 20):
        J = synth.grid3d_jacobian(g, 1)
        m, n = J.shape
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(J)
        K = synth.kkt_full_matrix(N, kc, kr, kd)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        assert fact.info("max_w") <= 128 and fact.info("nlevels") >= (6 if g == 12 else 12)  # (7 / 19 levels since round 6)
        b = np.random.default_rng(g).standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        assert scaled_residual(K, z, b) <= RESID_TOL
        assert rel_err(z, oracle.OracleLdl(N, kc, kr, kd).solve(b)) <= 1e-8
        gvec = np.zeros(N)
        gvec[:n] = np.random.default_rng(3).standard_normal(n)
        fact.solve(SleqpVec.from_raw(gvec))
        Pg = fact.solution_raw(0, n)
        A = J.tocsr()
        assert np.abs(A @ Pg).max() <= 1e-9 * np.abs(A).sum(axis=1).max() * np.abs(gvec).max()
        assert fact.info("dataflow_fallbacks") == 0


def test_full_size_krylov_loops(fact):
    """BASELINE.json configs[3] (n = 1e5, m = 5e4) under the device-resident Krylov loops of the EQP step, with the
    banded Hessian of the bench: the device-controlled CG against the host-driven loop (iteration count, step), the
    generalised Lanczos method against CG inside a large trust region (same minimiser of the projected model), the
    step in the null space of the working-set rows, the model decrease, the boundary case on the boundary."""
    from bench import make_problem
    from sleqp_amd.fact import SpMat
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
    m, n = J.shape
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    rng = np.random.default_rng(7)
    # (the Hessian of bench.py's spmv_setup: symmetric banded, half-bandwidth 5, diagonally dominant)
    Hl = sp.diags([np.full(n, 2.0)] + [rng.standard_normal(n - k) * 0.1 for k in range(1, 6)], [0, -1, -2, -3, -4, -5], format="csc")
    Hl.sort_indices()
    Hd = SpMat(fact, SleqpMat(n, n, Hl.indptr, Hl.indices, Hl.data))
    Hs = (Hl + Hl.T - sp.diags(Hl.diagonal())).tocsr()
    g = rng.standard_normal(n)
    A = J.tocsr()
    anorm = np.abs(A).sum(axis=1).max()
    fact.set_option("cg_device_loop", 0)
    want, _, its0 = fact.steihaug(Hd, g, 1e6, stat_tol=1e-3, max_iter=300)
    fact.set_option("cg_device_loop", 1)
    runs = fact.info("cg_device_runs")
    step, _, its1 = fact.steihaug(Hd, g, 1e6, stat_tol=1e-3, max_iter=300)
    assert fact.info("cg_device_runs") == runs + 1 and fact.info("cg_device_fallbacks") == 0
    assert 0 < its0 < 300 and its1 == its0
    # (the first projection of the host-driven run is the first solve of the factorisation: the ordinary tree launch;
    # every other one goes through the top block, which sums in another order - rounding-level differences that the
    # CG recurrence carries along)
    assert rel_err(step, want) <= 1e-8
    assert np.abs(A @ step).max() <= 1e-9 * anorm * max(1.0, np.abs(step).max())
    q = lambda s_: float(g @ s_ + 0.5 * s_ @ (Hs @ s_))
    assert q(step) < 0.0
    fact.set_option("lz_device_loop", 0)
    lz_host, _, its_host = fact.tr_solve(Hd, g, 1e6, method=1, stat_tol=1e-3, max_iter=300)
    fact.set_option("lz_device_loop", 1)
    lzi = fact.info("lz_device_iterations")
    lz, _, its2 = fact.tr_solve(Hd, g, 1e6, method=1, stat_tol=1e-3, max_iter=300)
    # (the whole run inside the trust region: every iteration on the device, the host solves ONE tridiagonal problem)
    assert its2 == its_host and fact.info("lz_device_iterations") - lzi == its2 and fact.info("lz_device_fallbacks") == 0
    assert rel_err(lz, lz_host) <= 1e-9
    # (both stop at stat_tol 1e-3 by their own tests: the iterates agree to that order, the model values much better)
    assert 0 < its2 < 300 and rel_err(lz, step) <= 2e-4 and abs(q(lz) - q(step)) <= 1e-7 * abs(q(step))
    # trust region active
    radius = 0.3 * np.linalg.norm(step)
    for method in (0, 1):
        sb, dual, _ = fact.tr_solve(Hd, g, radius, method=method, stat_tol=1e-6, max_iter=300)
        assert abs(np.linalg.norm(sb) - radius) <= 1e-8 * radius and dual >= 0.0
        assert np.abs(A @ sb).max() <= 1e-9 * anorm * max(1.0, np.abs(sb).max())
        assert q(sb) < 0.0
    Hd.free()


def test_adaptive_refinement_and_graphs(fact):
    """Refinement runs only when the residual asks for it; graph replay and direct launches agree bitwise."""
    from sleqp_amd.sparse import SleqpMat

    J, vi, ci, _ = _problem(800, 400, "b", 0.0, 8)
    N, kc, kr, kd = oracle.fill_aug_jac(800, 400, J.indptr, J.indices, J.data, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    b = np.random.default_rng(2).standard_normal(N)
    outs = {}
    for graph in (1, 0):
        fact.set_option("use_graph", graph)
        fact.set_option("refine_adaptive", 1)
        fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
        for _ in range(3):  # first call captures, later calls replay
            fact.solve(b)
        outs[graph] = fact.solution_raw(0, N)
        assert scaled_residual(K, outs[graph], b) <= RESID_TOL
    assert np.array_equal(outs[0], outs[1])
    assert fact.info("num_graphs") == 0
    fact.set_option("use_graph", 1)
    # well conditioned: the first pass meets the tolerance, the correction pass of the graph is skipped
    n0 = fact.info("num_refined")
    fact.solve(b)
    z_one = fact.solution_raw(0, N)
    assert fact.info("num_refined") == n0 and fact.info("last_iters") == 0 and fact.info("last_status") == 0
    assert fact.info("last_omega") <= 1e-14
    # non-adaptive: exactly refine_steps passes, whatever the residual says
    fact.set_option("refine_adaptive", 0)
    fact.set_option("refine_steps", 2)
    fact.solve(b)
    assert rel_err(fact.solution_raw(0, N), z_one) <= 1e-12 and fact.info("last_iters") == 2
    assert fact.info("num_refined") == n0 + 1
    fact.set_option("refine_adaptive", 1)
    fact.set_option("refine_steps", 1)
    # badly scaled rows: exact row equilibration makes them harmless
    Jb = sp.csc_matrix(sp.diags(np.logspace(0, 5, 400)) @ J)
    N2, c2, r2, d2 = oracle.fill_aug_jac(800, 400, Jb.indptr, Jb.indices, Jb.data, vi, ci)
    K2 = synth.kkt_full_matrix(N2, c2, r2, d2)
    fact.set_option("refine_tol", 1e-10)
    fact.set_matrix(SleqpMat(N2, N2, c2, r2, d2))
    fact.solve(b)
    assert scaled_residual(K2, fact.solution_raw(0, N2), b) <= RESID_TOL


def test_device_steihaug_known_answers(fact):
    """Reference known answers through the device-resident projected CG
    (constrained_newton_test.c:204-275, unconstrained_newton_test.c:67-205)."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    H = SpMat(fact, SleqpMat(2, 2, [0, 1, 2], [0, 1], [2.0, 2.0]))  # hess_prod = 2 * direction
    grad = np.array([2.0, 4.0])
    aug = StandardAugJac(2, fact)
    aug.set_iterate(SleqpMat(1, 2, [0, 0, 1], [0], [1.0]), [-1, -1], [0])  # c = x1 active
    step, dual, its = fact.steihaug(H, grad, 10.0)
    assert np.allclose(step, [-1.0, 0.0], atol=1e-8) and dual == -1.0
    aug.set_iterate(SleqpMat(0, 2, [0, 0, 0], [], []), [-1, -1], [])  # empty working set
    step, dual, its = fact.steihaug(H, grad, 10.0)
    assert np.allclose(step, [-1.0, -2.0], atol=1e-8)
    step, dual, its = fact.steihaug(H, grad, 1.0)  # trust region active
    assert np.allclose(step, [-0.44721359549995793, -0.89442719099991586], atol=1e-8)
    assert dual >= 0.0


@pytest.mark.parametrize("radius", [0.3, 5.0, 1e3])
def test_device_steihaug_vs_oracle(fact, radius):
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 400, 150
    J, vi, ci, W = _problem(n, m, "u", 0.05, 11)
    B = sp.random(n, n, density=0.02, random_state=3)
    Hm = (B @ B.T + 0.5 * sp.eye(n)).tocsc()
    HL = sp.tril(Hm, format="csc")
    HL.sort_indices()
    g = np.random.default_rng(5).standard_normal(n)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    # the interior case ends on |r.g| < (stat_tol 1e-2)^2; with the default 1e-6 that is 1e-16 absolute, which
    # the LAPACK-restating projections of the oracle never reach (it then runs into the iteration cap and,
    # like the reference, returns a zero step) while the device projections - active bounds eliminated
    # exactly - do: compare at a tolerance both can meet
    stat_tol = 1e-6 if radius < 100 else 1e-4
    want, its_ref = oracle.OracleFact(N, kc, kr, kd).steihaug(n, HL.indptr, HL.indices, HL.data, g, trust_radius=radius,
                                                              stat_tol=stat_tol)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    H = SpMat(fact, SleqpMat.from_scipy(HL))
    step, dual, its = fact.steihaug(H, g, radius, stat_tol=stat_tol)
    assert its_ref < 100 and abs(its - its_ref) <= 1
    assert rel_err(step, want) <= (1e-8 if radius < 100 else 1e-6)
    # the step lies in the null space of the working-set rows and inside the trust region
    A_W = sp.vstack([sp.eye(n, format="csr")[np.nonzero(vi >= 0)[0]], J.tocsr()])
    assert np.abs(A_W @ step).max() <= 1e-9 * max(1.0, np.abs(step).max()) * abs(A_W).sum(axis=1).max()
    assert np.linalg.norm(step) <= radius * (1 + 1e-10)


@pytest.mark.parametrize("mode", [1, 2], ids=["late_elimination", "low_rank_correction"])
@pytest.mark.parametrize("method", [0, 1], ids=["steihaug", "gltr"])
def test_krylov_loops_with_dense_jacobian_columns(fact, method, mode):
    """Dense Jacobian columns under the device Krylov loops: the projection is then K_0^-1 b plus
    a correction, and the partial dot products the x update of the tree launch leaves belong to the UNCORRECTED solve -
    the loops must form r.g / ||t||_P^2 from the corrected z.  Projected CG against the oracle's CG (iterates and
    iteration count), GLTR against the same minimiser; device-controlled and host-driven loop alike."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 1500, 700
    J, dcols = _with_dense_columns(synth.banded_jacobian(n, m, 10, 80, 17), 3, 5)
    rng = np.random.default_rng(9)
    vi, ci, W = _ws(n, m, rng, 1.0, 0.0)
    fact.set_option("dense_mode", mode)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    assert fact.info("dense_columns" if mode == 2 else "late_columns") == 3
    B = sp.random(n, n, density=3.0 / n, random_state=3)
    HL = sp.tril(B @ B.T + 0.5 * sp.eye(n), format="csc")
    HL.sort_indices()
    H = SpMat(fact, SleqpMat.from_scipy(HL))
    Hs = (HL + HL.T - sp.diags(HL.diagonal())).tocsr()
    g = rng.standard_normal(n)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    for radius, tol in ((1e3, 1e-4), (30.0, 1e-6)):
        want, its_ref = oracle.OracleFact(N, kc, kr, kd).steihaug(n, HL.indptr, HL.indices, HL.data, g,
                                                                  trust_radius=radius, stat_tol=tol)
        assert 0 < its_ref < 100
        for device_loop in (1, 0):
            fact.set_option("cg_device_loop", device_loop)
            fact.set_option("lz_device_loop", device_loop)
            step, dual, its = fact.tr_solve(H, g, radius, method=method, stat_tol=tol, max_iter=200)
            if method == 0:
                assert abs(its - its_ref) <= 1, (its, its_ref)
            if method == 1 and radius < 100:
                # on the boundary GLTR minimises the model over the whole Krylov space while Steihaug's CG stops where
                # its path leaves the region: another point of the boundary, with a model value at least as good
                q = lambda s_: float(g @ s_ + 0.5 * s_ @ (Hs @ s_))
                assert abs(np.linalg.norm(step) - radius) <= 1e-8 * radius and q(step) <= q(want) + 1e-9 * abs(q(want))
            else:
                # (interior case: GLTR and CG stop at stat_tol by tests of their own - the iterates agree to that order)
                assert rel_err(step, want) <= ((5e-5 if method == 1 else 1e-6) if radius > 100 else 1e-7), (radius, device_loop)
            assert np.abs(J @ step).max() <= 1e-9 * max(1.0, np.abs(step).max()) * abs(J).sum(axis=1).max()
            assert np.linalg.norm(step) <= radius * (1 + 1e-10)
    H.free()


@pytest.mark.parametrize("kind", ["positive_definite", "indefinite"])
def test_gltr_device_phase_matches_host_loop(fact, kind):
    """GLTR with the iterations of the positive definite / interior phase controlled on the device (krylov_device.inc:
    pivot and step recurrences of the Lanczos tridiagonal in a control block, three launches per iteration, the host
    looks every 8 iterations) against the host-driven loop (trlib's iteration, one tridiagonal trust-region solve per
    iteration): same iteration count, step and multiplier when the phase ends by convergence, by the iteration cap
    (at, one below and one above the chunk length; caps 1 and 2), by leaving the trust region or by an indefinite
    tridiagonal - the last two hand the loop over to the host in the middle of the Krylov space."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 1500, 700
    J = synth.banded_jacobian(n, m, 10, 80, 17)
    rng = np.random.default_rng(19)
    vi, ci, W = _ws(n, m, rng, 1.0, 0.0)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    B = sp.random(n, n, density=3.0 / n, random_state=3)
    shift = 0.5 if kind == "positive_definite" else -0.4
    HL = sp.tril(B @ B.T + shift * sp.eye(n), format="csc")
    HL.sort_indices()
    H = SpMat(fact, SleqpMat.from_scipy(HL))
    Hs = (HL + HL.T - sp.diags(HL.diagonal())).tocsr()
    q = lambda s_: float(g @ s_ + 0.5 * s_ @ (Hs @ s_))
    g = rng.standard_normal(n)
    big = 1e4 if kind == "positive_definite" else 40.0
    fact.set_option("lz_device_loop", 0)
    ref, _, _ = fact.tr_solve(H, g, big, method=1, stat_tol=1e-9, max_iter=400)
    nr = np.linalg.norm(ref)
    device_iterations = 0
    import scipy.linalg as sla

    Z = sla.null_space(J.toarray())
    lam = np.linalg.eigvalsh(Z.T @ (Hs @ Z))
    lam_lo, lam_hi = float(lam[0]), float(lam[-1])
    ray = {}
    for radius, tol, cap in ((big, 1e-4, 400), (big, 1e-9, 400), (0.3 * nr, 1e-7, 400), (0.999 * nr, 1e-9, 400),
                             (big, 1e-30, 7), (big, 1e-30, 8), (big, 1e-30, 9), (big, 1e-30, 17), (big, 1e-30, 1), (big, 1e-30, 2)):
        out = {}
        for dev in (0, 1):
            fact.set_option("lz_device_loop", dev)
            i0 = fact.info("lz_device_iterations")
            out[dev] = fact.tr_solve(H, g, radius, method=1, stat_tol=tol, max_iter=cap) + (fact.info("lz_device_iterations") - i0,)
            ray[dev] = dict(fact.last_tr)
        (s0, d0, it0, _), (s1, d1, it1, dits) = out[0], out[1]
        # the rayleigh slot (tr/tr_types.h:18-20): Rayleigh quotients of null-space directions - the same from the device
        # phase's coefficients and the host loop's, inside the spectrum of the projected Hessian, never a time-out
        # (the late Lanczos coefficients of a long indefinite run are rounding-sensitive - two runs of the SAME loop differ
        # in them while step and multiplier agree to 1e-15 -, so equality is asked of the positive definite family)
        for k_ in (("min_rayleigh", "max_rayleigh") if kind == "positive_definite" else ()):
            assert abs(ray[0][k_] - ray[1][k_]) <= 1e-9 * max(1.0, abs(ray[0][k_])), (radius, tol, cap, ray)
        assert lam_lo * (1 + 1e-9) - 1e-9 <= ray[1]["min_rayleigh"] <= ray[1]["max_rayleigh"] <= lam_hi * (1 + 1e-9) + 1e-9, (ray, lam_lo, lam_hi)
        assert not ray[0]["timed_out"] and not ray[1]["timed_out"]
        assert it1 == it0 and 0 < it0 <= cap, (radius, tol, cap, it0, it1)
        assert rel_err(s1, s0) <= 1e-9, (radius, tol, cap)
        assert abs(d1 - d0) <= 1e-9 * max(1.0, abs(d0))
        assert abs(q(s1) - q(s0)) <= 1e-10 * abs(q(s0))
        assert np.linalg.norm(s1) <= radius * (1 + 1e-10)
        assert np.abs(J @ s1).max() <= 1e-9 * max(1.0, np.abs(s1).max()) * abs(J).sum(axis=1).max()
        assert dits <= it1
        if cap >= 2:
            assert dits >= 1  # (the phase ran; with cap 1 there is nothing to hand to the device)
        if kind == "positive_definite" and radius == big and cap >= 2:
            assert dits == it1  # (never left: every iteration on the device)
        device_iterations += dits
    assert device_iterations > 0 and fact.info("lz_device_fallbacks") == 0
    # time_limit (tr/tr_types.h:9-16) through the C ABI: a limit that is over at the first look ends both loops with the
    # iterate reached (feasible, inside the region, fewer iterations than the cap), one of a minute changes nothing
    for method in ((0, 1) if kind == "positive_definite" else ()):
        for dev in (0, 1):
            fact.set_option("lz_device_loop", dev)
            fact.set_option("cg_device_loop", dev)
            full, _, its_full = fact.tr_solve(H, g, big, method=method, stat_tol=1e-30, max_iter=60)
            assert not fact.last_tr["timed_out"]
            s_, _, its = fact.tr_solve(H, g, big, method=method, stat_tol=1e-30, max_iter=60, time_limit=1e-4)
            assert fact.last_tr["timed_out"] and its < its_full, (method, dev, its, its_full)
            assert np.all(np.isfinite(s_)) and np.linalg.norm(s_) <= big
            assert np.abs(J @ s_).max() <= 1e-9 * max(1.0, np.abs(s_).max()) * abs(J).sum(axis=1).max()
            s_, _, its = fact.tr_solve(H, g, big, method=method, stat_tol=1e-30, max_iter=60, time_limit=60.0)
            assert not fact.last_tr["timed_out"] and its == its_full and rel_err(s_, full) <= 1e-9
    fact.set_option("lz_device_loop", 1)
    fact.set_option("cg_device_loop", 1)
    H.free()


def test_host_boundary_fast_path(fact):
    """The host boundary of the unmodified vtable (sleqp_fact_solve / sleqp_fact_solution): pinned staging, a right-hand
    side that is one contiguous run of indices without index upload or scatter, the whole solution sent to pinned
    memory behind the solve and read through hipfact_solution / hipfact_solution_view.  Every variant (kernel-read or
    copy-engine upload, copy-engine or kernel download, the round-3 path) must give the SAME bits, for run-shaped,
    scattered, empty and single-entry right-hand sides, several solution ranges per solve and two solves in a row
    without a solution in between; out-of-range indices are refused."""
    import ctypes as C

    from sleqp_amd import _lib
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 1500, 700
    J, vi, ci, W = _problem(n, m, "b", 0.05, 3)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
    ref = oracle.OracleFact(N, kc, kr, kd)
    rng = np.random.default_rng(8)
    cases = []
    cases.append(("dense", np.arange(N, dtype=np.int32), rng.standard_normal(N)))
    cases.append(("prefix_run", np.arange(n, dtype=np.int32), rng.standard_normal(n)))           # project / lsq: dense g
    cases.append(("shifted_run", np.arange(n, N, dtype=np.int32), rng.standard_normal(N - n)))   # min-norm: dense c
    idx = np.sort(rng.choice(N, N // 3, replace=False)).astype(np.int32)
    cases.append(("scattered", idx, rng.standard_normal(idx.size)))
    cases.append(("single", np.array([N // 2], dtype=np.int32), np.array([1.5])))
    cases.append(("empty", np.zeros(0, dtype=np.int32), np.zeros(0)))
    lib = _lib.load()
    for _ in range(2):  # (steady state of the factorisation: the top levels of the solve tree as one dense block)
        fact.solve(np.ones(N))
    variants = [dict(boundary_fast=1), dict(boundary_fast=1, validate_rhs=1), dict(boundary_fast=0)]
    for name, ix, vals in cases:
        ref.solve_sparse(ix, vals)
        want = ref.raw_solution()
        got = []
        for opts in variants:
            for k, v in opts.items():
                fact.set_option(k, v)
            fact.solve(SleqpVec(N, ix, vals))
            parts = [fact.solution_raw(0, n).copy(), fact.solution_raw(n, N).copy(), fact.solution_raw(5, 17).copy()]
            z = np.concatenate(parts[:2])
            assert np.array_equal(z[5:17], parts[2])
            view = C.c_void_p()
            assert lib.hipfact_solution_view(fact._h, C.byref(view), n, N) == 0
            zv = np.ctypeslib.as_array(C.cast(view, C.POINTER(C.c_double)), shape=(N - n,)).copy()
            assert np.array_equal(zv, parts[1]), name
            got.append(z)
            fact.set_option("validate_rhs", 0)
        for z in got[1:]:
            assert np.array_equal(z, got[0]), name
        assert rel_err(got[0], want) <= REL_TOL or np.abs(want).max() == 0.0, name
    fact.set_option("boundary_fast", 1)
    # two solves without a solution in between: the second one's result is what solution() returns
    b1, b2 = rng.standard_normal(N), rng.standard_normal(N)
    fact.solve(b1)
    fact.solve(b2)
    ref.solve_dense(b2)
    assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL
    # many solves in a row (both staging buffers, the unchecked steady state, every 8th solve checked)
    for t in range(20):
        bt = rng.standard_normal(N)
        fact.solve(SleqpVec(N, np.arange(N, dtype=np.int32), bt))
        ref.solve_dense(bt)
        assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL, t
    # indices outside [0, N): refused on the host when they are the first / last one, ignored by the device otherwise
    def raw_solve(ix, vals):  # (straight through the C ABI: the Python SleqpVec refuses such indices itself)
        ix = np.ascontiguousarray(ix, dtype=np.int32)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        return lib.hipfact_solve_sparse(fact._h, N, ix.size, ix.ctypes.data_as(C.c_void_p), vals.ctypes.data_as(C.c_void_p))

    HIPFACT_EINVAL = raw_solve([0, N], np.ones(2))
    assert HIPFACT_EINVAL != 0 and raw_solve([-1, 3], np.ones(2)) == HIPFACT_EINVAL
    fact.set_option("validate_rhs", 1)
    assert raw_solve([4, 2, 9], np.ones(3)) == HIPFACT_EINVAL  # not ascending: only the full walk sees it
    fact.set_option("validate_rhs", 0)
    assert raw_solve([4, N + 7, 9], np.ones(3)) == 0  # an index the host does not look at: ignored by the device scatter
    assert np.all(np.isfinite(fact.solution_raw(0, N)))
    fact.solve(b1)  # the handle is still usable
    ref.solve_dense(b1)
    assert rel_err(fact.solution_raw(0, N), ref.raw_solution()) <= REL_TOL


def test_device_controlled_cg_matches_the_host_driven_loop(fact):
    """krylov_device.inc: the Steihaug loop with alpha, beta and the exit tests in a control block in HBM (four
    launches per iteration, no host round trip inside a chunk) against the host-driven loop of the same handle -
    same tests in the same order, dot products summed in another fixed order: same iteration count, iterates equal to
    rounding; interior solution, boundary exit and negative curvature."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 3000, 1200
    J = synth.banded_jacobian(n, m, 8, 60, 3)
    vi, ci, W = synth.working_set_all_rows(n, m, 0.0, 1)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    B = sp.random(n, n, density=3.0 / n, random_state=3)
    spd = sp.tril(B @ B.T + 0.5 * sp.eye(n), format="csc")
    indef = sp.tril(B + B.T + sp.diags(np.linspace(-1.0, 2.0, n)), format="csc")
    g = np.random.default_rng(5).standard_normal(n)
    from sleqp_amd.sparse import SleqpVec

    for _ in range(2):  # (both loops in the steady state of the factorisation: the top of the solve tree as one dense block)
        aug.project_nullspace(SleqpVec.from_raw(g))
    # (interior case: |r.g| < (1e-2 tol)^2 must stay above the rounding level of r.g, else the exit is met by chance)
    for HL, radius, tol in ((spd, 1e3, 1e-3), (spd, 30.0, 1e-6), (indef, 2.0, 1e-6)):
        HL.sort_indices()
        H = SpMat(fact, SleqpMat.from_scipy(HL))
        fact.set_option("cg_device_loop", 0)
        want, dual0, its0 = fact.steihaug(H, g, radius, stat_tol=tol, max_iter=200)
        fact.set_option("cg_device_loop", 1)
        runs = fact.info("cg_device_runs")
        step, dual1, its1 = fact.steihaug(H, g, radius, stat_tol=tol, max_iter=200)
        assert fact.info("cg_device_runs") == runs + 1 and fact.info("cg_device_fallbacks") == 0
        assert its1 == its0 and its0 < 200 and (HL is indef or its0 > 0), (its0, its1)
        # the same with the x update (and with it the partials of r.g) as launches of their own behind the tree
        fact.set_option("xupd_fused", 0)
        step2, _, its2 = fact.steihaug(H, g, radius, stat_tol=tol, max_iter=200)
        fact.set_option("xupd_fused", 1)
        assert its2 == its0 and rel_err(step2, want) <= 1e-10 and fact.info("cg_device_fallbacks") == 0
        assert rel_err(step, want) <= 1e-10
        assert abs(dual1 - dual0) <= 1e-9 * max(1.0, abs(dual0))
        assert np.abs(J @ step).max() <= 1e-9 * max(1.0, np.abs(step).max()) * abs(J).sum(axis=1).max()
        if radius < 100:
            assert abs(np.linalg.norm(step) - radius) <= 1e-9 * radius
        H.free()


@pytest.mark.parametrize("per_row", [1, 3, 12, 40], ids=["lanes1", "lanes4", "lanes16", "lanes64"])
def test_device_controlled_cg_product_variants(fact, per_row):
    """The product kernel of the device-controlled loop (B d + the partials of five dot products) exists in four
    lane widths, chosen by the Hessian's average row length like the plain SpMV: each against the host-driven loop."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 2500, 900
    J = synth.banded_jacobian(n, m, 8, 60, 7)
    vi, ci, W = synth.working_set_all_rows(n, m, 0.0, 1)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    B = sp.random(n, n, density=max(per_row - 1, 0) / (2.0 * n), random_state=per_row)
    HL = sp.tril(B + B.T + sp.diags(4.0 + 2.0 * per_row + np.arange(n) / n), format="csc")  # diagonally dominant: convex
    HL.sort_indices()
    assert abs(2.0 * HL.nnz / n - (per_row + 0.5)) < 0.6 * per_row + 1.0
    g = np.random.default_rng(per_row).standard_normal(n)
    H = SpMat(fact, SleqpMat.from_scipy(HL))
    fact.set_option("cg_device_loop", 0)
    want, _, its0 = fact.steihaug(H, g, 1e3, stat_tol=1e-3, max_iter=200)
    fact.set_option("cg_device_loop", 1)
    runs = fact.info("cg_device_runs")
    step, _, its1 = fact.steihaug(H, g, 1e3, stat_tol=1e-3, max_iter=200)
    assert fact.info("cg_device_runs") == runs + 1 and fact.info("cg_device_fallbacks") == 0
    assert its1 == its0 and 0 < its0 < 200, (its0, its1)
    assert rel_err(step, want) <= 1e-10
    # stationarity of the projected problem: P (H step + g) = 0 up to the tolerance of the loop
    Hs = (HL + HL.T - sp.diags(HL.diagonal())).tocsr()
    from sleqp_amd.sparse import SleqpVec
    pr = aug.project_nullspace(SleqpVec.from_raw(Hs @ step + g)).to_raw()
    assert np.linalg.norm(pr) <= 1e-4 * np.linalg.norm(g)
    H.free()


def test_krylov_loops_on_a_plan_with_sliced_fronts(fact):
    """Dense Schur complement (tall fronts: row-sliced items in the fused solve launch, chain levels as small
    dataflow launches) under the device-resident Krylov loops.  With H = c I the EQP step inside a large trust
    region is -P g / c, P the projection onto the null space of the working-set rows: both loops must return it,
    and the projection must be idempotent and feasible."""
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat, SleqpVec

    n, m = 3000, 1500
    J = synth.uniform_jacobian(n, m, 10, 9)
    vi, ci, W = synth.working_set_all_rows(n, m, 0.0, 1)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    assert fact.info("fused_solve") == 1 and fact.info("solve_items") > fact.info("nsuper")
    g = np.random.default_rng(8).standard_normal(n)
    pg = aug.project_nullspace(SleqpVec.from_raw(g)).to_raw()
    ppg = aug.project_nullspace(SleqpVec.from_raw(pg)).to_raw()
    assert rel_err(ppg, pg) <= 1e-9
    assert np.abs(J @ pg).max() <= 1e-9 * np.abs(g).max() * abs(J).sum(axis=1).max()
    c = 2.5
    HL = sp.identity(n, format="csc") * c
    H = SpMat(fact, SleqpMat.from_scipy(sp.csc_matrix(HL)))
    for method in (0, 1):
        # (the reference's interior test is absolute, |r.g| < (1e-2 stat_tol)^2: 1e-10 here, above the rounding
        # level of r.g for |g| ~ 40 - at the default it is 1e-16 and only met by chance)
        step, dual, its = fact.tr_solve(H, g, 1e6, method=method, stat_tol=1e-3)
        assert rel_err(step, -pg / c) <= 1e-8 and its <= 3
        assert np.abs(J @ step).max() <= 1e-9 * np.abs(g).max() * abs(J).sum(axis=1).max()
    # trust region active: the step is the scaled projected gradient
    radius = 0.25 * np.linalg.norm(pg) / c
    for method in (0, 1):
        step, dual, its = fact.tr_solve(H, g, radius, method=method, stat_tol=1e-3)
        assert abs(np.linalg.norm(step) - radius) <= 1e-8 * radius
        assert rel_err(step, -pg * radius / np.linalg.norm(pg)) <= 1e-7


def test_device_steihaug_negative_curvature(fact):
    from sleqp_amd.fact import SpMat, StandardAugJac
    from sleqp_amd.sparse import SleqpMat

    n, m = 60, 20
    J, vi, ci, W = _problem(n, m, "u", 0.0, 4)
    Hd = sp.diags(np.linspace(-1.0, 2.0, n)).tocsc()  # indefinite Hessian
    g = np.random.default_rng(1).standard_normal(n)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    want, its_ref = oracle.OracleFact(N, kc, kr, kd).steihaug(n, Hd.indptr, Hd.indices, Hd.data, g, trust_radius=2.0)
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    step, dual, its = fact.steihaug(SpMat(fact, SleqpMat.from_scipy(Hd)), g, 2.0)
    assert rel_err(step, want) <= 1e-8
    assert abs(np.linalg.norm(step) - 2.0) <= 1e-9  # ends on the boundary


def test_concurrent_instances_like_thread_test():
    """thread_test.c:13-110: 8 threads, each builds and uses its own solver state; here 8 backend
    instances on one GPU (own stream / buffers / symbolic cache each), checked against the oracle."""
    import threading

    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    results = [None] * 8

    def worker(t):
        try:
            n, m = 300 + 40 * t, 120 + 10 * t
            J, vi, ci, W = _problem(n, m, "b" if t % 2 else "u", 0.05, 20 + t)
            N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
            b = np.random.default_rng(t).standard_normal(N)
            f = HipFact(device=0)
            errs = []
            for rep in range(3):
                f.set_matrix(SleqpMat(N, N, kc, kr, kd))
                f.solve(b)
                z = f.solution_raw(0, N)
                ref = oracle.OracleFact(N, kc, kr, kd)
                ref.solve_dense(b)
                errs.append(rel_err(z, ref.raw_solution()))
            f.free()
            results[t] = max(errs)
        except Exception as e:  # noqa: BLE001
            results[t] = e

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for r in results:
        assert not isinstance(r, Exception), r
        assert r <= REL_TOL


def test_two_full_size_handles_share_one_gpu():
    """Two handles with BASELINE's config 4 each, driven from two threads on one GPU (what `bench.py --gpus 2` does on a
    box with one device, and what two SLEQP instances of thread_test.c:77-110 would do): the dataflow launches of one
    handles would interleave on the same CUs.  A workgroup only waits for lower-indexed workgroups of its OWN launch -
    but the XCDs start their shares of a launch independently, so two such launches can keep each other's producers
    out of the chip until the bounded waits end (seen here in 1 run of ~8 before the handles took turns).  The handles
    of one device therefore queue their launch sequences behind one another's events (`turn_waits` > 0): no wait may
    time out (`dataflow_fallbacks`, `solve_timeouts` stay zero) and every solve meets the residual bound."""
    import threading

    from bench import make_problem
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    out = [None, None]

    def worker(t):
        try:
            J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", t)
            K = synth.kkt_full_matrix(N, cp, ri, vx)
            f = HipFact(device=0)
            worst = 0.0
            for rep in range(12):
                f.set_matrix(SleqpMat(N, N, cp, ri, vx))
                for _ in range(4):
                    f.solve(b)
                z = f.solution_raw(0, N)
                worst = max(worst, scaled_residual(K, z, b))
            out[t] = (worst, f.info("dataflow_fallbacks"), f.info("solve_timeouts"), f.info("turn_waits"))
            f.free()
        except Exception as e:  # noqa: BLE001
            out[t] = e

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for r in out:
        assert not isinstance(r, Exception), r
        assert r[0] <= RESID_TOL and r[1] == 0 and r[2] == 0, r
    assert out[0][3] + out[1][3] > 0, out
