"""The SLEQP-side shim (shim/fact_hipfact.c, shim/aug_jac_hipfact.c, shim/tr_hipfact.c) built against the
stand-alone harness: CPU — it loads and exports the reference's entry points; GPU —
driving it exactly like standard_aug_jac.c / the reference tests do reproduces the
golden vectors."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu
from util import REL_TOL, ZERO_EPS, golden_cases, rel_err

SO = os.path.join(ROOT, "shim", "libsleqp_hipfact_standalone.so")


class SleqpVecC(C.Structure):  # sparse/pub_vec.h:16-25
    _fields_ = [("data", C.POINTER(C.c_double)), ("indices", C.POINTER(C.c_int)), ("dim", C.c_int), ("nnz", C.c_int),
                ("nnz_max", C.c_int)]


@pytest.fixture(scope="module")
def shim(hipfact_lib):
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "shim")])
    lib = C.CDLL(SO)
    lib.sleqp_error_msg.restype = C.c_char_p
    lib.sleqp_fact_name.restype = C.c_char_p
    lib.sleqp_fact_version.restype = C.c_char_p
    lib.sleqp_mat_cols.restype = C.POINTER(C.c_int)
    lib.sleqp_mat_rows.restype = C.POINTER(C.c_int)
    lib.sleqp_mat_data.restype = C.POINTER(C.c_double)
    lib.sleqp_iterate_cons_jac.restype = C.c_void_p
    lib.sleqp_iterate_working_set.restype = C.c_void_p
    return lib


def test_shim_exports_reference_entry_points(shim):
    for name in ["sleqp_fact_create_default", "sleqp_fact_hipfact_create", "sleqp_hipfact_aug_jac_create",
                 "sleqp_hipfact_aug_jac_handle", "sleqp_hipfact_tr_solver_create", "sleqp_hipfact_tr_bind",
                 "sleqp_hipfact_tr_set_hessian", "sleqp_tr_solver_solve", "sleqp_tr_solver_release",
                 "sleqp_fact_set_matrix", "sleqp_fact_solve", "sleqp_fact_solution", "sleqp_fact_cond",
                 "sleqp_fact_flags", "sleqp_fact_release"]:
        assert hasattr(shim, name), name


def test_shim_reports_missing_gpu_as_sleqp_error(shim):
    if has_gpu():
        pytest.skip("GPU present")
    settings, fact = C.c_void_p(), C.c_void_p()
    assert shim.sleqp_settings_create(C.byref(settings)) == 0
    rc = shim.sleqp_fact_create_default(C.byref(fact), settings)
    assert rc == -1  # SLEQP_ERROR
    assert shim.sleqp_error_type() == 2  # SLEQP_INTERNAL_ERROR
    assert b"hipfact" in shim.sleqp_error_msg()
    shim.sleqp_settings_release(C.byref(settings))


def _vec(shim, dim, idx, dat):
    v = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create(C.byref(v), C.c_int(dim), C.c_int(max(len(idx), 1))) == 0
    for i, d in zip(idx, dat):
        assert shim.sleqp_vec_push(v, C.c_int(int(i)), C.c_double(float(d))) == 0
    return v


def _dense(v):
    out = np.zeros(v.contents.dim)
    for k in range(v.contents.nnz):
        out[v.contents.indices[k]] = v.contents.data[k]
    return out


def _fill_jac(shim, mat, c):
    assert shim.sleqp_mat_reserve(C.c_void_p(mat), C.c_int(len(c.jx))) == 0
    for j in range(c.n):
        assert shim.sleqp_mat_push_col(C.c_void_p(mat), C.c_int(j)) == 0
        for e in range(c.jp[j], c.jp[j + 1]):
            assert shim.sleqp_mat_push(C.c_void_p(mat), C.c_int(int(c.ji[e])), C.c_int(j), C.c_double(float(c.jx[e]))) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("c", golden_cases(), ids=lambda c: c.name)
def test_fact_shim_against_golden(shim, c):
    """sleqp_fact_create_default -> set_matrix(K lower) -> solve / solution, like standard_aug_jac.c."""
    settings, fact, K = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert shim.sleqp_settings_create(C.byref(settings)) == 0
    assert shim.sleqp_fact_create_default(C.byref(fact), settings) == 0, shim.sleqp_error_msg()
    assert shim.sleqp_fact_name(fact) == b"hipfact"
    assert shim.sleqp_fact_flags(fact) == 2  # SLEQP_FACT_FLAGS_LOWER only
    N = c.N
    assert shim.sleqp_mat_create(C.byref(K), N, N, max(len(c.K_data), 1)) == 0
    for j in range(N):
        assert shim.sleqp_mat_push_col(K, j) == 0
        for e in range(c.K_cols[j], c.K_cols[j + 1]):
            assert shim.sleqp_mat_push(K, int(c.K_rows[e]), j, C.c_double(float(c.K_data[e]))) == 0
    assert shim.sleqp_fact_set_matrix(fact, K) == 0, shim.sleqp_error_msg()
    cond = C.c_double()
    assert shim.sleqp_fact_cond(fact, C.byref(cond)) == 0 and cond.value >= 1.0
    # project_nullspace: rhs resized to N, solution(0, n)  (standard_aug_jac.c:396-435)
    rhs = _vec(shim, c.n, c.g_idx, c.g_dat)
    assert shim.sleqp_vec_resize(rhs, N) == 0
    assert shim.sleqp_fact_solve(fact, rhs) == 0, shim.sleqp_error_msg()
    sol = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create_empty(C.byref(sol), c.n) == 0
    assert shim.sleqp_fact_solution(fact, sol, 0, c.n, C.c_double(ZERO_EPS)) == 0
    want = np.zeros(c.n)
    want[c.proj_idx] = c.proj_dat
    assert rel_err(_dense(sol), want) <= REL_TOL
    # second solution() call on the same solve: the dual part (standard_aug_jac.c:382-386)
    sol2 = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create_empty(C.byref(sol2), N - c.n) == 0
    assert shim.sleqp_fact_solution(fact, sol2, c.n, N, C.c_double(ZERO_EPS)) == 0
    want = np.zeros(N - c.n)
    want[c.lsq_idx] = c.lsq_dat
    assert rel_err(_dense(sol2), want) <= REL_TOL
    for v in (rhs, sol, sol2):
        shim.sleqp_vec_free(C.byref(v))
    shim.sleqp_mat_release(C.byref(K))
    assert shim.sleqp_fact_release(C.byref(fact)) == 0 and not fact
    shim.sleqp_settings_release(C.byref(settings))


@pytest.mark.gpu
@pytest.mark.parametrize("c", golden_cases(), ids=lambda c: c.name)
def test_aug_jac_shim_against_golden(shim, c):
    """The optional second boundary: SleqpAugJac with device assembly, driven through sleqp_aug_jac_*."""
    settings, problem, iterate, aug = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert shim.sleqp_settings_create(C.byref(settings)) == 0
    assert shim.sleqp_problem_create_mini(C.byref(problem), c.n, c.m) == 0
    assert shim.sleqp_iterate_create_mini(C.byref(iterate), problem) == 0
    _fill_jac(shim, shim.sleqp_iterate_cons_jac(iterate), c)
    ws = C.c_void_p(shim.sleqp_iterate_working_set(iterate))
    for j in np.argsort(np.where(c.var_index >= 0, c.var_index, 1 << 30)):
        if c.var_index[j] >= 0:
            assert shim.sleqp_working_set_add_var(ws, int(j), 1) == 0
    for i in np.argsort(np.where(c.cons_index >= 0, c.cons_index, 1 << 30)):
        if c.cons_index[i] >= 0:
            assert shim.sleqp_working_set_add_cons(ws, int(i), 1) == 0
    W = c.N - c.n
    assert shim.sleqp_working_set_size(ws) == W
    assert shim.sleqp_hipfact_aug_jac_create(C.byref(aug), problem, settings) == 0, shim.sleqp_error_msg()
    assert shim.sleqp_aug_jac_set_iterate(aug, iterate) == 0, shim.sleqp_error_msg()
    g = _vec(shim, c.n, c.g_idx, c.g_dat)
    sol = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create_empty(C.byref(sol), c.n) == 0
    assert shim.sleqp_aug_jac_project_nullspace(aug, g, sol) == 0, shim.sleqp_error_msg()
    want = np.zeros(c.n)
    want[c.proj_idx] = c.proj_dat
    assert rel_err(_dense(sol), want) <= REL_TOL
    assert g.contents.dim == c.n  # rhs left untouched
    dual = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create_empty(C.byref(dual), W) == 0
    assert shim.sleqp_aug_jac_solve_lsq(aug, g, dual) == 0
    want = np.zeros(W)
    want[c.lsq_idx] = c.lsq_dat
    assert rel_err(_dense(dual), want) <= REL_TOL
    b = _vec(shim, W, c.b_idx, c.b_dat)
    assert shim.sleqp_aug_jac_solve_min_norm(aug, b, sol) == 0
    want = np.zeros(c.n)
    want[c.mn_idx] = c.mn_dat
    assert rel_err(_dense(sol), want) <= REL_TOL
    assert [b.contents.indices[k] for k in range(b.contents.nnz)] == list(c.b_idx)  # shift undone
    for v in (g, sol, dual, b):
        shim.sleqp_vec_free(C.byref(v))
    assert shim.sleqp_aug_jac_release(C.byref(aug)) == 0
    shim.sleqp_iterate_release(C.byref(iterate))
    shim.sleqp_problem_release(C.byref(problem))
    shim.sleqp_settings_release(C.byref(settings))


@pytest.mark.gpu
@pytest.mark.parametrize("radius", [0.3, 1e3])
def test_tr_solver_shim_against_oracle(shim, radius):
    """SleqpTRCallbacks on the device (shim/tr_hipfact.c): sleqp_tr_solver_solve with the hipfact
    augmented Jacobian reproduces steihaug_solver_solve (oracle restatement) — step, iteration
    behaviour at the trust-region boundary and in the interior."""
    import scipy.sparse as sp

    import oracle
    from sleqp_amd import synth

    n, m = 300, 120
    J = synth.uniform_jacobian(n, m, 4, 7)
    vi, ci, W = synth.working_set_all_rows(n, m, 0.05, 7)
    B = sp.random(n, n, density=0.02, random_state=3)
    HL = sp.tril((B @ B.T + 0.5 * sp.eye(n)).tocsc(), format="csc")
    HL.sort_indices()
    g = np.random.default_rng(5).standard_normal(n)
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    want, _ = oracle.OracleFact(N, kc, kr, kd).steihaug(n, HL.indptr, HL.indices, HL.data, g, trust_radius=radius)

    class Case:  # what _fill_jac reads
        pass

    c = Case()
    c.n, c.jp, c.ji, c.jx = n, J.indptr, J.indices, J.data
    settings, problem, iterate, aug = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert shim.sleqp_settings_create(C.byref(settings)) == 0
    assert shim.sleqp_problem_create_mini(C.byref(problem), n, m) == 0
    assert shim.sleqp_iterate_create_mini(C.byref(iterate), problem) == 0
    _fill_jac(shim, shim.sleqp_iterate_cons_jac(iterate), c)
    ws = C.c_void_p(shim.sleqp_iterate_working_set(iterate))
    for j in np.argsort(np.where(vi >= 0, vi, 1 << 30)):
        if vi[j] >= 0:
            assert shim.sleqp_working_set_add_var(ws, int(j), 1) == 0
    for i in np.argsort(np.where(ci >= 0, ci, 1 << 30)):
        if ci[i] >= 0:
            assert shim.sleqp_working_set_add_cons(ws, int(i), 1) == 0
    assert shim.sleqp_hipfact_aug_jac_create(C.byref(aug), problem, settings) == 0, shim.sleqp_error_msg()
    assert shim.sleqp_aug_jac_set_iterate(aug, iterate) == 0, shim.sleqp_error_msg()
    shim.sleqp_hipfact_aug_jac_handle.restype = C.c_void_p
    assert shim.sleqp_hipfact_aug_jac_handle(aug)

    tr, ctl, H = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert shim.sleqp_hipfact_tr_solver_create(C.byref(tr), C.byref(ctl), problem, settings) == 0
    assert shim.sleqp_hipfact_tr_bind(ctl, aug) == 0, shim.sleqp_error_msg()
    assert shim.sleqp_mat_create(C.byref(H), n, n, max(HL.nnz, 1)) == 0
    for j in range(n):
        assert shim.sleqp_mat_push_col(H, j) == 0
        for e in range(HL.indptr[j], HL.indptr[j + 1]):
            assert shim.sleqp_mat_push(H, int(HL.indices[e]), j, C.c_double(float(HL.data[e]))) == 0
    assert shim.sleqp_hipfact_tr_set_hessian(ctl, H) == 0, shim.sleqp_error_msg()
    assert shim.sleqp_hipfact_tr_set_hessian(ctl, H) == 0  # same pattern: values-only path

    grad = _vec(shim, n, np.arange(n), g)
    mult = _vec(shim, m, [], [])
    step = C.POINTER(SleqpVecC)()
    assert shim.sleqp_vec_create_empty(C.byref(step), n) == 0
    dual = C.c_double()
    assert shim.sleqp_tr_solver_solve(tr, aug, mult, grad, step, C.c_double(radius), C.byref(dual)) == 0, \
        shim.sleqp_error_msg()
    got = _dense(step)
    assert rel_err(got, want) <= 1e-8
    assert np.linalg.norm(got) <= radius * (1 + 1e-10)
    lo, hi = C.c_double(), C.c_double()
    assert shim.sleqp_tr_solver_current_rayleigh(tr, C.byref(lo), C.byref(hi)) == 0

    for v in (grad, mult, step):
        shim.sleqp_vec_free(C.byref(v))
    shim.sleqp_mat_release(C.byref(H))
    assert shim.sleqp_tr_solver_release(C.byref(tr)) == 0 and not tr
    assert shim.sleqp_aug_jac_release(C.byref(aug)) == 0
    shim.sleqp_iterate_release(C.byref(iterate))
    shim.sleqp_problem_release(C.byref(problem))
    shim.sleqp_settings_release(C.byref(settings))
