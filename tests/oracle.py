"""ctypes wrapper around oracle/liboracle.so (TEST INFRASTRUCTURE).

The oracle is the CPU restatement of the reference path; see oracle/README.md.
This module may only be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "oracle", "liboracle.so")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(["make", "-C", os.path.join(_ROOT, "oracle")], stdout=subprocess.DEVNULL)
        L = C.CDLL(_PATH)
        L.oracle_fact_set_matrix.restype = C.c_void_p
        L.oracle_fact_raw_solution.restype = C.POINTER(C.c_double)
        L.oracle_ldl_factor.restype = C.c_void_p
        L.oracle_ldl_lnz.restype = C.c_long
        L.oracle_ldl_flops.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def vec_to_raw(dim, indices, data):
    indices, data = _i32(indices), _f64(data)
    out = np.empty(dim)
    lib().oracle_vec_to_raw(C.c_int(dim), C.c_int(indices.size), _p(indices), _p(data), _p(out))
    return out


def vec_set_from_raw(values, zero_eps):
    values = _f64(values)
    idx = np.empty(max(values.size, 1), dtype=np.int32)
    dat = np.empty(max(values.size, 1))
    nnz = lib().oracle_vec_set_from_raw(_p(values), C.c_int(values.size), C.c_double(zero_eps), _p(idx), _p(dat))
    return idx[:nnz].copy(), dat[:nnz].copy()


def mat_mult_vec(num_rows, num_cols, cols, rows, data, x_idx, x_dat):
    cols, rows, data, x_idx, x_dat = _i32(cols), _i32(rows), _f64(data), _i32(x_idx), _f64(x_dat)
    out = np.empty(num_rows)
    lib().oracle_mat_mult_vec(C.c_int(num_rows), C.c_int(num_cols), _p(cols), _p(rows), _p(data),
                              C.c_int(x_idx.size), _p(x_idx), _p(x_dat), _p(out))
    return out


def mat_mult_vec_trans(num_rows, num_cols, cols, rows, data, x_idx, x_dat, eps):
    cols, rows, data, x_idx, x_dat = _i32(cols), _i32(rows), _f64(data), _i32(x_idx), _f64(x_dat)
    ri = np.empty(max(num_cols, 1), dtype=np.int32)
    rd = np.empty(max(num_cols, 1))
    nnz = lib().oracle_mat_mult_vec_trans(C.c_int(num_rows), C.c_int(num_cols), _p(cols), _p(rows), _p(data),
                                          C.c_int(x_idx.size), _p(x_idx), _p(x_dat), C.c_double(eps), _p(ri), _p(rd))
    return ri[:nnz].copy(), rd[:nnz].copy()


def hess_prod_lower(dim, jc, ir, pr, direction):
    jc, ir, pr, direction = _i32(jc), _i32(ir), _f64(pr), _f64(direction)
    out = np.empty(dim)
    rc = lib().oracle_hess_prod_lower(C.c_int(dim), _p(jc), _p(ir), _p(pr), _p(direction), _p(out))
    if rc != 0:
        raise ValueError("Hessian entry above the diagonal")
    return out


def fill_aug_jac(n, m_total, jp, ji, jx, var_index, cons_index, lower_only=True):
    """Returns (N, cols, rows, data) exactly as fill_aug_jac builds them."""
    jp, ji, jx = _i32(jp), _i32(ji), _f64(jx)
    var_index, cons_index = _i32(var_index), _i32(cons_index)
    nav = int((var_index >= 0).sum())
    W = nav + int((cons_index >= 0).sum())
    cap = lib().oracle_reserve_aug_jac(C.c_int(n), C.c_int(int(jp[n]) if n > 0 else 0), C.c_int(nav),
                                       C.c_int(1 if lower_only else 0))
    N = n + W
    cols = np.zeros(N + 2, dtype=np.int32)
    rows = np.zeros(max(cap, 1), dtype=np.int32)
    data = np.zeros(max(cap, 1))
    nnz = lib().oracle_fill_aug_jac(C.c_int(n), C.c_int(m_total), _p(jp), _p(ji), _p(jx), _p(var_index),
                                    _p(cons_index), C.c_int(W), C.c_int(1 if lower_only else 0), _p(cols), _p(rows),
                                    _p(data))
    assert nnz >= 0
    return N, cols[:N + 1].copy(), rows[:nnz].copy(), data[:nnz].copy()


class OracleFact:
    """fact_lapack.c restated: dense LU with partial pivoting of the symmetric densified K."""

    def __init__(self, N, cols, rows, data):
        self._keep = (_i32(cols), _i32(rows), _f64(data))
        self.N = int(N)
        self._f = lib().oracle_fact_set_matrix(C.c_int(self.N), _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]))
        if not self._f:
            raise ZeroDivisionError("Failed to factorize using LAPACK (oracle)")
        self._f = C.c_void_p(self._f)

    def solve_sparse(self, indices, data):
        indices, data = _i32(indices), _f64(data)
        lib().oracle_fact_solve(self._f, C.c_int(indices.size), _p(indices), _p(data))

    def solve_dense(self, rhs):
        rhs = _f64(rhs)
        assert rhs.size == self.N
        lib().oracle_fact_solve_dense(self._f, _p(rhs))

    def raw_solution(self):
        ptr = lib().oracle_fact_raw_solution(self._f)
        return np.ctypeslib.as_array(ptr, shape=(max(self.N, 1),))[: self.N].copy()

    def solution(self, begin, end, zero_eps):
        idx = np.empty(max(end - begin, 1), dtype=np.int32)
        dat = np.empty(max(end - begin, 1))
        nnz = lib().oracle_fact_solution(self._f, C.c_int(begin), C.c_int(end), C.c_double(zero_eps), _p(idx), _p(dat))
        return idx[:nnz].copy(), dat[:nnz].copy()

    def _aug(self, fn, n, rhs_idx, rhs_dat, zero_eps, out_dim):
        rhs_idx, rhs_dat = _i32(rhs_idx).copy(), _f64(rhs_dat)
        idx = np.empty(max(out_dim, 1), dtype=np.int32)
        dat = np.empty(max(out_dim, 1))
        nnz = fn(self._f, C.c_int(n), C.c_int(rhs_idx.size), _p(rhs_idx), _p(rhs_dat), C.c_double(zero_eps), _p(idx), _p(dat))
        return idx[:nnz].copy(), dat[:nnz].copy()

    def solve_min_norm(self, n, rhs_idx, rhs_dat, zero_eps=1e-20):
        return self._aug(lib().oracle_aug_jac_solve_min_norm, n, rhs_idx, rhs_dat, zero_eps, n)

    def solve_lsq(self, n, rhs_idx, rhs_dat, zero_eps=1e-20):
        return self._aug(lib().oracle_aug_jac_solve_lsq, n, rhs_idx, rhs_dat, zero_eps, self.N - n)

    def project_nullspace(self, n, rhs_idx, rhs_dat, zero_eps=1e-20):
        return self._aug(lib().oracle_aug_jac_project_nullspace, n, rhs_idx, rhs_dat, zero_eps, n)

    def steihaug(self, n, hc, hr, hx, gradient, trust_radius, stat_tol=1e-6, max_iter=100):
        hc, hr, hx, gradient = _i32(hc), _i32(hr), _f64(hx), _f64(gradient)
        step = np.empty(max(n, 1))
        its = lib().oracle_steihaug_solve(self._f, C.c_int(n), _p(hc), _p(hr), _p(hx), _p(gradient),
                                          C.c_double(trust_radius), C.c_double(stat_tol), C.c_int(max_iter), _p(step))
        assert its >= 0
        return step[:n].copy(), its

    def steihaug_rayleigh(self, n, hc, hr, hx, gradient, trust_radius, stat_tol=1e-6, max_iter=100):
        """steihaug() plus the Rayleigh bounds steihaug_solver_rayleigh would report: (step, its, min, max)."""
        hc, hr, hx, gradient = _i32(hc), _i32(hr), _f64(hx), _f64(gradient)
        step = np.empty(max(n, 1))
        lo, hi = C.c_double(), C.c_double()
        fn = lib().oracle_steihaug_solve_rayleigh
        fn.restype = C.c_int
        its = fn(self._f, C.c_int(n), _p(hc), _p(hr), _p(hx), _p(gradient), C.c_double(trust_radius),
                 C.c_double(stat_tol), C.c_int(max_iter), _p(step), C.byref(lo), C.byref(hi))
        assert its >= 0
        return step[:n].copy(), its, lo.value, hi.value

    def __del__(self):
        try:
            if self._f:
                lib().oracle_fact_free(self._f)
                self._f = None
        except Exception:
            pass


class OracleLdl:
    """Up-looking simplicial sparse LDL^T (large-N CPU baseline)."""

    def __init__(self, N, cols, rows, data, perm=None, symbolic_only=False):
        self._keep = (_i32(cols), _i32(rows), _f64(data), None if perm is None else _i32(perm))
        self.N = int(N)
        pp = None if perm is None else _p(self._keep[3])
        f = lib().oracle_ldl_factor(C.c_int(self.N), _p(self._keep[0]), _p(self._keep[1]), _p(self._keep[2]), pp,
                                    C.c_int(1 if symbolic_only else 0))
        if not f:
            raise ZeroDivisionError("oracle_ldl_factor failed (zero pivot or out of memory)")
        self._f = C.c_void_p(f)
        self.lnz = lib().oracle_ldl_lnz(self._f)
        self.flops = lib().oracle_ldl_flops(self._f)

    def solve(self, b):
        b = _f64(b)
        x = np.empty(self.N)
        lib().oracle_ldl_solve(self._f, _p(b), _p(x))
        return x

    def refactor(self, data):
        """Numeric-only refactorisation with new values of the same pattern (symbolic phase reused)."""
        d = _f64(data)
        if lib().oracle_ldl_refactor(self._f, _p(self._keep[0]), _p(self._keep[1]), _p(d)) != 0:
            raise ZeroDivisionError("oracle_ldl_refactor failed (zero pivot)")

    def __del__(self):
        try:
            if self._f:
                lib().oracle_ldl_free(self._f)
                self._f = None
        except Exception:
            pass


def reduced_aug_jac(n, m_total, jp, ji, jx, var_index, cons_index, W):
    """The matrix reduced_aug_jac.c hands to a PSD-flagged backend (compute_matrix_lower, :323-377): lower CSC of
    A_W A_W^T with EVERY entry below the diagonal stored (the reference's inner product reports nonzero
    unconditionally).  Returns (cols, rows, data)."""
    jp, ji = np.ascontiguousarray(jp, dtype=np.int32), np.ascontiguousarray(ji, dtype=np.int32)
    jx = np.ascontiguousarray(jx, dtype=np.float64)
    vi, ci = np.ascontiguousarray(var_index, dtype=np.int32), np.ascontiguousarray(cons_index, dtype=np.int32)
    cap = max(W * (W + 1) // 2, 1)
    cols, rows, data = np.zeros(W + 1, dtype=np.int32), np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.float64)
    f = lib().oracle_reduced_aug_jac
    f.restype = C.c_int
    nnz = f(C.c_int(n), C.c_int(m_total), _p(jp), _p(ji), _p(jx), _p(vi), _p(ci), C.c_int(W), _p(cols), _p(rows), _p(data))
    assert nnz == W * (W + 1) // 2
    return cols, rows[:nnz], data[:nnz]
