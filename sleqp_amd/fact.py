"""Python mirror of the reference's SleqpFact / SleqpAugJac interfaces on top of
the hipfact C ABI (ctypes plumbing only — every numeric step happens inside
libhipfact.so on the GPU).

``HipFact``  mirrors the five SleqpFact callbacks (fact/fact_types.h:25-32,
dispatch in fact/fact.c:59-118): set_matrix / solve / solution / cond / free.
``StandardAugJac`` mirrors aug_jac/standard_aug_jac.c: set_iterate,
solve_min_norm (:306-350), solve_lsq (:352-394), project_nullspace (:396-435).
The production binding is the C shim in shim/fact_hipfact.c; this module is
what the parity tests and bench.py drive.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import HipfactError
from .sparse import SleqpMat, SleqpVec

SLEQP_FACT_FLAGS_LOWER = 1 << 1  # fact/fact.h:13
SLEQP_NONE = -1


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class HipFact:
    """One backend instance (one SleqpFact object, fact/fact.c:21-45)."""

    name = "hipfact"
    flags = SLEQP_FACT_FLAGS_LOWER

    def __init__(self, device: int = -1, **options):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        rc = self._lib.hipfact_create(C.byref(self._h), device)
        if rc != 0:
            raise HipfactError(rc, self._lib.hipfact_last_error(None).decode())
        self.N = 0
        for k, v in options.items():
            self.set_option(k, v)

    # -- helpers -----------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise HipfactError(rc, self._lib.hipfact_last_error(self._h).decode())

    def set_option(self, name: str, value: float):
        self._check(self._lib.hipfact_set_option(self._h, name.encode(), float(value)))

    def info(self, name: str) -> float:
        v = C.c_double()
        self._check(self._lib.hipfact_get_info(self._h, name.encode(), C.byref(v)))
        return v.value

    @property
    def stream(self) -> int:
        s = C.c_void_p()
        self._check(self._lib.hipfact_stream(self._h, C.byref(s)))
        return s.value or 0

    # -- SleqpFact callbacks -------------------------------------------------
    def set_matrix(self, mat: SleqpMat):
        """SLEQP_FACT_SET_MATRIX (fact/fact_types.h:9): lower-triangular CSC K."""
        assert mat.num_rows == mat.num_cols
        self._keep = (mat.cols, mat.rows, mat.data)
        self.N = mat.num_cols
        self._check(self._lib.hipfact_set_matrix(self._h, mat.num_cols, _ptr(mat.cols), _ptr(mat.rows), _ptr(mat.data)))

    def solve(self, rhs):
        """SLEQP_FACT_SOLVE (fact/fact_types.h:12): sparse SleqpVec or dense array."""
        if isinstance(rhs, SleqpVec):
            idx = np.ascontiguousarray(rhs.indices, dtype=np.int32)
            val = np.ascontiguousarray(rhs.data, dtype=np.float64)
            self._check(self._lib.hipfact_solve_sparse(self._h, rhs.dim, rhs.nnz, _ptr(idx), _ptr(val)))
        else:
            b = np.ascontiguousarray(rhs, dtype=np.float64)
            assert b.size == self.N
            self._check(self._lib.hipfact_solve_dense(self._h, _ptr(b)))

    def solution_raw(self, begin: int, end: int) -> np.ndarray:
        out = np.empty(max(end - begin, 0), dtype=np.float64)
        self._check(self._lib.hipfact_solution(self._h, _ptr(out), begin, end))
        return out

    def solution(self, begin: int, end: int, zero_eps: float = 1e-20) -> SleqpVec:
        """SLEQP_FACT_SOLUTION (fact/fact_types.h:14-18) incl. the
        sleqp_vec_set_from_raw packing every reference backend applies."""
        return SleqpVec.from_raw(self.solution_raw(begin, end), zero_eps)

    def cond(self) -> float:
        v = C.c_double()
        self._check(self._lib.hipfact_condition(self._h, C.byref(v)))
        return v.value

    # -- device-resident variants ------------------------------------------
    def refactor_device(self, d_vals_ptr: int):
        self._check(self._lib.hipfact_refactor_device(self._h, C.c_void_p(d_vals_ptr)))

    def solve_device(self, d_rhs_ptr: int, d_sol_ptr: int):
        self._check(self._lib.hipfact_solve_device(self._h, C.c_void_p(d_rhs_ptr), C.c_void_p(d_sol_ptr)))

    def synchronize(self):
        self._check(self._lib.hipfact_synchronize(self._h))

    def check(self):
        """Blocks, finishes the refinement of the last solve and raises what the asynchronous
        device-resident entry points could not report (singular, stalled, timed out)."""
        self._check(self._lib.hipfact_check(self._h))

    def assemble_kkt(self, J: SleqpMat, var_index, cons_index, working_set_size: int, want_arrays: bool = True):
        """fill_aug_jac on the device (aug_jac/standard_aug_jac.c:135-237);
        returns the assembled lower CSC K (optional) and factors it."""
        n, m_total = J.num_cols, J.num_rows
        var_index = np.ascontiguousarray(var_index, dtype=np.int32)
        cons_index = np.ascontiguousarray(cons_index, dtype=np.int32)
        nav = int((var_index >= 0).sum())
        cap = n + J.nnz + nav
        N = n + working_set_size
        k_nnz = C.c_int(0)
        kp = np.empty(N + 1, dtype=np.int32) if want_arrays else None
        ki = np.empty(max(cap, 1), dtype=np.int32) if want_arrays else None
        kx = np.empty(max(cap, 1), dtype=np.float64) if want_arrays else None
        self._check(self._lib.hipfact_assemble_kkt(self._h, n, m_total, _ptr(J.cols), _ptr(J.rows), _ptr(J.data),
                                                   _ptr(var_index), _ptr(cons_index), working_set_size,
                                                   C.byref(k_nnz) if want_arrays else None, _ptr(kp), _ptr(ki), _ptr(kx)))
        self.N = N
        if not want_arrays:
            return None
        nnz = k_nnz.value
        return SleqpMat(N, N, kp, ki[:nnz].copy(), kx[:nnz].copy())

    def last_warning(self):
        """hipfact_last_warning: text, or None (rank-deficient working set factored with static pivoting)."""
        w = self._lib.hipfact_last_warning(self._h)
        return w.decode() if w else None

    def steihaug(self, hess: "SpMat", gradient, trust_radius: float, stat_tol: float = 1e-6, max_iter: int = 100):
        """Device-resident projected CG (tr/steihaug_solver.c:218-496): returns (step, tr_dual, iterations)."""
        g = np.ascontiguousarray(gradient, dtype=np.float64)
        step = np.empty_like(g)
        dual, its = C.c_double(), C.c_int()
        self._check(self._lib.hipfact_steihaug_solve(self._h, hess._m, _ptr(g), float(trust_radius), stat_tol * 1e-2,
                                                     int(max_iter), _ptr(step), C.byref(dual), C.byref(its)))
        return step, dual.value, its.value

    def reduced_matrix(self):
        """hipfact_reduced_matrix: S = A_W A_W^T (sparse, lower CSC, working-set row order) from the device."""
        import scipy.sparse as sp

        nnz = C.c_int()
        self._check(self._lib.hipfact_reduced_matrix(self._h, C.byref(nnz), None, None, None))
        m = int(self.info("m"))
        cp = np.empty(m + 1, dtype=np.int32)
        ri = np.empty(max(nnz.value, 1), dtype=np.int32)
        vx = np.empty(max(nnz.value, 1), dtype=np.float64)
        self._check(self._lib.hipfact_reduced_matrix(self._h, C.byref(nnz), _ptr(cp), _ptr(ri), _ptr(vx)))
        return sp.csc_matrix((vx[:nnz.value], ri[:nnz.value], cp), shape=(m, m))

    def tr_solve(self, hess, gradient, trust_radius: float, method: int = 1, stat_tol: float = 1e-6,
                 max_iter: int = 100, time_limit: float = -1.0):
        """hipfact_tr_solve_ex: method 0 = projected Steihaug CG (tr/steihaug_solver.c), 1 = generalised Lanczos
        (what trlib runs, tr/trlib_solver.c).  `hess` is an SpMat (explicit lower-triangular Hessian in HBM)
        or a callable d -> H d on host arrays (the matrix-free SLEQP_FUNC_HESS_PROD).  Returns
        (step, tr_dual, iterations); `time_limit` (seconds, < 0 = SLEQP_NONE) and the rest of the
        SleqpTRCallbacks contract (tr/tr_types.h:9-29) are left in `self.last_tr`: timed_out, min_rayleigh,
        max_rayleigh."""
        g = np.ascontiguousarray(gradient, dtype=np.float64)
        n = g.size
        step = np.empty_like(g)
        dual, its = C.c_double(), C.c_int()
        cb_type = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))
        if isinstance(hess, SpMat):
            mat, cb = hess._m, C.cast(None, cb_type)
        else:
            def _prod(_user, direction, product):
                try:
                    d = np.ctypeslib.as_array(direction, shape=(n,))
                    np.ctypeslib.as_array(product, shape=(n,))[:] = hess(d)
                    return 0
                except Exception:  # noqa: BLE001
                    return -1

            mat, cb = None, cb_type(_prod)
        extra = TrExtra(float(time_limit), 0, 1.0, 1.0)
        self._check(self._lib.hipfact_tr_solve_ex(self._h, int(method), mat, cb, None, _ptr(g), float(trust_radius),
                                                  stat_tol * 1e-2, int(max_iter), _ptr(step), C.byref(dual),
                                                  C.byref(its), C.byref(extra)))
        self.last_tr = {"timed_out": bool(extra.timed_out), "min_rayleigh": extra.min_rayleigh,
                        "max_rayleigh": extra.max_rayleigh}
        return step, dual.value, its.value

    def free(self):
        if self._h:
            self._lib.hipfact_free(C.byref(self._h))
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class TrExtra(C.Structure):
    """hipfact_tr_extra (include/hipfact.h)"""
    _fields_ = [("time_limit", C.c_double), ("timed_out", C.c_int), ("min_rayleigh", C.c_double),
                ("max_rayleigh", C.c_double)]


class SpMat:
    """Device-resident sparse matrix for the products around the EQP step
    (sleqp_mat_mult_vec / _trans, sparse/mat.c:282-363; symmetric Hessian
    product precedent bindings/mex/mex_hess.c:85-139)."""

    def __init__(self, fact: HipFact, mat: SleqpMat):
        self._fact = fact
        self._lib = fact._lib
        self.mat = mat
        self._m = C.c_void_p()
        fact._check(self._lib.hipfact_spmat_create(fact._h, mat.num_rows, mat.num_cols, _ptr(mat.cols), _ptr(mat.rows),
                                                   _ptr(mat.data), C.byref(self._m)))

    def mult_vec(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x.to_raw() if isinstance(x, SleqpVec) else x, dtype=np.float64)
        y = np.empty(self.mat.num_rows)
        self._fact._check(self._lib.hipfact_spmat_mult_vec(self._m, _ptr(x), _ptr(y)))
        return y

    def mult_vec_trans(self, x, eps: float = 0.0) -> SleqpVec:
        x = np.ascontiguousarray(x.to_raw() if isinstance(x, SleqpVec) else x, dtype=np.float64)
        y = np.empty(self.mat.num_cols)
        self._fact._check(self._lib.hipfact_spmat_mult_vec_trans(self._m, _ptr(x), _ptr(y)))
        return SleqpVec.from_raw(y, eps)

    def mult_vec_sym(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.mat.num_rows)
        self._fact._check(self._lib.hipfact_spmat_mult_vec_sym(self._m, _ptr(x), _ptr(y)))
        return y

    def mult_device(self, trans: int, d_x: int, d_y: int):
        self._fact._check(self._lib.hipfact_spmat_mult_device(self._m, trans, C.c_void_p(d_x), C.c_void_p(d_y)))

    def update_values(self, vals):
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        self._fact._check(self._lib.hipfact_spmat_update_values(self._m, _ptr(vals)))

    def free(self):
        if self._m:
            self._lib.hipfact_spmat_free(C.byref(self._m))
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class StandardAugJac:
    """Mirror of aug_jac/standard_aug_jac.c on top of a HipFact backend."""

    def __init__(self, num_variables: int, fact: HipFact, zero_eps: float = 1e-20, device_assembly: bool = True):
        self.n = int(num_variables)
        self.fact = fact
        self.zero_eps = zero_eps
        self.device_assembly = device_assembly
        self.working_set_size = 0
        self.condition = SLEQP_NONE
        self.K = None

    def set_iterate(self, cons_jac: SleqpMat, var_index, cons_index):
        """aug_jac_set_iterate (standard_aug_jac.c:239-293): assemble K (lower,
        since the backend declares SLEQP_FACT_FLAGS_LOWER), factor, condition."""
        var_index = np.asarray(var_index, dtype=np.int32)
        cons_index = np.asarray(cons_index, dtype=np.int32)
        self.working_set_size = int((var_index >= 0).sum() + (cons_index >= 0).sum())
        if self.device_assembly:
            self.K = self.fact.assemble_kkt(cons_jac, var_index, cons_index, self.working_set_size)
        else:
            from .synth import kkt_lower_from_jacobian

            N, cp, ri, vx = kkt_lower_from_jacobian(cons_jac.to_scipy(), var_index, cons_index)
            self.K = SleqpMat(N, N, cp, ri, vx)
            self.fact.set_matrix(self.K)
        self.condition = self.fact.cond()

    def solve_min_norm(self, rhs: SleqpVec) -> SleqpVec:
        assert rhs.dim == self.working_set_size
        total = self.n + self.working_set_size
        self.fact.solve(rhs.shifted(self.n, total))
        return self.fact.solution(0, self.n, self.zero_eps)

    def solve_lsq(self, rhs: SleqpVec) -> SleqpVec:
        assert rhs.dim == self.n
        total = self.n + self.working_set_size
        self.fact.solve(rhs.resized(total))
        return self.fact.solution(self.n, total, self.zero_eps)

    def project_nullspace(self, rhs: SleqpVec) -> SleqpVec:
        assert rhs.dim == self.n
        total = self.n + self.working_set_size
        self.fact.solve(rhs.resized(total))
        return self.fact.solution(0, self.n, self.zero_eps)
