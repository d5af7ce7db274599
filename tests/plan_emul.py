"""CPU emulator of the symbolic plan (TEST INFRASTRUCTURE, not product code).

Executes the multifrontal schedule that sleqp_amd/csrc/analysis.cpp produces
with dense numpy operations, front by front, exactly in the data layout the
device kernels use (L arena panels, U arena update matrices, relative indices,
product lists).  It exists so that the integer machinery of the analysis can be
validated in a container without a GPU; the device numerics are validated
separately against oracle/ on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

_NAMES = [
    "Kp", "Ki", "perm", "iperm", "Mp", "Mi", "Mtarget", "prod_ptr", "prod_a", "prod_b", "src", "sn_c0", "sn_r",
    "sn_rowptr", "sn_rows", "sn_parent", "sn_level", "sn_Loff", "sn_Uoff", "sn_uoff", "child_ptr", "child_idx", "dense_cols", "late_cols",
    "rel_ptr", "rel", "level_ptr", "level_sn", "Ar_ptr", "Ar_col", "Ar_src", "Kc_y",
    "bnd_row", "bnd_col", "row_ext", "ent_ext", "cut_ptr", "cut_row", "cut_ent",
]
_SCALARS = [
    "N", "n", "m", "my", "n_late", "n_late_rows", "saddle", "nnzK", "nsuper", "nlevels", "L_size", "U_size", "u_size", "nnzL", "nnzL_true", "flops",
    "flops_dense", "nprod", "max_r", "max_w", "max_u", "t_order", "t_symbolic", "t_total", "N_ext", "n_bounds",
]


class Plan:
    """Host copy of the symbolic plan, fetched through the C ABI."""

    def __init__(self, lib, N, colptr, rowidx, vals=None):
        colptr = np.ascontiguousarray(colptr, dtype=np.int32)
        rowidx = np.ascontiguousarray(rowidx, dtype=np.int32)
        lib.hipfact_plan_create.restype = C.c_int
        lib.hipfact_plan_error.restype = C.c_char_p
        p = C.c_void_p()
        vptr = None
        if vals is not None:
            vals = np.ascontiguousarray(vals, dtype=np.float64)
            vptr = vals.ctypes.data_as(C.c_void_p)
        rc = lib.hipfact_plan_create(C.c_int(N), colptr.ctypes.data_as(C.c_void_p),
                                     rowidx.ctypes.data_as(C.c_void_p), vptr, C.byref(p))
        if rc != 0:
            msg = lib.hipfact_plan_error(p).decode() if p else "?"
            lib.hipfact_plan_free(C.byref(p))
            raise RuntimeError(f"hipfact_plan_create failed ({rc}): {msg}")
        try:
            for name in _NAMES:
                data = C.c_void_p()
                ln = C.c_int64()
                es = C.c_int()
                rc = lib.hipfact_plan_array(p, name.encode(), C.byref(data), C.byref(ln), C.byref(es))
                assert rc == 0, name
                dt = np.int32 if es.value == 4 else np.int64
                if ln.value == 0:
                    arr = np.zeros(0, dtype=dt)
                else:
                    buf = (C.c_char * (ln.value * es.value)).from_address(data.value)
                    arr = np.frombuffer(buf, dtype=dt).copy()
                setattr(self, name, arr)
            for name in _SCALARS:
                v = C.c_double()
                rc = lib.hipfact_plan_scalar(p, name.encode(), C.byref(v))
                assert rc == 0, name
                setattr(self, name, v.value)
            for name in ("N", "n", "m", "my", "n_late", "n_late_rows", "nsuper", "nlevels", "L_size", "U_size", "u_size", "nnzK", "nprod", "N_ext", "n_bounds"):
                setattr(self, name, int(getattr(self, name)))
            self.saddle = bool(self.saddle)
        finally:
            lib.hipfact_plan_free(C.byref(p))


def _mvals(P: Plan, Kx: np.ndarray) -> np.ndarray:
    nM = len(P.Mi)
    if P.saddle:
        prods = Kx[P.prod_a] * Kx[P.prod_b]
        seg = np.repeat(np.arange(nM), np.diff(P.prod_ptr))
        return np.bincount(seg, weights=prods, minlength=nM)
    out = np.zeros(nM)
    ok = P.src >= 0
    out[ok] = Kx[P.src[ok]]
    return out


class EmulFactor:
    """Numeric multifrontal LDL^T executed on the plan with numpy."""

    def __init__(self, P: Plan, Kx: np.ndarray):
        self.P = P
        self.Kx = np.asarray(Kx, dtype=np.float64)
        # active bounds eliminated by the analysis (Plan::n_bounds): the plan is that of the reduced matrix K', whose
        # entries are a subset of the caller's (ent_ext); the caller's values are kept for the cut entries A'_B
        self.Kx_ext = self.Kx
        if P.n_bounds > 0:
            self.Kx = self.Kx_ext[P.ent_ext]
        L = np.zeros(max(P.L_size, 1))
        U = np.zeros(max(P.U_size, 1))
        mv = _mvals(P, self.Kx)
        np.add.at(L, P.Mtarget, mv)
        if P.saddle and P.n_late > 0:
            # late variables: M = [A_s A_s^T  A_d; A_d^T  -I] - the diagonal of their columns is -1 (k_diag_inactive)
            late_k = np.nonzero(P.perm >= P.my)[0]
            L[P.Mtarget[P.Mp[late_k]]] = -1.0
        self.d = np.zeros(P.m)
        for lev in range(P.nlevels):
            for s in P.level_sn[P.level_ptr[lev]:P.level_ptr[lev + 1]]:
                self._front(s, L, U)
        self.L = L

    def _panel(self, s, L):
        P = self.P
        w = P.sn_c0[s + 1] - P.sn_c0[s]
        r = P.sn_r[s]
        return L[P.sn_Loff[s]:P.sn_Loff[s] + r * w].reshape((w, r)).T, w, r  # column-major view

    def _umat(self, s, U):
        P = self.P
        w = P.sn_c0[s + 1] - P.sn_c0[s]
        u = P.sn_r[s] - w
        return U[P.sn_Uoff[s]:P.sn_Uoff[s] + u * u].reshape((u, u)).T, u

    def _front(self, s, L, U):
        P = self.P
        panel, w, r = self._panel(s, L)
        Us, u = self._umat(s, U)
        Us[:, :] = 0.0
        for c in P.child_idx[P.child_ptr[s]:P.child_ptr[s + 1]]:
            assert P.sn_level[c] < P.sn_level[s]
            Uc, uc = self._umat(c, U)
            rel = P.rel[P.rel_ptr[c]:P.rel_ptr[c] + uc]
            assert np.all(np.diff(rel) > 0)
            a, b = np.tril_indices(uc)
            ta, tb = rel[a], rel[b]
            inp = tb < w
            np.add.at(panel, (ta[inp], tb[inp]), Uc[a[inp], b[inp]])
            np.add.at(Us, (ta[~inp] - w, tb[~inp] - w), Uc[a[~inp], b[~inp]])
        # dense partial LDL^T without pivoting
        F11 = np.tril(panel[:w, :w]) + np.tril(panel[:w, :w], -1).T
        Lk = np.eye(w)
        d = np.zeros(w)
        A = F11.copy()
        for k in range(w):
            d[k] = A[k, k]
            if d[k] == 0.0 or not np.isfinite(d[k]):
                raise ZeroDivisionError(f"zero pivot in front {s} column {k}")
            Lk[k + 1:, k] = A[k + 1:, k] / d[k]
            A[k + 1:, k + 1:] -= np.outer(Lk[k + 1:, k], Lk[k + 1:, k]) * d[k]
        self.d[P.sn_c0[s]:P.sn_c0[s + 1]] = d
        if u > 0:
            Y = np.linalg.solve(Lk, panel[w:, :w].T).T  # L21 * D
            L21 = Y / d[None, :]
            panel[w:, :w] = L21
            upd = Y @ L21.T
            Us -= np.tril(upd)
        panel[:w, :w] = np.tril(Lk, -1) + np.diag(d)

    def solve_m(self, t):
        """Solve M y = t in pivot order (t, y indexed by pivot position)."""
        P = self.P
        y = np.array(t, dtype=np.float64)
        for s in range(P.nsuper):  # forward (children before parents)
            panel, w, r = self._panel(s, self.L)
            rows = P.sn_rows[P.sn_rowptr[s]:P.sn_rowptr[s] + r]
            L11 = np.tril(panel[:w, :w], -1) + np.eye(w)
            y[rows[:w]] = np.linalg.solve(L11, y[rows[:w]])
            if r > w:
                y[rows[w:]] -= panel[w:, :w] @ y[rows[:w]]
        y /= self.d
        for s in range(P.nsuper - 1, -1, -1):
            panel, w, r = self._panel(s, self.L)
            rows = P.sn_rows[P.sn_rowptr[s]:P.sn_rowptr[s] + r]
            L11 = np.tril(panel[:w, :w], -1) + np.eye(w)
            v = y[rows[:w]]
            if r > w:
                v = v - panel[w:, :w].T @ y[rows[w:]]
            y[rows[:w]] = np.linalg.solve(L11.T, v)
        return y

    def solve(self, b):
        """Solve K z = b (the caller's K, original ordering)."""
        P = self.P
        b = np.asarray(b, dtype=np.float64)
        if P.n_bounds == 0:
            return self._solve_plan(b)
        # x_B = beta; K' [x; y'] = [b_x; b_y' - A'_B beta]; y_B = b_B - beta - A'_B^T y'  (plan.h)
        n = P.n
        beta = b[n + P.bnd_row]
        cut_of = np.repeat(np.arange(P.n_bounds), np.diff(P.cut_ptr))
        cut_val = self.Kx_ext[P.cut_ent]
        b2 = np.concatenate([b[:n], b[n + P.row_ext]])
        b2[n:] -= np.bincount(P.cut_row, weights=cut_val * beta[cut_of], minlength=len(P.row_ext))
        z2 = self._solve_plan(b2)
        z = np.empty(P.N_ext)
        z[:n] = z2[:n]
        z[P.bnd_col] = beta
        z[n + P.row_ext] = z2[n:]
        z[n + P.bnd_row] = b[P.bnd_col] - beta - np.bincount(cut_of, weights=cut_val * z2[n + P.cut_row], minlength=P.n_bounds)
        return z

    def _solve_plan(self, b):
        """Solve K' z = b for the matrix the plan was built for."""
        P = self.P
        if not P.saddle:
            t = b[P.perm]
            y = self.solve_m(t)
            z = np.empty_like(b)
            z[P.perm] = y
            return z
        n, m, my = P.n, P.m, P.my
        bx, by = b[:n], b[n:]
        # t_p = A_p bx - by[perm]; rows of constraint rows leave the late columns out (they are unknowns of M), the
        # row of a late variable x_d is its own unit entry: t = b_d, no b_y term
        seg = np.repeat(np.arange(m), np.diff(P.Ar_ptr))
        is_y = P.perm < my
        late_col = np.zeros(max(n, 1), dtype=bool)
        late_col[P.late_cols] = True
        Av = self.Kx[P.Ar_src] * bx[P.Ar_col]
        Av[late_col[P.Ar_col] & is_y[seg]] = 0.0
        t = np.bincount(seg, weights=Av, minlength=m).astype(np.float64)
        t[is_y] -= by[P.perm[is_y]]
        yp = self.solve_m(t) if m > 0 else np.zeros(0)
        # x_j = b_j - sum_e K[e] y_p[Kc_y[e]] (every column, the late ones too: x_d = b_d - A_d^T y)
        x = bx.copy()
        off = P.Kc_y >= 0
        col_of = np.repeat(np.arange(n), np.diff(P.Kp[:n + 1]))
        x -= np.bincount(col_of[off], weights=self.Kx[off] * yp[P.Kc_y[off]], minlength=n)
        if P.n_late > 0:  # the solve's own value of the late variables is -x_d
            late_k = np.nonzero(~is_y)[0]
            assert np.allclose(-yp[late_k], x[P.late_cols[P.perm[late_k] - my]], rtol=1e-8, atol=1e-8 * max(1.0, np.abs(x).max()))
        z = np.empty(n + my)
        z[:n] = x
        z[n + P.perm[is_y]] = yp[is_y]
        return z
