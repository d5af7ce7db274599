"""Seeded synthetic KKT inputs.

Generators for the constraint Jacobian J (m x n, CSC like the reference's
``cons_jac``: iterate.c:87) and the lower-triangular augmented matrix
``K = [I A_W^T; A_W 0]`` in exactly the CSC layout ``fill_aug_jac`` produces
(aug_jac/standard_aug_jac.c:135-237): per column j < n the unit diagonal, then
the unit row of an active bound, then the active constraint rows in Jacobian
order; |W| empty trailing columns.

Everything here is input plumbing (numpy/scipy on the host); no numerics of the
product live in this file.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def banded_jacobian(n: int, m: int, nz_per_row: int = 20, width: int = 200, seed: int = 0) -> sp.csc_matrix:
    """Config 4 family: row i has ``nz_per_row`` entries at distinct random
    columns inside a width-``width`` window centred at floor(i*n/m), N(0,1)."""
    rng = np.random.default_rng(seed)
    width = min(width, n)
    nz_per_row = min(nz_per_row, width)
    centre = (np.arange(m, dtype=np.int64) * n) // max(m, 1)
    start = np.clip(centre - width // 2, 0, n - width)
    offs = np.argpartition(rng.random((m, width)), nz_per_row - 1, axis=1)[:, :nz_per_row]
    cols = (start[:, None] + offs).astype(np.int64)
    rows = np.repeat(np.arange(m, dtype=np.int64), nz_per_row)
    vals = rng.standard_normal(m * nz_per_row)
    J = sp.csc_matrix((vals, (rows, cols.ravel())), shape=(m, n))
    J.sort_indices()
    return J


def grid2d_jacobian(g: int, seed: int = 0) -> sp.csc_matrix:
    """PDE-constrained 2-D family: a g x g grid of states y and as many controls u,
    one constraint per cell - the 5-point Laplacian of the states plus the cell's control, ``A y + u = f`` with
    variable coefficients.  n = 2 g^2, m = g^2; ``J J^T`` is a 13-point stencil on the grid, so the separators of a nested
    dissection are ~g (2 g near the top) columns wide: wider than the 128 columns of one front."""
    rng = np.random.default_rng(seed)
    cell = np.arange(g * g, dtype=np.int64).reshape(g, g)
    rows, cols, vals = [], [], []
    for di, dj, base in ((0, 0, 4.0), (1, 0, -1.0), (-1, 0, -1.0), (0, 1, -1.0), (0, -1, -1.0)):
        src = cell[max(0, -di):g - max(0, di), max(0, -dj):g - max(0, dj)].ravel()
        dst = cell[max(0, di):g - max(0, -di), max(0, dj):g - max(0, -dj)].ravel()
        rows.append(src)
        cols.append(dst)
        vals.append(base * (1.0 + 0.1 * rng.random(src.size)))
    rows.append(cell.ravel())            # the cell's control
    cols.append(g * g + cell.ravel())
    vals.append(1.0 + 0.1 * rng.random(g * g))
    J = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(g * g, 2 * g * g))
    J.sort_indices()
    return J


def grid3d_jacobian(g: int, seed: int = 0) -> sp.csc_matrix:
    """PDE-constrained 3-D family (missing 3): a g x g x g grid of states and as many controls, one
    constraint per cell - the 7-point Laplacian of the states plus the cell's control.  n = 2 g^3, m = g^3 (g = 37:
    n = 101 306, m = 50 653, the size of BASELINE's config 4); ``J J^T`` is a 25-point stencil, the separators of a nested
    dissection are planes of ~2 g^2 cells: thousands of columns, chains of dozens of 128-column fronts."""
    rng = np.random.default_rng(seed)
    cell = np.arange(g * g * g, dtype=np.int64).reshape(g, g, g)
    rows, cols, vals = [], [], []
    for d, base in (((0, 0, 0), 6.0), ((1, 0, 0), -1.0), ((-1, 0, 0), -1.0), ((0, 1, 0), -1.0), ((0, -1, 0), -1.0),
                    ((0, 0, 1), -1.0), ((0, 0, -1), -1.0)):
        sl_src = tuple(slice(max(0, -k), g - max(0, k)) for k in d)
        sl_dst = tuple(slice(max(0, k), g - max(0, -k)) for k in d)
        src, dst = cell[sl_src].ravel(), cell[sl_dst].ravel()
        rows.append(src)
        cols.append(dst)
        vals.append(base * (1.0 + 0.1 * rng.random(src.size)))
    rows.append(cell.ravel())  # the cell's control
    cols.append(g ** 3 + cell.ravel())
    vals.append(1.0 + 0.1 * rng.random(g ** 3))
    J = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(g ** 3, 2 * g ** 3))
    J.sort_indices()
    return J


def uniform_jacobian(n: int, m: int, nz_per_row: int = 10, seed: int = 0) -> sp.csc_matrix:
    """Config 3 family: ``nz_per_row`` entries per row at uniform random distinct columns."""
    rng = np.random.default_rng(seed)
    nz_per_row = min(nz_per_row, n)
    cols = np.empty((m, nz_per_row), dtype=np.int64)
    # distinct columns per row: rejection on duplicates (cheap for nz_per_row << n)
    for i in range(m):
        c = rng.choice(n, size=nz_per_row, replace=False) if n < 4 * nz_per_row else None
        if c is None:
            c = rng.integers(0, n, size=nz_per_row)
            while len(np.unique(c)) < nz_per_row:
                c = rng.integers(0, n, size=nz_per_row)
        cols[i] = c
    rows = np.repeat(np.arange(m, dtype=np.int64), nz_per_row)
    vals = rng.standard_normal(m * nz_per_row)
    J = sp.csc_matrix((vals, (rows, cols.ravel())), shape=(m, n))
    J.sort_indices()
    return J


def working_set_all_rows(n: int, m_total: int, active_var_frac: float = 0.0, seed: int = 0):
    """Working-set index maps in the reference's convention
    (working_set.c:114-168): active variables are numbered first, active
    constraints continue after them; -1 = inactive.  All constraints active;
    a random fraction of the variable bounds active."""
    rng = np.random.default_rng(seed + 7919)
    var_index = np.full(n, -1, dtype=np.int32)
    if active_var_frac > 0:
        k = int(round(active_var_frac * n))
        act = np.sort(rng.choice(n, size=k, replace=False))
        var_index[act] = np.arange(k, dtype=np.int32)
    nav = int((var_index >= 0).sum())
    cons_index = (nav + np.arange(m_total)).astype(np.int32)
    return var_index, cons_index, nav + m_total


def kkt_lower_from_jacobian(J: sp.csc_matrix, var_index=None, cons_index=None):
    """Lower-triangular K in fill_aug_jac's CSC order, as (N, colptr, rowidx, vals).

    Vectorised restatement used to make large inputs quickly; the byte-for-byte
    oracle for this layout is oracle/kkt_oracle.c:oracle_fill_aug_jac and the
    parity tests compare the two."""
    J = sp.csc_matrix(J)
    J.sort_indices()
    m_total, n = J.shape
    if var_index is None:
        var_index = np.full(n, -1, dtype=np.int32)
    if cons_index is None:
        cons_index = np.arange(m_total, dtype=np.int32)
    var_index = np.asarray(var_index, dtype=np.int32)
    cons_index = np.asarray(cons_index, dtype=np.int32)
    size_w = int((var_index >= 0).sum() + (cons_index >= 0).sum())
    N = n + size_w
    jp, ji, jx = J.indptr.astype(np.int64), J.indices.astype(np.int64), J.data
    act = cons_index[ji] >= 0
    col_of = np.repeat(np.arange(n, dtype=np.int64), np.diff(jp))
    cnt_cons = np.bincount(col_of[act], minlength=n)
    cnt = 1 + (var_index >= 0).astype(np.int64) + cnt_cons
    colptr = np.zeros(N + 1, dtype=np.int64)
    colptr[1 : n + 1] = np.cumsum(cnt)
    colptr[n + 1 :] = colptr[n]
    nnz = int(colptr[n])
    rowidx = np.empty(nnz, dtype=np.int32)
    vals = np.empty(nnz, dtype=np.float64)
    rowidx[colptr[:n]] = np.arange(n, dtype=np.int32)
    vals[colptr[:n]] = 1.0
    av = np.nonzero(var_index >= 0)[0]
    rowidx[colptr[av] + 1] = n + var_index[av]
    vals[colptr[av] + 1] = 1.0
    # active Jacobian entries keep their in-column order
    base = colptr[:n] + 1 + (var_index >= 0)
    a_cols = col_of[act]
    first = np.zeros(n + 1, dtype=np.int64)
    first[1:] = np.cumsum(cnt_cons)
    rank = np.arange(a_cols.size, dtype=np.int64) - first[a_cols]
    dst = base[a_cols] + rank
    rowidx[dst] = n + cons_index[ji[act]]
    vals[dst] = jx[act]
    return N, colptr.astype(np.int32), rowidx, vals


def kkt_full_matrix(N, colptr, rowidx, vals) -> sp.csc_matrix:
    """Symmetric K as a scipy matrix (for residual checks in tests/bench)."""
    L = sp.csc_matrix((vals, rowidx, colptr), shape=(N, N))
    D = sp.diags(L.diagonal())
    return (L + L.T - D).tocsc()


def with_dense_columns(J: sp.csc_matrix, k: int, seed: int = 0, frac: float = 1.0, entries: int | None = None):
    """J plus k columns (existing variables, chosen at random) that get an entry in a share `frac` of the rows, or in
    exactly `entries` random rows each (a variable that appears in many constraints).  Returns (J', columns)."""
    m, n = J.shape
    rng = np.random.default_rng(seed)
    cols = np.sort(rng.choice(n, k, replace=False))
    rr, cc = [], []
    for c in cols:
        if entries is not None:
            rows = np.sort(rng.choice(m, min(entries, m), replace=False))
        elif frac < 1.0:
            rows = np.flatnonzero(rng.random(m) < frac)
        else:
            rows = np.arange(m)
        rr.append(rows)
        cc.append(np.full(rows.size, c))
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    D = sp.csc_matrix((rng.standard_normal(rr.size), (rr, cc)), shape=(m, n))
    Jd = (J + D).tocsc()
    Jd.sort_indices()
    return Jd, cols


def with_dense_rows(J: sp.csc_matrix, k: int, seed: int = 0, entries: int | None = None):
    """J plus k rows (existing constraints, chosen at random) that get an entry in every column, or in `entries`
    random columns each (a budget-type constraint sum_i x_i <= c).  Returns (J', rows)."""
    m, n = J.shape
    rng = np.random.default_rng(seed)
    rows = np.sort(rng.choice(m, k, replace=False))
    rr, cc = [], []
    for r in rows:
        cols = np.arange(n) if entries is None else np.sort(rng.choice(n, min(entries, n), replace=False))
        cc.append(cols)
        rr.append(np.full(cols.size, r))
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    D = sp.csc_matrix((rng.standard_normal(rr.size), (rr, cc)), shape=(m, n))
    Jd = (J + D).tocsc()
    Jd.sort_indices()
    return Jd, rows
