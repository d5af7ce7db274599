"""Shared helpers for the tests (golden loader, tolerances)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kkt_cases.npz")

# Parity tolerance for floating point (BASELINE.md "Parity"): solution agreement with the
# LAPACK-restating oracle <= 1e-9 relative; scaled residual <= 1e-12 after refinement.
REL_TOL = 1e-9
RESID_TOL = 1e-12
ZERO_EPS = 1e-20


class Case:
    def __init__(self, z, name):
        p = name + "/"
        self.name = name
        for k in z.files:
            if k.startswith(p):
                v = z[k]
                setattr(self, k[len(p):], v.item() if v.ndim == 0 else v)


def golden_cases():
    z = np.load(GOLDEN)
    return [Case(z, str(n)) for n in z["names"]]


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def scaled_residual(K, x, b):
    """||Kx - b||_inf / (||K||_inf ||x||_inf + ||b||_inf)"""
    r = np.abs(K @ x - b).max() if x.size else 0.0
    kn = abs(K).sum(axis=1).max() if x.size else 0.0
    den = kn * (np.abs(x).max() if x.size else 0.0) + (np.abs(b).max() if b.size else 0.0)
    return float(r / den) if den > 0 else float(r)


def reference_solution(K, b, ref, steps=6):
    """Extended-precision reference for K z = b: the oracle's LU solves, refined with residuals
    accumulated in long double (80-bit on x86).  Converges to the correctly rounded solution for
    cond(K) up to ~1e14 — far tighter than either fp64 solver it is used to judge."""
    Kl = np.asarray(K, dtype=np.longdouble)
    bl = np.asarray(b, dtype=np.longdouble)
    ref.solve_dense(np.asarray(b, dtype=np.float64))
    x = ref.raw_solution().astype(np.longdouble)
    for _ in range(steps):
        r = bl - Kl @ x
        ref.solve_dense(np.asarray(r, dtype=np.float64))
        x = x + ref.raw_solution().astype(np.longdouble)
    return np.asarray(x, dtype=np.float64)


def graded_family():
    """Working sets of graded difficulty for the a8 parity row (fact_ma57.c:444-507 handles them by
    threshold pivoting + scaling): row scalings 1e0..1e8, nearly parallel rows, column scalings,
    and mixtures, with and without active bounds.  Yields (name, J csc, var_index, cons_index)."""
    import scipy.sparse as sp

    from sleqp_amd import synth

    rng = np.random.default_rng(77)
    bases = [("band", synth.banded_jacobian(300, 150, 12, 80, 3)), ("unif", synth.uniform_jacobian(240, 100, 6, 4))]

    def ws(J, frac, seed):
        m, n = J.shape
        vi, ci, _ = synth.working_set_all_rows(n, m, frac, seed)
        return vi, ci

    def near_parallel(J, eps, k=10):
        Jr = sp.csr_matrix(J)
        pick = rng.choice(Jr.shape[0], k, replace=False)
        extra = Jr[pick].copy()
        extra.data = extra.data * (1.0 + eps * rng.standard_normal(extra.data.size))
        out = sp.vstack([Jr, extra]).tocsc()
        out.sort_indices()
        return out

    for bname, J in bases:
        m, n = J.shape
        for p in (0, 2, 4, 6, 8):
            d = np.logspace(0, p, m)
            rng.shuffle(d)
            Jp = sp.csc_matrix(sp.diags(d) @ J)
            Jp.sort_indices()
            yield (f"{bname}_rowscale_1e{p}", Jp) + ws(Jp, 0.05 if p % 4 == 0 else 0.0, p)
        for eps in (1e-2, 1e-3, 1e-4, 1e-5):
            Jp = near_parallel(J, eps)
            yield (f"{bname}_parallel_{eps:.0e}", Jp) + ws(Jp, 0.0, 1)
        for p in (2, 4):
            d = np.logspace(0, p, n)
            rng.shuffle(d)
            Jp = sp.csc_matrix(J @ sp.diags(d))
            Jp.sort_indices()
            yield (f"{bname}_colscale_1e{p}", Jp) + ws(Jp, 0.0, 2)
        d = np.logspace(0, 6, m + 10)
        rng.shuffle(d)
        Jp = near_parallel(J, 1e-3)
        Jp = sp.csc_matrix(sp.diags(d) @ Jp)
        Jp.sort_indices()
        yield (f"{bname}_mixed", Jp) + ws(Jp, 0.1, 5)
