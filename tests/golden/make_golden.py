"""Generates tests/golden/kkt_cases.npz from the CPU oracle (oracle/kkt_oracle.c).

Run from the repo root:  python tests/golden/make_golden.py
The fixtures are data only: inputs (Jacobian CSC, working-set maps, right-hand
sides) and the oracle's expected outputs (K arrays, the three AugJac solves,
SpMV results).  The problem data of the reference's own test fixtures
(HS71: src/test/constrained_fixture.c:91-120,207-242; Newton fixture:
constrained_newton_test.c:48-202; quadfunc bounds: quadfunc_fixture.c:113-131)
is restated numerically.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from sleqp_amd import synth  # noqa: E402

ZERO_EPS = 1e-20  # SLEQP_SETTINGS_REAL_ZERO_EPS default (settings.c)


def hs71_jac(x):
    jx = np.array([x[1] * x[2] * x[3], 2 * x[0], x[0] * x[2] * x[3], 2 * x[1], x[0] * x[1] * x[3], 2 * x[2],
                   x[0] * x[1] * x[2], 2 * x[3]])
    return np.array([0, 2, 4, 6, 8]), np.array([0, 1, 0, 1, 0, 1, 0, 1]), jx


def sparse_rhs(rng, dim, frac):
    k = max(1, int(round(frac * dim))) if dim > 0 else 0
    idx = np.sort(rng.choice(dim, size=k, replace=False)).astype(np.int32) if dim > 0 else np.zeros(0, np.int32)
    return idx, rng.standard_normal(k)


def make_cases():
    cases = {}
    x0 = np.array([1.0, 5.0, 5.0, 1.0])  # constrained_fixture.c:237-242
    jp, ji, jx = hs71_jac(x0)
    # HS71 at the initial point; working sets: (a) both constraints, (b) x0, x3 at lower bound + c1,
    # (c) three bounds + c0 (|W| = n), (d) only c1
    cases["hs71_a"] = (4, 2, jp, ji, jx, [-1, -1, -1, -1], [0, 1])
    cases["hs71_b"] = (4, 2, jp, ji, jx, [0, -1, -1, 1], [-1, 2])
    cases["hs71_c"] = (4, 2, jp, ji, jx, [0, 1, -1, 2], [3, -1])
    cases["hs71_d"] = (4, 2, jp, ji, jx, [-1, -1, -1, -1], [-1, 0])
    cases["newton"] = (2, 1, np.array([0, 0, 1]), np.array([0]), np.array([1.0]), [-1, -1], [0])
    cases["quadfunc_bounds"] = (2, 0, np.array([0, 0, 0]), np.zeros(0, int), np.zeros(0), [0, 1], [])
    cases["empty_ws"] = (3, 2, np.array([0, 1, 2, 3]), np.array([0, 1, 0]), np.array([1.0, 2.0, 3.0]), [-1, -1, -1],
                         [-1, -1])
    for name, n, m, kind, frac, seed in [("band60", 60, 30, "b", 0.1, 11), ("unif200", 200, 100, "u", 0.0, 12),
                                         ("band400", 400, 200, "b", 0.05, 13), ("ragged150", 150, 90, "u", 0.1, 14)]:
        J = synth.banded_jacobian(n, m, 8, 60, seed) if kind == "b" else synth.uniform_jacobian(n, m, 4, seed)
        vi, ci, _ = synth.working_set_all_rows(n, m, frac, seed)
        if name.startswith("ragged"):  # deactivate a third of the constraints, renumber
            rng = np.random.default_rng(seed)
            off = rng.choice(m, m // 3, replace=False)
            ci = ci.copy()
            ci[off] = -1
            nav = int((vi >= 0).sum())
            act = np.nonzero(ci >= 0)[0]
            ci[act] = nav + np.arange(act.size)
        cases[name] = (n, m, J.indptr, J.indices, J.data, vi, ci)
    return cases


def main():
    out = {}
    names = []
    for name, (n, m, jp, ji, jx, vi, ci) in make_cases().items():
        names.append(name)
        rng = np.random.default_rng(abs(hash(name)) % (2**32) if False else sum(map(ord, name)))
        jp, ji, jx = np.asarray(jp, np.int32), np.asarray(ji, np.int32), np.asarray(jx, float)
        vi, ci = np.asarray(vi, np.int32), np.asarray(ci, np.int32)
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, jp, ji, jx, vi, ci, lower_only=True)
        _, fc, fr, fd = oracle.fill_aug_jac(n, m, jp, ji, jx, vi, ci, lower_only=False)
        W = N - n
        f = oracle.OracleFact(N, kc, kr, kd)
        p = name + "/"
        out.update({p + "n": n, p + "m": m, p + "jp": jp, p + "ji": ji, p + "jx": jx, p + "var_index": vi,
                    p + "cons_index": ci, p + "N": N, p + "K_cols": kc, p + "K_rows": kr, p + "K_data": kd,
                    p + "Kfull_cols": fc, p + "Kfull_rows": fr, p + "Kfull_data": fd})
        # dense solve
        b = rng.standard_normal(N)
        f.solve_dense(b)
        out[p + "rhs_dense"] = b
        out[p + "sol_dense"] = f.raw_solution()
        # the three AugJac flavours with sparse right-hand sides
        gi, gd = sparse_rhs(rng, n, 0.6)
        pi_, pd = f.project_nullspace(n, gi, gd, ZERO_EPS)
        li, ld = f.solve_lsq(n, gi, gd, ZERO_EPS)
        out.update({p + "g_idx": gi, p + "g_dat": gd, p + "proj_idx": pi_, p + "proj_dat": pd, p + "lsq_idx": li,
                    p + "lsq_dat": ld})
        bi, bd = sparse_rhs(rng, W, 0.5) if W > 0 else (np.zeros(0, np.int32), np.zeros(0))
        mi, md = f.solve_min_norm(n, bi, bd, ZERO_EPS)
        out.update({p + "b_idx": bi, p + "b_dat": bd, p + "mn_idx": mi, p + "mn_dat": md})
        # SpMV with the Jacobian (sleqp_mat_mult_vec / _trans)
        xi, xd = sparse_rhs(rng, n, 0.5)
        out[p + "x_idx"], out[p + "x_dat"] = xi, xd
        out[p + "Jx"] = oracle.mat_mult_vec(m, n, jp, ji, jx, xi, xd)
        yi, yd = sparse_rhs(rng, m, 0.5) if m > 0 else (np.zeros(0, np.int32), np.zeros(0))
        ti, td = oracle.mat_mult_vec_trans(m, n, jp, ji, jx, yi, yd, 1e-10)
        out.update({p + "y_idx": yi, p + "y_dat": yd, p + "JTy_idx": ti, p + "JTy_dat": td})
    out["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kkt_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "cases")


if __name__ == "__main__":
    main()
