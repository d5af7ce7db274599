"""Randomised parity sweep: working sets of random Jacobians (banded / uniform, with dense columns, dense rows, active
bounds, partial working sets) through hipfact against an independent sparse LU of K (scipy SuperLU) - factor + solve,
a second right-hand side, a refactorisation with other values, and the trust-region solvers (Steihaug and GLTR, device
and host loops) against each other and against the KKT conditions of the projected problem.

    gpurun -- python scripts/fuzz_parity.py [cases] [seed]

The oracle is only the checker here (scipy), nothing of it is in the product path.  Prints one line per failure and a
summary; exit code 1 if anything failed.
"""
import os
import sys
import traceback

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sleqp_amd import synth  # noqa: E402
from sleqp_amd.fact import HipFact, SpMat, StandardAugJac  # noqa: E402
from sleqp_amd.sparse import SleqpMat, SleqpVec  # noqa: E402


LAST_TAG = [""]
STEPS = [0]  # working sets of the sequence mode that were factored and checked


def scaled_residual(K, z, b):
    r = K @ z - b
    return float(np.abs(r).max() / (abs(K).sum(axis=1).max() * max(np.abs(z).max(), 1e-300) + np.abs(b).max() + 1e-300))


def working_set(n, m, rng, row_frac, bound_frac):
    vi = np.full(n, -1, dtype=np.int32)
    av = np.sort(rng.choice(n, int(round(bound_frac * n)), replace=False))
    vi[av] = np.arange(av.size)
    ci = np.full(m, -1, dtype=np.int32)
    ac = np.sort(rng.choice(m, max(1, int(round(row_frac * m))), replace=False))
    ci[ac] = av.size + np.arange(ac.size)
    return vi, ci


def one_case(fact, rng, idx, tol=None, krylov_checks=("gltr", "cg")):
    """One random case; returns (tag, list of failures) or (tag, "skipped ...").  tol: stat_tol of the trust-region
    solvers (default: FUZZ_TOL or 1e-4)."""
    if tol is None:
        tol = float(os.environ.get("FUZZ_TOL", "1e-4"))
    kind = rng.choice(["banded", "uniform"])
    sizes = [int(v) for v in os.environ.get("FUZZ_SIZES", "40,150,600,1500,4000,12000").split(",")]
    n = int(rng.choice(sizes))
    m = max(1, int(n * rng.choice([0.2, 0.5, 0.8])))
    if kind == "banded":
        J = synth.banded_jacobian(n, m, int(min(rng.integers(3, 14), n)), int(min(rng.integers(20, 200), n)), int(rng.integers(1 << 30)))
    else:
        if n > 4000 and not os.environ.get("FUZZ_SIZES"):
            n, m = 1500, 700
        J = synth.uniform_jacobian(n, m, int(min(rng.integers(2, 6), n)), int(rng.integers(1 << 30)))
    extra = rng.choice(["none", "none", "dense_cols", "dense_rows", "both", "many_cols"])
    if extra in ("dense_cols", "both"):
        J, _ = synth.with_dense_columns(J, int(rng.integers(1, 6)), int(rng.integers(1 << 30)), frac=float(rng.choice([1.0, 0.6])))
    if extra == "many_cols":
        J, _ = synth.with_dense_columns(J, int(min(n // 2, rng.integers(20, 90))), int(rng.integers(1 << 30)), entries=int(min(m, rng.integers(30, 200))))
    if extra in ("dense_rows", "both"):
        J, _ = synth.with_dense_rows(J, int(rng.integers(1, 3)), int(rng.integers(1 << 30)))
    J = sp.csc_matrix(J)
    J.sort_indices()
    bound_frac = float(rng.choice([0.0, 0.0, 0.05, 0.2]))
    row_frac = float(rng.choice([1.0, 1.0, 0.7, 0.3]))
    # (keep the working set's rows independent: no more rows + bounds than variables)
    while int(round(bound_frac * n)) + int(round(row_frac * m)) > n - 1 and row_frac > 0.05:
        row_frac *= 0.7
    vi, ci = working_set(n, m, rng, row_frac, bound_frac)
    tag = f"case {idx}: {kind} n={n} m={J.shape[0]} extra={extra} rows={row_frac:.2f} bounds={bound_frac:.2f}"
    mode = int(rng.choice([1, 1, 2]))
    tag += f" dense_mode={mode}"
    LAST_TAG[0] = tag
    if os.environ.get("FUZZ_VERBOSE"):
        print("running", tag, file=sys.stderr, flush=True)
    # the matrix the reference would assemble, for the independent check
    N, kc, kr, kd = synth.kkt_lower_from_jacobian(J, vi, ci)
    K = synth.kkt_full_matrix(N, kc, kr, kd).tocsc()
    # (SuperLU has been seen to crash instead of reporting an exactly singular matrix: those are caught before it)
    from scipy.sparse.csgraph import structural_rank

    if structural_rank(K) < N or (N <= 500 and np.linalg.matrix_rank(K.toarray()) < N):
        return tag, "skipped (singular working set)"
    try:
        lu = spla.splu(K)
    except RuntimeError:
        return tag, "skipped (singular working set)"
    cond_proxy = np.abs(lu.U.diagonal()).max() / max(np.abs(lu.U.diagonal()).min(), 1e-300)
    if not np.isfinite(cond_proxy) or cond_proxy > 1e10:
        return tag, "skipped (ill conditioned)"
    fact.set_option("dense_mode", mode)
    for kv in filter(None, os.environ.get("FUZZ_OPTS", "").split(",")):  # e.g. FUZZ_OPTS=xupd_fused=0,rhs_fused=0
        fact.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    aug = StandardAugJac(n, fact)
    aug.set_iterate(SleqpMat.from_scipy(J), vi, ci)
    errs = []
    for rep in range(2):
        b = rng.standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        want = lu.solve(b)
        res = scaled_residual(K, z, b)
        rel = float(np.linalg.norm(z - want) / max(np.linalg.norm(want), 1e-300))
        # (the forward error is the condition's doing: SuperLU's own answer is no better)
        if not (res <= 1e-11) or not (rel <= max(1e-5, 1e-13 * cond_proxy)):
            errs.append(f"solve {rep}: scaled residual {res:.2e}, rel diff vs SuperLU {rel:.2e}")
    for rep in range(int(os.environ.get("FUZZ_EXTRA_SOLVES", "0"))):
        b = rng.standard_normal(N)
        fact.solve(b)
        z = fact.solution_raw(0, N)
        if os.environ.get("FUZZ_VERBOSE"):
            print(f"   extra solve {rep}: scaled residual {scaled_residual(K, z, b):.2e}, top block active {int(fact.info('top_block_active'))} "
                  f"cols {int(fact.info('top_block_cols'))} below {int(fact.info('top_block_below'))} levels {int(fact.info('nlevels'))} "
                  f"fronts {int(fact.info('nsuper'))} late rows {int(fact.info('late_rows'))}", file=sys.stderr, flush=True)
    # the plain vtable boundary with the assembled K (row dictionary behind it)
    f2 = HipFact()
    f2.set_option("dense_mode", mode)
    f2.set_matrix(SleqpMat(N, N, kc, kr, kd))
    b = rng.standard_normal(N)
    f2.solve(b)
    z2 = f2.solution_raw(0, N)
    res2 = scaled_residual(K, z2, b)
    if not (res2 <= 1e-11):
        errs.append(f"vtable boundary: scaled residual {res2:.2e}")
    # other values, same pattern
    kd2 = np.array(kd, copy=True)
    off = np.ones(len(kd2), dtype=bool)
    off[np.asarray(kc[:n])] = False
    unit = np.zeros(len(kd2), dtype=bool)
    if (vi >= 0).any():
        # the unit entries of the bound rows stay 1 (the dictionary recognises bound rows by them)
        Kl = sp.csc_matrix((np.arange(len(kd2)), kr, kc), shape=(N, N))
        for j in np.nonzero(vi >= 0)[0]:
            col = slice(kc[j], kc[j + 1])
            rows = np.asarray(kr[col])
            unit[np.arange(kc[j], kc[j + 1])[rows == n + vi[j]]] = True
        del Kl
    chg = off & ~unit
    kd2[chg] *= 1.0 + 0.2 * rng.standard_normal(int(chg.sum()))
    K2 = synth.kkt_full_matrix(N, kc, kr, kd2).tocsc()
    f2.set_matrix(SleqpMat(N, N, kc, kr, kd2))
    f2.solve(b)
    z3 = f2.solution_raw(0, N)
    res3 = scaled_residual(K2, z3, b)
    if not (res3 <= 1e-10):
        errs.append(f"vtable boundary, other values: scaled residual {res3:.2e}")
    del f2
    # trust-region solvers on the null space of the working set
    W = int((vi >= 0).sum() + (ci >= 0).sum())
    if W < n and n >= 40:
        B = sp.random(n, n, density=min(1.0, 3.0 / n), random_state=int(rng.integers(1 << 30)))
        shift = float(rng.choice([0.5, 0.5, -0.3]))
        HL = sp.tril(B @ B.T + shift * sp.eye(n), format="csc")
        HL.sort_indices()
        H = SpMat(fact, SleqpMat.from_scipy(HL))
        Hs = (HL + HL.T - sp.diags(HL.diagonal())).tocsr()
        g = rng.standard_normal(n)
        q = lambda s_: float(g @ s_ + 0.5 * s_ @ (Hs @ s_))
        radius = float(rng.choice([1e4, 5.0, 0.5])) if shift > 0 else float(rng.choice([5.0, 0.5]))
        # rows of the working set in x space: bounds (unit rows) and constraint rows
        rows = []
        if (vi >= 0).any():
            rows.append(sp.csr_matrix((np.ones(int((vi >= 0).sum())), (np.arange(int((vi >= 0).sum())), np.nonzero(vi >= 0)[0])), shape=(int((vi >= 0).sum()), n)))
        rows.append(J.tocsr()[np.nonzero(ci >= 0)[0], :])
        A = sp.vstack(rows).tocsr()
        anorm = max(abs(A).sum(axis=1).max(), 1.0)
        if os.environ.get("FUZZ_VERBOSE"):
            # reference Lanczos coefficients in numpy (dense projection)
            Ad = A.toarray()
            Pm = np.eye(n) - Ad.T @ np.linalg.solve(Ad @ Ad.T, Ad)
            t = g.copy(); y = Pm @ t; gam = [np.sqrt(t @ y)]; tprev = None; tcur = y.copy(); dl = []
            for k in range(4):
                qk = tcur / gam[k]
                Hq = Hs @ qk
                dl.append(qk @ Hq)
                tn = Hq - dl[k] / gam[k] * tcur - (gam[k] / gam[k - 1] * tprev if k > 0 else 0.0)
                yn = Pm @ tn
                gam.append(np.sqrt(max(tn @ yn, 0.0)))
                tprev, tcur = tcur, yn
            print("   numpy Lanczos: delta", dl, "gamma", gam, file=sys.stderr, flush=True)
        out = {}
        for name, method, opt, val in (("gltr_dev", 1, "lz_device_loop", 1), ("gltr_host", 1, "lz_device_loop", 0),
                                       ("cg_dev", 0, "cg_device_loop", 1), ("cg_host", 0, "cg_device_loop", 0)):
            try:
                fact.set_option(opt, val)
            except Exception:  # noqa: BLE001  (an older library without the option)
                pass
            # (the reference's interior test of CG is absolute, |r.g| < (1e-2 stat_tol)^2: 1e-12 here, above the rounding
            # level of r.g)
            s, dual, its = fact.tr_solve(H, g, radius, method=method, stat_tol=tol, max_iter=300)
            out[name] = (s, dual, its)
            if os.environ.get("FUZZ_VERBOSE"):
                print(f"   {name}: its {its} dual {dual:.6e} |s| {np.linalg.norm(s):.8e} q {q(s):.10e} radius {radius} shift {shift}", file=sys.stderr, flush=True)
            if name.split("_")[0] not in krylov_checks:
                continue
            feas = float(np.abs(A @ s).max() / (anorm * max(np.abs(s).max(), 1.0)))
            if not (feas <= 1e-8):
                errs.append(f"{name}: step leaves the null space ({feas:.2e})")
            if not (np.linalg.norm(s) <= radius * (1 + 1e-6)):  # (the basis is orthonormal to the accuracy of the projections)
                errs.append(f"{name}: step outside the trust region")
            if not (q(s) <= 1e-12 * abs(q(s))):
                errs.append(f"{name}: no model decrease ({q(s):.3e})")
        try:
            fact.set_option("lz_device_loop", 1)
        except Exception:  # noqa: BLE001
            pass
        fact.set_option("cg_device_loop", 1)
        if "gltr" in krylov_checks:
            # the matrix-free product of the problem (callback) instead of the Hessian in HBM, and small iteration caps
            s_mf, d_mf, i_mf = fact.tr_solve(lambda v: Hs @ v, g, radius, method=1, stat_tol=tol, max_iter=300)
            sa, da, ia = out["gltr_host"]
            if abs(i_mf - ia) > 1 or np.linalg.norm(s_mf - sa) > 1e-6 * max(np.linalg.norm(sa), 1e-300):
                errs.append(f"GLTR matrix-free vs explicit Hessian: its {i_mf}/{ia}, rel diff {np.linalg.norm(s_mf - sa) / max(np.linalg.norm(sa), 1e-300):.2e}")
            for cap in (1, 2, 9):
                if n - W < 2 * cap + 2:
                    continue  # (the Krylov space would be exhausted: what follows is rounding noise on either side)
                res = {}
                for dev in (1, 0):
                    fact.set_option("lz_device_loop", dev)
                    res[dev] = fact.tr_solve(H, g, radius, method=1, stat_tol=1e-30, max_iter=cap)
                fact.set_option("lz_device_loop", 1)
                (s1, d1, i1), (s0, d0, i0) = res[1], res[0]
                if i1 != i0 or np.linalg.norm(s1 - s0) > 1e-8 * max(np.linalg.norm(s0), 1e-300) or abs(d1 - d0) > 1e-8 * max(1.0, abs(d0)):
                    errs.append(f"GLTR cap {cap}: device phase vs host loop its {i1}/{i0}, rel diff {np.linalg.norm(s1 - s0) / max(np.linalg.norm(s0), 1e-300):.2e}")
        (sa, da, ia), (sb, db, ib) = out["gltr_dev"], out["gltr_host"]
        # (one iteration more or less where the convergence test is met to rounding is not a difference)
        if abs(ia - ib) > 1 or np.linalg.norm(sa - sb) > 1e-6 * max(np.linalg.norm(sb), 1e-300) or abs(da - db) > 1e-6 * max(1.0, abs(db)):
            errs.append(f"GLTR device phase vs host loop: its {ia}/{ib}, rel diff {np.linalg.norm(sa - sb) / max(np.linalg.norm(sb), 1e-300):.2e}, dual {da:.6e}/{db:.6e}")
        (sa, da, ia), (sb, db, ib) = out["cg_dev"], out["cg_host"]
        if "cg" in krylov_checks and (abs(ia - ib) > 1 or np.linalg.norm(sa - sb) > 1e-5 * max(np.linalg.norm(sb), 1e-300)):
            errs.append(f"CG device loop vs host loop: its {ia}/{ib}, rel diff {np.linalg.norm(sa - sb) / max(np.linalg.norm(sb), 1e-300):.2e}")
        # GLTR minimises over the whole Krylov space: at least as good as Steihaug's point
        if "cg" in krylov_checks and q(out["gltr_dev"][0]) > q(out["cg_dev"][0]) + 1e-6 * abs(q(out["cg_dev"][0])):
            errs.append(f"GLTR model value {q(out['gltr_dev'][0]):.8e} worse than CG {q(out['cg_dev'][0]):.8e}")
        H.free()
    if os.environ.get("FUZZ_VERBOSE"):
        try:
            print(f"   dense fallbacks {int(fact.info('dense_fallbacks'))}, probes {int(fact.info('dense_probes'))}", file=sys.stderr, flush=True)
        except Exception:  # noqa: BLE001
            pass
    return tag, errs


def sequence_case(rng, idx):
    """A run of working sets of ONE Jacobian (rows and bounds entering and leaving) through both boundaries - the plain
    vtable (K assembled on the host, row dictionary and superset plan behind hipfact_set_matrix) and the AugJac mirror
    (device assembly) - with min-norm, least-squares and projection solves against dense references."""
    kind = rng.choice(["banded", "uniform"])
    n = int(rng.choice([int(v) for v in os.environ.get("FUZZ_SEQ_SIZES", "60,300,1200,3000").split(",")]))
    m = max(2, int(n * rng.choice([0.3, 0.6])))
    if kind == "banded":
        J = synth.banded_jacobian(n, m, int(min(rng.integers(3, 10), n)), int(min(rng.integers(20, 120), n)), int(rng.integers(1 << 30)))
    else:
        J = synth.uniform_jacobian(n, m, int(min(rng.integers(2, 5), n)), int(rng.integers(1 << 30)))
    extra = rng.choice(["none", "dense_cols", "dense_rows"])
    if extra == "dense_cols":
        J, _ = synth.with_dense_columns(J, int(rng.integers(1, 5)), int(rng.integers(1 << 30)))
    if extra == "dense_rows":
        J, _ = synth.with_dense_rows(J, 1, int(rng.integers(1 << 30)))
    J = sp.csc_matrix(J)
    J.sort_indices()
    m = J.shape[0]
    tag = f"sequence {idx}: {kind} n={n} m={m} extra={extra}"
    LAST_TAG[0] = tag
    if os.environ.get("FUZZ_VERBOSE"):
        print("running", tag, file=sys.stderr, flush=True)
    fv = HipFact()   # plain vtable
    fa = HipFact()   # AugJac mirror
    aug = StandardAugJac(n, fa)
    rows_on = rng.random(m) < 0.6
    bnd_on = rng.random(n) < 0.05
    errs = []
    from scipy.sparse.csgraph import structural_rank

    for step in range(6):
        # a few rows / bounds enter and leave
        flip = rng.random(m) < 0.04
        rows_on ^= flip
        flipb = rng.random(n) < 0.01
        bnd_on ^= flipb
        if rows_on.sum() + bnd_on.sum() > n - 2:
            rows_on[np.nonzero(rows_on)[0][: int(rows_on.sum() + bnd_on.sum() - (n - 2))]] = False
        vi = np.full(n, -1, dtype=np.int32)
        av = np.nonzero(bnd_on)[0]
        vi[av] = np.arange(av.size)
        ci = np.full(m, -1, dtype=np.int32)
        ac = np.nonzero(rows_on)[0]
        ci[ac] = av.size + np.arange(ac.size)
        Jv = sp.csc_matrix((J.data * (1.0 + 0.1 * rng.standard_normal(J.nnz)), J.indices, J.indptr), shape=J.shape)
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(Jv, vi, ci)
        K = synth.kkt_full_matrix(N, kc, kr, kd).tocsc()
        if structural_rank(K) < N or (N <= 500 and np.linalg.matrix_rank(K.toarray()) < N):
            continue
        try:
            lu = spla.splu(K)
        except RuntimeError:
            continue
        dg = np.abs(lu.U.diagonal())
        if not np.isfinite(dg.max() / max(dg.min(), 1e-300)) or dg.max() / max(dg.min(), 1e-300) > 1e9:
            continue
        W = N - n
        # (the dense formulas below go through A A^T themselves: their own error grows with the condition)
        thr = max(1e-7, 1e-14 * dg.max() / max(dg.min(), 1e-300))
        STEPS[0] += 1
        b = rng.standard_normal(N)
        for name, f in (("vtable", fv), ("augjac", fa)):
            if name == "vtable":
                f.set_matrix(SleqpMat(N, N, kc, kr, kd))
            else:
                aug.set_iterate(SleqpMat.from_scipy(Jv), vi, ci)
            f.solve(b)
            z = f.solution_raw(0, N)
            res = scaled_residual(K, z, b)
            if not (res <= 1e-11):
                errs.append(f"step {step} {name}: scaled residual {res:.2e}")
        # the three AugJac solves against the dense formulas
        rows = []
        if av.size:
            rows.append(sp.csr_matrix((np.ones(av.size), (np.arange(av.size), av)), shape=(av.size, n)))
        rows.append(Jv.tocsr()[ac, :])
        A = sp.vstack(rows).toarray() if n <= 1200 else None
        if A is not None and W > 0:
            AAt = A @ A.T
            rhs_w = rng.standard_normal(W)
            want = A.T @ np.linalg.solve(AAt, rhs_w)
            got = aug.solve_min_norm(SleqpVec.from_raw(rhs_w)).to_raw()
            if np.linalg.norm(got - want) > thr * max(np.linalg.norm(want), 1e-300):
                errs.append(f"step {step}: min-norm solve rel diff {np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300):.2e}")
            rhs_n = rng.standard_normal(n)
            want = np.linalg.solve(AAt, A @ rhs_n)
            got = aug.solve_lsq(SleqpVec.from_raw(rhs_n)).to_raw()
            if np.linalg.norm(got - want) > thr * max(np.linalg.norm(want), 1e-300):
                errs.append(f"step {step}: least-squares solve rel diff {np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300):.2e}")
            want = rhs_n - A.T @ np.linalg.solve(AAt, A @ rhs_n)
            got = aug.project_nullspace(SleqpVec.from_raw(rhs_n)).to_raw()
            if np.linalg.norm(got - want) > thr * max(np.linalg.norm(want), 1e-300):
                errs.append(f"step {step}: projection rel diff {np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300):.2e}")
        # the trust-region loops on the superset plan of this run (H = c I inside a large region: the step is -P g / c)
        if W < n:
            cH = 2.5
            Hc = SpMat(fa, SleqpMat.from_scipy(sp.csc_matrix(sp.identity(n, format="csc") * cH)))
            gk = rng.standard_normal(n)
            pg = aug.project_nullspace(SleqpVec.from_raw(gk)).to_raw()
            for method, opt in ((1, "lz_device_loop"), (0, "cg_device_loop")):
                for dev in (1, 0):
                    fa.set_option(opt, dev)
                    st, _, its = fa.tr_solve(Hc, gk, 1e8, method=method, stat_tol=1e-5, max_iter=50)
                    if np.linalg.norm(st + pg / cH) > 1e-6 * max(np.linalg.norm(pg / cH), 1e-300) or its > 4:
                        errs.append(f"step {step}: {'GLTR' if method else 'CG'} (device loop {dev}) rel diff "
                                    f"{np.linalg.norm(st + pg / cH) / max(np.linalg.norm(pg / cH), 1e-300):.2e} in {its} iterations")
                fa.set_option(opt, 1)
            Hc.free()
    if fv.info("dataflow_fallbacks") or fa.info("dataflow_fallbacks"):
        errs.append("a dataflow launch timed out")
    return tag, errs


def spmv_case(fact, rng, idx):
    """The sparse products around the EQP step (J x, J^T y, the symmetric Hessian product from its lower triangle) on
    random shapes - empty rows and columns, single entries, rows of very different lengths - against scipy, and after
    update_values."""
    r = int(rng.choice([1, 2, 7, 64, 500, 3000]))
    c = int(rng.choice([1, 3, 50, 700, 4000]))
    dens = float(rng.choice([0.0, 0.002, 0.02, 0.3]))
    M = sp.random(r, c, density=dens, random_state=int(rng.integers(1 << 30)), format="csc")
    if rng.random() < 0.3 and r > 2:
        M = sp.csc_matrix(M + sp.csc_matrix((np.ones(c), (np.full(c, int(rng.integers(r))), np.arange(c))), shape=(r, c)))  # a dense row
    M.sort_indices()
    tag = f"spmv {idx}: {r} x {c}, nnz {M.nnz}"
    LAST_TAG[0] = tag
    errs = []
    S = SpMat(fact, SleqpMat.from_scipy(M))
    for rep in range(2):
        x, y = rng.standard_normal(c), rng.standard_normal(r)
        scale = max(abs(M).sum(axis=1).max() if M.nnz else 0.0, 1.0)
        got = S.mult_vec(x)
        if np.abs(got - M @ x).max() > 1e-13 * scale * max(np.abs(x).max(), 1.0):
            errs.append(f"M x: max diff {np.abs(got - M @ x).max():.2e}")
        got = S.mult_vec_trans(y).to_raw()
        scale_t = max(abs(M).sum(axis=0).max() if M.nnz else 0.0, 1.0)
        if np.abs(got - M.T @ y).max() > 1e-13 * scale_t * max(np.abs(y).max(), 1.0):
            errs.append(f"M^T y: max diff {np.abs(got - M.T @ y).max():.2e}")
        M = sp.csc_matrix((M.data * (1.0 + rng.standard_normal(M.nnz)), M.indices, M.indptr), shape=M.shape)
        S.update_values(M.data)
    S.free()
    k = min(r, c)
    B = sp.random(k, k, density=min(1.0, dens * 3 + 1.0 / k), random_state=int(rng.integers(1 << 30)))
    HL = sp.tril(B + B.T, format="csc")
    HL.sort_indices()
    Hs = (HL + HL.T - sp.diags(HL.diagonal())).tocsr()
    S = SpMat(fact, SleqpMat.from_scipy(HL))
    x = rng.standard_normal(k)
    got = S.mult_vec_sym(x)
    scale = max(abs(Hs).sum(axis=1).max() if Hs.nnz else 0.0, 1.0)
    if np.abs(got - Hs @ x).max() > 1e-13 * scale * max(np.abs(x).max(), 1.0):
        errs.append(f"symmetric product: max diff {np.abs(got - Hs @ x).max():.2e}")
    S.free()
    return tag, errs


def generic_case(fact, rng, idx):
    """Symmetric positive definite matrices without the saddle structure (the reduced / PSD backend of
    fact_cholmod.c's role): lower CSC through hipfact_set_matrix, dense and sparse right-hand sides, solution ranges."""
    N = int(rng.choice([1, 2, 9, 130, 900, 5000]))
    kind = rng.choice(["random", "banded"])
    if kind == "random":
        B = sp.random(N, N, density=min(1.0, float(rng.choice([1.0, 3.0, 8.0])) / N), random_state=int(rng.integers(1 << 30)))
        S = (B @ B.T + float(rng.choice([0.1, 1.0, 10.0])) * sp.eye(N)).tocsc()
    else:
        w = int(min(N - 1, rng.integers(1, 12))) if N > 1 else 0
        diags = [np.full(N, 4.0 * (w + 1))] + [rng.standard_normal(N - k) for k in range(1, w + 1)]
        L = sp.diags(diags, [-k for k in range(0, w + 1)], format="csc")
        S = (L + L.T - sp.diags(L.diagonal())).tocsc()
    SL = sp.tril(S, format="csc")
    SL.sort_indices()
    tag = f"generic {idx}: N={N} {kind} nnz {SL.nnz}"
    LAST_TAG[0] = tag
    errs = []
    fact.set_matrix(SleqpMat(N, N, SL.indptr.astype(np.int32), SL.indices.astype(np.int32), SL.data))
    lu = spla.splu(S.tocsc())
    b = rng.standard_normal(N)
    fact.solve(b)
    z = fact.solution_raw(0, N)
    if scaled_residual(S, z, b) > 1e-12 or np.linalg.norm(z - lu.solve(b)) > 1e-8 * max(np.linalg.norm(z), 1e-300):
        errs.append(f"dense rhs: scaled residual {scaled_residual(S, z, b):.2e}")
    # sparse right-hand side (a few entries, possibly none), ranges of the solution
    k = int(rng.integers(0, min(N, 6) + 1))
    ind = np.sort(rng.choice(N, k, replace=False)).astype(np.int32)
    val = rng.standard_normal(k)
    bs = np.zeros(N)
    bs[ind] = val
    fact.solve(SleqpVec(N, ind, val))
    lo = int(rng.integers(0, N + 1))
    hi = int(rng.integers(lo, N + 1))
    part = fact.solution_raw(lo, hi)
    full = fact.solution_raw(0, N)
    if not np.array_equal(part, full[lo:hi]):
        errs.append("solution range differs from the slice of the whole solution")
    if scaled_residual(S, full, bs) > 1e-12:
        errs.append(f"sparse rhs: scaled residual {scaled_residual(S, full, bs):.2e}")
    return tag, errs


def alternate_case(rng, idx):
    """ONE handle, several patterns visited in random order with new values every time (plan cache: parked states with
    their graphs, top blocks and refinement state swapped in and out, evictions beyond plan_cache), several solves each."""
    f = HipFact()
    npat = int(rng.integers(3, 8))
    f.set_option("plan_cache", int(rng.choice([1, 2, 4])))
    pats = []
    from scipy.sparse.csgraph import structural_rank

    tries = 0
    while len(pats) < npat and tries < 40:
        tries += 1
        n = int(rng.choice([50, 400, 2000, 6000]))
        m = max(2, int(n * rng.choice([0.3, 0.6])))
        J = synth.banded_jacobian(n, m, int(min(rng.integers(3, 10), n)), int(min(rng.integers(20, 120), n)), int(rng.integers(1 << 30))) \
            if rng.random() < 0.6 else synth.uniform_jacobian(n, m, int(min(rng.integers(2, 5), n)), int(rng.integers(1 << 30)))
        vi, ci, _ = synth.working_set_all_rows(n, m, float(rng.choice([0.0, 0.05])), int(rng.integers(1 << 30)))
        N, kc, kr, kd = synth.kkt_lower_from_jacobian(sp.csc_matrix(J), vi, ci)
        K0 = synth.kkt_full_matrix(N, kc, kr, kd).tocsc()
        if structural_rank(K0) < N or (N <= 500 and np.linalg.matrix_rank(K0.toarray()) < N):
            continue
        try:
            dg = np.abs(spla.splu(K0).U.diagonal())
        except RuntimeError:
            continue
        if not (dg.max() / max(dg.min(), 1e-300) <= 1e8):
            continue
        pats.append((n, N, kc, kr, np.array(kd, copy=True)))
    npat = len(pats)
    tag = f"alternate {idx}: {npat} patterns"
    if npat == 0:
        return tag, "skipped"
    LAST_TAG[0] = tag
    errs = []
    for visit in range(int(rng.integers(8, 20))):
        n, N, kc, kr, kd = pats[int(rng.integers(npat))]
        vals = np.array(kd, copy=True)
        off = np.ones(len(vals), dtype=bool)
        off[np.asarray(kc[:n])] = False
        unit = (vals == 1.0)
        chg = off & ~unit
        vals[chg] *= 1.0 + 0.1 * rng.standard_normal(int(chg.sum()))
        K = synth.kkt_full_matrix(N, kc, kr, vals).tocsc()
        f.set_matrix(SleqpMat(N, N, kc, kr, vals))
        for rep in range(int(rng.integers(1, 5))):
            b = rng.standard_normal(N)
            f.solve(b)
            z = f.solution_raw(0, N)
            res = scaled_residual(K, z, b)
            if not (res <= 1e-10):
                errs.append(f"visit {visit} (N={N}) solve {rep}: scaled residual {res:.2e}")
    if f.info("dataflow_fallbacks"):
        errs.append("a dataflow launch timed out")
    return tag, errs


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1  # one case again (every case has a generator of its own)
    fact = HipFact()
    failed = skipped = 0
    for idx in range(cases):
        if only >= 0 and idx != only:
            continue
        rng = np.random.default_rng([seed, idx])
        try:
            if os.environ.get("FUZZ_MODE") == "sequence":
                tag, res = sequence_case(rng, idx)
            elif os.environ.get("FUZZ_MODE") == "spmv":
                tag, res = spmv_case(fact, rng, idx)
            elif os.environ.get("FUZZ_MODE") == "generic":
                tag, res = generic_case(fact, rng, idx)
            elif os.environ.get("FUZZ_MODE") == "alternate":
                tag, res = alternate_case(rng, idx)
            else:
                tag, res = one_case(fact, rng, idx)
        except Exception as e:  # noqa: BLE001
            tag, res = f"case {idx} ({LAST_TAG[0]})", [f"exception {type(e).__name__}: {e}", traceback.format_exc(limit=3)]
            fact = HipFact()
        if isinstance(res, str):
            skipped += 1
            continue
        if res:
            failed += 1
            print(tag, flush=True)
            for r in res:
                print("   ", r, flush=True)
    print(f"{cases} cases (seed {seed}): {failed} failed, {skipped} skipped" + (f", {STEPS[0]} working sets checked" if STEPS[0] else ""), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
