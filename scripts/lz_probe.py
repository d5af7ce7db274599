"""GLTR with the device-controlled interior phase against the host-driven loop: same steps, same multipliers, same
iteration counts on interior / boundary / indefinite / capped cases, and the time per iteration on the bench workload.

    gpurun -- python scripts/lz_probe.py
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import make_problem  # noqa: E402
from sleqp_amd.fact import HipFact, SpMat  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "banded_n1e5_m5e4"
    J, N, cp, ri, vx, b = make_problem(workload, 0)
    m, n = J.shape
    fact = HipFact()
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    for _ in range(4):
        fact.solve(b)
    fact.check()
    rng = np.random.default_rng(7)
    diag = {"spd": 2.0, "indef": 0.2}
    g = rng.standard_normal(n)
    for kind, dv in diag.items():
        Hl = sp.diags([np.full(n, dv)] + [rng.standard_normal(n - k) * 0.1 for k in range(1, 6)], [0, -1, -2, -3, -4, -5], format="csc")
        Hl.sort_indices()
        Hd = SpMat(fact, SleqpMat(n, n, Hl.indptr, Hl.indices, Hl.data))
        Hs = (Hl + Hl.T - sp.diags(Hl.diagonal())).tocsr()
        q = lambda s_: float(g @ s_ + 0.5 * s_ @ (Hs @ s_))
        fact.set_option("lz_device_loop", 0)
        ref, _, _ = fact.tr_solve(Hd, g, 1e6 if kind == "spd" else 50.0, method=1, stat_tol=1e-8, max_iter=400)
        nr = np.linalg.norm(ref)
        for radius, tol, cap in ((1e6, 1e-3, 300), (1e6, 1e-8, 400), (0.3 * nr, 1e-6, 300), (0.999 * nr, 1e-8, 400),
                                 (1e6, 1e-30, 7), (1e6, 1e-30, 8), (1e6, 1e-30, 9), (1e6, 1e-30, 1), (1e6, 1e-30, 2)):
            if kind == "indef" and radius > 1e3:
                radius = 50.0
            out = {}
            for dev in (0, 1):
                fact.set_option("lz_device_loop", dev)
                r0, i0 = fact.info("lz_device_runs"), fact.info("lz_device_iterations")
                t0 = time.perf_counter()
                s, dual, its = fact.tr_solve(Hd, g, radius, method=1, stat_tol=tol, max_iter=cap)
                dt = time.perf_counter() - t0
                out[dev] = (s, dual, int(its), dt, int(fact.info("lz_device_runs") - r0), int(fact.info("lz_device_iterations") - i0))
            s0, d0, it0, t0_, _, _ = out[0]
            s1, d1, it1, t1_, runs, dits = out[1]
            print(f"{kind:5s} radius {radius:9.3g} tol {tol:7.1e} cap {cap:3d}: its {it0:3d}/{it1:3d} (device {dits:3d}, runs {runs}, "
                  f"fallbacks {int(fact.info('lz_device_fallbacks'))}) dual {d0:.6e}/{d1:.6e} |s| {np.linalg.norm(s0):.6e} "
                  f"rel diff {rel(s1, s0):.2e} q {q(s0):.10e}/{q(s1):.10e}  ms/it {t0_ * 1e3 / max(it0, 1):.4f} -> {t1_ * 1e3 / max(it1, 1):.4f}",
                  flush=True)
        Hd.free()
    # the bench's measurement
    Hl = sp.diags([np.full(n, 2.0)] + [rng.standard_normal(n - k) * 0.1 for k in range(1, 6)], [0, -1, -2, -3, -4, -5], format="csc")
    Hl.sort_indices()
    Hd = SpMat(fact, SleqpMat(n, n, Hl.indptr, Hl.indices, Hl.data))
    for dev, fold in ((0, 0), (1, 0), (0, 0), (1, 0)):
        fact.set_option("lz_device_loop", dev)
        fact.tr_solve(Hd, g, 1e6, method=1, stat_tol=1e-30, max_iter=3)
        for cap in (20, 100):
            t0 = time.perf_counter()
            _, _, its = fact.tr_solve(Hd, g, 1e6, method=1, stat_tol=1e-30, max_iter=cap)
            dt = time.perf_counter() - t0
            print(f"bench: device {dev} cap {cap}: {its} iterations, {dt * 1e3 / its:.4f} ms per iteration", flush=True)
    fact.steihaug(Hd, g, 1e6, stat_tol=1e-30, max_iter=3)
    t0 = time.perf_counter()
    _, _, its = fact.steihaug(Hd, g, 1e6, stat_tol=1e-30, max_iter=20)
    print(f"bench: CG {its} iterations, {(time.perf_counter() - t0) * 1e3 / its:.4f} ms per iteration")
    Hd.free()


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "matrix_free"):
    main()


def matrix_free():
    """The matrix-free iteration (Hessian product on the host through the callback): the callback's own time beside
    the time per iteration."""
    J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
    m, n = J.shape
    fact = HipFact()
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
    for _ in range(4):
        fact.solve(b)
    fact.check()
    rng = np.random.default_rng(7)
    g = rng.standard_normal(n)
    Hl = sp.diags([np.full(n, 2.0)] + [rng.standard_normal(n - k) * 0.1 for k in range(1, 6)], [0, -1, -2, -3, -4, -5], format="csc")
    Hs = (Hl + Hl.T - sp.diags(Hl.diagonal())).tocsr()
    d = rng.standard_normal(n)
    for _ in range(5):
        Hs @ d
    t0 = time.perf_counter()
    for _ in range(50):
        Hs @ d
    t_cb = (time.perf_counter() - t0) / 50
    spent = [0.0, 0]

    def prod(v):
        t = time.perf_counter()
        out = Hs @ v
        spent[0] += time.perf_counter() - t
        spent[1] += 1
        return out

    for dev in (1, 0, 1, 0, 1, 0, 1, 0):
        fact.set_option("lz_device_loop", dev)
        fact.tr_solve(prod, g, 1e6, method=1, stat_tol=1e-30, max_iter=20)
        spent[0], spent[1] = 0.0, 0
        t0 = time.perf_counter()
        s, dual, its = fact.tr_solve(prod, g, 1e6, method=1, stat_tol=1e-30, max_iter=20)
        dt = time.perf_counter() - t0
        print(f"matrix-free (device phase option {dev}): {its} iterations, {dt * 1e3 / its:.4f} ms per iteration; the callback itself "
              f"{spent[0] * 1e3 / max(spent[1], 1):.4f} ms x {spent[1]} calls (scipy product alone {t_cb * 1e3:.4f} ms); |s| {np.linalg.norm(s):.8e}; "
              f"dataflow fallbacks {int(fact.info('dataflow_fallbacks'))}, refined solves {int(fact.info('num_refined'))}, solves {int(fact.info('num_solve'))}",
              flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "matrix_free":
    matrix_free()
