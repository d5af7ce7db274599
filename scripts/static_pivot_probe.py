"""Probe (GPU): the draws of the random sweep that are numerically rank deficient, under static pivoting with several
shifts: backward error, distance to the extended-precision reference, the oracle's own distance, rank information."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle  # noqa: E402
from sleqp_amd import HipfactError, synth  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402
from test_gpu_parity import _problem  # noqa: E402
from util import reference_solution, rel_err, scaled_residual  # noqa: E402

rng = np.random.default_rng(2024)
fact = HipFact(device=0)
for trial in range(24):
    n = int(rng.integers(20, 600))
    m = int(rng.integers(1, max(2, n // 2)))
    kind = "b" if trial % 2 else "u"
    frac = float(rng.choice([0.0, 0.05, 0.3]))
    J, vi, ci, W = _problem(n, m, kind, frac, 100 + trial)
    if W > n:
        ci = ci.copy()
        ci[n - int((vi >= 0).sum()):] = -1
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = rng.standard_normal(N)
    if trial not in (4, 12, 20):
        continue
    K = synth.kkt_full_matrix(N, kc, kr, kd)
    sv = np.linalg.svd(K.toarray(), compute_uv=False)
    print(f"trial {trial}: N={N} sigma_min(K)={sv[-1]:.2e} sigma_2={sv[-2]:.2e} sigma_max={sv[0]:.2e}")
    try:
        ref = oracle.OracleFact(N, kc, kr, kd)
        ref.solve_dense(b)
        zo = ref.raw_solution()
        truth = reference_solution(K.toarray(), b, ref)
        print(f"   oracle err {rel_err(zo, truth):.2e}")
    except ZeroDivisionError:
        truth = None
        print("   oracle: singular")
    for delta in (0.0, 1e-8, 1e-10, 1e-12):
        fact.set_option("static_pivot", 1 if delta > 0 else 0)
        if delta > 0:
            fact.set_option("static_pivot_delta", delta)
        try:
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            fact.solve(b)
            z = fact.solution_raw(0, N)
            print(f"   delta {delta:.0e}: resid {scaled_residual(K, z, b):.2e} err {rel_err(z, truth) if truth is not None else float('nan'):.2e} "
                  f"perturbed {fact.info('num_perturbed')} omega {fact.info('last_omega'):.2e} iters {fact.info('last_iters')} |z| {np.abs(z).max():.2e}")
        except HipfactError as e:
            print(f"   delta {delta:.0e}: error {e.code} {str(e)[:120]}")
