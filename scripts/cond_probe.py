"""Graded-conditioning probe (GPU): device solve vs the LAPACK-restating oracle vs an
extended-precision reference, over row scalings, nearly parallel rows and column scalings.
    python scripts/cond_probe.py [equilibrate=1]
"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from sleqp_amd import HipfactError, synth  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402
from util import graded_family, reference_solution, rel_err  # noqa: E402


def main():
    eq = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    fact = HipFact(device=0)
    fact.set_option("equilibrate", eq)
    print(f"{'family':<22}{'cond(K)':>10}{'oracle err':>12}{'device err':>12}{'dev-vs-orc':>12}{'omega':>10}{'it':>4}{'st':>3}{'kappa':>10}")
    for name, J, vi, ci in graded_family():
        m, n = J.shape
        N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
        K = synth.kkt_full_matrix(N, kc, kr, kd).toarray()
        cond = np.linalg.cond(K)
        b = np.random.default_rng(1).standard_normal(N)
        try:
            ref = oracle.OracleFact(N, kc, kr, kd)
        except ZeroDivisionError:
            print(f"{name:<22}{cond:10.2e}  oracle: singular")
            continue
        ref.solve_dense(b)
        zo = ref.raw_solution()
        truth = reference_solution(K, b, ref)
        try:
            fact.set_matrix(SleqpMat(N, N, kc, kr, kd))
            fact.solve(b)
            z = fact.solution_raw(0, N)
        except HipfactError as e:
            print(f"{name:<22}{cond:10.2e}{rel_err(zo, truth):12.2e}  device: {e}")
            continue
        print(f"{name:<22}{cond:10.2e}{rel_err(zo, truth):12.2e}{rel_err(z, truth):12.2e}{rel_err(z, zo):12.2e}"
              f"{fact.info('last_omega'):10.1e}{int(fact.info('last_iters')):4d}{int(fact.info('last_status')):3d}"
              f"{fact.info('kappa_est'):10.1e}")


if __name__ == "__main__":
    main()
