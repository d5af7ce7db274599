"""Cold path through the plain vtable entry (hipfact_set_matrix with a pattern never seen): HIPFACT_TIMING=1 prints
the phases.  Two fresh handles in one process (the second one finds the code objects loaded).

    HIPFACT_TIMING=1 gpurun -- python scripts/cold_gpu_probe.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import make_problem  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402


def main():
    J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
    K = SleqpMat(N, N, cp, ri, vx)
    for rep in range(3):
        t0 = time.perf_counter()
        fact = HipFact()
        t1 = time.perf_counter()
        fact.set_matrix(K)
        t2 = time.perf_counter()
        fact.solve(b)
        x = fact.solution_raw(0, N)
        t3 = time.perf_counter()
        fact.set_matrix(K)
        fact.solve(b)
        x = fact.solution_raw(0, N)
        t4 = time.perf_counter()
        print(f"handle {rep}: create {1e3 * (t1 - t0):.1f} ms, cold set_matrix {1e3 * (t2 - t1):.1f} ms, first solve + solution "
              f"{1e3 * (t3 - t2):.2f} ms, second unit {1e3 * (t4 - t3):.2f} ms", file=sys.stderr, flush=True)
        del fact


if __name__ == "__main__":
    main()
