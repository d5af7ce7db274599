"""An environment switch of the library (read when a handle is created / a pattern analysed) against the time of the
factorisation and of the solve: a fresh handle per value, device-resident entry points.

    gpurun -- python scripts/wmax_probe.py [workload] [ENV_NAME value value ...]

Default: the supernode width cap HIPFACT_WMAX (the dataflow launch takes the LDS of its widest pivot block for every
workgroup: 133 KB at w = 128, one workgroup per CU).
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import make_problem  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "banded_n1e5_m5e4"
    env_name = sys.argv[2] if len(sys.argv) > 2 else "HIPFACT_WMAX"
    values = sys.argv[3:] if len(sys.argv) > 3 else ["128", "112", "96", "80", "64", "128"]
    J, N, cp, ri, vx, b = make_problem(workload, 0)
    K = SleqpMat(N, N, cp, ri, vx)
    dev = torch.device("cuda:0")
    d_vals = torch.from_numpy(np.asarray(vx)).to(dev)
    d_b = torch.from_numpy(np.asarray(b)).to(dev)
    d_x = torch.empty_like(d_b)
    for wmax in values:
        os.environ[env_name] = str(wmax)
        fact = HipFact()
        fact.set_matrix(K)
        for _ in range(12):
            fact.solve(b)
        fact.check()
        for _ in range(5):
            fact.set_matrix(K)
            fact.solve(b)
        fact.synchronize()
        reps = 40
        t0 = time.perf_counter()
        for _ in range(reps):
            fact.refactor_device(d_vals.data_ptr())
        fact.synchronize()
        tf = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            fact.refactor_device(d_vals.data_ptr())
            fact.solve_device(d_b.data_ptr(), d_x.data_ptr())
        fact.synchronize()
        tu = (time.perf_counter() - t0) / reps
        for _ in range(5):
            fact.solve_device(d_b.data_ptr(), d_x.data_ptr())
        fact.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fact.solve_device(d_b.data_ptr(), d_x.data_ptr())
        fact.synchronize()
        ts = (time.perf_counter() - t0) / 200
        print(f"{env_name} {wmax:>4s}: fronts {int(fact.info('nsuper'))} levels {int(fact.info('nlevels'))} nnzL {fact.info('nnzL'):.3e} "
              f"factor {tf * 1e3:.3f} ms, factor+solve {tu * 1e3:.3f} ms, solve {ts * 1e3:.4f} ms", flush=True)
        del fact


if __name__ == "__main__":
    main()
