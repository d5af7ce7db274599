"""Device-resident cost of structurally hard inputs next to the base: per kernel class (HIP events), factor / solve.
usage: python scripts/robust_probe.py [case ...]   cases: base hub_row hub_row5000 cols16 cols100 mix"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sleqp_amd import synth  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402

CLS = ("memset", "mvals", "gather", "factor", "factorA", "factorB", "factorC", "factorD", "factorT", "spanel", "fwd", "bwd",
       "tree", "rhs", "xupd", "resid", "axpy", "perm")


def run(tag, J):
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    f = HipFact(device=0)
    t0 = time.perf_counter()
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    cold = time.perf_counter() - t0
    d_vals = torch.from_numpy(vx).cuda()
    b = torch.randn(N, dtype=torch.float64, device="cuda")
    z = torch.empty_like(b)
    for _ in range(3):
        f.refactor_device(d_vals.data_ptr())
        f.solve_device(b.data_ptr(), z.data_ptr())
    f.synchronize()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        f.refactor_device(d_vals.data_ptr())
    f.synchronize()
    t_fac = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(100):
        f.solve_device(b.data_ptr(), z.data_ptr())
    f.synchronize()
    t_sol = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for _ in range(reps):
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    t_set = (time.perf_counter() - t0) / reps
    f.set_option("profile", -1)
    f.set_option("profile", 1)
    for _ in range(5):
        f.refactor_device(d_vals.data_ptr())
        f.solve_device(b.data_ptr(), z.data_ptr())
    f.synchronize()
    prof = {c: round(f.info(f"prof_{c}_ms") / 5 * 1e3, 1) for c in CLS if f.info(f"prof_{c}_count") > 0}
    f.set_option("profile", 0)
    print(f"{tag:14s} levels {int(f.info('nlevels')):3d} fronts {int(f.info('nsuper')):4d} nnzL {f.info('nnzL_true'):.3e} late {int(f.info('late_columns'))}/{int(f.info('late_rows'))} "
          f"cold {cold * 1e3:7.1f} ms | factor {t_fac * 1e3:6.3f} ms solve {t_sol * 1e3:6.3f} ms set_matrix(host K) {t_set * 1e3:6.3f} ms | us per class: {prof}", flush=True)
    f.free()


cases = sys.argv[1:] or ["base", "hub_row", "cols16"]
J4 = synth.banded_jacobian(100000, 50000, 20, 200, 0)
for c in cases:
    if c == "base":
        run(c, J4)
    elif c == "hub_row":
        run(c, synth.with_dense_rows(J4, 1, 1)[0])
    elif c == "hub_row5000":
        run(c, synth.with_dense_rows(J4, 1, 1, entries=5000)[0])
    elif c == "cols16":
        run(c, synth.with_dense_columns(J4, 16, 3)[0])
    elif c == "cols100":
        run(c, synth.with_dense_columns(J4, 100, 3)[0])
    elif c == "mix":
        run(c, synth.with_dense_columns(synth.with_dense_rows(J4, 3, 1)[0], 16, 2)[0])
