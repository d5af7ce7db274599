"""Structural robustness of the ordering (host-only plan API): the table of VERDICT round 3, item 2.
usage: python scripts/ordering_probe.py [big] [superlu]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from plan_emul import Plan  # noqa: E402
from sleqp_amd import _lib, synth  # noqa: E402

lib = _lib.load()


def stats(tag, J, superlu=False):
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    t0 = time.perf_counter()
    try:
        P = Plan(lib, N, cp, ri, vx)
    except RuntimeError as e:
        print(f"{tag:46s} FAILED after {time.perf_counter() - t0:.1f} s: {e}")
        return
    dt = time.perf_counter() - t0
    line = (f"{tag:46s} levels {P.nlevels:4d} fronts {P.nsuper:5d} nnzL {P.nnzL_true:10.3e} (stored {P.nnzL:9.3e}) flops {P.flops:9.2e} "
            f"late {P.n_late:4d} late_rows {P.n_late_rows:3d} nprod {P.nprod:9.2e} analysis {P.t_total:6.2f} s (fetch {dt:5.1f})")
    if superlu:
        import scipy.sparse.linalg as spla

        K = synth.kkt_full_matrix(N, cp, ri, vx).tocsc()
        t0 = time.perf_counter()
        lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options={"SymmetricMode": True})
        line += f" | SuperLU-MMD nnz(L) {lu.L.nnz:9.3e} in {time.perf_counter() - t0:5.1f} s"
    print(line, flush=True)


big = "big" in sys.argv
slu = "superlu" in sys.argv
n, m = 20000, 10000
J0 = synth.banded_jacobian(n, m, 20, 200, 0)
stats("n=2e4 m=1e4 base", J0, slu)
stats("  + 1 dense row", synth.with_dense_rows(J0, 1, 1)[0], slu)
stats("  + 64 dense columns", synth.with_dense_columns(J0, 64, 1)[0], slu)
stats("  + 65 dense columns", synth.with_dense_columns(J0, 65, 1)[0], slu)
stats("  + 100 dense columns", synth.with_dense_columns(J0, 100, 1)[0], slu)
stats("  + 200 columns x 300 entries", synth.with_dense_columns(J0, 200, 1, entries=300)[0], slu)
stats("  + 1000 columns x 150 entries", synth.with_dense_columns(J0, 1000, 1, entries=150)[0], slu)
if big:
    n, m = 100000, 50000
    J0 = synth.banded_jacobian(n, m, 20, 200, 0)
    stats("config 4 base", J0)
    stats("  + 1 dense row", synth.with_dense_rows(J0, 1, 1)[0])
    stats("  + 1 row with 5000 entries", synth.with_dense_rows(J0, 1, 1, entries=5000)[0])
    stats("  + 3 dense rows + 16 dense columns", synth.with_dense_columns(synth.with_dense_rows(J0, 3, 1)[0], 16, 2)[0])
    stats("  + 100 dense columns", synth.with_dense_columns(J0, 100, 1)[0])
