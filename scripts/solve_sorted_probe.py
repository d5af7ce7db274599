"""Solve-only rate: x update as a launch of its own / inside the tree launch (HIPFACT_XUPD_FUSED), with different numbers of workgroups."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
ref = None
for fused, blocks in ((1, 64), (1, 56), (1, 48), (1, 44), (1, 40), (1, 36), (0, 48)):
    os.environ["HIPFACT_XUPD_FUSED"] = str(fused)
    os.environ["HIPFACT_SOLVE_WHOLE_MAX"] = str(blocks)
    f = HipFact(device=0)
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    d_rhs = torch.tensor(b, device="cuda:0"); d_sol = torch.empty_like(d_rhs)
    for _ in range(40): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(400): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
        f.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    f.check()
    x = d_sol.cpu().numpy()
    if ref is None: ref = x.copy()
    print(f"fused={fused} blocks={blocks}: {1e6*best:.1f} us/solve  max diff {np.abs(x-ref).max():.1e} items {f.info('solve_items')}  timeouts {f.info('dataflow_fallbacks')}")
    del f
