"""Is the solve-only loop of bench.py bound by the host (python -> ctypes -> hipfact_solve_device) or by the device?
Enqueue time of 200 solves (no synchronisation) against the time until the device has finished them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
f = HipFact(device=0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
d_rhs = torch.tensor(b, device="cuda:0"); d_sol = torch.empty_like(d_rhs)
for g, xf in ((1, 1), (0, 1), (1, 0), (0, 0)):
    f.set_option("use_graph", g)
    f.set_option("xupd_fused", xf)
    for _ in range(20): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    t1 = time.perf_counter()
    f.synchronize()
    t2 = time.perf_counter()
    print(f"use_graph={g} xupd_fused={xf}: enqueue {1e6*(t1-t0)/200:.1f} us/solve, until done {1e6*(t2-t0)/200:.1f} us/solve")
