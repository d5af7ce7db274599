import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
f = HipFact(device=0)
f.set_option("refine_steps", 0)
for kv in sys.argv[1:]:
    f.set_option(kv.split("=")[0], float(kv.split("=")[1]))
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
d_rhs = torch.tensor(b, device="cuda:0"); d_sol = torch.empty_like(d_rhs)
print("spf MB", f.info("spf_bytes") / 1e6, "spb MB", f.info("spb_bytes") / 1e6)
for _ in range(30): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
f.synchronize()
t0 = time.perf_counter()
for _ in range(300): f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
f.synchronize()
print(f"solve {(time.perf_counter() - t0) / 300 * 1e6:.1f} us  timeouts {f.info('solve_timeouts'):.0f}")
