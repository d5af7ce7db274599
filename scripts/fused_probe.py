"""Timing + residual probe of experimental factorisation variants: options as name=value on the command line,
groups separated by '/'.  Prints refactor ms, solve us, scaled residual of one checked solve and the info counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import scipy.sparse as sp
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat

wl = os.environ.get("WORKLOAD", "banded_n1e5_m5e4")
J, N, cp, ri, vx, b = make_problem(wl, 0)
Kl = sp.csc_matrix((vx, ri, cp), shape=(N, N))
K = Kl + sp.tril(Kl, -1).T
groups = " ".join(sys.argv[1:]).split("/") if len(sys.argv) > 1 else [""]
d_vals = torch.from_numpy(vx).cuda()
d_rhs = torch.tensor(b, device="cuda:0")
d_sol = torch.empty_like(d_rhs)
for g in groups:
    f = HipFact(device=0)
    for kv in g.split():
        k, v = kv.split("=")
        f.set_option(k, float(v))
    try:
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    except Exception as e:  # (a timing experiment may leave a factor the check refuses: still timed)
        print("   set_matrix:", str(e)[:100], flush=True)
    if os.environ.get("STEPWISE"):
        # residual after the first factorisation and after each of a few refactorisations
        out = []
        for k in range(6):
            f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
            f.synchronize()
            x = d_sol.cpu().numpy()
            out.append(np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
            f.refactor_device(d_vals.data_ptr())
        print("   stepwise residuals:", " ".join(f"{v:.1e}" for v in out), flush=True)
    for _ in range(5):
        f.refactor_device(d_vals.data_ptr())
    f.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        f.refactor_device(d_vals.data_ptr())
    f.synchronize()
    tf = (time.perf_counter() - t0) * 10
    f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    x = d_sol.cpu().numpy()
    res = np.abs(K @ x - b).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max())
    for _ in range(20):
        f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    ts = (time.perf_counter() - t0) * 5e3
    print(f"[{g.strip()}] refactor {tf:.4f} ms  solve {ts:.1f} us  resid {res:.2e}  timeouts {f.info('solve_timeouts'):.0f} "
          f"fallbacks {f.info('dataflow_fallbacks'):.0f} items {f.info('factor_top_count'):.0f}", flush=True)
    f.free()
