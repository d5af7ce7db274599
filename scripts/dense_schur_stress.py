"""SURVEY 8(d) config 4b (optional stress): uniform Jacobian, 20 nonzeros per row - the Schur complement A A^T fills in
completely, the factorisation is a dense chain of fronts (nnz(L) ~ 1.1e9, ~4e13 flops at n = 1e5, m = 5e4).
Usage: python scripts/dense_schur_stress.py [n] [m] [per_row] [repetitions]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from sleqp_amd import synth
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from util import scaled_residual
n, m, per_row = (int(float(a)) for a in (sys.argv[1:4] if len(sys.argv) > 3 else ("1e5", "5e4", "20")))
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
J = synth.uniform_jacobian(n, m, per_row, 0)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
b = np.random.default_rng(1).standard_normal(N)
f = HipFact(device=0)
t0 = time.perf_counter()
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
f.synchronize()
t_cold = time.perf_counter() - t0
free, total = torch.cuda.mem_get_info(0)
print(f"n {n} m {m}: cold set_matrix {t_cold:.2f} s (analysis {f.info('analysis_s'):.2f} s), fronts {int(f.info('nsuper'))}, levels {int(f.info('nlevels'))}, "
      f"nnz(L) {f.info('nnzL'):.3e}, flops {f.info('flops'):.3e}, HBM in use {(total - free) / 1e9:.1f} GB", flush=True)
d_val = torch.tensor(vx, device="cuda:0")
d_rhs = torch.tensor(b, device="cuda:0"); d_sol = torch.empty_like(d_rhs)
for rep in range(reps):
    t0 = time.perf_counter(); f.refactor_device(d_val.data_ptr()); f.synchronize(); t1 = time.perf_counter()
    f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr()); f.synchronize(); t2 = time.perf_counter()
    f.check()
    print(f"  refactor {t1 - t0:.3f} s = {f.info('flops') / (t1 - t0) / 1e12:.1f} TFLOP/s, solve {1e3 * (t2 - t1):.2f} ms, timeouts {f.info('solve_timeouts')} fallbacks {f.info('dataflow_fallbacks')}", flush=True)
z = d_sol.cpu().numpy()
K = synth.kkt_full_matrix(N, cp, ri, vx)
print(f"  scaled residual {scaled_residual(K, z, b):.2e}  omega {f.info('last_omega'):.2e}")
