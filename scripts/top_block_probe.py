"""Solve with and without the top block (device-resident, steady state), and its correctness against the plain path."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
wl = sys.argv[1] if len(sys.argv) > 1 else "banded_n1e5_m5e4"
J, N, cp, ri, vx, b = make_problem(wl, 0)
d_vals = torch.from_numpy(vx).cuda()
db = torch.from_numpy(b).cuda()
ref = None
for after, cap, single in ((0, 2048, 0), (2, 2048, 0), (2, 2048, 1), (2, 1024, 1)):
    f = HipFact(device=0)
    f.set_option("top_block_after", after)
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    z = torch.empty_like(db)
    for _ in range(4):
        f.solve_device(db.data_ptr(), z.data_ptr())
    f.synchronize(); f.check()
    zz = z.cpu().numpy()
    if ref is None: ref = zz
    err = np.abs(zz - ref).max() / np.abs(ref).max()
    t0 = time.perf_counter()
    for _ in range(200):
        f.solve_device(db.data_ptr(), z.data_ptr())
    f.synchronize()
    t_sol = (time.perf_counter() - t0) / 200
    # the unit at 1 factor : 100 solves
    t0 = time.perf_counter()
    for _ in range(5):
        f.refactor_device(d_vals.data_ptr())
        for _ in range(100):
            f.solve_device(db.data_ptr(), z.data_ptr())
    f.synchronize()
    t_sqp = (time.perf_counter() - t0) / 5
    print(f"after {after} cap {cap} single {single}: top block cols {int(f.info('top_block_cols'))} levels {int(f.info('top_block_levels'))} items {int(f.info('top_block_items'))} "
          f"builds {int(f.info('top_block_builds'))} active {int(f.info('top_block_active'))} | solve {t_sol*1e3:.4f} ms  sqp(1:100) {t_sqp*1e3:.3f} ms  diff vs plain {err:.2e}", flush=True)
    f.free()
