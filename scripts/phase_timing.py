"""Timing-only ablation of k_factor_level phases (A assembly, B pivot block, C panel solve, D Schur update)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
wl = sys.argv[1] if len(sys.argv) > 1 else "banded_n1e5_m5e4"
J, N, cp, ri, vx, b = make_problem(wl, 0)
f = HipFact(device=0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
d_vals = torch.from_numpy(vx).cuda()
for mask in (1, 3, 7, 15, 2, 6, 14, 2|32, 2|64, 2|128, 2|32|64|128):
    f.set_option("debug_phases", mask)
    f.set_option("profile", -1); f.set_option("profile", 1)
    for _ in range(5):
        f.refactor_device(d_vals.data_ptr())
    f.synchronize()
    print(f"mask {mask:2d}: factor kernel {f.info('prof_factor_ms')/5:.3f} ms per factorisation")
    f.set_option("profile", 0)
