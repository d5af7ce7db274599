"""Times the numeric refactorisation only (no result check): for timing probes of experimental builds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from sleqp_amd import HipfactError
J, N, cp, ri, vx, b = make_problem(sys.argv[1] if len(sys.argv) > 1 else "banded_n1e5_m5e4", 0)
f = HipFact(device=0)
for k, v in (a.split("=") for a in sys.argv[2:]):
    f.set_option(k, float(v))
try:
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
except HipfactError as e:
    print("set_matrix:", e)
    f._lib.hipfact_set_option(f._h, b"fail_omega", 1e300)
d_vals = torch.from_numpy(vx).cuda()
for _ in range(5):
    f._lib.hipfact_refactor_device(f._h, d_vals.data_ptr())
f._lib.hipfact_synchronize(f._h)
t0 = time.perf_counter()
for _ in range(100):
    f._lib.hipfact_refactor_device(f._h, d_vals.data_ptr())
f._lib.hipfact_synchronize(f._h)
print("refactor ms", (time.perf_counter() - t0) * 10)
