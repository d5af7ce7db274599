import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from sleqp_amd import _lib
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
f = HipFact(device=0)
f.set_option("use_graph", 0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
for _ in range(3):
    f.solve(b)
lib = _lib.load()
out = np.zeros(512 * 8, dtype=np.int64)
lib.hipfact_debug_trace.argtypes = [C.c_void_p]
print("rc", lib.hipfact_debug_trace(out.ctypes.data_as(C.c_void_p)))
t = out.reshape(512, 8)[:135].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # us (100 MHz)
names = ["start", "prewait", "waited", "gathered", "prod1", "prod2", "published"]
for i in list(range(0, 135, 20)) + list(range(120, 135)):
    print(i, " ".join(f"{n}={t[i, k]:7.2f}" for k, n in enumerate(names)))
