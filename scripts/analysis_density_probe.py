"""Host-only probe: time of the symbolic analysis (hipfact_plan_create) of a banded Jacobian with `nz` entries per row in a
window of `width` columns (argv: nz width).  HIPFACT_TIMING=1 prints the phases, HIPFACT_NO_THP=1 drops the huge-page advice."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sleqp_amd import synth
import sleqp_amd, ctypes as C
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 60
width = int(sys.argv[2]) if len(sys.argv) > 2 else 200
J = synth.banded_jacobian(100000, 50000, nz, width, 0)
vi, ci, W = synth.working_set_all_rows(100000, 50000, 0.0, 0)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J, vi, ci)
lib = sleqp_amd.load()
plan = C.c_void_p()
cp = np.ascontiguousarray(cp, dtype=np.int32); ri = np.ascontiguousarray(ri, dtype=np.int32)
t0 = time.perf_counter()
rc = lib.hipfact_plan_create(C.c_int(N), cp.ctypes.data_as(C.c_void_p), ri.ctypes.data_as(C.c_void_p), None, C.byref(plan))
dt = time.perf_counter() - t0
val = C.c_double()
out = {}
for k in ("nlevels", "nsuper", "flops", "nnzL", "nprod", "max_r"):
    lib.hipfact_plan_scalar.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
    if lib.hipfact_plan_scalar(plan, k.encode(), C.byref(val)) == 0: out[k] = val.value
print(f"nz/row {nz} width {width}: rc {rc} analysis {dt:.3f} s", out)
