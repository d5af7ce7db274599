"""Phases of the host analysis (HIPFACT_TIMING ticks of analysis.cpp) on the box it runs on - no GPU needed.

    HIPFACT_TIMING=1 python scripts/cold_probe.py [reps]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from sleqp_amd import synth  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    J = synth.banded_jacobian(100000, 50000, 20, 200, 0)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    lib = C.CDLL(os.path.join(ROOT, "sleqp_amd", "csrc", "libhipfact.so"))
    lib.hipfact_plan_create.restype = C.c_int
    cp = np.ascontiguousarray(cp, dtype=np.int32)
    ri = np.ascontiguousarray(ri, dtype=np.int32)
    vx = np.ascontiguousarray(vx)
    for _ in range(reps):
        p = C.c_void_p()
        t0 = time.perf_counter()
        rc = lib.hipfact_plan_create(C.c_int(N), cp.ctypes.data_as(C.c_void_p), ri.ctypes.data_as(C.c_void_p),
                                     vx.ctypes.data_as(C.c_void_p), C.byref(p))
        print(f"rc {rc}: analysis {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
        lib.hipfact_plan_free(C.byref(p))


if __name__ == "__main__":
    main()
