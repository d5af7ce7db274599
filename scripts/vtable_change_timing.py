"""Where a changed working set through the plain vtable spends its time: bench.py's working_set_change case with
HIPFACT_TIMING=1 (the library prints its host phases to stderr)."""
import os
import sys

os.environ["HIPFACT_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

J, N, cp, ri, vx, b = bench.make_problem("banded_n1e5_m5e4", 0)
out = bench.working_set_change_bench(J, 0, steps=6)
print(out["fact_vtable"])
