"""Phase timing of the host analysis (HIPFACT_TIMING=1) for a bench workload; no GPU needed."""
import os, sys, time
os.environ["HIPFACT_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from plan_emul import Plan
from sleqp_amd import _lib, synth
n, m = 100000, 50000
J = synth.banded_jacobian(n, m, 20, 200, 0)
vi, ci, _ = synth.working_set_all_rows(n, m)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J, vi, ci)
for _ in range(3):
    t0 = time.perf_counter()
    p = Plan(_lib.load(), N, cp, ri)
    print("plan build (incl. export) %.1f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr)
