"""Two (or more) full-size handles from as many threads on one GPU, repeated: counts the dataflow fallbacks
(HIPFACT_TRACE_TIMEOUT=1 prints where a wait timed out).   python scripts/two_handles_probe.py [runs] [threads]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIPFACT_TRACE_TIMEOUT", "1")
from bench import make_problem  # noqa: E402
from sleqp_amd.fact import HipFact  # noqa: E402
from sleqp_amd.sparse import SleqpMat  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
probs = [make_problem("banded_n1e5_m5e4", t) for t in range(nth)]
total = 0
for run in range(runs):
    out = [None] * nth

    def worker(t):
        J, N, cp, ri, vx, b = probs[t]
        f = HipFact(device=0)
        for rep in range(12):
            f.set_matrix(SleqpMat(N, N, cp, ri, vx))
            for _ in range(4):
                f.solve(b)
            f.solution_raw(0, N)
        out[t] = (f.info("dataflow_fallbacks"), f.info("solve_timeouts"), f.info("turn_waits"))
        f.free()

    t0 = time.time()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(nth)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    total += sum(o[0] for o in out)
    print(run, "%.2f s" % (time.time() - t0), out, flush=True)
print("fallbacks in", runs, "runs:", total)
