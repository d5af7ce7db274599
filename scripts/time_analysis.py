import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from sleqp_amd import synth, _lib
from plan_emul import Plan
lib = _lib.load()
J = synth.banded_jacobian(100000, 50000, 20, 200, 0)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
for i in range(2):
    t = time.time(); P = Plan(lib, N, cp, ri, vx); dt = time.time() - t
    print(f"total {dt:.3f}s  (plan: order {P.t_order:.3f} symbolic {P.t_symbolic:.3f} total {P.t_total:.3f})")
