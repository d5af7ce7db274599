import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_problem
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
d_vals = torch.from_numpy(vx).cuda(); db = torch.from_numpy(b).cuda(); z = torch.empty_like(db)
f = HipFact(device=0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
def step(k):
    f.refactor_device(d_vals.data_ptr())
    for _ in range(k): f.solve_device(db.data_ptr(), z.data_ptr())
def timed(tag, k=100, reps=4):
    out = []
    for _ in range(reps):
        f.synchronize(); t0 = time.perf_counter(); step(k); f.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
    print(tag, [round(x, 2) for x in out], "builds", int(f.info("top_block_builds")), "graphs?", flush=True)
for _ in range(3): step(1)
timed("fresh handle")
f.set_option("profile", -1); f.set_option("profile", 1); step(1); step(1); f.synchronize(); f.set_option("profile", 0)
timed("after profile on/off")
f.set_option("top_block_after", 0); f.set_matrix(SleqpMat(N, N, cp, ri, vx)); timed("top block off")
f.set_option("top_block_after", 2); f.set_matrix(SleqpMat(N, N, cp, ri, vx)); step(12); step(12); timed("top block on again")
timed("k=1", k=1, reps=6)
