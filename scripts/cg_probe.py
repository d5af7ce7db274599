"""Probe: device-controlled projected CG against the host-driven loop (time per iteration, iterates)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import make_problem, spmv_setup
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
m, n = J.shape
f = HipFact(device=0)
f.set_matrix(SleqpMat(N, N, cp, ri, vx))
rng = np.random.default_rng(7)
Hl, Jd, Hd, xs, ys, ops = spmv_setup(f, J, n, m, "cuda:0", rng)
grad = rng.standard_normal(n)
for dev, three in ((1, 1), (0, 0), (1, 1)):
    f.set_option("cg_device_loop", dev)
    f.steihaug(Hd, grad, 1e6, stat_tol=1e-30, max_iter=3)
    t0 = time.perf_counter()
    step, dual, its = f.steihaug(Hd, grad, 1e6, stat_tol=1e-30, max_iter=24)
    dt = time.perf_counter() - t0
    print(f"cg_device_loop={dev} three_launch={three}: {its} iterations, {1e3*dt/max(its,1):.4f} ms/iteration, runs {f.info('cg_device_runs')} fallbacks {f.info('cg_device_fallbacks')} |step| {np.linalg.norm(step):.6e}")
    step2, dual2, its2 = f.steihaug(Hd, grad, 3.0, stat_tol=1e-8, max_iter=200)
    print(f"    radius 3: its {its2} dual {dual2:.6e} |step| {np.linalg.norm(step2):.12e}")
