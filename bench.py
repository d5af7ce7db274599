#!/usr/bin/env python3
"""KKT factor+solve benchmark (BASELINE.json metric) for the hipfact backend.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic KKT system whose data
is already resident in HBM: numeric refactorisation of K (SLEQP_FACT_SET_MATRIX
with an unchanged sparsity pattern, i.e. the steady state of an SQP run) plus
one solve K z = b with iterative refinement (SLEQP_FACT_SOLVE).  Workload at
N=1: BASELINE.json configs[3] (n=1e5, m=5e4, nnz(J)=1e6 `banded`, SURVEY.md
§8d).  For N>1 every rank factors an independent problem (seed = rank) on its
own GPU — BASELINE.json configs[4]; there is no data-path collective
("replicas only"), torch.distributed (RCCL) is used for the
barrier and the max-over-ranks reduction only.

Prints ONE JSON line on rank 0 (contract in the task description) carrying
`roofline` (dominant kernel, HIP-event timed on the handle's own stream) and
`cpu_baseline` (the oracle's simplicial sparse LDL^T on the host, rank 0, N=1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md)
MFMA_F64_PEAK_TFLOPS = 78.6  # dense fp64 matrix peak (same guide)


def make_problem(workload: str, seed: int):
    from sleqp_amd import synth

    if workload == "banded_n1e5_m5e4":
        J = synth.banded_jacobian(100000, 50000, 20, 200, seed)
    elif workload == "banded_n1e4_m5e3":
        J = synth.banded_jacobian(10000, 5000, 20, 200, seed)
    elif workload == "uniform_n1e4_m5e3":
        J = synth.uniform_jacobian(10000, 5000, 10, seed)
    elif workload == "tiny":
        J = synth.banded_jacobian(400, 200, 8, 60, seed)
    else:
        raise ValueError(workload)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
    b = np.random.default_rng(seed + 1).standard_normal(N)
    return J, N, cp, ri, vx, b


def algorithmic_bytes(fact):
    """SURVEY.md §8(d) figures from the backend's own symbolic analysis."""
    nnzK = fact.info("nnzK")
    nnzL = fact.info("nnzL") + fact.info("n") + (fact.info("nnzK") - fact.info("n") if fact.info("saddle") else 0)
    # stored row indices: one list per supernode (+ A's indices in saddle mode)
    nnz_idx = fact.info("rows_total") + (fact.info("nnzK") if fact.info("saddle") else 0)
    N = fact.info("N")
    factor = 12 * nnzK + 16 * nnzL + 4 * nnz_idx
    solve = 2 * (8 * nnzL + 4 * nnz_idx) + 8 * N + 3 * 8 * N
    return factor, solve, nnzL


def measured_ceilings(device):
    """STREAM triad and fp64 GEMM on this device (torch / rocBLAS: measurement plumbing only), so that the
    roofline can also be read against measured instead of datasheet ceilings."""
    import torch

    n = 1 << 27  # 3 x 1 GiB
    a = torch.empty(n, dtype=torch.float64, device=device)
    b = torch.ones(n, dtype=torch.float64, device=device)
    c = torch.ones(n, dtype=torch.float64, device=device)
    def triad():
        torch.add(b, c, alpha=3.0, out=a)
    for _ in range(3):
        triad()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        triad()
    torch.cuda.synchronize(device)
    triad_gbs = 3 * 8 * n * reps / (time.perf_counter() - t0) / 1e9
    del a, b, c
    m = 8192
    x = torch.randn(m, m, dtype=torch.float64, device=device)
    y = torch.randn(m, m, dtype=torch.float64, device=device)
    for _ in range(2):
        torch.matmul(x, y)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        torch.matmul(x, y)
    torch.cuda.synchronize(device)
    gemm_tf = 2.0 * m ** 3 * reps / (time.perf_counter() - t0) / 1e12
    return {"hbm_triad_GBps": triad_gbs, "fp64_gemm_TFLOPs": gemm_tf,
            "note": "torch.add triad on 3 x 1 GiB, torch.matmul fp64 8192^3 (rocBLAS)"}


def cpu_baseline(N, cp, ri, vx, b, budget_s=20.0):
    """Oracle (oracle/kkt_oracle.c) simplicial LDL^T, single thread, x-before-y natural order."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle  # test infrastructure; only the baseline leg uses it

    reps, t_total = 0, 0.0
    t_factor = t_solve = 0.0
    while reps < 1 or (t_total < budget_s and reps < 50):
        t0 = time.perf_counter()
        F = oracle.OracleLdl(N, cp, ri, vx)
        t1 = time.perf_counter()
        F.solve(b)
        t2 = time.perf_counter()
        t_factor += t1 - t0
        t_solve += t2 - t1
        t_total += t2 - t0
        reps += 1
        del F
    return {
        "value": reps / t_total,
        "unit": "factor+solve/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{reps} x (numeric+symbolic simplicial LDL^T + 1 solve) of the same K on 1 host core, "
                  f"factor {t_factor / reps * 1e3:.1f} ms, solve {t_solve / reps * 1e3:.2f} ms",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="banded_n1e5_m5e4")
    ap.add_argument("--refine", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the STREAM / DGEMM ceiling microbenchmarks")
    ap.add_argument("--solves-per-factor", type=int, default=1)
    args = ap.parse_args()

    import torch

    from sleqp_amd.replicas import Replicas

    rep = Replicas()
    rank, local_rank, world, dist = rep.rank, rep.local_rank, rep.world, rep.dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hipfact backend has no CPU path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem(args.workload, seed=rep.problem_seed())
    fact = HipFact(device=local_rank, refine_steps=args.refine)
    t0 = time.perf_counter()
    fact.set_matrix(SleqpMat(N, N, cp, ri, vx))  # cold call: analysis + upload + first factorisation
    fact.synchronize()
    t_cold = time.perf_counter() - t0
    d_vals = torch.from_numpy(vx).to(dev)
    d_rhs = torch.from_numpy(b).to(dev)
    d_sol = torch.empty_like(d_rhs)
    torch.cuda.synchronize()

    def step():
        fact.refactor_device(d_vals.data_ptr())
        for _ in range(args.solves_per_factor):
            fact.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())

    def barrier():
        rep.barrier()
        torch.cuda.synchronize()
        fact.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fact.synchronize()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0
    rep.barrier()
    t_max = rep.max_over_ranks(t_local)

    # correctness of what was timed (scaled residual of the last solve)
    from sleqp_amd import synth

    K = synth.kkt_full_matrix(N, cp, ri, vx)
    z = d_sol.cpu().numpy()
    resid = float(np.abs(K @ z - b).max() / (abs(K).sum(axis=1).max() * np.abs(z).max() + np.abs(b).max()))

    # ---- per-kernel-class timing with HIP events on the handle's stream
    fact.set_option("profile", -1)
    fact.set_option("profile", 1)
    prof_steps = max(3, min(args.steps, 10))
    for _ in range(prof_steps):
        step()
    fact.synchronize()
    prof = {}
    for cls in ("memset", "mvals", "gather", "factor", "factorA", "factorB", "factorC", "factorD", "factorT", "fwd", "bwd", "rhs",
                "xupd", "resid", "axpy", "perm"):
        ms, cnt = fact.info(f"prof_{cls}_ms"), fact.info(f"prof_{cls}_count")
        if cnt > 0:
            prof[cls] = {"ms_per_step": ms / prof_steps, "launches_per_step": cnt / prof_steps,
                         "avg_launch_us": ms / cnt * 1e3}
    fact.set_option("profile", 0)

    # ---- solve-only rate (the real ratio is ~1 factor : 100 solves, trlib_solver.c:768-776)
    nsolve = 50
    fact.synchronize()
    r0 = fact.info("num_refined")
    t0 = time.perf_counter()
    for _ in range(nsolve):
        fact.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    fact.synchronize()
    t_solve = (time.perf_counter() - t0) / nsolve
    # passes over the factor per solve: 1 + fraction of solves whose residual asked for a correction pass
    passes = 1.0 + (fact.info("num_refined") - r0) / nsolve * max(args.refine, 0)

    # ---- device-resident projected CG (tr/steihaug_solver.c loop): 20 iterations,
    # Hessian = symmetric banded SPD matrix, half-bandwidth 5 (§8d), every CG vector resident in HBM
    eqp = None
    if args.workload.startswith("banded") and rank == 0:
        import scipy.sparse as sp

        from sleqp_amd.fact import SpMat

        n = J.shape[1]
        rngh = np.random.default_rng(7)
        diags = [rngh.standard_normal(n - k) * 0.1 for k in range(1, 6)]
        Hl = sp.diags([np.full(n, 2.0)] + diags, [0, -1, -2, -3, -4, -5], format="csc")
        Hl.sort_indices()
        H = SpMat(fact, SleqpMat(n, n, Hl.indptr, Hl.indices, Hl.data))
        grad = rngh.standard_normal(n)
        fact.steihaug(H, grad, 1e6, stat_tol=1e-30, max_iter=3)  # warm-up (graph capture)
        t0 = time.perf_counter()
        _, _, its = fact.steihaug(H, grad, 1e6, stat_tol=1e-30, max_iter=20)
        t_cg = time.perf_counter() - t0
        eqp = {"iterations": its, "ms_total": t_cg * 1e3, "ms_per_iteration": t_cg * 1e3 / max(its, 1),
               "note": "1 projection (KKT solve) + 1 symmetric Hessian SpMV + 3 reductions per iteration, "
                       "host sees 3 scalars per iteration"}

    if rank == 0:
        fbytes, sbytes, nnzL = algorithmic_bytes(fact)
        dom = max(prof, key=lambda k: prof[k]["ms_per_step"]) if prof else "factor"
        kernel_names = {"factor": "k_factor_level", "factorA": "k_front_assemble", "factorB": "k_front_pivot",
                        "factorC": "k_front_panel", "factorD": "k_front_schur", "factorT": "k_factor_top",
                        "fwd": "k_fwd_top" if fact.info("top_level") == 0 else "k_fwd_level",
                        "bwd": "k_bwd_top" if fact.info("top_level") == 0 else "k_bwd_level", "mvals": "k_mvals_prod",
                        "memset": "hipMemsetAsync(L arena)"}
        launches = prof[dom]["launches_per_step"]
        # algorithmic bytes attributable to the dominant kernel, per launch:
        #   factor kernels: write L once + read it once for the updates (16 B/entry) + row indices (4 B) of the
        #   fronts that kernel family processes; the split phases A-D share their fronts' bytes.
        if dom == "factor":
            step_bytes = 16 * fact.info("ent_fused") + 4 * fact.info("rows_fused")
        elif dom in ("factorA", "factorB", "factorC", "factorD", "factorT"):
            # the per-level kernels and the single-launch top-of-tree kernel share the fronts' bytes by time
            fam = sum(prof[k]["ms_per_step"] for k in ("factorA", "factorB", "factorC", "factorD", "factorT") if k in prof)
            step_bytes = (16 * fact.info("ent_split") + 4 * fact.info("rows_split")) * prof[dom]["ms_per_step"] / fam
        elif dom in ("fwd", "bwd"):
            step_bytes = (sbytes / 2) * prof[dom]["launches_per_step"] / max(fact.info("nlevels"), 1)
        elif dom == "mvals":
            step_bytes = 12 * fact.info("nnzK") + 8 * fact.info("nnzM")
        else:
            step_bytes = fbytes
        bytes_per_launch = step_bytes / max(launches, 1)
        avg_s = prof[dom]["avg_launch_us"] * 1e-6
        achieved = bytes_per_launch / avg_s / 1e9
        factor_ms = sum(prof[k]["ms_per_step"] for k in ("memset", "mvals", "gather", "factor", "factorA", "factorB",
                                                          "factorC", "factorD", "factorT") if k in prof)
        # HBM traffic of the dominant kernel from the PMC counters (separate rocprofv3 --pmc passes of this
        # same command, profiles/r1_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE)
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")))
            rec = pmc.get(kernel_names.get(dom, dom))
            if rec and args.workload == "banded_n1e5_m5e4":
                traffic = rec["fetch_bytes_per_launch_x2"] + rec["write_bytes_per_launch"]
        except (OSError, ValueError):
            pass
        out = {
            "metric": "KKT factor+solve/sec (numeric refactor + 1 solve with residual-checked refinement, inputs resident in HBM)",
            "value": rep.aggregate_rate(args.steps, t_max),
            "unit": "factor+solve/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "n": int(fact.info("n")), "m": int(fact.info("m")), "N": N,
                       "nnz_J": int(J.nnz), "nnz_tril_K": int(fact.info("nnzK")), "nnz_L": int(nnzL),
                       "supernodes": int(fact.info("nsuper")), "etree_levels": int(fact.info("nlevels")),
                       "refine_steps": args.refine, "refine_adaptive": bool(fact.info("refine_adaptive")),
                       "refine_tol": 5e-13, "hip_graphs": int(fact.info("num_graphs")),
                       "solves_per_factor": args.solves_per_factor,
                       "parallelism": f"replicas{world}" if world > 1 else "single"},
            "roofline": {"bound": "hbm", "kernel": kernel_names.get(dom, dom), "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_us": prof[dom]["avg_launch_us"]},
            "kernels": prof,
            "solve_only": {"solves_per_s": 1.0 / t_solve, "ms_per_solve": t_solve * 1e3,
                           "passes_per_solve": passes, "algorithmic_GBps": sbytes * passes / t_solve / 1e9},
            "factor_only_ms": factor_ms,
            "factor_family_GBps": fbytes / (factor_ms * 1e-3) / 1e9,
            "eqp_cg_device": eqp,
            "cold_set_matrix_s": t_cold,
            "analysis_s": fact.info("analysis_s"),
            "scaled_residual": resid,
        }
        # dense-front workloads (config 3): the Schur kernel is bound by the fp64 matrix cores, not by HBM
        flops = fact.info("flops")
        if dom == "factorD" and flops / max(fbytes, 1.0) > MFMA_F64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9):
            tf = flops / (prof[dom]["ms_per_step"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": kernel_names[dom], "achieved": tf, "peak": MFMA_F64_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": tf / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                               "flops_per_step": flops, "avg_launch_us": prof[dom]["avg_launch_us"],
                               "note": "all factor flops attributed to the Schur kernel (upper bound)"}
        if world == 1 and not args.no_ceilings:
            out["measured_ceilings"] = measured_ceilings(f"cuda:{local_rank}")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, cp, ri, vx, b)
        print(json.dumps(out), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
