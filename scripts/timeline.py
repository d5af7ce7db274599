"""In-kernel timeline of the single-launch top-of-tree factorisation (wall_clock64 stamps).

Builds an instrumented copy of the library (the product sources are patched in a scratch directory,
never in place), runs the config-4 workload on the GPU and prints, for the workgroups on the
critical path of the top levels, when they started, finished waiting, finished their work and
published.  The numbers quoted in DESIGN.md section 4 come from this script
(profiles/r1_timeline_factor_top.txt).

    python scripts/timeline.py build     # here (hipcc cross-compiles)
    python scripts/timeline.py run       # on the GPU box (gpurun)
"""
import ctypes as C
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, "sleqp_amd", "_timeline_build")  # sibling of csrc: same relative include paths
SRC = os.path.join(ROOT, "sleqp_amd", "csrc")


def build():
    shutil.rmtree(SCRATCH, ignore_errors=True)
    shutil.copytree(SRC, SCRATCH, ignore=shutil.ignore_patterns("*.so", "*.o"))
    flags = "-O3 -std=c++17 -fPIC -pthread -DHIPFACT_TRACE"
    subprocess.check_call(["make", "-C", SCRATCH, "CXXFLAGS=" + flags])
    print("built", os.path.join(SCRATCH, "libhipfact.so"))


def run():
    import numpy as np

    os.environ["HIPFACT_LIBRARY"] = os.path.join(SCRATCH, "libhipfact.so")
    sys.path.insert(0, ROOT)
    from bench import make_problem
    from sleqp_amd import _lib
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem(sys.argv[2] if len(sys.argv) > 2 else "banded_n1e5_m5e4", 0)
    f = HipFact(device=0)
    f.set_option("use_graph", 0)
    for kv in os.environ.get("HIPFACT_TIMELINE_OPTS", "").split():
        f.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    for _ in range(3):
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    lib = _lib.load()
    NW = 8192
    out = np.zeros(NW * 8, dtype=np.int64)
    lib.hipfact_debug_trace.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace(out.ctypes.data_as(C.c_void_p)) == 0
    n = min(int(f.info("factor_top_count")), NW)
    t = out.reshape(NW, 8)[:n]
    role, front = t[:, 7] // 100000, t[:, 7] % 100000
    tt = (t[:, :6].astype(np.float64) - t[:, 0][t[:, 0] > 0].min()) / 100.0  # wall clock: 100 MHz
    piv = np.zeros(NW * 24, dtype=np.int64)
    lib.hipfact_debug_trace_pivot.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_pivot(piv.ctypes.data_as(C.c_void_p)) == 0
    piv = piv.reshape(NW, 24)
    own = np.zeros(NW * 8, dtype=np.int64)
    lib.hipfact_debug_trace_owner.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_owner(own.ctypes.data_as(C.c_void_p)) == 0
    own = own.reshape(NW, 8)
    # per level: when its first pivot workgroup got its children, when its last Schur workgroup published
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from plan_emul import Plan

    P = Plan(lib, N, cp, ri, vx)
    lev = P.sn_level
    if os.environ.get("HIPFACT_TIMELINE_RAW"):
        np.savez(os.environ["HIPFACT_TIMELINE_RAW"], tt=tt, role=role, front=front, lev=lev[front], w=np.diff(P.sn_c0)[front], r=P.sn_r[front])
    print("# per level of the dataflow launch (us): fronts, pivots waited (first..last), pivots done (last), panels published (last), Schur published (last)")
    for l in range(int(f.info("factor_top_level")), P.nlevels):
        m_ = lev[front] == l
        pv, pn, sc = m_ & ((role == 0) | (role == 5) | (role == 6)), m_ & (role == 1), m_ & ((role == 2) | (role == 4) | (role == 5))
        # (a stamp that was never taken - a workgroup beyond the trace slots, a branch without that mark - is 0 in the raw
        # record, i.e. far below zero after the shift to the launch's first stamp: left out, not averaged in)
        def mx(a):
            a = a[a > -1e6]
            return a.max() if a.size else float("nan")

        def mn(a):
            a = a[a > -1e6]
            return a.min() if a.size else float("nan")
        w_ = tt[pv, 1]; w_ = w_[w_ > 0]
        print(f"  level {l:2d}  fronts {int(pv.sum()):4d}  panel wgs {int(pn.sum()):4d}  schur wgs {int(sc.sum()):5d}  pivots start {mn(tt[pv, 0]):7.1f}  waited {mn(w_) if w_.size else 0:7.1f} .. {mx(w_) if w_.size else 0:7.1f}"
              f"  pivots done {mx(tt[pv, 2]):7.1f}  panels published {mx(tt[pn, 3]):7.1f}  schur published {mx(tt[sc, 3]):7.1f}")
    print("# per level and role: workgroups, mean us resident before their wait ended, mean us of work behind the wait, mean us to publish; "
          "(panel / Schur: the wait is for the own front's pivot / panel workgroups)")
    for l in range(int(f.info("factor_top_level")), P.nlevels):
        m_ = (lev[front] == l) & (role != 3)
        parts = []
        for rr, nm in ((0, "pivot"), (1, "panel"), (2, "schur"), (4, "fused"), (5, "whole"), (6, "pivot+panel")):
            k_ = m_ & (role == rr) & (tt[:, 1] > 0) & (tt[:, 2] > 0)
            if k_.any():
                parts.append(f"{nm} {int(k_.sum()):4d}: wait {np.mean(tt[k_, 1] - tt[k_, 0]):6.1f}  work {np.mean(tt[k_, 2] - tt[k_, 1]):5.1f}  publish {np.mean(tt[k_, 3] - tt[k_, 2]):4.1f}")
        print(f"  level {l:2d}  " + "   ".join(parts))
    names = ["pivot", "panel", "schur", "spanl", "fused", "whole", "pv+pn"]
    sp = role == 3
    if sp.any():
        # solve-panel items (role 3): stamped at start only (slot 0); their end is not on the record, so the window
        # they run in is given by the starts
        idx = np.nonzero(sp)[0]
        print(f"# solve-panel items: {int(sp.sum())}, resident from {tt[sp, 0].min():.1f} .. {tt[sp, 0].max():.1f} us, finished {tt[sp, 2].min():.1f} .. {tt[sp, 2].max():.1f}, "
              f"mean duration after their wait {np.mean(tt[sp, 2] - tt[sp, 1]):.1f} us; by position in the launch:")
        for a in range(0, len(idx), max(1, len(idx) // 12)):
            b = idx[a:a + max(1, len(idx) // 12)]
            print(f"    items {b[0]:5d}..{b[-1]:5d}: resident {tt[b, 0].min():7.1f} .. {tt[b, 0].max():7.1f}  finished {tt[b, 2].min():7.1f} .. {tt[b, 2].max():7.1f}")
    print("# workgroup role front | us since the first workgroup started: start, before its (last) wait, after it, "
          "work done, published")
    crit = [i for i in range(n) if role[i] != 3]
    for i in crit[-int(os.environ.get("HIPFACT_TIMELINE_TAIL", "60")):]:
        extra = ""
        pre = tt[i, 5] if t[i, 5] > 0 else float("nan")  # (a stamp that was never taken is 0 in the raw record)
        print(f"{i:5d} {names[role[i]]:5s} f{front[i]:4d}  start {tt[i, 0]:8.2f}  prewait {pre:8.2f}  waited {tt[i, 1]:8.2f}"
              f"  done {tt[i, 2]:8.2f}  published {tt[i, 3]:8.2f}{extra}")
        if role[i] in (1, 4) and i >= crit[-30]:
            base = t[:, 0][t[:, 0] > 0].min()
            print("        panel, per block row: polled " + " ".join(f"{(x - base) / 100.0:.2f}" for x in piv[i][8:16] if x > 0)
                  + " | barrier " + " ".join(f"{(x - base) / 100.0:.2f}" for x in piv[i][0:8] if x > 0)
                  + " | computed " + " ".join(f"{(x - base) / 100.0:.2f}" for x in piv[i][16:24] if x > 0))
        if role[i] == 0 and i >= crit[-16]:
            base = t[:, 0][t[:, 0] > 0].min()
            st = [(x - base) / 100.0 for x in piv[i] if x > 0]
            print("        chain wave (block column 0 in LDS, first diagonal block, then per step: row handed over, "
                  "diagonal tile updated, diagonal block done): " + " ".join(f"{x:.2f}" for x in st))
            print(f"        chain finished {(t[i, 4] - base) / 100.0:.2f} (then: inverse of L11, store)")
            print("        rows 2.. handed over by their owners at: " + " ".join(f"{(x - base) / 100.0:.2f}" for x in own[i][2:] if x > 0))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "run"]()

