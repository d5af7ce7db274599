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
    p = os.path.join(SCRATCH, "kernels.hip")
    s = open(p).read()
    s = s.replace("typedef double d4_t __attribute__((ext_vector_type(4)));",
                  "__device__ long long g_trace[4096 * 8];\n"
                  "#define TRW(slot) if (threadIdx.x == 0) g_trace[blockIdx.x * 8 + (slot)] = wall_clock64()\n"
                  "typedef double d4_t __attribute__((ext_vector_type(4)));", 1)
    a = s.index("__global__ __launch_bounds__(512) void k_factor_top(")
    b = s.index("// titems: fronts of the levels >= top_level, children before parents")
    seg = s[a:b]
    seg = seg.replace("  const FrontItem& S = T.it;\n",
                      "  const FrontItem& S = T.it;\n  TRW(0);\n  if (threadIdx.x == 0) g_trace[blockIdx.x * 8 + 7] = T.role * 100000 + T.front;\n", 1)
    seg = seg.replace("    dev_build_solve_panel(sitems[S.part], L, SPf, SPb, lds);\n    return;",
                      "    TRW(1);\n    dev_build_solve_panel(sitems[S.part], L, SPf, SPb, lds);\n    __syncthreads();\n    TRW(2);\n    return;", 1)
    seg = seg.replace("    flag_publish_add(&bdone[T.front]);", "    TRW(2);\n    flag_publish_add(&bdone[T.front]);\n    TRW(3);", 1)
    seg = seg.replace("    dev_panel_rows_product_posted(", "    TRW(1);\n    dev_panel_rows_product_posted(", 1)
    seg = seg.replace("    flag_publish_add(&cdone[T.front]);", "    TRW(2);\n    flag_publish_add(&cdone[T.front]);\n    TRW(3);", 1)
    seg = seg.replace("    flag_publish_add(&ddone[T.front]);", "    TRW(2);\n    flag_publish_add(&ddone[T.front]);\n    TRW(3);", 1)
    s = s[:a] + seg + s[b:]
    s = s.replace("      cw.wait();  // top-of-tree launch: everything above was requested before the children are awaited",
                  "      if (threadIdx.x == 0 && cw.n > 0) g_trace[blockIdx.x * 8 + 5] = wall_clock64();\n"
                  "      cw.wait();\n"
                  "      if (threadIdx.x == 0 && cw.n > 0) g_trace[blockIdx.x * 8 + 1] = wall_clock64();", 1)
    s = s.replace("    flag_wait_ge(wait_addr, wait_target, info);\n  }\n  const int si = tid & 63;",
                  "    if (threadIdx.x == 0) g_trace[blockIdx.x * 8 + 5] = wall_clock64();\n"
                  "    flag_wait_ge(wait_addr, wait_target, info);\n"
                  "    if (threadIdx.x == 0) g_trace[blockIdx.x * 8 + 1] = wall_clock64();\n  }\n  const int si = tid & 63;", 1)
    # pivot role detail: after the children gather, after the first diagonal block, after every step
    s = s.replace("__device__ long long g_trace[4096 * 8];", "__device__ long long g_trace[4096 * 8];\n__device__ long long g_piv[4096 * 24];\n"
                  "#define TRP(slot) if (threadIdx.x == 0) g_piv[blockIdx.x * 24 + (slot)] = wall_clock64()", 1)
    s = s.replace("      __syncthreads();\n      if (wave == 0) {\n        if (!(phases & 32)) dev_diag_block(c, scratch, 0, info);\n      } else {\n#pragma unroll\n        for (int cc = 0; cc < 2; ++cc)",
                  "      __syncthreads();\n      TRP(0);\n      if (wave == 0) {\n        if (!(phases & 32)) dev_diag_block(c, scratch, 0, info);\n        TRP(1);\n      } else {\n#pragma unroll\n        for (int cc = 0; cc < 2; ++cc)", 1)
    s = s.replace("    __syncthreads();\n    // S3 + look-ahead S1: tile t = 0 is the next diagonal block (kb+1, kb+1)", "    __syncthreads();\n    TRP(2 + 2 * kb);\n    // S3 + look-ahead S1: tile t = 0 is the next diagonal block (kb+1, kb+1)", 1)
    s = s.replace("      xpend = x;\n      xrow = kb;\n      xcol = j;\n    }\n    __syncthreads();\n  }", "      xpend = x;\n      xrow = kb;\n      xcol = j;\n    }\n    __syncthreads();\n    TRP(3 + 2 * kb);\n  }", 1)
    # finer: inside a step, wave 0 (next diagonal tile, diagonal block) and wave 1 (trailing tiles, inverse row)
    s = s.replace("#define TRP(slot)", "__device__ long long g_fine[4096 * 40];\n"
                  "#define TRF(w, slot) if (threadIdx.x == 64 * (w)) g_fine[blockIdx.x * 40 + (slot)] = wall_clock64()\n#define TRP(slot)", 1)
    s = s.replace("        if (!(phases & 64)) dev_trailing_tile(c, k0, kb + 1, kb + 1);\n        if (!(phases & 32)) dev_diag_block(c, scratch, k0 + 16, info);",
                  "        if (!(phases & 64)) dev_trailing_tile(c, k0, kb + 1, kb + 1);\n        TRF(0, 4 * kb);\n        const long long cyc0 = __builtin_readcyclecounter();\n        if (!(phases & 32)) dev_diag_block(c, scratch, k0 + 16, info);\n        if (threadIdx.x == 0 && kb == 0) g_fine[blockIdx.x * 40 + 39] = __builtin_readcyclecounter() - cyc0;\n        TRF(0, 4 * kb + 1);", 1)
    s = s.replace("    if (ROWINV && wave == 1 + (kb + 3) % 7) {", "    TRF(1, 4 * kb + 2);\n    if (ROWINV && wave == 1 + (kb + 3) % 7) {", 1)
    s = s.replace("      xpend = x;\n      xrow = kb;\n      xcol = j;\n    }\n    __syncthreads();", "      xpend = x;\n      xrow = kb;\n      xcol = j;\n    }\n    TRF(1, 4 * kb + 3);\n    __syncthreads();", 1)
    # anatomy of one diagonal block (the fourth of a front): shader clocks at entry, operands loaded, steps 0-7
    # done, steps 8-14 done, results stored
    s = s.replace("  double a[16], x[4];\n", "  double a[16], x[4];\n  const long long dcA = __builtin_readcyclecounter();\n", 1)
    s = s.replace("  double dsel = a[0];  // pivot 0 (lane li = 0 keeps it)",
                  "  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n  const long long dcB = __builtin_readcyclecounter();\n  double dsel = a[0];  // pivot 0 (lane li = 0 keeps it)", 1)
    s = s.replace("  nl = diag_step<7>(a, x, dsel, li, nl);\n", "  nl = diag_step<7>(a, x, dsel, li, nl);\n  const long long dcC = __builtin_readcyclecounter();\n", 1)
    s = s.replace("  nl = diag_step<14>(a, x, dsel, li, nl);\n", "  nl = diag_step<14>(a, x, dsel, li, nl);\n  const long long dcD = __builtin_readcyclecounter();\n", 1)
    s = s.replace("    if (nneg) atomicAdd(&info[INFO_NEG_PIVOT], nneg);\n  }\n}",
                  "    if (nneg) atomicAdd(&info[INFO_NEG_PIVOT], nneg);\n  }\n"
                  "  if (threadIdx.x == 0 && k0 == 48) {\n    const long long dcE = __builtin_readcyclecounter();\n"
                  "    long long* g = g_fine + blockIdx.x * 40 + 32;\n    g[0] = dcB - dcA;\n    g[1] = dcC - dcB;\n    g[2] = dcD - dcC;\n    g[3] = dcE - dcD;\n  }\n}", 1)
    assert "dcE" in s and "dcC" in s and "dcB" in s and "dcA" in s
    assert s.count("TRF(") >= 5
    assert "TRP(0)" in s and "TRP(1)" in s and s.count("g_trace[blockIdx.x * 8 + 1]") >= 2
    assert s.count("TRW(") >= 7
    open(p, "w").write(s)
    h = os.path.join(SCRATCH, "hipfact.hip")
    t = open(h).read()
    t += ('\nextern "C" int hipfact_debug_trace(long long* out) {\n'
          "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_trace), sizeof(long long) * 4096 * 8);\n}\n"
          'extern "C" int hipfact_debug_trace_fine(long long* out) {\n'
          "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_fine), sizeof(long long) * 4096 * 40);\n}\n"
          'extern "C" int hipfact_debug_trace_pivot(long long* out) {\n'
          "  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_piv), sizeof(long long) * 4096 * 24);\n}\n")
    open(h, "w").write(t)
    subprocess.check_call(["make", "-C", SCRATCH])
    print("built", os.path.join(SCRATCH, "libhipfact.so"))


def run():
    import numpy as np

    os.environ["HIPFACT_LIBRARY"] = os.path.join(SCRATCH, "libhipfact.so")
    sys.path.insert(0, ROOT)
    from bench import make_problem
    from sleqp_amd import _lib
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
    f = HipFact(device=0)
    f.set_option("use_graph", 0)
    for _ in range(3):
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    lib = _lib.load()
    out = np.zeros(4096 * 8, dtype=np.int64)
    lib.hipfact_debug_trace.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace(out.ctypes.data_as(C.c_void_p)) == 0
    n = min(int(f.info("factor_top_count")), 4096)
    t = out.reshape(4096, 8)[:n]
    role, front = t[:, 7] // 100000, t[:, 7] % 100000
    tt = (t[:, :6].astype(np.float64) - t[:, 0][t[:, 0] > 0].min()) / 100.0  # wall clock: 100 MHz
    piv = np.zeros(4096 * 24, dtype=np.int64)
    lib.hipfact_debug_trace_pivot.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_pivot(piv.ctypes.data_as(C.c_void_p)) == 0
    piv = piv.reshape(4096, 24)
    fine = np.zeros(4096 * 40, dtype=np.int64)
    lib.hipfact_debug_trace_fine.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_fine(fine.ctypes.data_as(C.c_void_p)) == 0
    fine = fine.reshape(4096, 40)
    # per level: when its first pivot workgroup got its children, when its last Schur workgroup published
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from plan_emul import Plan

    P = Plan(lib, N, cp, ri, vx)
    lev = P.sn_level
    print("# per level of the dataflow launch (us): fronts, pivots waited (first..last), pivots done (last), panels published (last), Schur published (last)")
    for l in range(int(f.info("factor_top_level")), P.nlevels):
        m_ = lev[front] == l
        pv, pn, sc = m_ & (role == 0), m_ & (role == 1), m_ & (role == 2)
        def mx(a): return a.max() if a.size else float("nan")
        def mn(a): return a.min() if a.size else float("nan")
        w_ = tt[pv, 1]; w_ = w_[w_ > 0]
        print(f"  level {l:2d}  fronts {int(pv.sum()):4d}  panel wgs {int(pn.sum()):4d}  schur wgs {int(sc.sum()):5d}  pivots start {mn(tt[pv, 0]):7.1f}  waited {mn(w_) if w_.size else 0:7.1f} .. {mx(w_) if w_.size else 0:7.1f}"
              f"  pivots done {mx(tt[pv, 2]):7.1f}  panels published {mx(tt[pn, 3]):7.1f}  schur published {mx(tt[sc, 3]):7.1f}")
    print("# per level and role: workgroups, mean us resident before their wait ended, mean us of work behind the wait, mean us to publish; "
          "(panel / Schur: the wait is for the own front's pivot / panel workgroups)")
    for l in range(int(f.info("factor_top_level")), P.nlevels):
        m_ = (lev[front] == l) & (role < 3)
        parts = []
        for rr, nm in ((0, "pivot"), (1, "panel"), (2, "schur")):
            k_ = m_ & (role == rr) & (tt[:, 1] > 0) & (tt[:, 2] > 0)
            if k_.any():
                parts.append(f"{nm} {int(k_.sum()):4d}: wait {np.mean(tt[k_, 1] - tt[k_, 0]):6.1f}  work {np.mean(tt[k_, 2] - tt[k_, 1]):5.1f}  publish {np.mean(tt[k_, 3] - tt[k_, 2]):4.1f}")
        print(f"  level {l:2d}  " + "   ".join(parts))
    names = ["pivot", "panel", "schur", "spanl"]
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
    for i in crit[-60:]:
        extra = ""
        print(f"{i:5d} {names[role[i]]:5s} f{front[i]:4d}  start {tt[i, 0]:8.2f}  prewait {tt[i, 5]:8.2f}  waited {tt[i, 1]:8.2f}"
              f"  done {tt[i, 2]:8.2f}  published {tt[i, 3]:8.2f}{extra}")
        if role[i] == 0 and i >= crit[-16]:
            base = t[:, 0][t[:, 0] > 0].min()
            st = [(x - base) / 100.0 for x in piv[i] if x > 0]
            print("        pivot detail (gathered, first diagonal block, then per step: block column done, step done): "
                  + " ".join(f"{x:.2f}" for x in st))
            print("        shader-clock ticks of the diagonal block of step 0:", fine[i][39])
            print("        shader clocks of the fourth diagonal block: operands loaded, steps 0-7, steps 8-14, stored + checked:", list(fine[i][32:36]))
            fs = [(x - base) / 100.0 for x in fine[i][:32] if x > 0]
            print("        per step: wave 0 next diagonal tile updated, diagonal block done; wave 1 trailing tiles done, inverse row done: "
                  + " ".join(f"{x:.2f}" for x in fs))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "run"]()

