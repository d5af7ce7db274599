"""In-kernel timeline of the fused solve launch (k_solve_tree), wall_clock64 stamps per workgroup:
entry, panel entries requested, dependency wait over, posted.

    python scripts/timeline_tree.py build     # here
    python scripts/timeline_tree.py run [workload]   # on the GPU box (gpurun)
"""
import ctypes as C
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, "sleqp_amd", "_timeline_build")
SRC = os.path.join(ROOT, "sleqp_amd", "csrc")
NB = 4096


def inline_parts(path):
    """The kernels live in kernels_*.inc files included by kernels_solve.hip: expand them in the scratch copy, so that the
    patches below see one text."""
    import re

    d = os.path.dirname(path)
    text = open(path).read()
    text = re.sub(r'^#include "(kernels_\w+\.inc)"$', lambda m: open(os.path.join(d, m.group(1))).read(), text, flags=re.M)
    open(path, "w").write(text)


def build():
    shutil.rmtree(SCRATCH, ignore_errors=True)
    shutil.copytree(SRC, SCRATCH, ignore=shutil.ignore_patterns("*.so", "*.o"))
    p = os.path.join(SCRATCH, "kernels_solve.hip")
    inline_parts(p)
    s = open(p).read()
    s = s.replace("typedef double d4_t __attribute__((ext_vector_type(4)));",
                  f"__device__ long long g_st[{NB} * 8];\n"
                  f"#define TRS(slot) if (threadIdx.x == 0) g_st[blockIdx.x * 8 + (slot)] = wall_clock64()\n"
                  "typedef double d4_t __attribute__((ext_vector_type(4)));", 1)
    a = s.index("__device__ __forceinline__ void dev_solve_fwd(")
    b = s.index("void k_solve_tree(", a)  # (the launch itself: both sweeps' device functions sit in front of it)
    seg = s[a:b]
    # forward
    seg = seg.replace("  // staged row tid (front row fr): own right-hand side and, per child, which of its update rows lands here", "  TRS(1);\n  // staged row tid (front row fr): own right-hand side and, per child, which of its update rows lands here", 1)
    seg = seg.replace("  if (tid < rl) f[tid] = f0;\n  __syncthreads();", "  if (tid < rl) f[tid] = f0;\n  __syncthreads();\n  TRS(2);", 1)
    seg = seg.replace("      post_f64(uvec + T.uoff + a0 + j, f[w + j] + s2);\n    }\n  }\n}", "      post_f64(uvec + T.uoff + a0 + j, f[w + j] + s2);\n    }\n  }\n  TRS(3);\n}", 1)
    # backward
    seg = seg.replace("  const int myrow = (tid >= top && tid < ro) ? rows[T.rowoff + w + a0 + (tid - top)] : -1;", "  TRS(1);\n  const int myrow = (tid >= top && tid < ro) ? rows[T.rowoff + w + a0 + (tid - top)] : -1;", 1)
    seg = seg.replace("    if (tid >= top && tid < ro) sent_f64_agent(uvec + T.uoff + a0 + (tid - top));\n  }\n  __syncthreads();", "    if (tid >= top && tid < ro) sent_f64_agent(uvec + T.uoff + a0 + (tid - top));\n  }\n  __syncthreads();\n  TRS(2);", 1)
    seg = seg.replace("      y[T.c0 + tid] = s2;\n      post_f64(ysol + T.c0 + tid, s2);\n    }\n  }\n", "      y[T.c0 + tid] = s2;\n      post_f64(ysol + T.c0 + tid, s2);\n    }\n  }\n  TRS(3);\n", 1)
    assert seg.count("TRS(") == 6, seg.count("TRS(")
    s = s[:a] + seg + s[b:]
    assert s.count("  const int par = *epoch & 1;") >= 1
    s = s.replace("  const int par = *epoch & 1;", "  TRS(0);\n  const int par = *epoch & 1;", 1)
    open(p, "w").write(s)
    h = os.path.join(SCRATCH, "kernels_solve.hip")
    t = open(h).read()
    t += ('\nextern "C" int hipfact_debug_trace_tree(long long* out) {\n'
          f"  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_st), sizeof(long long) * {NB} * 8);\n}}\n")
    open(h, "w").write(t)
    subprocess.check_call(["make", "-C", SCRATCH])
    print("built", os.path.join(SCRATCH, "libhipfact.so"))


def run():
    import numpy as np

    os.environ["HIPFACT_LIBRARY"] = os.path.join(SCRATCH, "libhipfact.so")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from bench import make_problem
    from plan_emul import Plan
    from sleqp_amd import _lib
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem(sys.argv[2] if len(sys.argv) > 2 else "banded_n1e5_m5e4", 0)
    f = HipFact(device=0)
    f.set_option("use_graph", 0)
    f.set_option("refine_steps", 0)
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    for _ in range(3):
        f.solve(b)
        f.solution_raw(0, 1)
    lib = _lib.load()
    out = np.zeros(NB * 8, dtype=np.int64)
    lib.hipfact_debug_trace_tree.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_tree(out.ctypes.data_as(C.c_void_p)) == 0
    nf = int(f.info("solve_items"))
    t = out.reshape(NB, 8)[: 2 * nf, :4].astype(np.float64)
    t = (t - t[:, 0].min()) / 100.0  # 100 MHz
    P = Plan(lib, N, cp, ri, vx)
    # level of the item at each forward position (a sliced front has several items): from the items themselves
    item_dt = np.dtype([("spf", "<i8"), ("spb", "<i8"), ("uoff", "<i8"), ("rowoff", "<i8"), ("c0", "<i4"), ("w", "<i4"), ("r", "<i4"),
                        ("nchild", "<i4"), ("Qf", "<i4"), ("Ef", "<i4"), ("Pb", "<i4"), ("Eb", "<i4"), ("c_uoff", "<i8", 4),
                        ("c_invoff", "<i4", 4), ("Loff", "<i8"), ("xbegin", "<i4"), ("xend", "<i4"), ("a0", "<i4"), ("a1", "<i4"),
                        ("sl", "<i4"), ("nsl", "<i4"), ("poff", "<i8"), ("plevel", "<i4"), ("pad_", "<i4")])
    raw = np.empty(2 * nf * item_dt.itemsize, dtype=np.uint8)  # forward order, then backward order
    lib.hipfact_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    assert lib.hipfact_debug_copy(f._h, b"sitems", raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    items = raw.view(item_dt)
    front_of_c0 = {int(P.sn_c0[s_]): s_ for s_ in range(P.nsuper)}
    lev = np.array([int(P.sn_level[front_of_c0[int(c)]]) for c in items["c0"]])
    print(f"# {P.nsuper} fronts, {nf} items ({int((items['nsl'] > 1).sum())} of them row slices)")
    print("# fused solve launch, us since the first workgroup started.  per level: workgroups, entry (min..max), waited (max), posted (max)")
    for name, sl, levels in (("forward", slice(0, nf), lev[:nf]), ("backward", slice(nf, 2 * nf), lev[nf:])):
        tt = t[sl]
        print(name)
        for l in (range(P.nlevels) if name == "forward" else range(P.nlevels - 1, -1, -1)):
            m = levels == l
            print(f"  level {l:2d}  fronts {int(m.sum()):4d}  entry {tt[m, 0].min():7.2f} .. {tt[m, 0].max():7.2f}  requested {tt[m, 1].max():7.2f}"
                  f"  waited {tt[m, 2].max():7.2f}  posted {tt[m, 3].max():7.2f}")
    print(f"# launch span {t[:, 3].max():.2f} us; forward done {t[:nf, 3].max():.2f}")
    # the two bottom levels in detail: quartiles of the stamps and of the time an item stays resident
    for name, sl, levels in (("forward", slice(0, nf), lev[:nf]), ("backward", slice(nf, 2 * nf), lev[nf:])):
        tt = t[sl]
        for l in (0, 1):
            m = levels == l
            q = lambda a: " ".join(f"{v:6.2f}" for v in np.percentile(a, [0, 25, 50, 75, 100]))
            print(f"# {name} level {l}: entry [{q(tt[m, 0])}]  requested-entry [{q(tt[m, 1] - tt[m, 0])}]  waited-requested [{q(tt[m, 2] - tt[m, 1])}]"
                  f"  posted-waited [{q(tt[m, 3] - tt[m, 2])}]  resident [{q(tt[m, 3] - tt[m, 0])}]  late entries (> 2 us) {int((tt[m, 0] > tt[m, 0].min() + 2).sum())}")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "run"]()
