"""In-kernel timeline of the single-launch solve sweeps (k_fwd_top / k_bwd_top), wall_clock64 stamps.

Same mechanism as scripts/timeline.py: an instrumented copy of the sources is built in a scratch
directory; per workgroup the stamps are kernel entry, begin / end of its last dependency wait,
work done, published.

    python scripts/timeline_solve.py build     # here
    python scripts/timeline_solve.py run [workload]      # on the GPU box (gpurun); default banded_n1e5_m5e4
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
                  f"__device__ long long g_st[2 * {NB} * 8];\n__device__ int g_kid;\n"
                  f"#define TRS(slot) if (threadIdx.x == 0) g_st[(g_kid * {NB} + blockIdx.x) * 8 + (slot)] = wall_clock64()\n"
                  "typedef double d4_t __attribute__((ext_vector_type(4)));", 1)
    # kernel entries
    a = s.index("__global__ __launch_bounds__(SB) void k_fwd_top(")
    a = s.index("  extern __shared__", a)
    s = s[:a] + "  if (threadIdx.x == 0) g_kid = 0;\n  TRS(0);\n" + s[a:]
    a = s.index("__global__ __launch_bounds__(SB) void k_bwd_top(")
    a = s.index("  extern __shared__", a)
    s = s[:a] + "  if (threadIdx.x == 0) g_kid = 1;\n  TRS(0);\n" + s[a:]
    # waits and publishes
    a = s.index("__device__ __forceinline__ void top_wait(int* __restrict__ flags, int who, int* __restrict__ info, int target) {")
    a = s.index("  if (threadIdx.x == 0) {", a)
    s = s[:a] + "  TRS(1);\n" + s[a:]
    a = s.index("  __syncthreads();\n}\n\n__device__ __forceinline__ void top_publish_add(", a)
    s = s[:a] + "  TRS(2);\n" + s[a:]
    for name in ("top_publish_add", "top_publish"):
        a = s.index(f"__device__ __forceinline__ void {name}(int* __restrict__ flags, int who) {{")
        a = s.index("  asm volatile", a)
        s = s[:a] + "  TRS(3);\n" + s[a:]
        a = s.index("  }\n}\n", a)
        s = s[:a] + "    g_st[(g_kid * %d + blockIdx.x) * 8 + 4] = wall_clock64();\n" % NB + s[a:]
    assert s.count("TRS(") >= 6
    open(p, "w").write(s)
    h = os.path.join(SCRATCH, "kernels_solve.hip")
    t = open(h).read()
    t += ('\nextern "C" int hipfact_debug_trace_solve(long long* out) {\n'
          f"  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_st), sizeof(long long) * 2 * {NB} * 8);\n}}\n")
    open(h, "w").write(t)
    subprocess.check_call(["make", "-C", SCRATCH])
    print("built", os.path.join(SCRATCH, "libhipfact.so"))


def run():
    import numpy as np
    import torch

    os.environ["HIPFACT_LIBRARY"] = os.path.join(SCRATCH, "libhipfact.so")
    sys.path.insert(0, ROOT)
    from bench import make_problem
    from sleqp_amd import _lib
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem(sys.argv[2] if len(sys.argv) > 2 else "banded_n1e5_m5e4", 0)
    f = HipFact(device=0)
    f.set_option("use_graph", 0)
    f.set_option("solve_fused", 0)  # the two-launch sweeps are what this script instruments
    f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    dev = torch.device("cuda", 0)
    d_rhs = torch.from_numpy(b).to(dev)
    d_sol = torch.empty_like(d_rhs)
    for _ in range(3):
        f.solve_device(d_rhs.data_ptr(), d_sol.data_ptr())
    f.synchronize()
    lib = _lib.load()
    out = np.zeros(2 * NB * 8, dtype=np.int64)
    lib.hipfact_debug_trace_solve.argtypes = [C.c_void_p]
    assert lib.hipfact_debug_trace_solve(out.ctypes.data_as(C.c_void_p)) == 0
    n = min(int(f.info("top_count")), NB)
    t = out.reshape(2, NB, 8)
    for kid, name in ((0, "forward (last workgroups = top of the tree)"), (1, "backward (first workgroups = top of the tree)")):
        tt = t[kid, :n, :5].astype(np.float64)
        base = tt[:, 0][tt[:, 0] > 0].min()
        tt = (tt - base) / 100.0
        print(f"# {name}: us since the first workgroup started: entry, last wait begin, last wait end, work done, published")
        rng = range(max(0, n - 40), n) if kid == 0 else range(0, min(n, 40))
        for i in rng:
            print(f"{i:5d}  entry {tt[i, 0]:8.2f}  wait {tt[i, 1]:8.2f} .. {tt[i, 2]:8.2f}  done {tt[i, 3]:8.2f}  published {tt[i, 4]:8.2f}")
        print(f"# kernel span (first entry to last publish): {tt[:, 4].max():.2f} us")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "run"]()
