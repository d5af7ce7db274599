"""In-kernel timeline of the fused solve launch WITH the top block: wall_clock64 stamps per workgroup.
    python scripts/timeline_top.py build   (here)      python scripts/timeline_top.py run   (GPU box)"""
import ctypes as C
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRATCH = os.path.join(ROOT, "sleqp_amd", "_timeline_build")
SRC = os.path.join(ROOT, "sleqp_amd", "csrc")
NB = 4096


def build():
    import re

    shutil.rmtree(SCRATCH, ignore_errors=True)
    shutil.copytree(SRC, SCRATCH, ignore=shutil.ignore_patterns("*.so", "*.o"))
    p = os.path.join(SCRATCH, "kernels_solve.hip")
    text = open(p).read()
    text = re.sub(r'^#include "(kernels_\w+\.inc)"$', lambda m: open(os.path.join(SCRATCH, m.group(1))).read(), text, flags=re.M)
    s = text
    s = s.replace("typedef double d4_t __attribute__((ext_vector_type(4)));",
                  f"__device__ long long g_st[{NB} * 8];\n"
                  f"#define TRS(slot) if (threadIdx.x == 0) g_st[blockIdx.x * 8 + (slot)] = wall_clock64()\n"
                  "typedef double d4_t __attribute__((ext_vector_type(4)));", 1)
    def patch(text, reps):
        for r in reps:
            assert text.count(r[0]) == 1, (r[0], text.count(r[0]))
            if len(r) == 2:
                text = text.replace(r[0], r[1] + r[0], 1)
            elif r[2] == "TAIL":
                text = text.replace(r[0], r[0][:-2] + "  TRS(3);\n}", 1)
            else:
                text = text.replace(r[0], r[0] + r[2], 1)
        return text

    def segment(text, start, end, reps):
        a, b = text.index(start), text.index(end)
        return text[:a] + patch(text[a:b], reps) + text[b:]

    s = segment(s, "__device__ __forceinline__ void dev_solve_fwd(", "__device__ __forceinline__ void dev_solve_bwd(", [
        ("  // staged row tid (front row fr): own right-hand side and, per child, which of its update rows lands here", "  TRS(1);\n"),
        ("  if (tid < rl) f[tid] = f0;\n  __syncthreads();", None, "\n  TRS(2);"),
        ("      post_f64(uvec + T.uoff + a0 + j, f[w + j] + s2);\n    }\n  }\n}", None, "TAIL")])
    s = segment(s, "__device__ __forceinline__ void dev_solve_bwd(", "// ---- top block (device_types.h: TopBlockIn)", [
        ("  const int myrow = (tid >= top && tid < ro) ? rows[T.rowoff + w + a0 + (tid - top)] : -1;", "  TRS(1);\n"),
        ("    if (tid >= top && tid < ro) sent_f64_agent(uvec + T.uoff + a0 + (tid - top));\n  }\n  __syncthreads();", None, "\n  TRS(2);"),
        ("      y[T.c0 + tid] = s2;\n      post_f64(ysol + T.c0 + tid, s2);\n    }\n  }\n", None, "\n  TRS(3);\n")])
    s = patch(s, [("  const int par = *epoch & 1;\n  if ((int)blockIdx.x > nall) {", "  TRS(0);\n")])
    # the top block as one product (dev_top_one): entry | prefetch issued | lists in LDS | gathered | posted
    s = segment(s, "__device__ __forceinline__ void dev_top_one(", "__global__ __launch_bounds__(FB) void k_top_dinv(", [
        ("  if (tid < nr) {\n    sent_f64(ysol_prev + B.tpos[I.r0 + tid]);", "  TRS(1);\n"),
        ("  constexpr int GK = 6;", "  TRS(4);\n"),
        ("  __syncthreads();\n  double acc = 0.0;", None, "\n  TRS(2);"),
        ("      y[k] = s2;\n      post_f64(ysol + k, s2);\n    }\n  }\n", None, "  TRS(3);\n")])
    s = s.replace("  __syncthreads();\n  double acc = 0.0;\n  TRS(2);", "  __syncthreads();\n  TRS(2);\n  double acc = 0.0;")
    open(p, "w").write(s)
    h = os.path.join(SCRATCH, "kernels_solve.hip")
    t = open(h).read()
    t += ('\nextern "C" int hipfact_debug_trace_tree(long long* out) {\n'
          f"  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_st), sizeof(long long) * {NB} * 8);\n}}\n")
    open(h, "w").write(t)
    subprocess.check_call(["make", "-C", SCRATCH])


def run():
    import numpy as np

    os.environ["HIPFACT_LIBRARY"] = os.path.join(SCRATCH, "libhipfact.so")
    sys.path.insert(0, ROOT)
    from bench import make_problem
    from sleqp_amd import _lib
    from sleqp_amd.fact import HipFact
    from sleqp_amd.sparse import SleqpMat

    J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
    lib = _lib.load()
    lib.hipfact_debug_trace_tree.argtypes = [C.c_void_p]
    for after, cap in ((0, 2048), (2, 2048)):
        f = HipFact(device=0)
        f.set_option("use_graph", 0)
        f.set_option("refine_steps", 0)
        f.set_option("top_block_after", after)
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
        for _ in range(5):
            f.solve(b)
            f.solution_raw(0, 1)
        out = np.zeros(NB * 8, dtype=np.int64)
        assert lib.hipfact_debug_trace_tree(out.ctypes.data_as(C.c_void_p)) == 0
        nit = int(f.info("solve_items"))
        nT, ntf = int(f.info("top_block_cols")), int(f.info("top_block_items"))
        nfb = int(f.info("top_block_below")) if nT else nit
        ntr = (nT + 63) // 64 if nT else 0
        t = out.reshape(NB, 8).astype(np.float64)
        base = t[ntr:ntr + nfb, 0].min()  # (first forward item of this launch; slots of other blocks may hold older stamps)
        t = (t - base) / 100.0
        fw = t[ntr:ntr + nfb]
        single = nT > 0
        ntb = 0 if single else ntf
        tf = t[ntr + nfb:ntr + nfb + ntf]
        bw = t[ntr + nfb + ntf + ntb:ntr + 2 * nfb + ntf + ntb]
        print(f"== top_block_after {after} cap {cap}: nT {nT}, {ntf} top items, single product {single}, {nfb} items below, rhs items {ntr}")
        print(f"forward below: first entry {fw[:, 0].min():7.2f} last posted {fw[:, 3].max():7.2f}   (last 12 items posted: {np.sort(fw[:, 3])[-12:].round(1)})")
        if ntf and single:
            q = lambda a: " ".join(f"{v:7.2f}" for v in np.percentile(a, [0, 50, 100]))
            print(f"top items: entry [{q(tf[:, 0])}] prefetch issued [{q(tf[:, 1])}] lists in LDS [{q(tf[:, 4])}] gathered [{q(tf[:, 2])}] posted [{q(tf[:, 3])}]")
        print(f"backward below: first waited {bw[:, 2].min():7.2f} first posted {bw[:, 3].min():7.2f} last posted {bw[:, 3].max():7.2f}")
        print(f"(first 12 backward items: waited {bw[:12, 2].round(1)} posted {bw[:12, 3].round(1)})")
        f.free()


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1] if len(sys.argv) > 1 else "run"]()
