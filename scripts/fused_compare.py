"""Debug: factor arena (L) and update arena (U) of an experimental schedule against the reference schedule, front by front."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from bench import make_problem
from sleqp_amd import _lib
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from plan_emul import Plan
lib = _lib.load()
lib.hipfact_debug_copy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
J, N, cp, ri, vx, b = make_problem(os.environ.get("WORKLOAD", "banded_n1e5_m5e4"), 0)
P = Plan(lib, N, cp, ri, vx)
def run(opts):
    f = HipFact(device=0)
    f.set_option("refine_steps", 0)
    for kv in opts.split():
        f.set_option(kv.split("=")[0], float(kv.split("=")[1]))
    try:
        f.set_matrix(SleqpMat(N, N, cp, ri, vx))
    except Exception as e:  # (a schedule under test may leave a factor the check refuses: still compared)
        print('   set_matrix:', str(e)[:120])
    f.synchronize()
    L = np.empty(P.L_size, dtype=np.float64)
    assert lib.hipfact_debug_copy(f._h, b"L", L.ctypes.data_as(C.c_void_p), L.nbytes) == 0
    nsp = int(f.info("solve_panel_bytes")) // 16  # (both copies have the same size)
    SPf = np.empty(int(f.info("spf_bytes")) // 8, dtype=np.float64)
    SPb = np.empty(int(f.info("spb_bytes")) // 8, dtype=np.float64)
    assert lib.hipfact_debug_copy(f._h, b"SPf", SPf.ctypes.data_as(C.c_void_p), SPf.nbytes) == 0
    assert lib.hipfact_debug_copy(f._h, b"SPb", SPb.ctypes.data_as(C.c_void_p), SPb.nbytes) == 0
    isz = int(f.info("sitem_bytes"))
    raw = np.empty(int(f.info("solve_items")) * isz, dtype=np.uint8)
    assert lib.hipfact_debug_copy(f._h, b"sitems", raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    f.free()
    return L, SPf, SPb, raw.reshape(-1, isz)
ref, rSPf, rSPb, ritems = run(sys.argv[1] if len(sys.argv) > 1 else "factor_top_fused=0")
loff2front = {int(P.sn_Loff[s]): s for s in range(P.nsuper)}
for trial in range(int(os.environ.get("TRIALS", "3"))):
    got, gSPf, gSPb, gitems = run(sys.argv[2] if len(sys.argv) > 2 else "factor_top_fused=64")
    if not np.array_equal(ritems, gitems):
        print('   (solve items differ)')
    # solve panels, item by item (offsets spf / spb lead the item; the front from its panel offset)
    off = gitems[:, :16].copy().view(np.int64).reshape(-1, 2)
    # Loff sits behind: 2 ll, 2 ll, 4 int, 4 int, 4 ll, 4 int -> byte 112
    loffs = gitems[:, 112:120].copy().view(np.int64).ravel()
    order = np.argsort(off[:, 0]); ends_f = np.append(off[order, 0][1:], len(gSPf)); 
    orderb = np.argsort(off[:, 1]); ends_b = np.append(off[orderb, 1][1:], len(gSPb))
    nbad = 0
    for which, A_, B_, col, ordr, ends in (("SPf", rSPf, gSPf, 0, order, ends_f), ("SPb", rSPb, gSPb, 1, orderb, ends_b)):
        for k, it in enumerate(ordr):
            a0, a1 = int(off[it, col]), int(ends[k])
            a, g = A_[a0:a1], B_[a0:a1]
            d = np.abs(a - g); sc = np.abs(a).max() + 1e-300
            if not (np.nanmax(d) / sc < 1e-11) or np.isnan(g).any():
                s_ = loff2front.get(int(loffs[it]), -1)
                nbad += 1
                if nbad <= 12:
                    idx = np.nonzero(~(d <= 1e-11 * sc))[0]
                    print(f"   {which} item {it} front {s_} level {P.sn_level[s_] if s_ >= 0 else -1} w {np.diff(P.sn_c0)[s_]} r {P.sn_r[s_]}: rel {np.nanmax(d) / sc:.1e}, {len(idx)} of {a1 - a0} entries differ, first {idx[:3]} last {idx[-3:]}")
    print(f"trial {trial}: {nbad} solve-panel items differ")
    w = np.diff(P.sn_c0)
    bad = []
    for s in range(P.nsuper):
        r, ws = int(P.sn_r[s]), int(w[s])
        a = ref[P.sn_Loff[s]:P.sn_Loff[s] + r * ws].reshape((ws, r)).T
        g = got[P.sn_Loff[s]:P.sn_Loff[s] + r * ws].reshape((ws, r)).T
        d = np.abs(a - g)
        sc = np.abs(a).max() + 1e-300
        if not (d.max() / sc < 1e-11):
            i, k = np.unravel_index(np.nanargmax(d), d.shape)
            rows_bad = np.nonzero((d > 1e-11 * sc).any(axis=1))[0]
            cols_bad = np.nonzero((d > 1e-11 * sc).any(axis=0))[0]
            bad.append((int(P.sn_level[s]), s, ws, r, d.max() / sc, int(i), int(k), rows_bad.min(), rows_bad.max(), len(rows_bad), cols_bad.min(), cols_bad.max(), len(cols_bad)))
            if len(bad) == 1 and os.environ.get("SHOW"):
                np.set_printoptions(precision=4, linewidth=200)
                i0 = int(rows_bad.min())
                print("ref rows", i0, ":\n", a[i0:i0 + 4, :8], "\ngot:\n", g[i0:i0 + 4, :8], "\ncount of zeros in got panel:", int((g == 0).sum()), "of", g.size, " nan:", int(np.isnan(g).sum()))
    print(f"trial {trial}: {len(bad)} fronts differ")
    for t in sorted(bad)[:12]:
        print("   level %d front %d w %d r %d  rel %.1e at (%d,%d)  bad rows %d..%d (%d)  bad cols %d..%d (%d)" % t)
