"""Debug: compare the solve panels and the fused solve against numpy on a small problem (GPU)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sleqp_amd import synth, _lib
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from plan_emul import Plan
lib = _lib.load()
n, m = (int(a) for a in sys.argv[1:3]) if len(sys.argv) > 2 else (60, 30)
J = synth.banded_jacobian(n, m, 6, 30, 0)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J)
fact = HipFact(device=0)
fact.set_option("equilibrate", 0)
fact.set_option("refine_steps", 0)
fact.set_matrix(SleqpMat(N, N, cp, ri, vx))
print("fused", fact.info("fused_solve"), "nsuper", fact.info("nsuper"))
P = Plan(lib, N, cp, ri, vx)
def dcopy(name, count, dtype=np.float64):
    out = np.empty(count, dtype=dtype)
    rc = lib.hipfact_debug_copy(fact._h, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes)
    assert rc == 0, lib.hipfact_last_error(fact._h)
    return out
L = dcopy("L", P.L_size)
spbytes = int(fact.info("solve_panel_bytes"))
# item layout: 2 ll, 2 ll, 4 int, 4 int, 4 ll, 4 int, 1 ll  = 16+16+16+16+32+16+8 = 120 bytes
item_dt = np.dtype([("spf", "<i8"), ("spb", "<i8"), ("uoff", "<i8"), ("rowoff", "<i8"), ("c0", "<i4"), ("w", "<i4"), ("r", "<i4"),
                    ("nchild", "<i4"), ("Qf", "<i4"), ("Ef", "<i4"), ("Pb", "<i4"), ("Eb", "<i4"), ("c_uoff", "<i8", 4),
                    ("c_invoff", "<i4", 4), ("Loff", "<i8"), ("xbegin", "<i4"), ("xend", "<i4"), ("a0", "<i4"), ("a1", "<i4"),
                    ("sl", "<i4"), ("nsl", "<i4"), ("poff", "<i8")])
print("itemsize", item_dt.itemsize)
items = dcopy("sitems", int(fact.info("solve_items")) * item_dt.itemsize // 8).view(item_dt)
def rows_of(it):
    return (int(it["w"]) if int(it["sl"]) == 0 else 0) + int(it["a1"]) - int(it["a0"])
SPf = dcopy("SPf", int(sum(((int(it["Ef"]) * rows_of(it) * int(it["Qf"]) + 1) & ~1) for it in items)))
SPb = dcopy("SPb", int(sum(((int(it["Eb"]) * int(it["w"]) * int(it["Pb"]) + 1) & ~1) for it in items)))
worst = 0
for it in items:
    w, r = int(it["w"]), int(it["r"])
    panel = L[it["Loff"]: it["Loff"] + r * w].reshape((w, r)).T
    X = np.tril(panel[:w, :w], -1) + np.eye(w)
    d = np.diag(panel[:w, :w])
    S = np.vstack([X, -(panel[w:, :w] @ X)])
    Sb = S.copy(); Sb[:w] /= d[:, None]
    # the item's rows: the pivot rows (slice 0 only) and the update rows [a0, a1)
    sel = (list(range(w)) if int(it["sl"]) == 0 else []) + [w + a for a in range(int(it["a0"]), int(it["a1"]))]
    ro = len(sel)
    Q, E = int(it["Qf"]), int(it["Ef"])
    TS = ro * Q
    got = np.zeros((ro, w))
    for k in range(w):
        got[:, k] = SPf[it["spf"] + (k // Q) * TS + (k % Q) * ro: it["spf"] + (k // Q) * TS + (k % Q) * ro + ro]
    e1 = np.abs(got - S[sel]).max()
    Pb, Eb = int(it["Pb"]), int(it["Eb"])
    TSb = w * Pb
    gotb = np.zeros((ro, w))
    for i in range(ro):
        gotb[i, :] = SPb[it["spb"] + (i // Pb) * TSb + (i % Pb) * w: it["spb"] + (i // Pb) * TSb + (i % Pb) * w + w]
    e2 = np.abs(gotb - Sb[sel]).max()
    worst = max(worst, e1, e2)
print("panel max err", worst)
b = np.random.default_rng(0).standard_normal(N)
fact.solve(b)
xh = dcopy("xhat", P.m); ys = dcopy("ysol", P.m); uv = dcopy("uvec", P.u_size); yy = dcopy("y", P.m)
def sent(a):
    return a.view(np.uint64) == 0xFFFFFFFFFFFFFFFF
print("after launch: xhat sentinel", sent(xh).sum(), "ysol sentinel", sent(ys).sum(), "uvec sentinel", sent(uv).sum(), "of", P.m, P.m, P.u_size)
for pos, it in enumerate(items):
    c0, w, r = int(it["c0"]), int(it["w"]), int(it["r"])
    print(pos, "c0", c0, "w", w, "r", r, "nch", int(it["nchild"]), "Qf", int(it["Qf"]), "Ef", int(it["Ef"]), "Pb", int(it["Pb"]), "Eb", int(it["Eb"]),
          "xhat_sent", int(sent(xh[c0:c0 + w]).sum()), "ysol_sent", int(sent(ys[c0:c0 + w]).sum()),
          "uvec_sent", int(sent(uv[it["uoff"]: it["uoff"] + r - w]).sum()))
try:
    z = fact.solution_raw(0, N)
    K = synth.kkt_full_matrix(N, cp, ri, vx)
    print("resid", np.abs(K @ z - b).max())
except Exception as e:
    print("solve failed:", e)
