"""Per-level front statistics of the symbolic plan (counts, widths, update sizes) for a bench workload."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from plan_emul import Plan  # noqa: E402
from sleqp_amd import _lib, synth  # noqa: E402

n, m, per_row = (int(float(a)) for a in (sys.argv[1:4] if len(sys.argv) > 3 else ("1e5", "5e4", "20")))
kind = sys.argv[4] if len(sys.argv) > 4 else "banded"
J = synth.banded_jacobian(n, m, per_row, 200, 0) if kind == "banded" else synth.uniform_jacobian(n, m, per_row, 0)
vi, ci, _ = synth.working_set_all_rows(n, m)
N, cp, ri, vx = synth.kkt_lower_from_jacobian(J, vi, ci)
p = Plan(_lib.load(), N, cp, ri)
w = np.diff(p.sn_c0).astype(np.int64) if len(p.sn_c0) == p.nsuper + 1 else None
r = p.sn_r.astype(np.int64)
lev = p.sn_level
print(f"nsuper {p.nsuper} levels {p.nlevels} L_size {p.L_size} U_size {p.U_size}")
for l in range(p.nlevels):
    s = np.where(lev == l)[0]
    u = r[s] - w[s]
    fl = (w[s] ** 3 / 3 + w[s] ** 2 * u + w[s] * u * u).sum()
    print(f"level {l:2d} fronts {len(s):4d}  w mean {w[s].mean():6.1f} max {w[s].max():4d}  u mean {u.mean():7.1f} max {u.max():5d}"
          f"  r max {r[s].max():5d}  flops {fl:.2e}")
nch = np.diff(p.child_ptr)
for l in range(p.nlevels):
    s = np.where(lev == l)[0]
    print(f"level {l:2d} children per front: mean {nch[s].mean():5.2f} max {nch[s].max():3d}  hist {np.bincount(nch[s], minlength=6)[:8]}")
print(f"nM {len(p.Mi)} nprod {p.nprod} avg products per entry {p.nprod / max(len(p.Mi), 1):.2f} nnzK {p.nnzK}")
pl = np.diff(p.prod_ptr)
print("products per entry: hist", np.bincount(np.minimum(pl, 12)))
