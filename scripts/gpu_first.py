"""First GPU bring-up: factor + solve small/medium KKT systems, compare with scipy."""
import sys, time, os
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sleqp_amd import synth
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat, SleqpVec

def run(n, m, kind, frac=0.0, seed=1, refine=1, **opts):
    J = synth.banded_jacobian(n, m, min(20, max(n // 4, 1)), min(200, n), seed) if kind == 'b' else synth.uniform_jacobian(n, m, min(5, n), seed)
    vi, ci, W = synth.working_set_all_rows(n, m, frac, seed)
    N, cp, ri, vx = synth.kkt_lower_from_jacobian(J, vi, ci)
    K = synth.kkt_full_matrix(N, cp, ri, vx)
    f = HipFact(refine_steps=refine, **opts)
    t = time.time(); f.set_matrix(SleqpMat(N, N, cp, ri, vx)); t1 = time.time() - t
    b = np.random.default_rng(seed).standard_normal(N)
    t = time.time(); f.solve(b); z = f.solution_raw(0, N); t2 = time.time() - t
    res = np.abs(K @ z - b).max()
    if N <= 20000:
        zr = spla.spsolve(K.tocsc(), b); err = np.abs(z - zr).max() / np.abs(zr).max()
    else:
        err = float('nan')
    print(f"n={n} m={m} {kind} frac={frac} N={N} saddle={f.info('saddle')} nsuper={f.info('nsuper')} nlev={f.info('nlevels')} set_matrix={t1*1e3:.1f}ms solve={t2*1e3:.2f}ms resid={res:.2e} err={err:.2e} cond={f.cond():.3g}", flush=True)
    return res

ok = True
for args in [(2,1,'u'), (4,2,'u',0.5), (40,20,'b'), (40,20,'u',0.2), (300,150,'b',0.1), (1000,500,'u'), (2000,1000,'b',0.05), (10000,5000,'b')]:
    for refine in (0, 1):
        r = run(*args, refine=refine)
        ok &= bool(r < 1e-8)
print("generic mode:")
# generic: SPD matrix
rng = np.random.default_rng(0)
B = sp.random(500, 500, density=0.01, random_state=0, format='csc'); M = (B @ B.T + sp.eye(500)*5).tocsc()
L = sp.tril(M, format='csc'); L.sort_indices()
f = HipFact(refine_steps=1)
f.set_matrix(SleqpMat(500, 500, L.indptr, L.indices, L.data))
b = rng.standard_normal(500); f.solve(b); z = f.solution_raw(0, 500)
print("generic saddle=", f.info('saddle'), "resid", np.abs(M @ z - b).max()); ok &= bool(np.abs(M@z-b).max() < 1e-8)
print("ALL OK" if ok else "FAILURES")
