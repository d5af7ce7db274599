"""Probe: the graded ill-conditioned family through hipfact_set_matrix with / without the row dictionary."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import util as T
import oracle
from sleqp_amd.fact import HipFact
from sleqp_amd.sparse import SleqpMat
from util import reference_solution, rel_err
fam = {name: (J, vi, ci) for name, J, vi, ci in T.graded_family()}
def run(f, name, tag):
    J, vi, ci = fam[name]
    m, n = J.shape
    N, kc, kr, kd = oracle.fill_aug_jac(n, m, J.indptr, J.indices, J.data, vi, ci)
    b = np.random.default_rng(1).standard_normal(N)
    f.set_matrix(SleqpMat(N, N, kc, kr, kd))
    p0 = f.info("num_passes")
    f.solve(b)
    z = f.solution_raw(0, N)
    ref = oracle.OracleFact(N, kc, kr, kd); ref.solve_dense(b)
    zo = ref.raw_solution()
    from sleqp_amd import synth
    zr = reference_solution(synth.kkt_full_matrix(N, kc, kr, kd).toarray(), b, ref)
    print(tag, name, "err vs ref %.2e (oracle %.2e)" % (rel_err(z, zr), rel_err(zo, zr)), "passes", f.info("num_passes") - p0, "omega %.1e" % f.info("last_omega"),
          "status", f.info("last_status"), "iters", f.info("last_iters"), "mstruct", f.info("m_struct"), "W", N - n)
f = HipFact(device=0); run(f, "band_parallel_1e-05", "fresh vt=1"); f.free()
f = HipFact(device=0); f.set_option("superset_vtable", 0); run(f, "band_parallel_1e-05", "fresh vt=0"); f.free()
f = HipFact(device=0)
for nm in ("band_parallel_1e-02", "band_parallel_1e-03", "band_parallel_1e-04", "band_parallel_1e-05"):
    run(f, nm, "seq vt=1")
f.free()
