"""Streaming CSR SpMV (k_spmv_stream) against the lanes-per-row kernel: same results (to rounding: another summation
order), time on the matrix of bench.py's spmv.large and on the workload's Jacobian."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import scipy.sparse as sp
from bench import make_problem, spmv_bytes
from sleqp_amd.fact import HipFact, SpMat
from sleqp_amd.sparse import SleqpMat

fact = HipFact(device=0)
fact.set_option("spmv_stream_min", 0)
dev = "cuda:0"
def run(M, Msp, tag):
    r, c = Msp.shape
    x = torch.randn(max(r, c), dtype=torch.float64, device=dev)
    y = torch.empty(max(r, c), dtype=torch.float64, device=dev)
    xh = x.cpu().numpy()
    for trans, name, ref in ((0, "M x", Msp @ xh[:c]), (1, "M^T x", Msp.T @ xh[:r])):
        out = {}
        for stream in (0, 1):
            fact.set_option("spmv_stream", stream)
            for _ in range(3):
                M.mult_device(trans, x.data_ptr(), y.data_ptr())
            fact.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                M.mult_device(trans, x.data_ptr(), y.data_ptr())
            fact.synchronize()
            dt = (time.perf_counter() - t0) / 20
            yh = y.cpu().numpy()[:len(ref)]
            err = np.abs(yh - ref).max() / max(1.0, np.abs(ref).max())
            by = spmv_bytes(len(ref), (c if trans == 0 else r), Msp.nnz)
            out[stream] = (dt * 1e6, by / dt / 1e9, err)
        print(f"{tag:14s} {name:6s}: lanes-per-row {out[0][0]:8.1f} us {out[0][1]:7.0f} GB/s err {out[0][2]:.1e} | stream {out[1][0]:8.1f} us {out[1][1]:7.0f} GB/s err {out[1][2]:.1e}", flush=True)

J, N, cp, ri, vx, b = make_problem("banded_n1e5_m5e4", 0)
run(SpMat(fact, SleqpMat.from_scipy(J)), sp.csc_matrix(J), "J (1e6 nnz)")
# ragged: rows of 0 .. 300 entries, a few long rows
rng = np.random.default_rng(1)
R = sp.random(5000, 4000, density=0.01, random_state=3, format="lil")
R[7, :] = 1.0
R[4000:4003, ::2] = 2.0
R = sp.csc_matrix(R); R.sort_indices()
run(SpMat(fact, SleqpMat(5000, 4000, R.indptr, R.indices, R.data)), R, "ragged")
n = 1 << 21
offs = np.sort(rng.choice(np.arange(-(1 << 15), 1 << 15), 40, replace=False)).astype(np.int64)
cols = np.arange(n, dtype=np.int64)
rows = (cols[:, None] + offs[None, :]) % n
rows.sort(axis=1)
cpl = (np.arange(n + 1, dtype=np.int64) * 40).astype(np.int32)
ril = rows.reshape(-1).astype(np.int32)
vxl = rng.standard_normal(ril.size)
Ml = sp.csc_matrix((vxl, ril, cpl), shape=(n, n))
run(SpMat(fact, SleqpMat(n, n, cpl, ril, vxl)), Ml, "large (84M)")
