"""CPU baseline through Intel MKL PARDISO (test / bench infrastructure, never part of the product path).

The reference's CPU backends for the KKT system are third-party sparse direct solvers behind `SleqpFact`
(MA57 `fact/fact_ma57.c:529-711`, UMFPACK `fact/fact_umfpack.c:119-233`, CHOLMOD on the reduced system
`fact/fact_cholmod.c:87-195`); none of them is installed in the image.  MKL PARDISO with `mtype = -2` (real symmetric
indefinite, supernodal LDL^T with Bunch-Kaufman pivoting inside supernodes, multithreaded) is the nearest solver that
IS in the image (`libmkl_rt.so`, no headers: the published C interface of `pardisoinit` / `pardiso` is bound with
ctypes).  It is driven the way `fact_ma57.c:529-625` drives MA57: analysis + numeric factorisation per `set_matrix`
(phase 12), one forward/backward solve per `solve` (phase 33); `numeric_only` keeps the analysis (phase 22 + 33), which
is what the GPU `value` measures.

K arrives as the lower-triangular CSC arrays of `fill_aug_jac` (standard_aug_jac.c:135-237).  PARDISO wants the UPPER
triangle in CSR with every diagonal entry stored: lower CSC of a symmetric matrix IS its upper CSR, the empty diagonal
of the (2,2) block is added as explicit zeros.
"""
import ctypes as C
import os
import time

import numpy as np

_CANDIDATES = ("libmkl_rt.so.2", "libmkl_rt.so.1", "libmkl_rt.so", "/opt/conda/lib/libmkl_rt.so.2",
               "/opt/conda/lib/libmkl_rt.so.1", "/opt/conda/lib/libmkl_rt.so")


def _load():
    for name in _CANDIDATES:
        try:
            return C.CDLL(name, mode=C.RTLD_GLOBAL), name
        except OSError:
            continue
    return None, None


def upper_csr_with_diagonal(N, cp, ri, vx):
    """Lower CSC (cp, ri, vx) of a symmetric matrix -> upper CSR arrays with all N diagonal entries present."""
    cp = np.asarray(cp, dtype=np.int64)
    ri = np.asarray(ri, dtype=np.int64)
    vx = np.asarray(vx, dtype=np.float64)
    cnt = np.diff(cp)
    first = np.where(cnt > 0, ri[np.minimum(cp[:-1], max(len(ri) - 1, 0))], -1) if len(ri) else np.full(N, -1)
    has_diag = (cnt > 0) & (first == np.arange(N))
    add = (~has_diag).astype(np.int64)
    ia = np.zeros(N + 1, dtype=np.int64)
    np.cumsum(cnt + add, out=ia[1:])
    ja = np.empty(ia[N], dtype=np.int32)
    a = np.empty(ia[N], dtype=np.float64)
    # rows with a stored diagonal are copied as they are; the others get (row, 0.0) in front
    dst0 = ia[:-1] + add
    src_row = np.repeat(np.arange(N), cnt)
    pos = dst0[src_row] + (np.arange(len(ri)) - cp[:-1][src_row])
    ja[pos] = ri
    a[pos] = vx
    miss = np.nonzero(add)[0]
    ja[ia[:-1][miss]] = miss
    a[ia[:-1][miss]] = 0.0
    return ia.astype(np.int32), ja, a


class Pardiso:
    """One PARDISO instance (mtype -2) on a fixed pattern."""

    def __init__(self, N, cp, ri, vx, threads=None):
        lib, name = _load()
        if lib is None:
            raise OSError("libmkl_rt not found")
        self.lib, self.libname = lib, name
        if threads is not None:
            lib.MKL_Set_Num_Threads(C.c_int(int(threads)))
        self.N = int(N)
        self.ia, self.ja, self.a = upper_csr_with_diagonal(N, cp, ri, vx)
        self.pt = (C.c_void_p * 64)()
        self.iparm = (C.c_int * 64)()
        self.mtype = C.c_int(-2)
        lib.pardisoinit(self.pt, C.byref(self.mtype), self.iparm)
        self.iparm[34] = 1   # zero-based indices
        self.iparm[7] = 0    # no iterative refinement steps beyond the default rule (0 = automatic: 2 if pivots perturbed)
        self.iparm[17] = -1  # report nnz(L)
        self.iparm[18] = -1  # report factorisation flops
        self.x = np.empty(self.N)
        self._alive = False

    def _call(self, phase, b=None):
        one, zero, err = C.c_int(1), C.c_int(0), C.c_int(0)
        n = C.c_int(self.N)
        bb = np.ascontiguousarray(b if b is not None else self.x, dtype=np.float64)
        self.lib.pardiso(self.pt, C.byref(one), C.byref(one), C.byref(self.mtype), C.byref(C.c_int(phase)), C.byref(n),
                         self.a.ctypes.data_as(C.c_void_p), self.ia.ctypes.data_as(C.c_void_p),
                         self.ja.ctypes.data_as(C.c_void_p), None, C.byref(one), self.iparm, C.byref(zero),
                         bb.ctypes.data_as(C.c_void_p), self.x.ctypes.data_as(C.c_void_p), C.byref(err))
        if err.value != 0:
            raise RuntimeError(f"pardiso phase {phase} failed with error {err.value}")

    def analyse(self):
        self._call(11)
        self._alive = True

    def factor(self):
        self._call(22)

    def analyse_and_factor(self):
        self._call(12)
        self._alive = True

    def solve(self, b):
        self._call(33, b)
        return self.x

    @property
    def nnz_factor(self):
        return int(self.iparm[17])

    def free(self):
        if self._alive:
            self._call(-1)
            self._alive = False

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def time_pardiso(N, cp, ri, vx, b, threads, budget_s=4.0, with_analysis=True):
    """One thread count: the unit with analysis (phase 12 + 33, what fact_ma57.c:529-625 / fact_umfpack.c:145-160 do on
    every set_matrix) and numeric-only (phase 22 + 33, the analysis kept)."""
    P = Pardiso(N, cp, ri, vx, threads=threads)
    out = {"cores": int(threads), "unit": "factor+solve/s"}
    t0 = time.perf_counter()
    P.analyse_and_factor()
    t1 = time.perf_counter()
    x = P.solve(b).copy()
    t2 = time.perf_counter()
    reps, t_tot, t_solve = 1, t2 - t0, t2 - t1
    while with_analysis and t_tot < budget_s / 2 and reps < 50:
        t0 = time.perf_counter()
        P.analyse_and_factor()
        t1 = time.perf_counter()
        P.solve(b)
        t2 = time.perf_counter()
        t_tot += t2 - t0
        t_solve += t2 - t1
        reps += 1
    out.update({"value": reps / t_tot, "ms_per_unit": t_tot / reps * 1e3, "solve_ms": t_solve / reps * 1e3,
                "nnz_L": P.nnz_factor})
    nrep, t_num, t_ns = 0, 0.0, 0.0
    while nrep < 2 or (t_num < budget_s / 2 and nrep < 200):
        t0 = time.perf_counter()
        P.factor()
        t1 = time.perf_counter()
        P.solve(b)
        t2 = time.perf_counter()
        t_num += t2 - t0
        t_ns += t2 - t1
        nrep += 1
    P.free()
    out["numeric_only"] = {"value": nrep / t_num, "ms_per_unit": t_num / nrep * 1e3, "solve_ms": t_ns / nrep * 1e3}
    out["sample"] = f"{reps} x (phase 12 analysis + factorisation, phase 33 solve) and {nrep} x (phase 22 + 33) of the same K"
    return out, x


def baseline(N, cp, ri, vx, b, budget_s=8.0):
    """The bench line's `cpu_baseline.pardiso` object: one thread, and the thread count that factors this K fastest (a
    sweep over powers of two up to the cores of the box - a matrix of this size does not scale to hundreds of threads);
    residual of the solution checked."""
    lib, name = _load()
    if lib is None:
        return {"present": False}
    cores = len(os.sched_getaffinity(0))
    out = {"present": True, "library": name, "mtype": -2, "host_cores": cores,
           "role": "stand-in for the reference's third-party CPU backends (MA57 / UMFPACK / CHOLMOD are not installed): "
                   "multithreaded supernodal symmetric-indefinite LDL^T on the same K and right-hand side"}
    try:
        sweep = {}
        counts = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, cores) if c <= cores})
        x = None
        for c in counts:
            r, x = time_pardiso(N, cp, ri, vx, b, c, budget_s / (2.0 * len(counts)), with_analysis=False)
            sweep[str(c)] = r
        best = max(sweep.values(), key=lambda r: r["numeric_only"]["value"])
        out["one_thread"] = sweep["1"]
        out["best_threads"] = best
        out["all_cores"] = sweep[str(cores)]
        out["threads_sweep_numeric_only"] = {k: round(v["numeric_only"]["value"], 3) for k, v in sweep.items()}
        # scaled residual of the PARDISO solution on K (same measure as the parity tests)
        import scipy.sparse as sp

        L = sp.csc_matrix((vx, ri, cp), shape=(N, N))
        K = L + sp.tril(L, -1).T
        r = K @ x - b
        out["scaled_residual"] = float(np.abs(r).max() / (abs(K).sum(axis=1).max() * np.abs(x).max() + np.abs(b).max()))
    except Exception as e:  # noqa: BLE001
        out["error"] = repr(e)[:300]
    return out


def baseline_subprocess(N, cp, ri, vx, b, budget_s=8.0, timeout_s=180.0):
    """`baseline` in a fresh interpreter: MKL's threading runtime stays apart from whatever OpenMP runtime the calling
    process (torch) has loaded, and a crash of the third-party library cannot take the bench down."""
    import json
    import subprocess
    import sys
    import tempfile

    if _load()[0] is None:
        return {"present": False}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "k.npz")
        np.savez(path, N=N, cp=cp, ri=ri, vx=vx, b=b)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), path, str(budget_s)], capture_output=True,
                               text=True, timeout=timeout_s)
            if r.returncode != 0:
                return {"present": True, "error": (r.stderr or r.stdout)[-300:]}
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:  # noqa: BLE001
            return {"present": True, "error": repr(e)[:300]}


if __name__ == "__main__":
    import json
    import sys

    z = np.load(sys.argv[1])
    print(json.dumps(baseline(int(z["N"]), z["cp"], z["ri"], z["vx"], z["b"], float(sys.argv[2]))))
