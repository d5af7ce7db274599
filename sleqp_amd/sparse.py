"""Host-side mirrors of the reference's sparse containers (numpy backed).

``SleqpVec`` follows the public struct of sparse/pub_vec.h:16-25 (sorted
indices, ``dim``, ``nnz``); ``SleqpMat`` is the CSC container of
sparse/mat.c:11-25 with the push-style construction of mat.c:178-225.  They
exist so that the Python tests and the bench can drive the C ABI with exactly
the data the SLEQP shim would pass.  No numerics live here.
"""
from __future__ import annotations

import numpy as np


class SleqpVec:
    """Sparse vector: ascending ``indices``, ``data``, dimension ``dim``."""

    def __init__(self, dim: int, indices=None, data=None):
        self.dim = int(dim)
        self.indices = np.zeros(0, dtype=np.int32) if indices is None else np.asarray(indices, dtype=np.int32)
        self.data = np.zeros(0, dtype=np.float64) if data is None else np.asarray(data, dtype=np.float64)
        assert self.indices.shape == self.data.shape
        assert np.all(np.diff(self.indices) > 0), "indices must be strictly ascending"
        assert self.indices.size == 0 or (self.indices[0] >= 0 and self.indices[-1] < self.dim)

    @property
    def nnz(self) -> int:
        return int(self.indices.size)

    # sleqp_vec_set_from_raw (sparse/vec.c:71-103): keep entries with |v| > zero_eps
    @classmethod
    def from_raw(cls, values, zero_eps: float = 0.0) -> "SleqpVec":
        values = np.asarray(values, dtype=np.float64)
        keep = ~(np.abs(values) <= zero_eps)
        idx = np.nonzero(keep)[0].astype(np.int32)
        return cls(values.size, idx, values[idx])

    # sleqp_vec_to_raw (sparse/vec.c:105-119)
    def to_raw(self) -> np.ndarray:
        out = np.zeros(self.dim)
        out[self.indices] = self.data
        return out

    # sleqp_vec_resize (used by standard_aug_jac.c:376,389 to pad / unpad the rhs)
    def resized(self, dim: int) -> "SleqpVec":
        keep = self.indices < dim
        return SleqpVec(dim, self.indices[keep], self.data[keep])

    def shifted(self, offset: int, dim: int) -> "SleqpVec":
        """rhs->indices[k] += offset (standard_aug_jac.c:328-333)."""
        return SleqpVec(dim, self.indices + offset, self.data)


class SleqpMat:
    """CSC matrix: ``cols`` (num_cols+1), ``rows``, ``data``; rows strictly
    ascending inside a column (sleqp_mat_is_valid, sparse/mat.c:772-821)."""

    def __init__(self, num_rows: int, num_cols: int, cols=None, rows=None, data=None):
        self.num_rows, self.num_cols = int(num_rows), int(num_cols)
        self.cols = np.zeros(num_cols + 1, dtype=np.int32) if cols is None else np.ascontiguousarray(cols, dtype=np.int32)
        self.rows = np.zeros(0, dtype=np.int32) if rows is None else np.ascontiguousarray(rows, dtype=np.int32)
        self.data = np.zeros(0, dtype=np.float64) if data is None else np.ascontiguousarray(data, dtype=np.float64)

    @property
    def nnz(self) -> int:
        return int(self.cols[-1]) if self.cols.size else 0

    @classmethod
    def from_scipy(cls, M) -> "SleqpMat":
        import scipy.sparse as sp

        M = sp.csc_matrix(M)
        M.sort_indices()
        return cls(M.shape[0], M.shape[1], M.indptr, M.indices, M.data)

    def to_scipy(self):
        import scipy.sparse as sp

        return sp.csc_matrix((self.data[: self.nnz], self.rows[: self.nnz], self.cols), shape=(self.num_rows, self.num_cols))

    def is_valid(self) -> bool:
        if self.cols[0] != 0 or np.any(np.diff(self.cols) < 0):
            return False
        for j in range(self.num_cols):
            r = self.rows[self.cols[j] : self.cols[j + 1]]
            if r.size and (np.any(np.diff(r) <= 0) or r[0] < 0 or r[-1] >= self.num_rows):
                return False
        return True

    def is_lower(self) -> bool:
        col_of = np.repeat(np.arange(self.num_cols), np.diff(self.cols))
        return bool(np.all(self.rows[: self.nnz] >= col_of))
