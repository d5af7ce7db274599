"""Shared helpers for the tests (golden loader, tolerances)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kkt_cases.npz")

# Parity tolerance for floating point (BASELINE.md "Parity"): solution agreement with the
# LAPACK-restating oracle <= 1e-9 relative; scaled residual <= 1e-12 after refinement.
REL_TOL = 1e-9
RESID_TOL = 1e-12
ZERO_EPS = 1e-20


class Case:
    def __init__(self, z, name):
        p = name + "/"
        self.name = name
        for k in z.files:
            if k.startswith(p):
                v = z[k]
                setattr(self, k[len(p):], v.item() if v.ndim == 0 else v)


def golden_cases():
    z = np.load(GOLDEN)
    return [Case(z, str(n)) for n in z["names"]]


def rel_err(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def scaled_residual(K, x, b):
    """||Kx - b||_inf / (||K||_inf ||x||_inf + ||b||_inf)"""
    r = np.abs(K @ x - b).max() if x.size else 0.0
    kn = abs(K).sum(axis=1).max() if x.size else 0.0
    den = kn * (np.abs(x).max() if x.size else 0.0) + (np.abs(b).max() if b.size else 0.0)
    return float(r / den) if den > 0 else float(r)
