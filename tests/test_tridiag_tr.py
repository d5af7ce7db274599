"""Host part of the generalised Lanczos trust-region solver: the tridiagonal subproblem
(sleqp_amd/csrc/tridiag_tr.cpp) against a dense eigen-decomposition."""
import ctypes as C

import numpy as np
import pytest


def _exact(T, g0, radius):
    lam, V = np.linalg.eigh(T)
    gt = V.T @ g0

    def step(mu):
        return -gt / (lam + mu)

    if lam[0] > 0 and np.linalg.norm(step(0.0)) <= radius:
        return V @ step(0.0), 0.0
    lo = max(0.0, -lam[0]) + 1e-15
    hi = lo + 1.0
    while np.linalg.norm(step(hi)) > radius:
        hi = lo + 2 * (hi - lo)
    for _ in range(300):
        mid = 0.5 * (lo + hi)
        if np.linalg.norm(step(mid)) > radius:
            lo = mid
        else:
            hi = mid
    return V @ step(hi), hi


def _solve(lib, delta, gamma, gamma0, radius):
    k = len(delta)
    d = np.ascontiguousarray(delta, dtype=np.float64)
    g = np.zeros(max(k, 1))
    g[1:k] = gamma
    h = np.zeros(k)
    lam = C.c_double()
    rc = lib.hipfact_tridiag_tr(k, d.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), C.c_double(gamma0),
                                C.c_double(radius), h.ctypes.data_as(C.c_void_p), C.byref(lam))
    assert rc == 0
    return h, lam.value


@pytest.mark.parametrize("k", [1, 2, 5, 40, 100])
@pytest.mark.parametrize("kind", ["spd", "indefinite"])
@pytest.mark.parametrize("radius", [1e-2, 1.0, 1e3])
def test_tridiagonal_trust_region_subproblem(hipfact_lib, k, kind, radius):
    rng = np.random.default_rng(k * 7 + (kind == "spd"))
    gamma = np.abs(rng.standard_normal(max(k - 1, 0))) + 0.1
    delta = rng.standard_normal(k)
    if kind == "spd":
        delta = np.abs(delta) + 2.5 * np.concatenate([gamma, [0]])[:k] + 2.5 * np.concatenate([[0], gamma])[:k] + 0.1
    T = np.diag(delta) + np.diag(gamma, 1) + np.diag(gamma, -1)
    gamma0 = 1.7
    g0 = np.zeros(k)
    g0[0] = gamma0
    h, lam = _solve(hipfact_lib, delta, gamma, gamma0, radius)
    want, mu = _exact(T, g0, radius)
    def model(x):
        return g0 @ x + 0.5 * x @ (T @ x)

    assert np.linalg.norm(h) <= radius * (1 + 1e-10)
    # near the hard case ||h(lambda)|| changes by 1e8 per unit of lambda: the multiplier and the model value are
    # what is determined to working accuracy, the vector only to ~1e-7
    assert model(h) <= model(want) + 1e-10 * max(1.0, abs(model(want)))
    # (numerically) hard case: e_1 has no component along the leftmost eigenvector, the dense reference stays
    # inside the region while the solver adds that eigenvector and reaches the boundary - a lower model value
    hard = mu > 0 and np.linalg.norm(want) < radius * (1 - 1e-6)
    if not hard:
        assert np.abs(h - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    assert abs(lam - mu) <= 1e-8 * max(1.0, mu)
    # KKT conditions of the subproblem
    assert np.abs((T + lam * np.eye(k)) @ h + g0).max() <= 1e-9 * max(1.0, np.abs(g0).max(), lam * radius)
    if lam > 0:
        assert abs(np.linalg.norm(h) - radius) <= 1e-6 * radius


def test_tridiagonal_hard_case(hipfact_lib):
    """e_1 orthogonal to the leftmost eigenvector: the boundary is reached by adding that eigenvector."""
    # T = diag(1, -2) decoupled (gamma -> 0): the leftmost eigenvector is e_2, untouched by gamma0 e_1
    delta = np.array([1.0, -2.0])
    gamma = np.array([0.0])
    h, lam = _solve(hipfact_lib, delta, gamma, 0.5, 3.0)
    assert abs(lam - 2.0) <= 1e-8
    assert abs(np.linalg.norm(h) - 3.0) <= 1e-8
    assert abs(h[0] + 0.5 / 3.0) <= 1e-8  # (1 + lam) h_0 = -gamma0
