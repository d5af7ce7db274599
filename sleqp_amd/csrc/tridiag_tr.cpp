// Trust-region subproblem on a symmetric tridiagonal matrix (host, O(k) per secular step):
//
//     min  1/2 h^T T h + gamma0 e_1^T h   subject to  ||h||_2 <= radius,
//
// T = tridiag(gamma_1..gamma_{k-1}; delta_0..delta_{k-1}; gamma_1..gamma_{k-1}).  This is the inner
// problem of the generalised Lanczos trust-region method (Gould, Lucidi, Roma, Toint 1999), which is
// what the reference's EQP step runs through the third-party trlib (tr/trlib_solver.c:322-352; trlib
// is absent from the reference tree and unpinned, CMakeLists.txt:96-98 - the published algorithm is
// restated here).  More'-Sorensen iteration on the secular equation with LDL^T factorisations of
// T + lambda I, leftmost eigenvalue by Sturm bisection, hard case by inverse iteration.
#include "tridiag_tr.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace hipfact {

namespace {

// LDL^T of T + lam I (no pivoting); returns false when a pivot is not positive
bool ldl_shifted(int k, const double* delta, const double* gamma, double lam, std::vector<double>& d,
                 std::vector<double>& l) {
  d.resize(k);
  l.resize(std::max(k - 1, 0));
  d[0] = delta[0] + lam;
  if (!(d[0] > 0.0)) return false;
  for (int i = 1; i < k; ++i) {
    l[i - 1] = gamma[i] / d[i - 1];
    d[i] = delta[i] + lam - l[i - 1] * gamma[i];
    if (!(d[i] > 0.0)) return false;
  }
  return true;
}

// solves (T + lam I) x = rhs with the factors above
void ldl_solve(int k, const std::vector<double>& d, const std::vector<double>& l, const double* rhs, double* x) {
  for (int i = 0; i < k; ++i) x[i] = rhs[i] - (i > 0 ? l[i - 1] * x[i - 1] : 0.0);
  for (int i = 0; i < k; ++i) x[i] /= d[i];
  for (int i = k - 2; i >= 0; --i) x[i] -= l[i] * x[i + 1];
}

// number of eigenvalues of T below x (Sturm sequence)
int sturm_count(int k, const double* delta, const double* gamma, double x) {
  int count = 0;
  double q = delta[0] - x;
  if (q < 0.0) ++count;
  for (int i = 1; i < k; ++i) {
    if (q == 0.0) q = 1e-300;
    q = delta[i] - x - gamma[i] * gamma[i] / q;
    if (q < 0.0) ++count;
  }
  return count;
}

double leftmost_eigenvalue(int k, const double* delta, const double* gamma) {
  double lo = delta[0], hi = delta[0];
  for (int i = 0; i < k; ++i) {
    const double r = (i > 0 ? std::fabs(gamma[i]) : 0.0) + (i + 1 < k ? std::fabs(gamma[i + 1]) : 0.0);
    lo = std::min(lo, delta[i] - r);
    hi = std::max(hi, delta[i] + r);
  }
  for (int it = 0; it < 200 && hi - lo > 1e-15 * std::max(1.0, std::max(std::fabs(lo), std::fabs(hi))); ++it) {
    const double mid = 0.5 * (lo + hi);
    if (sturm_count(k, delta, gamma, mid) >= 1)
      hi = mid;
    else
      lo = mid;
  }
  return 0.5 * (lo + hi);
}

double norm2(int k, const double* x) {
  double s = 0.0;
  for (int i = 0; i < k; ++i) s += x[i] * x[i];
  return std::sqrt(s);
}

}  // namespace

int tridiag_tr_solve(int k, const double* delta, const double* gamma, double gamma0, double radius, double* h,
                     double* lambda) {
  if (k <= 0 || !(radius > 0.0)) return -1;
  std::vector<double> d, l, rhs(k, 0.0), w(k);
  rhs[0] = -gamma0;
  double lam = 0.0;
  // interior solution?
  if (ldl_shifted(k, delta, gamma, 0.0, d, l)) {
    ldl_solve(k, d, l, rhs.data(), h);
    if (norm2(k, h) <= radius) {
      *lambda = 0.0;
      return 0;
    }
  }
  const double theta = leftmost_eigenvalue(k, delta, gamma);
  const double scale = std::max(1.0, std::fabs(theta));
  double lam_lo = std::max(0.0, -theta);  // the multiplier lies in [lam_lo, inf)
  lam = std::max(lam, lam_lo);
  // start to the right of the pole; the secular Newton iteration then converges monotonically from the left
  // of the root once ||h|| > radius, so first make sure the factorisation exists
  double shift = 1e-10 * scale;
  while (!ldl_shifted(k, delta, gamma, lam, d, l)) {
    lam = lam_lo + shift;
    shift *= 10.0;
    if (shift > 1e10 * scale) return -2;
  }
  ldl_solve(k, d, l, rhs.data(), h);
  double hn = norm2(k, h);
  // Hard case (or nearly): the boundary is not reached to the right of the pole.  The multiple of the
  // leftmost eigenvector that reaches it is added (More'-Sorensen); the factors are those of the nearly
  // singular T + lam I, so inverse iteration converges in a step or two.
  auto complete_with_eigenvector = [&]() {
    std::vector<double> u(k, 1.0), t(k);
    for (int i = 0; i < k; ++i) u[i] = 1.0 + 0.37 * ((i * 2654435761u) % 1000) / 1000.0;  // no accidental orthogonality
    for (int it = 0; it < 4; ++it) {
      ldl_solve(k, d, l, u.data(), t.data());
      const double nu = norm2(k, t.data());
      if (!(nu > 0.0) || !(nu < 1.7e308)) break;
      for (int i = 0; i < k; ++i) u[i] = t[i] / nu;
    }
    const double un = norm2(k, u.data());
    if (!(un > 0.0)) return;
    for (int i = 0; i < k; ++i) u[i] /= un;
    double hu = 0.0;
    for (int i = 0; i < k; ++i) hu += h[i] * u[i];
    const double hcur = norm2(k, h);
    const double disc = hu * hu + (radius * radius - hcur * hcur);
    if (disc < 0.0) {  // (only possible from outside the region) pull back radially
      for (int i = 0; i < k; ++i) h[i] *= radius / hcur;
      return;
    }
    // root of tau^2 + 2 hu tau - (radius^2 - ||h||^2) = 0 of smaller magnitude, without cancellation
    const double sq = std::sqrt(disc);
    const double tau = (radius * radius - hcur * hcur) / (hu + (hu >= 0.0 ? sq : -sq));
    for (int i = 0; i < k; ++i) h[i] += tau * u[i];
  };
  if (hn < radius) {
    // if lam_lo == 0 this is the interior solution up to rounding
    if (lam_lo > 0.0) complete_with_eigenvector();
    *lambda = lam;
    return 0;
  }
  // secular Newton: phi(lam) = 1 / ||h(lam)|| - 1 / radius
  for (int it = 0; it < 100; ++it) {
    if (std::fabs(hn - radius) <= 1e-14 * radius) break;
    // w = L^-1 h scaled: ||w||^2 = h^T (T + lam I)^-1 h
    ldl_solve(k, d, l, h, w.data());
    double hw = 0.0;
    for (int i = 0; i < k; ++i) hw += h[i] * w[i];
    if (!(hw > 0.0)) break;
    double lam_new = lam + (hn * hn / hw) * ((hn - radius) / radius);
    if (!(lam_new > lam_lo)) lam_new = 0.5 * (lam + lam_lo);  // safeguard: stay right of the pole
    if (lam_new == lam) break;
    if (!ldl_shifted(k, delta, gamma, lam_new, d, l)) {
      lam_lo = std::max(lam_lo, lam_new);
      lam_new = 0.5 * (lam + lam_new);
      if (!ldl_shifted(k, delta, gamma, lam_new, d, l)) break;
    }
    lam = lam_new;
    ldl_solve(k, d, l, rhs.data(), h);
    hn = norm2(k, h);
  }
  // nearly hard case: the secular iteration has run into the pole (||h|| changes by many orders of magnitude
  // per ulp of lambda there) without meeting the boundary: move along the leftmost eigenvector instead
  if (lam > 0.0 && std::fabs(hn - radius) > 1e-12 * radius && ldl_shifted(k, delta, gamma, lam, d, l))
    complete_with_eigenvector();
  *lambda = lam;
  return 0;
}

}  // namespace hipfact

extern "C" int hipfact_tridiag_tr(int k, const double* delta, const double* gamma, double gamma0, double radius,
                                  double* h, double* lambda) {
  if (!delta || !h || !lambda || (k > 1 && !gamma)) return -1;
  return hipfact::tridiag_tr_solve(k, delta, gamma, gamma0, radius, h, lambda);
}
