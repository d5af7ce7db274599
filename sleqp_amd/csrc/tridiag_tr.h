// Tridiagonal trust-region subproblem of the generalised Lanczos method (host side).
#pragma once

namespace hipfact {

// min 1/2 h^T T h + gamma0 e_1^T h, ||h|| <= radius; T has diagonal delta[0..k) and off-diagonal
// gamma[1..k) (gamma[i] couples i-1 and i; gamma[0] is not read).  Returns 0 on success; *lambda is the
// multiplier of the trust-region constraint (0: interior solution).
int tridiag_tr_solve(int k, const double* delta, const double* gamma, double gamma0, double radius, double* h,
                     double* lambda);

}  // namespace hipfact
