/* tr_hipfact.h — SleqpTRSolver that runs the Krylov loop of the EQP step on the device. */
#ifndef SLEQP_TR_HIPFACT_H
#define SLEQP_TR_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "pub_settings.h"
#include "sparse/mat.h"
#include "tr/tr_solver.h"
#endif

struct hipfact_handle;

/* Control block of the solver (owned by the SleqpTRSolver, valid until it is released). */
typedef struct SleqpHipfactTR SleqpHipfactTR;

/* Created like sleqp_trlib_solver_create / sleqp_steihaug_solver_create (tr/trlib_solver.c:723-817,
 * tr/steihaug_solver.c:498-536).  SLEQP_SETTINGS_ENUM_TR_SOLVER selects the method exactly as
 * newton.c:97-109 does: CG -> projected Steihaug CG, TRLIB / AUTO -> generalised Lanczos (what trlib
 * runs).  The Hessian is the problem's matrix-free product (sleqp_problem_hess_prod with the
 * multipliers of the solve call) unless an explicit matrix is supplied below. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_solver_create(SleqpTRSolver** star,
                               SleqpHipfactTR** ctl,
                               SleqpProblem* problem,
                               SleqpSettings* settings);

/* The factorisation to project with: the handle that sleqp_hipfact_aug_jac_create returned for the
 * augmented Jacobian the solver will be called with.  The solver takes its own reference
 * (hipfact_retain), so the two objects may be released in any order. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_bind(SleqpHipfactTR* ctl, struct hipfact_handle* handle);

/* Optional: Hessian of the Lagrangian at the current iterate and multipliers as an explicit matrix,
 * lower triangle, CSC (the prod_from_hess_matrix precedent, bindings/mex/mex_hess.c:85-139); it then
 * replaces the matrix-free product and stays resident in HBM (no PCIe traffic inside the loop).
 * The pattern may change between calls; NULL goes back to the matrix-free product. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_set_hessian(SleqpHipfactTR* ctl, const SleqpMat* hess_lower);

#endif /* SLEQP_TR_HIPFACT_H */
