/* tr_hipfact.h — SleqpTRSolver that runs the projected CG of the EQP step on the device. */
#ifndef SLEQP_TR_HIPFACT_H
#define SLEQP_TR_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "pub_settings.h"
#include "sparse/mat.h"
#include "tr/tr_solver.h"
#endif

/* Control block of the solver (owned by the SleqpTRSolver, valid until it is released). */
typedef struct SleqpHipfactTR SleqpHipfactTR;

/* Created like sleqp_steihaug_solver_create (tr/steihaug_solver.c:498-536).  `ctl` receives the
 * control block through which the caller supplies the Hessian of the Lagrangian. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_solver_create(SleqpTRSolver** star,
                               SleqpHipfactTR** ctl,
                               SleqpProblem* problem,
                               SleqpSettings* settings);

/* Bind the solver to the augmented Jacobian of aug_jac_hipfact.c whose factorisation it projects with
 * (once after creation, again if the augmented Jacobian object is replaced). */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_bind(SleqpHipfactTR* ctl, SleqpAugJac* aug_jac);

/* Hessian of the Lagrangian at the current iterate and multipliers as an explicit matrix, lower
 * triangle, CSC (the prod_from_hess_matrix precedent, bindings/mex/mex_hess.c:85-139).  The
 * pattern may change between calls; values are copied to the device. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_tr_set_hessian(SleqpHipfactTR* ctl, const SleqpMat* hess_lower);

#endif /* SLEQP_TR_HIPFACT_H */
