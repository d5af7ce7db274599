/* aug_jac_hipfact.h — SleqpAugJac with on-device KKT assembly (optional second boundary). */
#ifndef SLEQP_AUG_JAC_HIPFACT_H
#define SLEQP_AUG_JAC_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "aug_jac/aug_jac.h"
#include "pub_settings.h"
#endif

SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_aug_jac_create(SleqpAugJac** star, SleqpProblem* problem, SleqpSettings* settings);

/* The hipfact handle (device factorisation) behind an augmented Jacobian created above, NULL for any
 * other SleqpAugJac: used by tr_hipfact.c, which runs the projected CG on the same handle. */
struct hipfact_handle;
struct hipfact_handle*
sleqp_hipfact_aug_jac_handle(SleqpAugJac* aug_jac);

#endif /* SLEQP_AUG_JAC_HIPFACT_H */
