/* aug_jac_hipfact.h — SleqpAugJac with on-device KKT assembly (optional second boundary). */
#ifndef SLEQP_AUG_JAC_HIPFACT_H
#define SLEQP_AUG_JAC_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "aug_jac/aug_jac.h"
#include "pub_settings.h"
#endif

struct hipfact_handle;

/* Created like sleqp_standard_aug_jac_create (aug_jac/standard_aug_jac.c:525-547), without a
 * SleqpFact: the factorisation lives in a hipfact handle owned by the new object.  `handle`
 * (may be NULL) receives a borrowed pointer to it for objects that share the factorisation - the
 * TR solver of tr_hipfact.c takes its own reference with hipfact_retain; no process-global state
 * connects the two. */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_aug_jac_create(SleqpAugJac** star,
                             SleqpProblem* problem,
                             SleqpSettings* settings,
                             struct hipfact_handle** handle);

#endif /* SLEQP_AUG_JAC_HIPFACT_H */
