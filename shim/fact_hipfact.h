/* fact_hipfact.h — SleqpFact backend over the hipfact C ABI (include/hipfact.h).
 * Drop into src/main/fact/ of chrhansk/sleqp; see INTEGRATION.md. */
#ifndef SLEQP_FACT_HIPFACT_H
#define SLEQP_FACT_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "fact.h"
#include "sparse/mat.h"
#include "types.h"
#endif

SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_fact_hipfact_create(SleqpFact** star, SleqpSettings* settings);

/* PSD | LOWER: the backend behind the reduced AugJac (A_W A_W^T, symmetric
 * positive definite; pattern fact_cholmod.c:231-262) */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_fact_hipfact_psd_create(SleqpFact** star, SleqpSettings* settings);

#endif /* SLEQP_FACT_HIPFACT_H */
