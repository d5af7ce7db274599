/* mat_hipfact.h — sparse products of SleqpMat on the device: drop-in forms of
 * sleqp_mat_mult_vec / sleqp_mat_mult_vec_trans (sparse/mat.c:282-363) for the once-per-EQP Jacobian
 * products (newton.c:377, working_step.c:341, direction.c:66). */
#ifndef SLEQP_MAT_HIPFACT_H
#define SLEQP_MAT_HIPFACT_H

#ifdef HIPFACT_STANDALONE
#include "sleqp_mini.h"
#else
#include "sparse/mat.h"
#include "sparse/pub_vec.h"
#endif

struct hipfact_handle;

/* A device-resident copy of one SleqpMat (both orientations, so that either product is a gather-only CSR
 * kernel).  Owns a reference to the hipfact handle it lives on. */
typedef struct SleqpHipfactMat SleqpHipfactMat;

SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_mat_create(SleqpHipfactMat** star, struct hipfact_handle* handle);

/* Makes the device copy current: uploads the values; pattern and transposed index structure only when the
 * pattern differs from the last call (hash of cols / rows). */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_mat_set(SleqpHipfactMat* mat, const SleqpMat* matrix);

/* result = matrix * vector, dense result of length num_rows: sleqp_mat_mult_vec (sparse/mat.c:282-310). */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_mat_mult_vec(SleqpHipfactMat* mat, const SleqpVec* vector, double* result);

/* result = matrix^T * vector as a sparse vector, entries with |value| <= eps dropped:
 * sleqp_mat_mult_vec_trans (sparse/mat.c:312-363). */
SLEQP_WARNUNUSED
SLEQP_RETCODE
sleqp_hipfact_mat_mult_vec_trans(SleqpHipfactMat* mat, const SleqpVec* vector, double eps, SleqpVec* result);

SLEQP_RETCODE
sleqp_hipfact_mat_release(SleqpHipfactMat** star);

#endif /* SLEQP_MAT_HIPFACT_H */
