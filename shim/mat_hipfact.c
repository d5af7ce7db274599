/* mat_hipfact.c — SleqpMat products on the device.
 *
 * The reference computes cons_jac^T * multipliers (newton.c:377), cons_jac * step (working_step.c:341,
 * direction.c:66) with the scatter / merge-join loops of sparse/mat.c:282-363 on the host.  Here the matrix
 * lives in HBM in both orientations (hipfact_spmat: CSC as given = CSR of the transpose, plus the CSR built
 * once per pattern) and either product is one gather-only kernel; per call one sparse vector goes up
 * (densified on the host like sleqp_vec_to_raw, sparse/vec.c:105-119) and one dense vector comes back, packed
 * with sleqp_vec_set_from_raw (sparse/vec.c:71-103) for the transposed product.
 *
 * Call sites:
 *   newton.c:377      sleqp_mat_mult_vec_trans(cons_jac, violated_multipliers, zero_eps, sparse_cache)
 *   working_step.c:341, direction.c:66   sleqp_mat_mult_vec(cons_jac, direction, dense_cache)
 */
#include "mat_hipfact.h"

#include <assert.h>
#include <stdint.h>

#ifndef HIPFACT_STANDALONE
#include "fail.h"
#include "mem.h"
#endif

#include "hipfact.h"

struct SleqpHipfactMat
{
  hipfact_handle* handle; /* own reference */
  hipfact_spmat* device;
  int num_rows, num_cols, nnz;
  uint64_t pattern_hash;
  double* dense_in;  /* max(num_rows, num_cols) */
  double* dense_out; /* max(num_rows, num_cols) */
  int dense_size;
};

static uint64_t
pattern_hash(const SleqpMat* matrix)
{
  uint64_t h        = 1469598103934665603ull;
  const int* cols   = sleqp_mat_cols(matrix);
  const int* rows   = sleqp_mat_rows(matrix);
  const int numcols = sleqp_mat_num_cols(matrix);
  const int nnz     = sleqp_mat_nnz(matrix);
  for (int j = 0; j <= numcols; ++j)
  {
    h = (h ^ (uint64_t)(unsigned)cols[j]) * 1099511628211ull;
  }
  for (int k = 0; k < nnz; ++k)
  {
    h = (h ^ (uint64_t)(unsigned)rows[k]) * 1099511628211ull;
  }
  return h;
}

#define HIPFACT_MAT_CALL(mat, x)                                               \
  do                                                                           \
  {                                                                            \
    const int hipfact_status = (x);                                            \
    if (hipfact_status != HIPFACT_OK)                                          \
    {                                                                          \
      sleqp_raise(SLEQP_INTERNAL_ERROR,                                        \
                  "Caught hipfact error <%d> (%s)",                            \
                  hipfact_status,                                              \
                  hipfact_last_error((mat)->handle));                          \
    }                                                                          \
  } while (0)

SLEQP_RETCODE
sleqp_hipfact_mat_create(SleqpHipfactMat** star, struct hipfact_handle* handle)
{
  if (!handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact matrix needs a hipfact handle");
  }

  SleqpHipfactMat* mat = NULL;

  SLEQP_CALL(sleqp_malloc(&mat));

  *mat = (SleqpHipfactMat){0};

  if (hipfact_retain(handle) != HIPFACT_OK)
  {
    sleqp_free(&mat);
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact matrix: cannot retain the handle");
  }

  mat->handle = handle;
  *star       = mat;

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_mat_set(SleqpHipfactMat* mat, const SleqpMat* matrix)
{
  const int num_rows  = sleqp_mat_num_rows(matrix);
  const int num_cols  = sleqp_mat_num_cols(matrix);
  const int nnz       = sleqp_mat_nnz(matrix);
  const uint64_t hash = pattern_hash(matrix);

  if (mat->device && num_rows == mat->num_rows && num_cols == mat->num_cols && nnz == mat->nnz
      && hash == mat->pattern_hash)
  {
    HIPFACT_MAT_CALL(mat, hipfact_spmat_update_values(mat->device, sleqp_mat_data(matrix)));
    return SLEQP_OKAY;
  }

  if (mat->device)
  {
    hipfact_spmat_free(&mat->device);
  }

  HIPFACT_MAT_CALL(mat,
                   hipfact_spmat_create(mat->handle,
                                        num_rows,
                                        num_cols,
                                        sleqp_mat_cols(matrix),
                                        sleqp_mat_rows(matrix),
                                        sleqp_mat_data(matrix),
                                        &mat->device));

  mat->num_rows     = num_rows;
  mat->num_cols     = num_cols;
  mat->nnz          = nnz;
  mat->pattern_hash = hash;

  const int size = num_rows > num_cols ? num_rows : num_cols;

  if (size > mat->dense_size)
  {
    SLEQP_CALL(sleqp_realloc(&mat->dense_in, size));
    SLEQP_CALL(sleqp_realloc(&mat->dense_out, size));
    mat->dense_size = size;
  }

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_mat_mult_vec(SleqpHipfactMat* mat, const SleqpVec* vector, double* result)
{
  if (!mat->device)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact matrix: no matrix set");
  }

  assert(vector->dim == mat->num_cols);

  SLEQP_CALL(sleqp_vec_to_raw(vector, mat->dense_in));

  HIPFACT_MAT_CALL(mat, hipfact_spmat_mult_vec(mat->device, mat->dense_in, result));

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_mat_mult_vec_trans(SleqpHipfactMat* mat, const SleqpVec* vector, double eps, SleqpVec* result)
{
  if (!mat->device)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact matrix: no matrix set");
  }

  assert(vector->dim == mat->num_rows);
  assert(result->dim == mat->num_cols);

  SLEQP_CALL(sleqp_vec_to_raw(vector, mat->dense_in));

  HIPFACT_MAT_CALL(mat, hipfact_spmat_mult_vec_trans(mat->device, mat->dense_in, mat->dense_out));

  /* entries with |sum| <= eps are not pushed (mat.c:349-358); sleqp_vec_set_from_raw applies the same test */
  SLEQP_CALL(sleqp_vec_set_from_raw(result, mat->dense_out, mat->num_cols, eps));

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_mat_release(SleqpHipfactMat** star)
{
  SleqpHipfactMat* mat = *star;

  if (!mat)
  {
    return SLEQP_OKAY;
  }

  if (mat->device)
  {
    hipfact_spmat_free(&mat->device);
  }

  hipfact_free(&mat->handle);

  sleqp_free(&mat->dense_out);
  sleqp_free(&mat->dense_in);

  sleqp_free(&mat);

  *star = NULL;

  return SLEQP_OKAY;
}
