/* fact_hipfact.c — the SLEQP side of the drop-in boundary.
 *
 * A SleqpFact backend (reference interface: fact/fact_types.h:9-32, creation
 * pattern fact/fact_ma57.c:809-839, fact/fact_lapack.c:186-221) whose five
 * callbacks forward to the MI355X-native library libhipfact.so through its C
 * ABI (include/hipfact.h).  Selected with -DSLEQP_FACT=HIPFACT
 * (shim/SearchFactHIPFACT.cmake).  Declares SLEQP_FACT_FLAGS_LOWER only, so the
 * augmented matrix arrives as the lower triangle built by fill_aug_jac
 * (aug_jac/standard_aug_jac.c:135-237) and the standard (not the reduced)
 * AugJac is used (trial_point.c:91-109).
 *
 * Host memory goes through sleqp_alloc_array / sleqp_realloc / sleqp_free so
 * that SLEQP_NOMEM is raised consistently; device errors and singular
 * matrices become sleqp_raise(SLEQP_INTERNAL_ERROR, ...), like
 * "Failed to factorize using LAPACK" (fact/fact_lapack.c:117-120).
 *
 * With -DHIPFACT_STANDALONE the file builds against shim/sleqp_mini.h (a
 * minimal stand-alone implementation of the few SLEQP entry points used here)
 * so that the boundary can be exercised without libsleqp.
 */
#include "fact_hipfact.h"

#include <assert.h>

#ifndef HIPFACT_STANDALONE
#include "defs.h"
#include "error.h"
#include "fail.h"
#include "log.h"
#include "mem.h"
#endif

#include "hipfact.h"

#ifndef SLEQP_FACT_HIPFACT_NAME
#define SLEQP_FACT_HIPFACT_NAME "hipfact"
#endif
#ifndef SLEQP_FACT_HIPFACT_VERSION
#define SLEQP_FACT_HIPFACT_VERSION HIPFACT_VERSION
#endif

typedef struct
{
  hipfact_handle* handle;

  /* dense staging buffer for hipfact_solution -> sleqp_vec_set_from_raw */
  double* slice;
  int slice_size;

  int num_rows;
} HipFactData;

#define HIPFACT_CALL(data, x)                                                  \
  do                                                                           \
  {                                                                            \
    const int hipfact_status = (x);                                            \
    if (hipfact_status != HIPFACT_OK)                                          \
    {                                                                          \
      sleqp_raise(SLEQP_INTERNAL_ERROR,                                        \
                  "Caught hipfact error <%d> (%s)",                            \
                  hipfact_status,                                              \
                  hipfact_last_error((data)->handle));                         \
    }                                                                          \
  } while (0)

static SLEQP_RETCODE
hipfact_fact_set_matrix(void* fact_data, SleqpMat* matrix)
{
  HipFactData* data = (HipFactData*)fact_data;

  const int num_cols = sleqp_mat_num_cols(matrix);
  const int num_rows = sleqp_mat_num_rows(matrix);

  assert(num_cols == num_rows);

  if (data->slice_size < num_rows)
  {
    SLEQP_CALL(sleqp_realloc(&data->slice, num_rows));
    data->slice_size = num_rows;
  }

  data->num_rows = num_rows;

  /* uploads K (or only its values when the pattern is unchanged) and runs the
   * numeric factorisation on the device */
  HIPFACT_CALL(data,
               hipfact_set_matrix(data->handle,
                                  num_cols,
                                  sleqp_mat_cols(matrix),
                                  sleqp_mat_rows(matrix),
                                  sleqp_mat_data(matrix)));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_fact_solve(void* fact_data, const SleqpVec* rhs)
{
  HipFactData* data = (HipFactData*)fact_data;

  assert(rhs->dim == data->num_rows);

  HIPFACT_CALL(data,
               hipfact_solve_sparse(data->handle,
                                    rhs->dim,
                                    rhs->nnz,
                                    rhs->indices,
                                    rhs->data));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_fact_solution(void* fact_data,
                      SleqpVec* sol,
                      int begin,
                      int end,
                      double zero_eps)
{
  HipFactData* data = (HipFactData*)fact_data;

  assert(begin <= end);

  HIPFACT_CALL(data, hipfact_solution(data->handle, data->slice, begin, end));

  SLEQP_CALL(sleqp_vec_set_from_raw(sol, data->slice, end - begin, zero_eps));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_fact_condition(void* fact_data, double* condition)
{
  HipFactData* data = (HipFactData*)fact_data;

  HIPFACT_CALL(data, hipfact_condition(data->handle, condition));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_fact_free(void** star)
{
  HipFactData* data = (HipFactData*)(*star);

  if (!data)
  {
    return SLEQP_OKAY;
  }

  hipfact_free(&data->handle);

  sleqp_free(&data->slice);

  sleqp_free(&data);

  *star = NULL;

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_data_create(HipFactData** star)
{
  SLEQP_CALL(sleqp_malloc(star));

  HipFactData* data = *star;

  *data = (HipFactData){0};

  /* device ordinal: SLEQP_HIP_DEVICE, then LOCAL_RANK, else 0 (one backend
   * instance per solver object, cf. thread_test.c:77-110) */
  const int status = hipfact_create(&data->handle, -1);

  if (status != HIPFACT_OK)
  {
    const char* message = hipfact_last_error(NULL);
    sleqp_free(star);
    sleqp_raise(SLEQP_INTERNAL_ERROR,
                "Failed to create hipfact backend <%d> (%s)",
                status,
                message);
  }

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_fact_hipfact_create(SleqpFact** star, SleqpSettings* settings)
{
  SleqpFactCallbacks callbacks = {.set_matrix = hipfact_fact_set_matrix,
                                  .solve      = hipfact_fact_solve,
                                  .solution   = hipfact_fact_solution,
                                  .condition  = hipfact_fact_condition,
                                  .free       = hipfact_fact_free};

  HipFactData* data = NULL;

  SLEQP_CALL(hipfact_data_create(&data));

  SLEQP_CALL(sleqp_fact_create(star,
                               SLEQP_FACT_HIPFACT_NAME,
                               SLEQP_FACT_HIPFACT_VERSION,
                               settings,
                               &callbacks,
                               SLEQP_FACT_FLAGS_LOWER,
                               (void*)data));

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_fact_create_default(SleqpFact** star, SleqpSettings* settings)
{
  SLEQP_CALL(sleqp_fact_hipfact_create(star, settings));

  return SLEQP_OKAY;
}
