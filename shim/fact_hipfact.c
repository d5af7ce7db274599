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
#include <math.h>

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

  int num_rows;

  /* counters of the handle at the last look (debug log of what a call cost) */
  double seen_analyses, seen_swaps, seen_fallbacks, seen_hits, seen_refine;

  bool psd; /* created by sleqp_fact_hipfact_psd_create */
} HipFactData;

#ifdef HIPFACT_STANDALONE
/* harness only: the handle behind the backend created last (the tests read
 * hipfact_get_info through it; SleqpFact keeps its fact_data private) */
static hipfact_handle* last_handle = NULL;
hipfact_handle*
sleqp_fact_hipfact_last_handle(void)
{
  return last_handle;
}
#endif

/* What a call did inside the library, at debug level (the pattern is
 * aug_jac.c:45-76: the reference wraps its callbacks in timers; here the
 * interesting events are symbolic analyses, plan swaps, fallbacks of the
 * dataflow launches and refinement passes behind the first) */
static void
hipfact_log_events(HipFactData* data, const char* where)
{
  if (sleqp_log_level() < SLEQP_LOG_DEBUG)
  {
    return;
  }

  double analyses = 0., swaps = 0., fallbacks = 0., hits = 0., refine = 0.;
  double analysis_s = 0., nlevels = 0., nsuper = 0., maps = 0.;

  hipfact_get_info(data->handle, "analyses", &analyses);
  hipfact_get_info(data->handle, "plan_swaps", &swaps);
  hipfact_get_info(data->handle, "dataflow_fallbacks", &fallbacks);
  hipfact_get_info(data->handle, "cache_hits", &hits);
  hipfact_get_info(data->handle, "num_refined", &refine);

  if (analyses != data->seen_analyses)
  {
    hipfact_get_info(data->handle, "analysis_s", &analysis_s);
    hipfact_get_info(data->handle, "nlevels", &nlevels);
    hipfact_get_info(data->handle, "nsuper", &nsuper);
    hipfact_get_info(data->handle, "maps_on", &maps);
    sleqp_log_debug("hipfact %s: symbolic analysis #%d (%.1f ms, %d fronts on "
                    "%d levels, %s structure)",
                    where,
                    (int)analyses,
                    1e3 * analysis_s,
                    (int)nsuper,
                    (int)nlevels,
                    maps != 0. ? "working-set superset" : "exact");
  }
  else if (swaps != data->seen_swaps)
  {
    sleqp_log_debug("hipfact %s: cached plan swapped in (%d swaps, %d cache "
                    "hits so far)",
                    where,
                    (int)swaps,
                    (int)hits);
  }
  if (fallbacks != data->seen_fallbacks)
  {
    sleqp_log_debug("hipfact %s: a dataflow launch timed out, per-level "
                    "launches from now on (%s)",
                    where,
                    hipfact_last_error(data->handle));
  }
  if (refine != data->seen_refine)
  {
    sleqp_log_debug("hipfact %s: iterative refinement continued behind the "
                    "passes the solve graph carries (%d solves so far)",
                    where,
                    (int)refine);
  }

  data->seen_analyses  = analyses;
  data->seen_swaps     = swaps;
  data->seen_fallbacks = fallbacks;
  data->seen_hits      = hits;
  data->seen_refine    = refine;
}

#define HIPFACT_CALL(data, x)                                                  \
  do                                                                           \
  {                                                                            \
    const int hipfact_status = (x);                                            \
    if (hipfact_status != HIPFACT_OK)                                          \
    {                                                                          \
      sleqp_raise(SLEQP_INTERNAL_ERROR,                                        \
                  "Caught hipfact error <%d> (%s)%s",                          \
                  hipfact_status,                                              \
                  hipfact_last_error((data)->handle),                          \
                  hipfact_status == HIPFACT_ESINGULAR                          \
                    ? " - K is singular and the right-hand side is not in "    \
                      "its range (rank-deficient working sets as such are "   \
                      "factored with static pivoting, like MA57's 'rank "      \
                      "deficient' success, fact_ma57.c:41-42)"                 \
                    : "");                                                     \
    }                                                                          \
  } while (0)

static SLEQP_RETCODE
hipfact_fact_set_matrix(void* fact_data, SleqpMat* matrix)
{
  HipFactData* data = (HipFactData*)fact_data;

  const int num_cols = sleqp_mat_num_cols(matrix);
  const int num_rows = sleqp_mat_num_rows(matrix);

  assert(num_cols == num_rows);

  data->num_rows = num_rows;

  /* uploads K (or only its values when the pattern is unchanged) and runs the
   * numeric factorisation on the device */
  HIPFACT_CALL(data,
               hipfact_set_matrix(data->handle,
                                  num_cols,
                                  sleqp_mat_cols(matrix),
                                  sleqp_mat_rows(matrix),
                                  sleqp_mat_data(matrix)));

  /* a rank-deficient working set is factored with static pivoting, as MA57 factors it ("Success - rank deficient",
   * fact_ma57.c:41-42: a positive status MA57_CHECK_ERROR lets pass, :118-133) - and said, which MA57's backend does
   * not do */
  {
    const char* warning = hipfact_last_warning(data->handle);
    if (warning)
    {
      sleqp_log_warn("hipfact: %s", warning);
    }
  }

  hipfact_log_events(data, "set_matrix");

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

  /* the dense solution stays in (page-locked) memory of the backend, like
   * ma57_data->rhs_sol (fact_ma57.c:713-730); it was sent there behind the
   * solve, this call waits for that transfer */
  const double* slice = NULL;

  HIPFACT_CALL(data, hipfact_solution_view(data->handle, &slice, begin, end));

  hipfact_log_events(data, "solution");

  /* sleqp_vec_set_from_raw (vec.c:71-103) walks the slice twice - count, then push - which is 57 us of the 250 us a
   * solve + solution costs at n = 1e5.  One walk: the capacity of the dense case is reserved (the vector keeps it
   * across calls, and projections / duals are dense), entries are pushed exactly as vec.c:88-100 does, in
   * ascending order with the same zero test */
  {
    const int dim = end - begin;
    SLEQP_CALL(sleqp_vec_clear(sol));
    SLEQP_CALL(sleqp_vec_resize(sol, dim));
    SLEQP_CALL(sleqp_vec_reserve(sol, dim));

    double* data = sol->data;
    int* indices = sol->indices;
    int k        = 0;

    for (int i = 0; i < dim; ++i)
    {
      const double v = slice[i];

      data[k]    = v;
      indices[k] = i;
      k += !(fabs(v) <= zero_eps);
    }

    sol->nnz = k;
  }

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

#ifdef HIPFACT_STANDALONE
  if (last_handle == data->handle)
  {
    last_handle = NULL;
  }
#endif

  hipfact_free(&data->handle);

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

#ifdef HIPFACT_STANDALONE
  last_handle = data->handle;
#endif

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

/* The PSD flavour (pattern: fact_cholmod.c:15, 231-262): declares
 * SLEQP_FACT_FLAGS_PSD | SLEQP_FACT_FLAGS_LOWER, so that create_aug_jac
 * (trial_point.c:94-101) puts the REDUCED AugJac in front of it and the matrix
 * handed over is the lower triangle of the symmetric positive definite
 * A_W A_W^T (reduced_aug_jac.c:323-377, 473).  Inside the library this is the
 * generic mode: the same supernodal engine on M = K with static 1x1 pivots. */
SLEQP_RETCODE
sleqp_fact_hipfact_psd_create(SleqpFact** star, SleqpSettings* settings)
{
  SleqpFactCallbacks callbacks = {.set_matrix = hipfact_fact_set_matrix,
                                  .solve      = hipfact_fact_solve,
                                  .solution   = hipfact_fact_solution,
                                  .condition  = hipfact_fact_condition,
                                  .free       = hipfact_fact_free};

  HipFactData* data = NULL;

  SLEQP_CALL(hipfact_data_create(&data));

  data->psd = true;

  /* never the saddle interpretation, whatever the pattern looks like */
  {
    int status = hipfact_set_option(data->handle, "force_generic", 1.);

    if (status == HIPFACT_OK)
    {
      status = hipfact_set_option(data->handle, "exact_pattern", 1.);
    }

    if (status != HIPFACT_OK)
    {
      /* nothing owns the backend data yet: release it before raising */
      void* fact_data = (void*)data;
      SLEQP_CALL(hipfact_fact_free(&fact_data));
      sleqp_raise(SLEQP_INTERNAL_ERROR,
                  "Failed to configure the hipfact PSD backend <%d>",
                  status);
    }
  }

  SLEQP_CALL(sleqp_fact_create(star,
                               SLEQP_FACT_HIPFACT_NAME "-psd",
                               SLEQP_FACT_HIPFACT_VERSION,
                               settings,
                               &callbacks,
                               SLEQP_FACT_FLAGS_PSD | SLEQP_FACT_FLAGS_LOWER,
                               (void*)data));

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_fact_create_default(SleqpFact** star, SleqpSettings* settings)
{
  SLEQP_CALL(sleqp_fact_hipfact_create(star, settings));

  return SLEQP_OKAY;
}
