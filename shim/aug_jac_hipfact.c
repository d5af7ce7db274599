/* aug_jac_hipfact.c — the optional second boundary: a SleqpAugJac
 * (aug_jac/aug_jac_types.h:27-35, created like aug_jac/standard_aug_jac.c:525-547)
 * that never builds K on the host.  set_iterate hands the constraint Jacobian
 * and the working-set index maps to hipfact_assemble_kkt, which restates
 * fill_aug_jac (standard_aug_jac.c:135-237) on the device, factors K there and
 * keeps it resident; the three solves are the same rhs padding / shifting as
 * standard_aug_jac.c:306-435, driven through the C ABI.
 *
 * Hook: in create_aug_jac (trial_point.c:105-108) replace
 *   sleqp_standard_aug_jac_create(star, problem, settings, fact)
 * by sleqp_hipfact_aug_jac_create(star, problem, settings, &handle) when
 * SLEQP_FACT_HIPFACT is the configured backend.
 */
#include "aug_jac_hipfact.h"

#include <assert.h>

#ifndef HIPFACT_STANDALONE
#include "fail.h"
#include "iterate.h"
#include "mem.h"
#include "problem.h"
#include "working_set.h"
#endif

#include "hipfact.h"

#include <stdbool.h>
#include <string.h>

typedef struct AugJacData
{
  SleqpProblem* problem;
  hipfact_handle* handle; /* one reference; the TR solver of tr_hipfact.c may hold another (hipfact_retain) */

  int working_set_size;
  double condition;
  double zero_eps;

  /* "Do not recompute for linear problems & unchanged working set" (standard_aug_jac.c:247-259):
   * K depends on the constraint Jacobian and on the two index maps only, so the previous maps
   * stand in for the reference's copy of the working set */
  bool fixed_jacobian;
  bool has_factorization;
  int* prev_var_index;
  int* prev_cons_index;

  int* var_index;  /* num_variables */
  int* cons_index; /* num_constraints */
} AugJacData;

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
aug_jac_set_iterate(SleqpIterate* iterate, void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  SleqpProblem* problem        = jacobian->problem;
  SleqpMat* cons_jac           = sleqp_iterate_cons_jac(iterate);
  SleqpWorkingSet* working_set = sleqp_iterate_working_set(iterate);

  const int num_variables   = sleqp_problem_num_vars(problem);
  const int num_constraints = sleqp_problem_num_cons(problem);

  for (int j = 0; j < num_variables; ++j)
  {
    jacobian->var_index[j] = sleqp_working_set_var_index(working_set, j);
  }

  for (int i = 0; i < num_constraints; ++i)
  {
    jacobian->cons_index[i] = sleqp_working_set_cons_index(working_set, i);
  }

  if (jacobian->fixed_jacobian && jacobian->has_factorization
      && memcmp(jacobian->var_index, jacobian->prev_var_index, (size_t)num_variables * sizeof(int)) == 0
      && memcmp(jacobian->cons_index, jacobian->prev_cons_index, (size_t)num_constraints * sizeof(int)) == 0)
  {
    return SLEQP_OKAY;
  }

  jacobian->has_factorization = false;
  jacobian->working_set_size  = sleqp_working_set_size(working_set);
  jacobian->condition         = SLEQP_NONE;

  /* assembly + factorisation on the device; K never exists on the host.  A working set inside the
   * structure analysed before costs a numeric refactorisation only (superset plan). */
  HIPFACT_CALL(jacobian,
               hipfact_assemble_kkt(jacobian->handle,
                                    num_variables,
                                    num_constraints,
                                    sleqp_mat_cols(cons_jac),
                                    sleqp_mat_rows(cons_jac),
                                    sleqp_mat_data(cons_jac),
                                    jacobian->var_index,
                                    jacobian->cons_index,
                                    jacobian->working_set_size,
                                    NULL,
                                    NULL,
                                    NULL,
                                    NULL));

  {
    /* rank-deficient working set: factored with static pivoting (fact_ma57.c:41-42), and said */
    const char* warning = hipfact_last_warning(jacobian->handle);
    if (warning)
    {
      sleqp_log_warn("hipfact: %s", warning);
    }
  }

  HIPFACT_CALL(jacobian, hipfact_condition(jacobian->handle, &jacobian->condition));

  memcpy(jacobian->prev_var_index, jacobian->var_index, (size_t)num_variables * sizeof(int));
  memcpy(jacobian->prev_cons_index, jacobian->cons_index, (size_t)num_constraints * sizeof(int));
  jacobian->has_factorization = true;

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
aug_jac_condition(bool* exact, double* condition, void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  *exact     = false;
  *condition = jacobian->condition;

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
solve_and_extract(AugJacData* jacobian, const SleqpVec* rhs, int total_size, SleqpVec* sol, int begin, int end)
{
  HIPFACT_CALL(jacobian, hipfact_solve_sparse(jacobian->handle, total_size, rhs->nnz, rhs->indices, rhs->data));

  /* the dense solution stays in page-locked memory of the backend (sent there
   * behind the solve); sparsified straight out of it like fact_ma57.c:713-730 */
  const double* slice = NULL;

  HIPFACT_CALL(jacobian, hipfact_solution_view(jacobian->handle, &slice, begin, end));

  SLEQP_CALL(sleqp_vec_set_from_raw(sol, slice, end - begin, jacobian->zero_eps));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
aug_jac_solve_min_norm(const SleqpVec* _rhs, SleqpVec* sol, void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  /* Cast away constness: the indices are shifted in place around the call,
   * exactly like standard_aug_jac.c:328-345 */
  SleqpVec* rhs = (SleqpVec*)_rhs;

  const int num_variables = sleqp_problem_num_vars(jacobian->problem);
  const int total_size    = num_variables + jacobian->working_set_size;

  assert(rhs->dim == jacobian->working_set_size);

  for (int k = 0; k < rhs->nnz; ++k)
  {
    rhs->indices[k] += num_variables;
  }

  SLEQP_RETCODE status = solve_and_extract(jacobian, rhs, total_size, sol, 0, num_variables);

  for (int k = 0; k < rhs->nnz; ++k)
  {
    rhs->indices[k] -= num_variables;
  }

  return status;
}

static SLEQP_RETCODE
aug_jac_solve_lsq(const SleqpVec* rhs, SleqpVec* sol, void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  const int num_variables = sleqp_problem_num_vars(jacobian->problem);
  const int total_size    = num_variables + jacobian->working_set_size;

  assert(rhs->dim == num_variables);

  /* "just add some zeros" (standard_aug_jac.c:375-376): the padded vector has
   * the same nonzeros, only a larger dimension */
  SLEQP_CALL(solve_and_extract(jacobian, rhs, total_size, sol, num_variables, total_size));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
aug_jac_project_nullspace(const SleqpVec* rhs, SleqpVec* sol, void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  const int num_variables = sleqp_problem_num_vars(jacobian->problem);
  const int total_size    = num_variables + jacobian->working_set_size;

  assert(rhs->dim == num_variables);

  SLEQP_CALL(solve_and_extract(jacobian, rhs, total_size, sol, 0, num_variables));

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
aug_jac_free(void* data)
{
  AugJacData* jacobian = (AugJacData*)data;

  hipfact_free(&jacobian->handle);

  sleqp_free(&jacobian->prev_cons_index);
  sleqp_free(&jacobian->prev_var_index);
  sleqp_free(&jacobian->cons_index);
  sleqp_free(&jacobian->var_index);

  SLEQP_CALL(sleqp_problem_release(&jacobian->problem));

  sleqp_free(&jacobian);

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_aug_jac_create(SleqpAugJac** star,
                             SleqpProblem* problem,
                             SleqpSettings* settings,
                             struct hipfact_handle** handle)
{
  AugJacData* jacobian = NULL;

  if (handle)
  {
    *handle = NULL;
  }

  SLEQP_CALL(sleqp_malloc(&jacobian));

  *jacobian = (AugJacData){0};

  const int num_variables   = sleqp_problem_num_vars(problem);
  const int num_constraints = sleqp_problem_num_cons(problem);

  SLEQP_CALL(sleqp_problem_capture(problem));
  jacobian->problem        = problem;
  jacobian->condition      = SLEQP_NONE;
  jacobian->fixed_jacobian = !(sleqp_problem_has_nonlinear_cons(problem));

#ifdef HIPFACT_STANDALONE
  jacobian->zero_eps = sleqp_settings_zero_eps(settings);
#else
  jacobian->zero_eps = sleqp_settings_real_value(settings, SLEQP_SETTINGS_REAL_ZERO_EPS);
#endif

  /* (the reference's allocation macros are comma expressions, pub_mem.h:14-45: parenthesise them) */
  SLEQP_RETCODE status = (sleqp_alloc_array(&jacobian->var_index, num_variables));
  if (status == SLEQP_OKAY) status = (sleqp_alloc_array(&jacobian->cons_index, num_constraints));
  if (status == SLEQP_OKAY) status = (sleqp_alloc_array(&jacobian->prev_var_index, num_variables));
  if (status == SLEQP_OKAY) status = (sleqp_alloc_array(&jacobian->prev_cons_index, num_constraints));

  int hipfact_status = HIPFACT_OK;

  if (status == SLEQP_OKAY)
  {
    hipfact_status = hipfact_create(&jacobian->handle, -1);
  }

  if (status != SLEQP_OKAY || hipfact_status != HIPFACT_OK)
  {
    /* nothing allocated above outlives a failed creation */
    (void)aug_jac_free(jacobian);

    if (status != SLEQP_OKAY)
    {
      return status;
    }

    sleqp_raise(SLEQP_INTERNAL_ERROR,
                "Failed to create hipfact backend <%d> (%s)",
                hipfact_status,
                hipfact_last_error(NULL));
  }

  SleqpAugJacCallbacks callbacks = {.set_iterate       = aug_jac_set_iterate,
                                    .solve_min_norm    = aug_jac_solve_min_norm,
                                    .solve_lsq         = aug_jac_solve_lsq,
                                    .project_nullspace = aug_jac_project_nullspace,
                                    .condition         = aug_jac_condition,
                                    .free              = aug_jac_free};

  SLEQP_CALL(sleqp_aug_jac_create(star, problem, &callbacks, jacobian));

  if (handle)
  {
    *handle = jacobian->handle; /* borrowed: hipfact_retain it to keep it beyond the AugJac's life */
  }

  return SLEQP_OKAY;
}
