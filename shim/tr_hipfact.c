/* tr_hipfact.c — the EQP trust-region subproblem solved on the device.
 *
 * A SleqpTRSolver (tr/tr_types.h:9-30, created like tr/steihaug_solver.c:498-536) whose `solve`
 * runs hipfact_steihaug_solve: the loop of steihaug_solver_solve (tr/steihaug_solver.c:218-496)
 * with every CG vector resident in HBM — per iteration one KKT projection
 * (sleqp_aug_jac_project_nullspace), one symmetric Hessian product and three reductions; the host
 * sees three scalars per iteration instead of two PCIe hops and a sparse <-> dense marshal.
 *
 * The Hessian of the Lagrangian must be available as an explicit lower-triangular CSC matrix
 * (sleqp_hipfact_tr_set_hessian) instead of the matrix-free hess_prod callback; `multipliers`
 * of the solve callback are therefore not used.  The projection comes from the SleqpAugJac of
 * aug_jac_hipfact.c (the factorisation lives in its hipfact handle), found through
 * sleqp_hipfact_aug_jac_handle.
 */
#include "tr_hipfact.h"

#include <assert.h>

#ifndef HIPFACT_STANDALONE
#include "error.h"
#include "fail.h"
#include "mem.h"
#include "problem.h"
#endif

#include "aug_jac_hipfact.h"
#include "hipfact.h"

struct SleqpHipfactTR
{
  SleqpProblem* problem;
  SleqpSettings* settings;

  hipfact_handle* hess_owner; /* handle the device copy of the Hessian was created on */
  hipfact_spmat* hessian;
  int hess_nnz;

  int max_iter;
  double rel_tol; /* stat_eps * tolerance_factor (steihaug_solver.c:21,241) */
  double zero_eps;

  double* dense_gradient; /* num_variables */
  double* dense_step;     /* num_variables */
};

static SLEQP_RETCODE
hipfact_tr_free(void** star)
{
  SleqpHipfactTR* solver = (SleqpHipfactTR*)(*star);

  if (solver->hessian)
  {
    hipfact_spmat_free(&solver->hessian);
  }

  sleqp_free(&solver->dense_step);
  sleqp_free(&solver->dense_gradient);

  SLEQP_CALL(sleqp_settings_release(&solver->settings));
  SLEQP_CALL(sleqp_problem_release(&solver->problem));

  sleqp_free(&solver);
  *star = NULL;

  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_tr_rayleigh(double* min_rayleigh, double* max_rayleigh, void* solver_data)
{
  /* not tracked on the device; 1 / 1 is what steihaug_solver_solve starts from (:229-230) */
  (void)solver_data;
  *min_rayleigh = 1.;
  *max_rayleigh = 1.;
  return SLEQP_OKAY;
}

static SLEQP_RETCODE
hipfact_tr_solve(SleqpAugJac* jacobian,
                 const SleqpVec* multipliers,
                 const SleqpVec* gradient,
                 SleqpVec* newton_step,
                 double trust_radius,
                 double* tr_dual,
                 double time_limit,
                 void* solver_data)
{
  SleqpHipfactTR* solver = (SleqpHipfactTR*)solver_data;
  (void)multipliers; /* the Hessian was supplied for the current multipliers */
  (void)time_limit;

  const int num_variables = sleqp_problem_num_vars(solver->problem);

  hipfact_handle* handle = sleqp_hipfact_aug_jac_handle(jacobian);

  if (!handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver needs the hipfact augmented Jacobian");
  }

  if (!solver->hessian || solver->hess_owner != handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver: no Hessian set for this augmented Jacobian");
  }

  assert(gradient->dim == num_variables);

  SLEQP_CALL(sleqp_vec_to_raw(gradient, solver->dense_gradient));

  int iterations   = 0;
  const int status = hipfact_steihaug_solve(handle,
                                            solver->hessian,
                                            solver->dense_gradient,
                                            trust_radius,
                                            solver->rel_tol,
                                            solver->max_iter,
                                            solver->dense_step,
                                            tr_dual,
                                            &iterations);

  if (status != HIPFACT_OK)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "Caught hipfact error <%d> (%s)", status, hipfact_last_error(handle));
  }

  SLEQP_CALL(sleqp_vec_set_from_raw(newton_step, solver->dense_step, num_variables, solver->zero_eps));

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_tr_set_hessian(SleqpHipfactTR* solver, const SleqpMat* hess_lower)
{
  /* bound to the factorisation handle at the first solve's augmented Jacobian: the device copy is
   * (re)created on demand by sleqp_hipfact_tr_bind */
  const int num_variables = sleqp_problem_num_vars(solver->problem);

  assert(sleqp_mat_num_rows(hess_lower) == num_variables);
  assert(sleqp_mat_num_cols(hess_lower) == num_variables);

  if (!solver->hess_owner)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver: bind an augmented Jacobian before setting the Hessian");
  }

  const int nnz = sleqp_mat_nnz(hess_lower);

  if (solver->hessian && nnz == solver->hess_nnz)
  {
    /* same pattern assumed by the caller (SQP iterations): values only */
    const int status = hipfact_spmat_update_values(solver->hessian, sleqp_mat_data(hess_lower));
    if (status == HIPFACT_OK)
    {
      return SLEQP_OKAY;
    }
  }

  if (solver->hessian)
  {
    hipfact_spmat_free(&solver->hessian);
  }

  const int status = hipfact_spmat_create(solver->hess_owner,
                                          num_variables,
                                          num_variables,
                                          sleqp_mat_cols(hess_lower),
                                          sleqp_mat_rows(hess_lower),
                                          sleqp_mat_data(hess_lower),
                                          &solver->hessian);

  if (status != HIPFACT_OK)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "Caught hipfact error <%d> (%s)", status, hipfact_last_error(solver->hess_owner));
  }

  solver->hess_nnz = nnz;

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_tr_bind(SleqpHipfactTR* solver, SleqpAugJac* jacobian)
{
  hipfact_handle* handle = sleqp_hipfact_aug_jac_handle(jacobian);

  if (!handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver needs the hipfact augmented Jacobian");
  }

  if (solver->hess_owner != handle && solver->hessian)
  {
    hipfact_spmat_free(&solver->hessian);
  }

  solver->hess_owner = handle;

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_tr_solver_create(SleqpTRSolver** star,
                               SleqpHipfactTR** ctl,
                               SleqpProblem* problem,
                               SleqpSettings* settings)
{
  SleqpHipfactTR* solver = NULL;

  const int num_variables = sleqp_problem_num_vars(problem);

  SLEQP_CALL(sleqp_malloc(&solver));

  *solver = (SleqpHipfactTR){0};

  solver->problem = problem;
  SLEQP_CALL(sleqp_problem_capture(solver->problem));

  SLEQP_CALL(sleqp_settings_capture(settings));
  solver->settings = settings;

#ifdef HIPFACT_STANDALONE
  solver->max_iter = sleqp_settings_max_newton_iterations(settings);
  solver->rel_tol  = sleqp_settings_stat_tol(settings) * 1e-2;
  solver->zero_eps = sleqp_settings_zero_eps(settings);
#else
  solver->max_iter = sleqp_settings_int_value(settings, SLEQP_SETTINGS_INT_MAX_NEWTON_ITERATIONS);
  solver->rel_tol  = sleqp_settings_real_value(settings, SLEQP_SETTINGS_REAL_STAT_TOL) * 1e-2;
  solver->zero_eps = sleqp_settings_real_value(settings, SLEQP_SETTINGS_REAL_ZERO_EPS);
#endif

  SLEQP_CALL(sleqp_alloc_array(&solver->dense_gradient, num_variables));
  SLEQP_CALL(sleqp_alloc_array(&solver->dense_step, num_variables));

  SleqpTRCallbacks callbacks = {.solve = hipfact_tr_solve, .rayleigh = hipfact_tr_rayleigh, .free = hipfact_tr_free};

  SLEQP_CALL(sleqp_tr_solver_create(star, &callbacks, (void*)solver));

  *ctl = solver;

  return SLEQP_OKAY;
}
