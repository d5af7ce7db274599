/* tr_hipfact.c — the EQP trust-region subproblem solved on the device.
 *
 * A SleqpTRSolver (tr/tr_types.h:9-30) whose `solve` runs hipfact_tr_solve: the Krylov loop of the
 * reference's two solvers — the generalised Lanczos method behind trlib_krylov_min
 * (tr/trlib_solver.c:244-651) and the projected Steihaug CG (tr/steihaug_solver.c:218-496) — with
 * every n-vector resident in HBM.  Per iteration: one KKT projection (sleqp_aug_jac_project_nullspace
 * in the reference), one Hessian product, a few reductions; the host sees a handful of scalars.
 *
 * The Hessian of the Lagrangian is matrix-free in SLEQP (SLEQP_FUNC_HESS_PROD, func.c:373-408):
 * by default the product is computed by the problem's own callback on the host, with one n-vector
 * crossing PCIe in each direction per iteration (trlib_solver.c:542-602 does the same product per
 * iteration, plus the projection's two crossings that stay on the device here).  Callers that can
 * provide the Hessian as an explicit matrix (sleqp_hipfact_tr_set_hessian) avoid that as well.
 *
 * The projection comes from the factorisation inside the hipfact handle of the augmented Jacobian
 * (aug_jac_hipfact.c); the solver holds its own reference to that handle (sleqp_hipfact_tr_bind) —
 * no process-global state.
 */
#include "tr_hipfact.h"

#include <assert.h>
#include <stdint.h>
#include <string.h>

#ifndef HIPFACT_STANDALONE
#include "fail.h"
#include "mem.h"
#include "problem.h"
#endif

#include "hipfact.h"

struct SleqpHipfactTR
{
  SleqpProblem* problem;
  SleqpSettings* settings;

  hipfact_handle* handle; /* own reference (hipfact_retain) */
  hipfact_spmat* hessian; /* explicit Hessian, optional */
  int hess_nnz;
  uint64_t hess_pattern_hash;

  int method; /* HIPFACT_TR_GLTR / HIPFACT_TR_STEIHAUG */
  int max_iter;
  double rel_tol; /* stat_eps * tolerance_factor (steihaug_solver.c:21,241; trlib_solver.c:265) */
  double zero_eps;

  /* matrix-free product: sparse staging around sleqp_problem_hess_prod */
  const SleqpVec* multipliers; /* of the solve call in progress */
  SleqpVec* sparse_direction;
  SleqpVec* sparse_product;
  SLEQP_RETCODE callback_status;

  double* dense_gradient; /* num_variables */
  double* dense_step;     /* num_variables */

  /* of the last solve (steihaug_solver.c:229-230: 1 / 1 before the first) */
  double min_rayleigh;
  double max_rayleigh;
};

static uint64_t
pattern_hash(const SleqpMat* matrix)
{
  /* FNV-1a over the column pointers and row indices */
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

static SLEQP_RETCODE
hipfact_tr_free(void** star)
{
  SleqpHipfactTR* solver = (SleqpHipfactTR*)(*star);

  if (solver->hessian)
  {
    hipfact_spmat_free(&solver->hessian);
  }

  hipfact_free(&solver->handle); /* our reference */

  SLEQP_CALL(sleqp_vec_free(&solver->sparse_product));
  SLEQP_CALL(sleqp_vec_free(&solver->sparse_direction));

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
  /* steihaug_solver_rayleigh (steihaug_solver.c:173-185) / trlib_rayleigh (trlib_solver.c:654-662): the extremes the
   * last solve collected, here on the device (hipfact_tr_extra) */
  SleqpHipfactTR* solver = (SleqpHipfactTR*)solver_data;
  *min_rayleigh          = solver->min_rayleigh;
  *max_rayleigh          = solver->max_rayleigh;
  return SLEQP_OKAY;
}

/* hipfact_hess_prod_fn: dense host vectors <-> sleqp_problem_hess_prod on sparse vectors, exactly
 * the conversion the reference applies around the same call (trlib_solver.c:575-590: the direction
 * is a SleqpVec there already) */
static int
hess_prod_callback(void* user, const double* direction, double* product)
{
  SleqpHipfactTR* solver  = (SleqpHipfactTR*)user;
  const int num_variables = sleqp_problem_num_vars(solver->problem);

  SLEQP_RETCODE status
    = sleqp_vec_set_from_raw(solver->sparse_direction, (double*)direction, num_variables, solver->zero_eps);

  if (status == SLEQP_OKAY)
  {
    status = sleqp_problem_hess_prod(solver->problem,
                                     solver->sparse_direction,
                                     solver->multipliers,
                                     solver->sparse_product);
  }

  if (status == SLEQP_OKAY)
  {
    status = sleqp_vec_to_raw(solver->sparse_product, product);
  }

  solver->callback_status = status;

  return status == SLEQP_OKAY ? 0 : -1;
}

static SLEQP_RETCODE
tr_callback_solve(SleqpAugJac* jacobian,
                 const SleqpVec* multipliers,
                 const SleqpVec* gradient,
                 SleqpVec* newton_step,
                 double trust_radius,
                 double* tr_dual,
                 double time_limit,
                 void* solver_data)
{
  SleqpHipfactTR* solver = (SleqpHipfactTR*)solver_data;
  (void)jacobian; /* its factorisation is the bound handle */

  const int num_variables = sleqp_problem_num_vars(solver->problem);

  if (!solver->handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver: no factorisation bound (sleqp_hipfact_tr_bind)");
  }

  assert(gradient->dim == num_variables);

  SLEQP_CALL(sleqp_vec_to_raw(gradient, solver->dense_gradient));

  solver->multipliers     = multipliers;
  solver->callback_status = SLEQP_OKAY;

  int iterations = 0;
  /* time_limit: seconds or SLEQP_NONE (-1), tr/tr_solver.c:18-21,47 - the same convention as hipfact_tr_extra */
  hipfact_tr_extra extra = {.time_limit = time_limit, .timed_out = 0, .min_rayleigh = 1., .max_rayleigh = 1.};
  const int status       = hipfact_tr_solve_ex(solver->handle,
                                         solver->method,
                                         solver->hessian,
                                         solver->hessian ? NULL : hess_prod_callback,
                                         solver,
                                         solver->dense_gradient,
                                         trust_radius,
                                         solver->rel_tol,
                                         solver->max_iter,
                                         solver->dense_step,
                                         tr_dual,
                                         &iterations,
                                         &extra);

  solver->min_rayleigh = extra.min_rayleigh;
  solver->max_rayleigh = extra.max_rayleigh;

  solver->multipliers = NULL;

  /* an error raised inside the problem's Hessian product keeps its own message */
  SLEQP_CALL(solver->callback_status);

  if (status != HIPFACT_OK)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR,
                "Caught hipfact error <%d> (%s)",
                status,
                hipfact_last_error(solver->handle));
  }

  SLEQP_CALL(sleqp_vec_set_from_raw(newton_step, solver->dense_step, num_variables, solver->zero_eps));

  if (extra.timed_out)
  {
    /* steihaug_solver.c:490-492, trlib_solver.c:641-644 */
    return SLEQP_ABORT_TIME;
  }

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_tr_set_hessian(SleqpHipfactTR* solver, const SleqpMat* hess_lower)
{
  if (!hess_lower)
  {
    if (solver->hessian)
    {
      hipfact_spmat_free(&solver->hessian);
    }
    return SLEQP_OKAY;
  }

  const int num_variables = sleqp_problem_num_vars(solver->problem);

  assert(sleqp_mat_num_rows(hess_lower) == num_variables);
  assert(sleqp_mat_num_cols(hess_lower) == num_variables);

  if (!solver->handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver: bind a factorisation before setting the Hessian");
  }

  const int nnz       = sleqp_mat_nnz(hess_lower);
  const uint64_t hash = pattern_hash(hess_lower);

  if (solver->hessian && nnz == solver->hess_nnz && hash == solver->hess_pattern_hash)
  {
    /* same pattern (checked, not assumed): values only */
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

  const int status = hipfact_spmat_create(solver->handle,
                                          num_variables,
                                          num_variables,
                                          sleqp_mat_cols(hess_lower),
                                          sleqp_mat_rows(hess_lower),
                                          sleqp_mat_data(hess_lower),
                                          &solver->hessian);

  if (status != HIPFACT_OK)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "Caught hipfact error <%d> (%s)", status, hipfact_last_error(solver->handle));
  }

  solver->hess_nnz          = nnz;
  solver->hess_pattern_hash = hash;

  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_hipfact_tr_bind(SleqpHipfactTR* solver, struct hipfact_handle* handle)
{
  if (!handle)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver needs the handle of the hipfact augmented Jacobian");
  }

  if (solver->handle == handle)
  {
    return SLEQP_OKAY;
  }

  if (solver->hessian)
  {
    hipfact_spmat_free(&solver->hessian); /* lives on the old handle */
  }

  if (solver->handle)
  {
    hipfact_free(&solver->handle);
  }

  if (hipfact_retain(handle) != HIPFACT_OK)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "hipfact TR solver: cannot retain the factorisation handle");
  }

  solver->handle = handle;

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

  solver->min_rayleigh = 1.;
  solver->max_rayleigh = 1.;

  solver->problem = problem;
  SLEQP_CALL(sleqp_problem_capture(solver->problem));

  SLEQP_CALL(sleqp_settings_capture(settings));
  solver->settings = settings;

#ifdef HIPFACT_STANDALONE
  solver->max_iter                 = sleqp_settings_max_newton_iterations(settings);
  solver->rel_tol                  = sleqp_settings_stat_tol(settings) * 1e-2;
  solver->zero_eps                 = sleqp_settings_zero_eps(settings);
  const SLEQP_TR_SOLVER tr_solver = sleqp_settings_tr_solver(settings);
#else
  solver->max_iter = sleqp_settings_int_value(settings, SLEQP_SETTINGS_INT_MAX_NEWTON_ITERATIONS);
  solver->rel_tol  = sleqp_settings_real_value(settings, SLEQP_SETTINGS_REAL_STAT_TOL) * 1e-2;
  solver->zero_eps = sleqp_settings_real_value(settings, SLEQP_SETTINGS_REAL_ZERO_EPS);
  const SLEQP_TR_SOLVER tr_solver
    = (SLEQP_TR_SOLVER)sleqp_settings_enum_value(settings, SLEQP_SETTINGS_ENUM_TR_SOLVER);
#endif

  /* newton.c:97-109: CG is chosen explicitly (or by AUTO for PSD Hessians, which the caller decides
   * by creating this solver with CG); everything else is trlib's Lanczos method */
  solver->method = (tr_solver == SLEQP_TR_SOLVER_CG) ? HIPFACT_TR_STEIHAUG : HIPFACT_TR_GLTR;

  SLEQP_CALL(sleqp_alloc_array(&solver->dense_gradient, num_variables));
  SLEQP_CALL(sleqp_alloc_array(&solver->dense_step, num_variables));
  SLEQP_CALL(sleqp_vec_create_empty(&solver->sparse_direction, num_variables));
  SLEQP_CALL(sleqp_vec_create_empty(&solver->sparse_product, num_variables));

  SleqpTRCallbacks callbacks = {.solve = tr_callback_solve, .rayleigh = hipfact_tr_rayleigh, .free = hipfact_tr_free};

  SLEQP_CALL(sleqp_tr_solver_create(star, &callbacks, (void*)solver));

  *ctl = solver;

  return SLEQP_OKAY;
}
