/* sleqp_mini.h — stand-alone harness for the hipfact shim (HIPFACT_STANDALONE).
 *
 * A minimal, independently written implementation of the handful of SLEQP
 * entry points the shim touches, with the reference's names, argument meaning
 * and error behaviour, so that shim/fact_hipfact.c and shim/aug_jac_hipfact.c
 * can be compiled, loaded and exercised without libsleqp.  Inside a SLEQP
 * checkout the shim includes the real headers instead and this file is unused.
 *
 * Interfaces mirrored (chrhansk/sleqp v1.0.2, src/main/): pub_types.h:27-88
 * (return codes, SLEQP_CALL), pub_error.h:16-48 (sleqp_raise), pub_mem.h:14-48,
 * sparse/pub_vec.h:16-25 (public SleqpVec struct), sparse/pub_mat.h:65-114
 * (SleqpMat accessors), fact/fact.h:9-70, fact/fact_types.h:9-32,
 * aug_jac/aug_jac.h:11-92, aug_jac/aug_jac_types.h:9-35,
 * pub_working_set.h (index accessors), iterate / problem accessors.
 */
#ifndef SLEQP_MINI_H
#define SLEQP_MINI_H

#include <stdbool.h>
#include <stddef.h>
#include <stdlib.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLEQP_WARNUNUSED __attribute__((warn_unused_result))
#define SLEQP_NONE (-1)

typedef enum
{
  SLEQP_ERROR      = -1,
  SLEQP_OKAY       = 0,
  SLEQP_ABORT_TIME = 1,
} SLEQP_RETCODE;

typedef enum
{
  SLEQP_FAILED_ASSERTION,
  SLEQP_NOMEM,
  SLEQP_INTERNAL_ERROR,
  SLEQP_FUNC_EVAL_ERROR,
  SLEQP_CALLBACK_ERROR,
  SLEQP_MATH_ERROR,
  SLEQP_INVALID_DERIV,
  SLEQP_ILLEGAL_ARGUMENT
} SLEQP_ERROR_TYPE;

SLEQP_ERROR_TYPE
sleqp_error_type(void);
const char*
sleqp_error_msg(void);

/* pub_log.h:9-95: levels, global level / handler, the sleqp_log_* macros */
typedef enum
{
  SLEQP_LOG_SILENT     = 0,
  SLEQP_LOG_ERROR      = 1,
  SLEQP_LOG_WARN       = 2,
  SLEQP_LOG_INFO       = 3,
  SLEQP_LOG_DEBUG      = 4,
  SLEQP_NUM_LOG_LEVELS = 5
} SLEQP_LOG_LEVEL;
#include <time.h>
typedef void (*SLEQP_LOG_HANDLER)(SLEQP_LOG_LEVEL level, time_t time, const char* message);
SLEQP_LOG_LEVEL
sleqp_log_level(void);
void
sleqp_log_set_level(SLEQP_LOG_LEVEL level);
void
sleqp_log_set_handler(SLEQP_LOG_HANDLER handler);
void
sleqp_log_msg_level(int level, const char* fmt, ...);
#define sleqp_log_log_msg(level, ...)                                          \
  do                                                                           \
  {                                                                            \
    if (sleqp_log_level() >= level)                                            \
    {                                                                          \
      sleqp_log_msg_level(level, __VA_ARGS__);                                 \
    }                                                                          \
  } while (0)
#define sleqp_log_info(...) sleqp_log_log_msg(SLEQP_LOG_INFO, __VA_ARGS__)
#define sleqp_log_warn(...) sleqp_log_log_msg(SLEQP_LOG_WARN, __VA_ARGS__)
#define sleqp_log_error(...) sleqp_log_log_msg(SLEQP_LOG_ERROR, __VA_ARGS__)
#define sleqp_log_debug(...) sleqp_log_log_msg(SLEQP_LOG_DEBUG, __VA_ARGS__)
/* harness only: the messages logged so far (newline separated), cleared by the call */
const char*
sleqp_mini_log_drain(void);
void
sleqp_set_error(const char* file, int line, const char* func, SLEQP_ERROR_TYPE error_type, const char* fmt, ...)
  __attribute__((__format__(__printf__, 5, 6)));

#define sleqp_raise(error_type, fmt, ...)                                                                    \
  do                                                                                                         \
  {                                                                                                          \
    sleqp_set_error(__FILE__, __LINE__, __func__, error_type, fmt, ##__VA_ARGS__);                           \
    return SLEQP_ERROR;                                                                                      \
  } while (false)

#define SLEQP_CALL(x)                                                                                        \
  do                                                                                                         \
  {                                                                                                          \
    const SLEQP_RETCODE _status = (x);                                                                       \
    if (_status != SLEQP_OKAY)                                                                               \
    {                                                                                                        \
      return _status;                                                                                        \
    }                                                                                                        \
  } while (0)

/* ---- memory (pub_mem.h) ---- */
#define sleqp_allocate_memory(ptr, size)                                                                     \
  (((size) == 0) ? ((*(ptr) = NULL), SLEQP_OKAY)                                                             \
                 : (((*(ptr) = malloc(size)) != NULL) ? SLEQP_OKAY : sleqp_mini_nomem(__FILE__, __LINE__)))
#define sleqp_reallocate_memory(ptr, size)                                                                   \
  (((size) == 0) ? ((free(*(ptr)), (*(ptr) = NULL)), SLEQP_OKAY) : sleqp_mini_realloc((void**)(ptr), (size)))
#define sleqp_malloc(ptr) sleqp_allocate_memory(ptr, sizeof(**ptr))
#define sleqp_alloc_array(ptr, count) sleqp_allocate_memory(ptr, ((size_t)(count)) * sizeof(**ptr))
#define sleqp_realloc(ptr, count) sleqp_reallocate_memory(ptr, ((size_t)(count)) * sizeof(**ptr))
#define sleqp_free(ptr)                                                                                      \
  do                                                                                                         \
  {                                                                                                          \
    free(*(ptr));                                                                                            \
    *(ptr) = NULL;                                                                                           \
  } while (false)
SLEQP_RETCODE
sleqp_mini_nomem(const char* file, int line);
SLEQP_RETCODE
sleqp_mini_realloc(void** ptr, size_t size);

/* ---- settings: opaque, only captured / released ---- */
typedef struct SleqpSettings SleqpSettings;
SLEQP_RETCODE
sleqp_settings_create(SleqpSettings** star);
SLEQP_RETCODE
sleqp_settings_release(SleqpSettings** star);
double
sleqp_settings_zero_eps(const SleqpSettings* settings); /* SLEQP_SETTINGS_REAL_ZERO_EPS, default 1e-20 */
SLEQP_RETCODE
sleqp_settings_capture(SleqpSettings* settings);
double
sleqp_settings_stat_tol(const SleqpSettings* settings); /* SLEQP_SETTINGS_REAL_STAT_TOL, default 1e-6 */
int
sleqp_settings_max_newton_iterations(const SleqpSettings* settings); /* SLEQP_SETTINGS_INT_MAX_NEWTON_ITERATIONS, default 100 */
SLEQP_RETCODE
sleqp_settings_set_newton(SleqpSettings* settings, double stat_tol, int max_newton_iterations); /* harness only */
/* SLEQP_SETTINGS_ENUM_TR_SOLVER (pub_types.h:134-140), default AUTO */
typedef enum
{
  SLEQP_TR_SOLVER_TRLIB = 0,
  SLEQP_TR_SOLVER_CG,
  SLEQP_TR_SOLVER_LSQR,
  SLEQP_TR_SOLVER_AUTO
} SLEQP_TR_SOLVER;
SLEQP_TR_SOLVER
sleqp_settings_tr_solver(const SleqpSettings* settings);
void
sleqp_settings_set_tr_solver(SleqpSettings* settings, SLEQP_TR_SOLVER value); /* harness only */

/* ---- sparse vector (public struct, sparse/pub_vec.h:16-25) ---- */
typedef struct SleqpVec
{
  double* data;
  int* indices;

  int dim;
  int nnz;
  int nnz_max;
} SleqpVec;

SLEQP_RETCODE
sleqp_vec_create(SleqpVec** vec, int dim, int nnz_max);
SLEQP_RETCODE
sleqp_vec_create_empty(SleqpVec** vec, int dim);
SLEQP_RETCODE
sleqp_vec_create_full(SleqpVec** vec, int dim);
SLEQP_RETCODE
sleqp_vec_push(SleqpVec* vec, int idx, double value);
SLEQP_RETCODE
sleqp_vec_clear(SleqpVec* vec);
SLEQP_RETCODE
sleqp_vec_reserve(SleqpVec* vec, int nnz);
SLEQP_RETCODE
sleqp_vec_resize(SleqpVec* vec, int dim);
SLEQP_RETCODE
sleqp_vec_set_from_raw(SleqpVec* vec, const double* values, int dim, double zero_eps);
SLEQP_RETCODE
sleqp_vec_to_raw(const SleqpVec* vec, double* values);
SLEQP_RETCODE
sleqp_vec_free(SleqpVec** vec);

/* ---- sparse matrix (CSC, sparse/pub_mat.h) ---- */
typedef struct SleqpMat SleqpMat;
SLEQP_RETCODE
sleqp_mat_create(SleqpMat** matrix, int num_rows, int num_cols, int nnz_max);
SLEQP_RETCODE
sleqp_mat_reserve(SleqpMat* matrix, int nnz);
SLEQP_RETCODE
sleqp_mat_resize(SleqpMat* matrix, int num_rows, int num_cols);
SLEQP_RETCODE
sleqp_mat_clear(SleqpMat* matrix);
SLEQP_RETCODE
sleqp_mat_push(SleqpMat* matrix, int row, int col, double value);
SLEQP_RETCODE
sleqp_mat_push_col(SleqpMat* matrix, int col);
int
sleqp_mat_num_cols(const SleqpMat* matrix);
int
sleqp_mat_num_rows(const SleqpMat* matrix);
int
sleqp_mat_nnz(const SleqpMat* matrix);
double*
sleqp_mat_data(const SleqpMat* matrix);
int*
sleqp_mat_cols(const SleqpMat* matrix);
int*
sleqp_mat_rows(const SleqpMat* matrix);
SLEQP_RETCODE
sleqp_mat_release(SleqpMat** matrix);
/* harness only: bulk load of CSC arrays (what nnz calls of sleqp_mat_push_col / sleqp_mat_push would build) */
SLEQP_RETCODE
sleqp_mat_set_arrays_mini(SleqpMat* matrix, const int* cols, const int* rows, const double* data, int nnz);

/* ---- SleqpFact (fact/fact.h, fact/fact_types.h) ---- */
typedef struct SleqpFact SleqpFact;

typedef SLEQP_RETCODE (*SLEQP_FACT_SET_MATRIX)(void* fact_data, SleqpMat* matrix);
typedef SLEQP_RETCODE (*SLEQP_FACT_SOLVE)(void* fact_data, const SleqpVec* rhs);
typedef SLEQP_RETCODE (*SLEQP_FACT_SOLUTION)(void* fact_data, SleqpVec* sol, int begin, int end, double zero_eps);
typedef SLEQP_RETCODE (*SLEQP_FACT_CONDITION)(void* fact_data, double* condition);
typedef SLEQP_RETCODE (*SLEQP_FACT_FREE)(void** star);

typedef struct
{
  SLEQP_FACT_SET_MATRIX set_matrix;
  SLEQP_FACT_SOLVE solve;
  SLEQP_FACT_SOLUTION solution;
  SLEQP_FACT_CONDITION condition;
  SLEQP_FACT_FREE free;
} SleqpFactCallbacks;

typedef enum
{
  SLEQP_FACT_FLAGS_NONE  = 0,
  SLEQP_FACT_FLAGS_PSD   = (1 << 0),
  SLEQP_FACT_FLAGS_LOWER = (1 << 1)
} SLEQP_FACT_FLAGS;

SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_create(SleqpFact** star, const char* name, const char* version, SleqpSettings* settings,
                  SleqpFactCallbacks* callbacks, SLEQP_FACT_FLAGS flags, void* fact_data);
const char*
sleqp_fact_name(SleqpFact* factorization);
const char*
sleqp_fact_version(SleqpFact* factorization);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_set_matrix(SleqpFact* factorization, SleqpMat* matrix);
SLEQP_FACT_FLAGS
sleqp_fact_flags(SleqpFact* factorization);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_create_default(SleqpFact** star, SleqpSettings* settings);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_capture(SleqpFact* factorization);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_solve(SleqpFact* factorization, const SleqpVec* rhs);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_solution(SleqpFact* factorization, SleqpVec* sol, int begin, int end, double zero_eps);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_cond(SleqpFact* factorization, double* condition);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_fact_release(SleqpFact** star);

/* ---- problem / working set / iterate: only what an AugJac reads ---- */
typedef struct SleqpProblem SleqpProblem;
typedef struct SleqpWorkingSet SleqpWorkingSet;
typedef struct SleqpIterate SleqpIterate;

typedef enum
{
  SLEQP_INACTIVE     = 0,
  SLEQP_ACTIVE_LOWER = (1 << 0),
  SLEQP_ACTIVE_UPPER = (1 << 1),
  SLEQP_ACTIVE_BOTH  = (SLEQP_ACTIVE_LOWER | SLEQP_ACTIVE_UPPER),
} SLEQP_ACTIVE_STATE;

SLEQP_RETCODE
sleqp_problem_create_mini(SleqpProblem** star, int num_vars, int num_cons);
int
sleqp_problem_num_vars(const SleqpProblem* problem);
int
sleqp_problem_num_cons(const SleqpProblem* problem);
/* problem.h:14; the mini problem is nonlinear unless told otherwise */
bool
sleqp_problem_has_nonlinear_cons(SleqpProblem* problem);
void
sleqp_problem_set_nonlinear_cons_mini(SleqpProblem* problem, bool value);
/* problem.h:54-59: product = Hessian of the Lagrangian (x, cons_duals) times direction - the matrix-free
 * SLEQP_FUNC_HESS_PROD of the user (pub_func.h:168-172).  The mini problem forwards to a callback on dense
 * arrays installed by the test. */
typedef int (*SleqpMiniHessProd)(const double* direction, const double* cons_duals, double* product, void* data);
void
sleqp_problem_set_hess_prod_mini(SleqpProblem* problem, SleqpMiniHessProd callback, void* data);
SLEQP_RETCODE
sleqp_problem_capture(SleqpProblem* problem);
SLEQP_RETCODE
sleqp_problem_release(SleqpProblem** star);

SLEQP_RETCODE
sleqp_problem_hess_prod(SleqpProblem* problem, const struct SleqpVec* direction, const struct SleqpVec* cons_duals,
                        struct SleqpVec* product);

SLEQP_RETCODE
sleqp_working_set_create(SleqpWorkingSet** star, SleqpProblem* problem);
SLEQP_RETCODE
sleqp_working_set_reset(SleqpWorkingSet* working_set);
SLEQP_RETCODE
sleqp_working_set_add_var(SleqpWorkingSet* working_set, int index, SLEQP_ACTIVE_STATE state);
SLEQP_RETCODE
sleqp_working_set_add_cons(SleqpWorkingSet* working_set, int index, SLEQP_ACTIVE_STATE state);
int
sleqp_working_set_var_index(const SleqpWorkingSet* working_set, int index);
int
sleqp_working_set_cons_index(const SleqpWorkingSet* working_set, int index);
int
sleqp_working_set_num_active_vars(const SleqpWorkingSet* working_set);
int
sleqp_working_set_num_active_cons(const SleqpWorkingSet* working_set);
int
sleqp_working_set_size(const SleqpWorkingSet* working_set);
SLEQP_RETCODE
sleqp_working_set_release(SleqpWorkingSet** star);

SLEQP_RETCODE
sleqp_iterate_create_mini(SleqpIterate** star, SleqpProblem* problem);
SleqpMat*
sleqp_iterate_cons_jac(const SleqpIterate* iterate);
SleqpWorkingSet*
sleqp_iterate_working_set(const SleqpIterate* iterate);
SLEQP_RETCODE
sleqp_iterate_release(SleqpIterate** star);

/* ---- SleqpAugJac (aug_jac/aug_jac.h, aug_jac/aug_jac_types.h) ---- */
typedef struct SleqpAugJac SleqpAugJac;

typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_SET_ITERATE)(SleqpIterate* iterate, void* data);
typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_SOLVE_MIN_NORM)(const SleqpVec* rhs, SleqpVec* sol, void* data);
typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_SOLVE_LSQ)(const SleqpVec* rhs, SleqpVec* sol, void* data);
typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_PROJECT_NULLSPACE)(const SleqpVec* rhs, SleqpVec* sol, void* data);
typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_CONDITION)(bool* exact, double* condition, void* data);
typedef SLEQP_RETCODE (*SLEQP_AUG_JAC_FREE)(void* data);

typedef struct
{
  SLEQP_AUG_JAC_SET_ITERATE set_iterate;
  SLEQP_AUG_JAC_SOLVE_MIN_NORM solve_min_norm;
  SLEQP_AUG_JAC_SOLVE_LSQ solve_lsq;
  SLEQP_AUG_JAC_PROJECT_NULLSPACE project_nullspace;
  SLEQP_AUG_JAC_CONDITION condition;
  SLEQP_AUG_JAC_FREE free;
} SleqpAugJacCallbacks;

SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_create(SleqpAugJac** star, SleqpProblem* problem, SleqpAugJacCallbacks* callbacks, void* aug_jac_data);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_set_iterate(SleqpAugJac* aug_jac, SleqpIterate* iterate);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_solve_min_norm(SleqpAugJac* aug_jac, const SleqpVec* rhs, SleqpVec* sol);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_solve_lsq(SleqpAugJac* aug_jac, const SleqpVec* rhs, SleqpVec* sol);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_project_nullspace(SleqpAugJac* aug_jac, const SleqpVec* rhs, SleqpVec* sol);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_condition(SleqpAugJac* aug_jac, bool* exact, double* condition);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_aug_jac_release(SleqpAugJac** star);

/* ---- SleqpTRSolver (tr/tr_solver.h, tr/tr_types.h) ---- */
typedef struct SleqpTRSolver SleqpTRSolver;

typedef SLEQP_RETCODE (*SLEQP_TR_SOLVER_SOLVE)(SleqpAugJac* jacobian, const SleqpVec* multipliers,
                                               const SleqpVec* gradient, SleqpVec* newton_step, double trust_radius,
                                               double* tr_dual, double time_limit, void* solver_data);
typedef SLEQP_RETCODE (*SLEQP_TR_SOLVER_RAYLEIGH)(double* min_rayleigh, double* max_rayleigh, void* solver_data);
typedef SLEQP_RETCODE (*SLEQP_TR_SOLVER_FREE)(void** solver_data);

typedef struct
{
  SLEQP_TR_SOLVER_SOLVE solve;
  SLEQP_TR_SOLVER_RAYLEIGH rayleigh;
  SLEQP_TR_SOLVER_FREE free;
} SleqpTRCallbacks;

SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_tr_solver_create(SleqpTRSolver** star, SleqpTRCallbacks* callbacks, void* solver_data);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_tr_solver_solve(SleqpTRSolver* solver, SleqpAugJac* jacobian, const SleqpVec* multipliers,
                      const SleqpVec* gradient, SleqpVec* newton_step, double trust_radius, double* tr_dual);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_tr_solver_set_time_limit(SleqpTRSolver* solver, double time_limit); /* tr/tr_solver.h:26 */
SLEQP_RETCODE
sleqp_tr_solver_current_rayleigh(SleqpTRSolver* solver, double* min_rayleigh, double* max_rayleigh);
SLEQP_WARNUNUSED SLEQP_RETCODE
sleqp_tr_solver_release(SleqpTRSolver** star);

#ifdef __cplusplus
}
#endif

#endif /* SLEQP_MINI_H */
