/* sleqp_mini.c — stand-alone implementation of the SLEQP entry points declared
 * in sleqp_mini.h (harness for the hipfact shim; not part of the product
 * library).  Written from the interface descriptions, semantics as in the
 * reference (file:line cited in sleqp_mini.h). */
#define _POSIX_C_SOURCE 200809L
#include "sleqp_mini.h"

#include <assert.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

/* ---- error slot (thread local, like error.c:7-8) ---- */
static _Thread_local SLEQP_ERROR_TYPE g_error_type = SLEQP_INTERNAL_ERROR;
static _Thread_local char g_error_msg[1024];

SLEQP_ERROR_TYPE
sleqp_error_type(void) { return g_error_type; }
const char*
sleqp_error_msg(void) { return g_error_msg; }

/* ---- log (log.c): global level + handler; the harness keeps the messages for the tests */
static SLEQP_LOG_LEVEL g_log_level = SLEQP_LOG_INFO;
static SLEQP_LOG_HANDLER g_log_handler = NULL;
static char g_log_buf[1 << 16];
static char g_log_out[1 << 16];
static size_t g_log_len = 0;
SLEQP_LOG_LEVEL
sleqp_log_level(void) { return g_log_level; }
void
sleqp_log_set_level(SLEQP_LOG_LEVEL level) { g_log_level = level; }
void
sleqp_log_set_handler(SLEQP_LOG_HANDLER handler) { g_log_handler = handler; }
void
sleqp_log_msg_level(int level, const char* fmt, ...)
{
  char msg[2048];
  va_list args;
  va_start(args, fmt);
  vsnprintf(msg, sizeof msg, fmt, args);
  va_end(args);
  if (g_log_handler)
  {
    g_log_handler((SLEQP_LOG_LEVEL)level, time(NULL), msg);
  }
  const size_t len = strlen(msg);
  if (g_log_len + len + 2 < sizeof g_log_buf)
  {
    memcpy(g_log_buf + g_log_len, msg, len);
    g_log_len += len;
    g_log_buf[g_log_len++] = '\n';
    g_log_buf[g_log_len]   = 0;
  }
}
const char*
sleqp_mini_log_drain(void)
{
  memcpy(g_log_out, g_log_buf, g_log_len + 1);
  g_log_out[g_log_len] = 0;
  g_log_len            = 0;
  g_log_buf[0]         = 0;
  return g_log_out;
}

void
sleqp_set_error(const char* file, int line, const char* func, SLEQP_ERROR_TYPE error_type, const char* fmt, ...)
{
  (void)file;
  (void)line;
  (void)func;
  g_error_type = error_type;
  va_list args;
  va_start(args, fmt);
  vsnprintf(g_error_msg, sizeof g_error_msg, fmt, args);
  va_end(args);
}

SLEQP_RETCODE
sleqp_mini_nomem(const char* file, int line)
{
  sleqp_set_error(file, line, "alloc", SLEQP_NOMEM, "Failed to allocate memory");
  return SLEQP_ERROR;
}

SLEQP_RETCODE
sleqp_mini_realloc(void** ptr, size_t size)
{
  void* p = realloc(*ptr, size);
  if (!p)
    return sleqp_mini_nomem(__FILE__, __LINE__);
  *ptr = p;
  return SLEQP_OKAY;
}

/* ---- settings ---- */
struct SleqpSettings
{
  int refcount;
  double zero_eps;
  double stat_tol;
  int max_newton_iterations;
  SLEQP_TR_SOLVER tr_solver;
};

SLEQP_TR_SOLVER
sleqp_settings_tr_solver(const SleqpSettings* settings) { return settings->tr_solver; }
void
sleqp_settings_set_tr_solver(SleqpSettings* settings, SLEQP_TR_SOLVER value) { settings->tr_solver = value; }

SLEQP_RETCODE
sleqp_settings_create(SleqpSettings** star)
{
  SLEQP_CALL(sleqp_malloc(star));
  (*star)->refcount = 1;
  (*star)->zero_eps = 1e-20;
  (*star)->stat_tol = 1e-6;
  (*star)->max_newton_iterations = 100;
  (*star)->tr_solver             = SLEQP_TR_SOLVER_AUTO;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_settings_release(SleqpSettings** star)
{
  if (*star && --(*star)->refcount == 0)
    free(*star);
  *star = NULL;
  return SLEQP_OKAY;
}

double
sleqp_settings_zero_eps(const SleqpSettings* settings) { return settings ? settings->zero_eps : 1e-20; }

SLEQP_RETCODE
sleqp_settings_capture(SleqpSettings* settings)
{
  ++settings->refcount;
  return SLEQP_OKAY;
}

double
sleqp_settings_stat_tol(const SleqpSettings* settings) { return settings ? settings->stat_tol : 1e-6; }

int
sleqp_settings_max_newton_iterations(const SleqpSettings* settings) { return settings ? settings->max_newton_iterations : 100; }

SLEQP_RETCODE
sleqp_settings_set_newton(SleqpSettings* settings, double stat_tol, int max_newton_iterations)
{
  settings->stat_tol              = stat_tol;
  settings->max_newton_iterations = max_newton_iterations;
  return SLEQP_OKAY;
}

/* ---- vectors ---- */
SLEQP_RETCODE
sleqp_vec_create(SleqpVec** vstar, int dim, int nnz_max)
{
  SLEQP_CALL(sleqp_malloc(vstar));
  SleqpVec* vec = *vstar;
  vec->dim      = dim;
  vec->nnz      = 0;
  vec->nnz_max  = nnz_max;
  vec->data     = NULL;
  vec->indices  = NULL;
  SLEQP_CALL(sleqp_alloc_array(&vec->data, nnz_max));
  SLEQP_CALL(sleqp_alloc_array(&vec->indices, nnz_max));
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_create_empty(SleqpVec** vec, int dim) { return sleqp_vec_create(vec, dim, 0); }
SLEQP_RETCODE
sleqp_vec_create_full(SleqpVec** vec, int dim) { return sleqp_vec_create(vec, dim, dim); }

SLEQP_RETCODE
sleqp_vec_reserve(SleqpVec* vec, int nnz)
{
  if (vec->nnz_max >= nnz)
    return SLEQP_OKAY;
  SLEQP_CALL(sleqp_realloc(&vec->data, nnz));
  SLEQP_CALL(sleqp_realloc(&vec->indices, nnz));
  vec->nnz_max = nnz;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_push(SleqpVec* vec, int idx, double value)
{
  if (vec->nnz >= vec->nnz_max || idx >= vec->dim || (vec->nnz > 0 && idx <= vec->indices[vec->nnz - 1]))
    sleqp_raise(SLEQP_ILLEGAL_ARGUMENT, "Invalid vector push at index %d", idx);
  vec->data[vec->nnz]    = value;
  vec->indices[vec->nnz] = idx;
  ++vec->nnz;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_clear(SleqpVec* vec)
{
  vec->nnz = 0;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_resize(SleqpVec* vec, int dim)
{
  if (dim < vec->dim)
    while (vec->nnz > 0 && vec->indices[vec->nnz - 1] >= dim)
      --vec->nnz;
  vec->dim = dim;
  return SLEQP_OKAY;
}

/* Two passes like vec.c:71-103 (count, reserve, push).  The pushes of the
 * second pass cannot fail - capacity was reserved for exactly these entries,
 * indices ascend by construction -, and the reference's sleqp_vec_push checks
 * them with assert only (vec.c:45-69: compiled out of a release build), so the
 * stores are done in place here. */
SLEQP_RETCODE
sleqp_vec_set_from_raw(SleqpVec* vec, const double* values, int dim, double zero_eps)
{
  int nnz = 0;
  for (int i = 0; i < dim; ++i)
    nnz += !(fabs(values[i]) <= zero_eps);
  SLEQP_CALL(sleqp_vec_clear(vec));
  SLEQP_CALL(sleqp_vec_resize(vec, dim));
  SLEQP_CALL(sleqp_vec_reserve(vec, nnz));
  double* data = vec->data;
  int* indices = vec->indices;
  int k        = 0;
  for (int i = 0; i < dim; ++i)
  {
    const double v = values[i];
    if (!(fabs(v) <= zero_eps))
    {
      data[k]    = v;
      indices[k] = i;
      ++k;
    }
  }
  vec->nnz = k;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_to_raw(const SleqpVec* vec, double* values)
{
  for (int i = 0; i < vec->dim; ++i)
    values[i] = 0.;
  for (int k = 0; k < vec->nnz; ++k)
    values[vec->indices[k]] = vec->data[k];
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_vec_free(SleqpVec** vstar)
{
  if (*vstar)
  {
    free((*vstar)->data);
    free((*vstar)->indices);
    free(*vstar);
  }
  *vstar = NULL;
  return SLEQP_OKAY;
}

/* ---- matrices ---- */
struct SleqpMat
{
  int refcount;
  int num_rows, num_cols;
  int nnz, nnz_max;
  double* data;
  int* cols;
  int* rows;
  int cols_cap;
};

SLEQP_RETCODE
sleqp_mat_create(SleqpMat** mstar, int num_rows, int num_cols, int nnz_max)
{
  SLEQP_CALL(sleqp_malloc(mstar));
  SleqpMat* m = *mstar;
  memset(m, 0, sizeof *m);
  m->refcount = 1;
  m->num_rows = num_rows;
  m->num_cols = num_cols;
  m->nnz_max  = nnz_max;
  m->cols_cap = num_cols + 1;
  SLEQP_CALL(sleqp_alloc_array(&m->data, nnz_max));
  SLEQP_CALL(sleqp_alloc_array(&m->rows, nnz_max));
  SLEQP_CALL(sleqp_alloc_array(&m->cols, num_cols + 1));
  for (int j = 0; j <= num_cols; ++j)
    m->cols[j] = 0;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_reserve(SleqpMat* m, int nnz)
{
  if (m->nnz_max >= nnz)
    return SLEQP_OKAY;
  SLEQP_CALL(sleqp_realloc(&m->data, nnz));
  SLEQP_CALL(sleqp_realloc(&m->rows, nnz));
  m->nnz_max = nnz;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_set_arrays_mini(SleqpMat* m, const int* cols, const int* rows, const double* data, int nnz)
{
  SLEQP_CALL(sleqp_mat_reserve(m, nnz));
  memcpy(m->cols, cols, (size_t)(m->num_cols + 1) * sizeof(int));
  if (nnz > 0)
  {
    memcpy(m->rows, rows, (size_t)nnz * sizeof(int));
    memcpy(m->data, data, (size_t)nnz * sizeof(double));
  }
  m->nnz = nnz;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_resize(SleqpMat* m, int num_rows, int num_cols)
{
  if (num_cols + 1 > m->cols_cap)
  {
    SLEQP_CALL(sleqp_realloc(&m->cols, num_cols + 1));
    m->cols_cap = num_cols + 1;
  }
  for (int j = m->num_cols + 1; j <= num_cols; ++j)
    m->cols[j] = m->nnz;
  m->num_rows = num_rows;
  m->num_cols = num_cols;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_clear(SleqpMat* m)
{
  m->nnz = 0;
  for (int j = 0; j <= m->num_cols; ++j)
    m->cols[j] = 0;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_push(SleqpMat* m, int row, int col, double value)
{
  if (m->nnz >= m->nnz_max || row < 0 || row >= m->num_rows || col < 0 || col >= m->num_cols)
    sleqp_raise(SLEQP_ILLEGAL_ARGUMENT, "Invalid matrix push at (%d, %d)", row, col);
  m->data[m->nnz] = value;
  m->rows[m->nnz] = row;
  ++m->nnz;
  for (int j = col + 1; j <= m->num_cols; ++j) /* keep the trailing column pointers consistent */
    m->cols[j] = m->nnz;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_mat_push_col(SleqpMat* m, int col)
{
  if (col < 0 || col >= m->num_cols)
    sleqp_raise(SLEQP_ILLEGAL_ARGUMENT, "Invalid column %d", col);
  m->cols[col] = m->nnz;
  for (int j = col + 1; j <= m->num_cols; ++j)
    m->cols[j] = m->nnz;
  return SLEQP_OKAY;
}

int
sleqp_mat_num_cols(const SleqpMat* m) { return m->num_cols; }
int
sleqp_mat_num_rows(const SleqpMat* m) { return m->num_rows; }
int
sleqp_mat_nnz(const SleqpMat* m) { return m->nnz; }
double*
sleqp_mat_data(const SleqpMat* m) { return m->data; }
int*
sleqp_mat_cols(const SleqpMat* m) { return m->cols; }
int*
sleqp_mat_rows(const SleqpMat* m) { return m->rows; }

SLEQP_RETCODE
sleqp_mat_release(SleqpMat** mstar)
{
  SleqpMat* m = *mstar;
  if (m && --m->refcount == 0)
  {
    free(m->data);
    free(m->rows);
    free(m->cols);
    free(m);
  }
  *mstar = NULL;
  return SLEQP_OKAY;
}

/* ---- SleqpFact dispatch ---- */
struct SleqpFact
{
  int refcount;
  char* name;
  char* version;
  SleqpFactCallbacks callbacks;
  SLEQP_FACT_FLAGS flags;
  void* fact_data;
};

SLEQP_RETCODE
sleqp_fact_create(SleqpFact** star, const char* name, const char* version, SleqpSettings* settings,
                  SleqpFactCallbacks* callbacks, SLEQP_FACT_FLAGS flags, void* fact_data)
{
  (void)settings;
  SLEQP_CALL(sleqp_malloc(star));
  SleqpFact* f = *star;
  f->refcount  = 1;
  f->name      = strdup(name);
  f->version   = strdup(version);
  f->callbacks = *callbacks;
  f->flags     = flags;
  f->fact_data = fact_data;
  return SLEQP_OKAY;
}

const char*
sleqp_fact_name(SleqpFact* f) { return f->name; }
const char*
sleqp_fact_version(SleqpFact* f) { return f->version; }
SLEQP_FACT_FLAGS
sleqp_fact_flags(SleqpFact* f) { return f->flags; }

SLEQP_RETCODE
sleqp_fact_set_matrix(SleqpFact* f, SleqpMat* matrix) { return f->callbacks.set_matrix(f->fact_data, matrix); }
SLEQP_RETCODE
sleqp_fact_solve(SleqpFact* f, const SleqpVec* rhs) { return f->callbacks.solve(f->fact_data, rhs); }
SLEQP_RETCODE
sleqp_fact_solution(SleqpFact* f, SleqpVec* sol, int begin, int end, double zero_eps)
{
  return f->callbacks.solution(f->fact_data, sol, begin, end, zero_eps);
}

SLEQP_RETCODE
sleqp_fact_cond(SleqpFact* f, double* condition)
{
  if (f->callbacks.condition)
    return f->callbacks.condition(f->fact_data, condition);
  *condition = SLEQP_NONE;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_fact_capture(SleqpFact* f)
{
  ++f->refcount;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_fact_release(SleqpFact** star)
{
  SleqpFact* f = *star;
  if (f && --f->refcount == 0)
  {
    if (f->callbacks.free)
      SLEQP_CALL(f->callbacks.free(&f->fact_data));
    free(f->name);
    free(f->version);
    free(f);
  }
  *star = NULL;
  return SLEQP_OKAY;
}

/* ---- problem / working set / iterate ---- */
struct SleqpProblem
{
  int refcount;
  int num_vars, num_cons;
  bool nonlinear_cons;
  SleqpMiniHessProd hess_prod;
  void* hess_data;
};

void
sleqp_problem_set_hess_prod_mini(SleqpProblem* p, SleqpMiniHessProd callback, void* data)
{
  p->hess_prod = callback;
  p->hess_data = data;
}

SLEQP_RETCODE
sleqp_problem_hess_prod(SleqpProblem* p, const SleqpVec* direction, const SleqpVec* cons_duals, SleqpVec* product)
{
  if (!p->hess_prod)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "mini problem: no Hessian product installed");
  }
  double* dir  = calloc((size_t)(p->num_vars > 0 ? p->num_vars : 1), sizeof(double));
  double* dual = calloc((size_t)(p->num_cons > 0 ? p->num_cons : 1), sizeof(double));
  double* prod = calloc((size_t)(p->num_vars > 0 ? p->num_vars : 1), sizeof(double));
  SLEQP_RETCODE status = SLEQP_OKAY;
  if (!dir || !dual || !prod)
  {
    status = SLEQP_ERROR;
  }
  else
  {
    for (int k = 0; k < direction->nnz; ++k) dir[direction->indices[k]] = direction->data[k];
    if (cons_duals)
      for (int k = 0; k < cons_duals->nnz; ++k) dual[cons_duals->indices[k]] = cons_duals->data[k];
    if (p->hess_prod(dir, dual, prod, p->hess_data) != 0)
      status = SLEQP_ERROR;
    else
      status = sleqp_vec_set_from_raw(product, prod, p->num_vars, 0.0);
  }
  free(dir);
  free(dual);
  free(prod);
  if (status != SLEQP_OKAY)
  {
    sleqp_raise(SLEQP_INTERNAL_ERROR, "mini problem: Hessian product failed");
  }
  return SLEQP_OKAY;
}

bool
sleqp_problem_has_nonlinear_cons(SleqpProblem* p) { return p->nonlinear_cons; }
void
sleqp_problem_set_nonlinear_cons_mini(SleqpProblem* p, bool value) { p->nonlinear_cons = value; }

SLEQP_RETCODE
sleqp_problem_create_mini(SleqpProblem** star, int num_vars, int num_cons)
{
  SLEQP_CALL(sleqp_malloc(star));
  (*star)->refcount = 1;
  (*star)->nonlinear_cons = true;
  (*star)->hess_prod      = NULL;
  (*star)->hess_data      = NULL;
  (*star)->num_vars = num_vars;
  (*star)->num_cons = num_cons;
  return SLEQP_OKAY;
}
int
sleqp_problem_num_vars(const SleqpProblem* p) { return p->num_vars; }
int
sleqp_problem_num_cons(const SleqpProblem* p) { return p->num_cons; }
SLEQP_RETCODE
sleqp_problem_capture(SleqpProblem* p)
{
  ++p->refcount;
  return SLEQP_OKAY;
}
SLEQP_RETCODE
sleqp_problem_release(SleqpProblem** star)
{
  if (*star && --(*star)->refcount == 0)
    free(*star);
  *star = NULL;
  return SLEQP_OKAY;
}

struct SleqpWorkingSet
{
  int refcount;
  int num_vars, num_cons;
  int num_active_vars, num_active_cons;
  int* var_indices;
  int* cons_indices;
};

SLEQP_RETCODE
sleqp_working_set_create(SleqpWorkingSet** star, SleqpProblem* problem)
{
  SLEQP_CALL(sleqp_malloc(star));
  SleqpWorkingSet* ws = *star;
  memset(ws, 0, sizeof *ws);
  ws->refcount = 1;
  ws->num_vars = problem->num_vars;
  ws->num_cons = problem->num_cons;
  SLEQP_CALL(sleqp_alloc_array(&ws->var_indices, ws->num_vars));
  SLEQP_CALL(sleqp_alloc_array(&ws->cons_indices, ws->num_cons));
  return sleqp_working_set_reset(ws);
}

SLEQP_RETCODE
sleqp_working_set_reset(SleqpWorkingSet* ws)
{
  for (int j = 0; j < ws->num_vars; ++j)
    ws->var_indices[j] = SLEQP_NONE;
  for (int i = 0; i < ws->num_cons; ++i)
    ws->cons_indices[i] = SLEQP_NONE;
  ws->num_active_vars = ws->num_active_cons = 0;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_working_set_add_var(SleqpWorkingSet* ws, int index, SLEQP_ACTIVE_STATE state)
{
  if (ws->num_active_cons != 0)
    sleqp_raise(SLEQP_INTERNAL_ERROR, "Must add variables before constraints");
  if (index < 0 || index >= ws->num_vars || state == SLEQP_INACTIVE || ws->var_indices[index] != SLEQP_NONE)
    sleqp_raise(SLEQP_ILLEGAL_ARGUMENT, "Invalid working set variable %d", index);
  ws->var_indices[index] = ws->num_active_vars++;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_working_set_add_cons(SleqpWorkingSet* ws, int index, SLEQP_ACTIVE_STATE state)
{
  if (index < 0 || index >= ws->num_cons || state == SLEQP_INACTIVE || ws->cons_indices[index] != SLEQP_NONE)
    sleqp_raise(SLEQP_ILLEGAL_ARGUMENT, "Invalid working set constraint %d", index);
  ws->cons_indices[index] = ws->num_active_vars + ws->num_active_cons++;
  return SLEQP_OKAY;
}

int
sleqp_working_set_var_index(const SleqpWorkingSet* ws, int index) { return ws->var_indices[index]; }
int
sleqp_working_set_cons_index(const SleqpWorkingSet* ws, int index) { return ws->cons_indices[index]; }
int
sleqp_working_set_num_active_vars(const SleqpWorkingSet* ws) { return ws->num_active_vars; }
int
sleqp_working_set_num_active_cons(const SleqpWorkingSet* ws) { return ws->num_active_cons; }
int
sleqp_working_set_size(const SleqpWorkingSet* ws) { return ws->num_active_vars + ws->num_active_cons; }

SLEQP_RETCODE
sleqp_working_set_release(SleqpWorkingSet** star)
{
  SleqpWorkingSet* ws = *star;
  if (ws && --ws->refcount == 0)
  {
    free(ws->var_indices);
    free(ws->cons_indices);
    free(ws);
  }
  *star = NULL;
  return SLEQP_OKAY;
}

struct SleqpIterate
{
  int refcount;
  SleqpMat* cons_jac;
  SleqpWorkingSet* working_set;
};

SLEQP_RETCODE
sleqp_iterate_create_mini(SleqpIterate** star, SleqpProblem* problem)
{
  SLEQP_CALL(sleqp_malloc(star));
  SleqpIterate* it = *star;
  it->refcount     = 1;
  it->cons_jac     = NULL;
  it->working_set  = NULL;
  SLEQP_CALL(sleqp_mat_create(&it->cons_jac, problem->num_cons, problem->num_vars, 0));
  SLEQP_CALL(sleqp_working_set_create(&it->working_set, problem));
  return SLEQP_OKAY;
}

SleqpMat*
sleqp_iterate_cons_jac(const SleqpIterate* it) { return it->cons_jac; }
SleqpWorkingSet*
sleqp_iterate_working_set(const SleqpIterate* it) { return it->working_set; }

SLEQP_RETCODE
sleqp_iterate_release(SleqpIterate** star)
{
  SleqpIterate* it = *star;
  if (it && --it->refcount == 0)
  {
    SLEQP_CALL(sleqp_mat_release(&it->cons_jac));
    SLEQP_CALL(sleqp_working_set_release(&it->working_set));
    free(it);
  }
  *star = NULL;
  return SLEQP_OKAY;
}

/* ---- SleqpAugJac dispatch ---- */
struct SleqpAugJac
{
  int refcount;
  SleqpProblem* problem;
  SleqpAugJacCallbacks callbacks;
  void* data;
};

SLEQP_RETCODE
sleqp_aug_jac_create(SleqpAugJac** star, SleqpProblem* problem, SleqpAugJacCallbacks* callbacks, void* aug_jac_data)
{
  SLEQP_CALL(sleqp_malloc(star));
  SleqpAugJac* aj = *star;
  aj->refcount    = 1;
  aj->problem     = problem;
  SLEQP_CALL(sleqp_problem_capture(problem));
  aj->callbacks = *callbacks;
  aj->data      = aug_jac_data;
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_aug_jac_set_iterate(SleqpAugJac* aj, SleqpIterate* iterate) { return aj->callbacks.set_iterate(iterate, aj->data); }
SLEQP_RETCODE
sleqp_aug_jac_solve_min_norm(SleqpAugJac* aj, const SleqpVec* rhs, SleqpVec* sol)
{
  return aj->callbacks.solve_min_norm(rhs, sol, aj->data);
}
SLEQP_RETCODE
sleqp_aug_jac_solve_lsq(SleqpAugJac* aj, const SleqpVec* rhs, SleqpVec* sol)
{
  return aj->callbacks.solve_lsq(rhs, sol, aj->data);
}
SLEQP_RETCODE
sleqp_aug_jac_project_nullspace(SleqpAugJac* aj, const SleqpVec* rhs, SleqpVec* sol)
{
  return aj->callbacks.project_nullspace(rhs, sol, aj->data);
}
SLEQP_RETCODE
sleqp_aug_jac_condition(SleqpAugJac* aj, bool* exact, double* condition)
{
  return aj->callbacks.condition(exact, condition, aj->data);
}

SLEQP_RETCODE
sleqp_aug_jac_release(SleqpAugJac** star)
{
  SleqpAugJac* aj = *star;
  if (aj && --aj->refcount == 0)
  {
    if (aj->callbacks.free)
      SLEQP_CALL(aj->callbacks.free(aj->data));
    SLEQP_CALL(sleqp_problem_release(&aj->problem));
    free(aj);
  }
  *star = NULL;
  return SLEQP_OKAY;
}

/* ---- SleqpTRSolver dispatch (tr/tr_solver.c) ---- */
struct SleqpTRSolver
{
  int refcount;
  SleqpTRCallbacks callbacks;
  void* data;
  double time_limit; /* tr/tr_solver.c:13 */
};

SLEQP_RETCODE
sleqp_tr_solver_set_time_limit(SleqpTRSolver* solver, double time_limit)
{
  solver->time_limit = time_limit; /* tr/tr_solver.c:18-21 */
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_tr_solver_create(SleqpTRSolver** star, SleqpTRCallbacks* callbacks, void* solver_data)
{
  SLEQP_CALL(sleqp_malloc(star));
  (*star)->refcount  = 1;
  (*star)->callbacks  = *callbacks;
  (*star)->data       = solver_data;
  (*star)->time_limit = SLEQP_NONE; /* tr/tr_solver.c:47 */
  return SLEQP_OKAY;
}

SLEQP_RETCODE
sleqp_tr_solver_solve(SleqpTRSolver* solver, SleqpAugJac* jacobian, const SleqpVec* multipliers, const SleqpVec* gradient,
                      SleqpVec* newton_step, double trust_radius, double* tr_dual)
{
  return solver->callbacks.solve(jacobian, multipliers, gradient, newton_step, trust_radius, tr_dual, solver->time_limit,
                                 solver->data);
}

SLEQP_RETCODE
sleqp_tr_solver_current_rayleigh(SleqpTRSolver* solver, double* min_rayleigh, double* max_rayleigh)
{
  return solver->callbacks.rayleigh(min_rayleigh, max_rayleigh, solver->data);
}

SLEQP_RETCODE
sleqp_tr_solver_release(SleqpTRSolver** star)
{
  SleqpTRSolver* solver = *star;
  if (solver && --solver->refcount == 0)
  {
    if (solver->callbacks.free)
      SLEQP_CALL(solver->callbacks.free(&solver->data));
    free(solver);
  }
  *star = NULL;
  return SLEQP_OKAY;
}
