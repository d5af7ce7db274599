/* kkt_oracle.c — CPU restatement of the reference's KKT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is shipped, linked into or
 * called by the product (sleqp_amd/, shim/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only
 * as the checker / reported CPU baseline.
 *
 * Reference: chrhansk/sleqp v1.0.2; every function cites the file:line (under
 * src/main/ unless noted) it restates.  The reference's own build needs
 * cmake-generated headers (sleqp/defs.h from defs.h.in, sleqp/export.h) and
 * therefore cannot be compiled here as oracle/_ref ("Oracle").
 *
 * The factorisation itself lives in third-party dependencies that are absent
 * from /root/reference: LAPACK dgetrf/dgetrs (fact/fact_lapack.c:6-19, found via
 * find_package(LAPACK), no version pin) and MA57 / CHOLMOD / UMFPACK (no pins
 * either).  Restated here:
 *   - oracle_lu_*     : LAPACK's published dgetf2/dgetrs algorithm (partial
 *                       pivoting, right-looking) on the densified K exactly as
 *                       fact_lapack.c:52-154 drives it;
 *   - oracle_ldl_*    : the published up-looking simplicial sparse LDL^T
 *                       (T. Davis, "Algorithm 849: a concise sparse Cholesky
 *                       factorization package", the kernel of CHOLMOD's
 *                       simplicial path), used as the large-N CPU baseline.
 * Pinning: tests/test_oracle_pins.py checks this file against the reference
 * tests' known answers (sparse/sleqp_sparse_matrix_test.c:12-56,
 * constrained_newton_test.c:204-275, unconstrained_newton_test.c:67-205,
 * dual_estimation_test.c:15-103) and against the real LAPACK dgetrf/dgetrs
 * shipped inside scipy.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_ERROR (-1)

/* ------------------------------------------------------------------------ */
/* sparse vectors (sparse/vec.c)                                            */
/* ------------------------------------------------------------------------ */

/* sleqp_vec_to_raw, sparse/vec.c:105-119 */
void oracle_vec_to_raw(int dim, int nnz, const int* indices, const double* data, double* values)
{
  for (int i = 0; i < dim; ++i)
    values[i] = 0.;
  for (int k = 0; k < nnz; ++k)
    values[indices[k]] = data[k];
}

/* sleqp_vec_set_from_raw, sparse/vec.c:71-103 with sleqp_is_zero, cmp.c:89-92:
 * keeps entries with !(|v| <= zero_eps).  Returns nnz. */
int oracle_vec_set_from_raw(const double* values, int dim, double zero_eps, int* indices, double* data)
{
  int nnz = 0;
  for (int i = 0; i < dim; ++i)
  {
    const double v = values[i];
    if (!(fabs(v) <= zero_eps))
    {
      indices[nnz] = i;
      data[nnz]    = v;
      ++nnz;
    }
  }
  return nnz;
}

/* ------------------------------------------------------------------------ */
/* sparse matrix products (sparse/mat.c)                                    */
/* ------------------------------------------------------------------------ */

/* sleqp_mat_mult_vec, sparse/mat.c:282-310: result = M * x, x sparse */
void oracle_mat_mult_vec(int num_rows, int num_cols, const int* cols, const int* rows, const double* data,
                         int x_nnz, const int* x_indices, const double* x_data, double* result)
{
  (void)num_cols;
  for (int i = 0; i < num_rows; ++i)
    result[i] = 0.;
  for (int k = 0; k < x_nnz; ++k)
  {
    const int col       = x_indices[k];
    const double factor = x_data[k];
    for (int e = cols[col]; e < cols[col + 1]; ++e)
      result[rows[e]] += factor * data[e];
  }
}

/* sleqp_mat_mult_vec_trans, sparse/mat.c:312-363: result = M^T x by per-column
 * merge joins, entries pushed when !(|sum| <= eps).  Returns nnz. */
int oracle_mat_mult_vec_trans(int num_rows, int num_cols, const int* cols, const int* rows, const double* data,
                              int x_nnz, const int* x_indices, const double* x_data, double eps,
                              int* res_indices, double* res_data)
{
  (void)num_rows;
  int nnz = 0;
  for (int col = 0; col < num_cols; ++col)
  {
    int k_vec = 0, k_mat = cols[col];
    double sum = 0.;
    while (k_vec < x_nnz && k_mat < cols[col + 1])
    {
      const int vec_idx = x_indices[k_vec];
      const int row_idx = rows[k_mat];
      if (vec_idx < row_idx)
        ++k_vec;
      else if (vec_idx > row_idx)
        ++k_mat;
      else
        sum += x_data[k_vec++] * data[k_mat++];
    }
    if (!(fabs(sum) <= eps))
    {
      res_indices[nnz] = col;
      res_data[nnz]    = sum;
      ++nnz;
    }
  }
  return nnz;
}

/* prod_from_hess_matrix, bindings/mex/mex_hess.c:85-139: symmetric product
 * from a lower-triangular CSC Hessian.  Returns ORACLE_ERROR for an entry
 * above the diagonal (the reference raises SLEQP_FUNC_EVAL_ERROR). */
int oracle_hess_prod_lower(int dim, const int* jc, const int* ir, const double* pr, const double* direction,
                           double* product)
{
  for (int row = 0; row < dim; ++row)
    product[row] = 0.;
  for (int col = 0; col < dim; ++col)
    for (int index = jc[col]; index < jc[col + 1]; ++index)
    {
      const int row      = ir[index];
      const double value = pr[index];
      if (row < col)
        return ORACLE_ERROR;
      if (row == col)
        product[row] += value * direction[col];
      else
      {
        product[row] += value * direction[col];
        product[col] += value * direction[row];
      }
    }
  return ORACLE_OK;
}

/* ------------------------------------------------------------------------ */
/* KKT assembly (aug_jac/standard_aug_jac.c)                                */
/* ------------------------------------------------------------------------ */

/* reserve_aug_jac, standard_aug_jac.c:106-133: capacity of the K arrays */
int oracle_reserve_aug_jac(int num_vars, int cons_nnz, int num_active_vars, int lower_only)
{
  int max_nnz = num_vars + (cons_nnz + num_active_vars);
  if (!lower_only)
    max_nnz += (cons_nnz + num_active_vars);
  return max_nnz;
}

/* add_upper, standard_aug_jac.c:34-104 (non-LOWER backends only) */
static int add_upper(int num_variables, int aug_num_cols, int* aug_cols, int* aug_rows, double* aug_data,
                     int nnz)
{
  int* col_indices = (int*)calloc((size_t)aug_num_cols + 2, sizeof(int));
  if (!col_indices)
    return ORACLE_ERROR;
  for (int column = num_variables + 1; column < aug_num_cols + 1; ++column)
    aug_cols[column] = 0;
  for (int column = 0; column < num_variables; ++column)
    for (int index = aug_cols[column]; index < aug_cols[column + 1]; ++index)
    {
      if (aug_rows[index] < num_variables)
        continue;
      ++aug_cols[aug_rows[index] + 1];
    }
  for (int column = num_variables + 1; column < aug_num_cols + 1; ++column)
  {
    aug_cols[column] += aug_cols[column - 1];
    col_indices[column] = 0;
  }
  int aug_total_nnz = nnz;
  for (int column = 0; column < num_variables; ++column)
    for (int index = aug_cols[column]; index < aug_cols[column + 1]; ++index)
    {
      if (aug_rows[index] < num_variables)
        continue;
      const int target_column = aug_rows[index];
      const int target_index  = aug_cols[target_column] + col_indices[target_column + 1]++;
      aug_data[target_index]  = aug_data[index];
      aug_rows[target_index]  = column;
      ++aug_total_nnz;
    }
  free(col_indices);
  return aug_total_nnz;
}

/* fill_aug_jac, standard_aug_jac.c:135-237.  cons_jac is CSC (num_cons x
 * num_variables); var_index / cons_index are sleqp_working_set_var_index /
 * _cons_index (working_set.c:191-205; -1 = SLEQP_NONE).  Writes K into
 * aug_cols[num_variables + working_set_size + 1], aug_rows, aug_data (capacity
 * oracle_reserve_aug_jac) and returns nnz, or ORACLE_ERROR. */
int oracle_fill_aug_jac(int num_variables, int num_constraints, const int* cons_jac_cols,
                        const int* cons_jac_rows, const double* cons_jac_data, const int* var_index,
                        const int* cons_index, int working_set_size, int lower_only, int* aug_cols,
                        int* aug_rows, double* aug_data)
{
  (void)num_constraints;
  const int augmented_size = num_variables + working_set_size;
  int nnz                  = 0;
  for (int column = 0; column < num_variables; ++column)
  {
    aug_cols[column] = nnz; /* sleqp_mat_push_col */
    /* push identity part first... */
    aug_rows[nnz] = column;
    aug_data[nnz] = 1.;
    ++nnz;
    {
      const int variable_index = var_index[column];
      if (variable_index != -1)
      {
        aug_rows[nnz] = num_variables + variable_index;
        aug_data[nnz] = 1.;
        ++nnz;
      }
    }
    for (int index = cons_jac_cols[column]; index < cons_jac_cols[column + 1]; ++index)
    {
      const int jac_row        = cons_jac_rows[index];
      const int act_cons_index = cons_index[jac_row];
      if (act_cons_index != -1)
      {
        aug_rows[nnz] = num_variables + act_cons_index;
        aug_data[nnz] = cons_jac_data[index];
        ++nnz;
      }
    }
  }
  for (int j = num_variables; j <= augmented_size; ++j)
    aug_cols[j] = nnz;
  if (!lower_only)
    nnz = add_upper(num_variables, augmented_size, aug_cols, aug_rows, aug_data, nnz);
  return nnz;
}

/* ------------------------------------------------------------------------ */
/* dense LU oracle (fact/fact_lapack.c + LAPACK dgetf2 / dgetrs)            */
/* ------------------------------------------------------------------------ */

/* store_matrix_values, fact_lapack.c:52-73: symmetric densification of the
 * lower-triangular CSC K into an N x N array (zeroed first, :100-103). */
/* ---------------------------------------------------------------------------
 * Reduced AugJac: the matrix A_W A_W^T handed to a PSD-flagged backend.
 * Restates aug_jac/reduced_aug_jac.c: the CSR copy of the working-set rows (count_rows :112-191, fill_system
 * :193-262: unit rows of the active variables first, then the constraint rows in the column order of cons_jac) and
 * compute_matrix_lower (:323-377) with compute_inner_product (:283-321).  Faithful to the reference in one
 * noteworthy detail: compute_inner_product starts from *nonzero = true (:290), so EVERY entry below the diagonal is
 * pushed, zero products included - the matrix the backend receives is a dense lower triangle.
 * Output: lower CSC (cols[size + 1], rows, data) with capacity size (size + 1) / 2; returns nnz.
 * ------------------------------------------------------------------------- */
int oracle_reduced_aug_jac(int num_variables, int num_constraints, const int* cons_jac_cols, const int* cons_jac_rows,
                           const double* cons_jac_data, const int* var_index, const int* cons_index,
                           int working_set_size, int* out_cols, int* out_rows, double* out_data)
{
  const int size = working_set_size;
  int num_active_vars = 0;
  for (int j = 0; j < num_variables; ++j) num_active_vars += (var_index[j] >= 0);
  /* count_rows */
  int* row_counts = (int*)calloc((size_t)size + 2, sizeof(int));
  int nnz = num_active_vars;
  for (int i = 0; i < num_active_vars; ++i) row_counts[i] = 1;
  for (int col = 0; col < num_variables; ++col)
    for (int index = cons_jac_cols[col]; index < cons_jac_cols[col + 1]; ++index)
    {
      const int cons = cons_index[cons_jac_rows[index]];
      if (cons < 0) continue;
      ++row_counts[cons];
      ++nnz;
    }
  int offset = 0;
  for (int i = 0; i <= size; ++i)
  {
    const int current = row_counts[i];
    row_counts[i]     = offset;
    offset += current;
  }
  /* fill_system */
  int* col_indices = (int*)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  double* values   = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  int* row_offsets = (int*)calloc((size_t)size + 1, sizeof(int));
  for (int i = 0; i < num_active_vars; ++i) values[i] = 1.;
  for (int j = 0; j < num_variables; ++j)
    if (var_index[j] >= 0) col_indices[var_index[j]] = j;
  for (int col = 0; col < num_variables; ++col)
    for (int index = cons_jac_cols[col]; index < cons_jac_cols[col + 1]; ++index)
    {
      const int cons = cons_index[cons_jac_rows[index]];
      if (cons < 0) continue;
      const int jac_index    = row_counts[cons] + row_offsets[cons];
      values[jac_index]      = cons_jac_data[index];
      col_indices[jac_index] = col;
      ++row_offsets[cons];
    }
  /* compute_matrix_lower */
  int out = 0;
  for (int col = 0; col < size; ++col)
  {
    out_cols[col] = out;
    if (col < num_active_vars)
    {
      out_rows[out] = col;
      out_data[out] = 1.;
      ++out;
    }
    else
    {
      double product = 0.;
      for (int k = row_counts[col]; k < row_counts[col + 1]; ++k) product += values[k] * values[k];
      out_rows[out] = col;
      out_data[out] = product;
      ++out;
    }
    for (int row = col + 1; row < size; ++row)
    {
      double product = 0.;
      int col_index = row_counts[col], row_index = row_counts[row];
      const int col_bound = row_counts[col + 1], row_bound = row_counts[row + 1];
      while (col_index < col_bound && row_index < row_bound)
      {
        const int col_col = col_indices[col_index], row_col = col_indices[row_index];
        if (row_col < col_col)
          ++row_index;
        else if (row_col > col_col)
          ++col_index;
        else
        {
          product += values[row_index] * values[col_index];
          ++row_index;
          ++col_index;
        }
      }
      out_rows[out] = row; /* nonzero is always true in the reference (:290) */
      out_data[out] = product;
      ++out;
    }
  }
  out_cols[size] = out;
  free(row_counts);
  free(col_indices);
  free(values);
  free(row_offsets);
  (void)num_constraints;
  return out;
}

void oracle_lapack_store_matrix_values(int num_cols, const int* cols, const int* rows, const double* data,
                                       double* values)
{
  for (long i = 0; i < (long)num_cols * num_cols; ++i)
    values[i] = 0.;
  for (int col = 0; col < num_cols; ++col)
    for (int index = cols[col]; index < cols[col + 1]; ++index)
    {
      const int row                          = rows[index];
      values[(long)row * num_cols + col]     = data[index];
      values[(long)col * num_cols + row]     = data[index];
    }
}

/* LAPACK dgetf2 (unblocked right-looking LU with partial pivoting; dgetrf
 * computes the same factorisation blocked), as called at fact_lapack.c:108-115.
 * A is column-major N x N, overwritten by L (unit) and U; ipiv is 1-based
 * like LAPACK's.  Returns INFO (0, or k > 0 if U(k,k) is exactly zero). */
int oracle_dgetrf(int N, double* A, int* ipiv)
{
  int info = 0;
  for (int j = 0; j < N; ++j)
  {
    /* idamax: first index of maximal |A(i,j)|, i >= j */
    int p       = j;
    double best = fabs(A[(long)j * N + j]);
    for (int i = j + 1; i < N; ++i)
    {
      const double v = fabs(A[(long)j * N + i]);
      if (v > best)
      {
        best = v;
        p    = i;
      }
    }
    ipiv[j] = p + 1;
    if (A[(long)j * N + p] != 0.)
    {
      if (p != j) /* dswap of full rows */
        for (int c = 0; c < N; ++c)
        {
          const double t       = A[(long)c * N + j];
          A[(long)c * N + j]   = A[(long)c * N + p];
          A[(long)c * N + p]   = t;
        }
      const double piv = A[(long)j * N + j];
      for (int i = j + 1; i < N; ++i)
        A[(long)j * N + i] /= piv;
    }
    else if (info == 0)
      info = j + 1;
    /* dger: trailing rank-1 update */
    for (int c = j + 1; c < N; ++c)
    {
      const double ujc = A[(long)c * N + j];
      if (ujc != 0.)
      {
        double* col       = A + (long)c * N;
        const double* lj  = A + (long)j * N;
        for (int i = j + 1; i < N; ++i)
          col[i] -= lj[i] * ujc;
      }
    }
  }
  return info;
}

/* LAPACK dgetrs, TRANS = 'N', NRHS = 1 (fact_lapack.c:134-146): dlaswp,
 * unit-lower forward substitution, upper back substitution, in place. */
void oracle_dgetrs(int N, const double* A, const int* ipiv, double* b)
{
  for (int j = 0; j < N; ++j)
  {
    const int p = ipiv[j] - 1;
    if (p != j)
    {
      const double t = b[j];
      b[j]           = b[p];
      b[p]           = t;
    }
  }
  for (int j = 0; j < N; ++j)
  {
    const double bj = b[j];
    if (bj != 0.)
      for (int i = j + 1; i < N; ++i)
        b[i] -= bj * A[(long)j * N + i];
  }
  for (int j = N - 1; j >= 0; --j)
  {
    b[j] /= A[(long)j * N + j];
    const double bj = b[j];
    for (int i = 0; i < j; ++i)
      b[i] -= bj * A[(long)j * N + i];
  }
}

typedef struct
{
  int rows;
  double* values;
  int* ipiv;
  double* sol;
} OracleFact;

/* lapack_set_matrix, fact_lapack.c:75-123 */
OracleFact* oracle_fact_set_matrix(int num_rows, const int* cols, const int* rows, const double* data)
{
  OracleFact* f = (OracleFact*)calloc(1, sizeof(OracleFact));
  if (!f)
    return NULL;
  f->rows   = num_rows;
  f->values = (double*)malloc(sizeof(double) * (size_t)(num_rows > 0 ? num_rows : 1) * (num_rows > 0 ? num_rows : 1));
  f->ipiv   = (int*)malloc(sizeof(int) * (size_t)(num_rows + 1));
  f->sol    = (double*)malloc(sizeof(double) * (size_t)(num_rows + 1));
  if (!f->values || !f->ipiv || !f->sol)
    return NULL;
  oracle_lapack_store_matrix_values(num_rows, cols, rows, data, f->values);
  const int info = oracle_dgetrf(num_rows, f->values, f->ipiv);
  if (info != 0) /* "Failed to factorize using LAPACK", fact_lapack.c:117-120 */
  {
    free(f->values);
    free(f->ipiv);
    free(f->sol);
    free(f);
    return NULL;
  }
  return f;
}

/* lapack_solve, fact_lapack.c:125-154 (rhs given as the SleqpVec fields) */
void oracle_fact_solve(OracleFact* f, int rhs_nnz, const int* rhs_indices, const double* rhs_data)
{
  oracle_vec_to_raw(f->rows, rhs_nnz, rhs_indices, rhs_data, f->sol);
  oracle_dgetrs(f->rows, f->values, f->ipiv, f->sol);
}

void oracle_fact_solve_dense(OracleFact* f, const double* rhs)
{
  memcpy(f->sol, rhs, sizeof(double) * (size_t)f->rows);
  oracle_dgetrs(f->rows, f->values, f->ipiv, f->sol);
}

/* lapack_solution, fact_lapack.c:156-171: returns nnz of the packed slice */
int oracle_fact_solution(OracleFact* f, int begin, int end, double zero_eps, int* indices, double* data)
{
  return oracle_vec_set_from_raw(f->sol + begin, end - begin, zero_eps, indices, data);
}

const double* oracle_fact_raw_solution(OracleFact* f) { return f->sol; }

void oracle_fact_free(OracleFact* f)
{
  if (!f)
    return;
  free(f->values);
  free(f->ipiv);
  free(f->sol);
  free(f);
}

/* ------------------------------------------------------------------------ */
/* the three AugJac solves (aug_jac/standard_aug_jac.c)                     */
/* ------------------------------------------------------------------------ */

/* aug_jac_solve_min_norm, standard_aug_jac.c:306-350: rhs (dim |W|) is shifted
 * by +n in place, K [x;y] = [0;rhs], sol = solution(0, n).  Returns nnz. */
int oracle_aug_jac_solve_min_norm(OracleFact* f, int num_variables, int rhs_nnz, int* rhs_indices,
                                  const double* rhs_data, double zero_eps, int* sol_indices, double* sol_data)
{
  for (int k = 0; k < rhs_nnz; ++k)
    rhs_indices[k] += num_variables;
  oracle_fact_solve(f, rhs_nnz, rhs_indices, rhs_data);
  const int nnz = oracle_fact_solution(f, 0, num_variables, zero_eps, sol_indices, sol_data);
  for (int k = 0; k < rhs_nnz; ++k)
    rhs_indices[k] -= num_variables;
  return nnz;
}

/* aug_jac_solve_lsq, standard_aug_jac.c:352-394: rhs (dim n) padded with
 * zeros to n + |W|, sol = solution(n, n + |W|) */
int oracle_aug_jac_solve_lsq(OracleFact* f, int num_variables, int rhs_nnz, const int* rhs_indices,
                             const double* rhs_data, double zero_eps, int* sol_indices, double* sol_data)
{
  oracle_fact_solve(f, rhs_nnz, rhs_indices, rhs_data);
  return oracle_fact_solution(f, num_variables, f->rows, zero_eps, sol_indices, sol_data);
}

/* aug_jac_project_nullspace, standard_aug_jac.c:396-435: same system,
 * sol = solution(0, n) */
int oracle_aug_jac_project_nullspace(OracleFact* f, int num_variables, int rhs_nnz, const int* rhs_indices,
                                     const double* rhs_data, double zero_eps, int* sol_indices, double* sol_data)
{
  oracle_fact_solve(f, rhs_nnz, rhs_indices, rhs_data);
  return oracle_fact_solution(f, 0, num_variables, zero_eps, sol_indices, sol_data);
}

/* ------------------------------------------------------------------------ */
/* projected Steihaug CG (tr/steihaug_solver.c:218-496, tr/tr_util.c:8-58)  */
/* ------------------------------------------------------------------------ */

static double dotn(int n, const double* a, const double* b)
{
  double s = 0.;
  for (int i = 0; i < n; ++i)
    s += a[i] * b[i];
  return s;
}

static void project_dense(OracleFact* f, int n, const double* v, double* out)
{
  for (int i = 0; i < f->rows; ++i)
    f->sol[i] = (i < n) ? v[i] : 0.;
  oracle_dgetrs(f->rows, f->values, f->ipiv, f->sol);
  memcpy(out, f->sol, sizeof(double) * (size_t)n);
}

/* steihaug_solver_solve with the Hessian given as a lower-triangular CSC
 * matrix (hess_prod = oracle_hess_prod_lower).  stat_tol = SLEQP_SETTINGS_REAL_
 * STAT_TOL (1e-6, settings.c), tolerance_factor 1e-2 (steihaug_solver.c:21),
 * max_iter = MAX_NEWTON_ITERATIONS (100, settings.c:62; -1 = none).
 * Returns the number of CG iterations, or ORACLE_ERROR. */
/* ... and the extremes of the Rayleigh quotients d.Bd / d.d it collects on the way (steihaug_collect_rayleigh,
 * steihaug_solver.c:150-171, called right behind the product at :282; both start at 1, :229-230) - what
 * steihaug_solver_rayleigh (:173-185) hands to newton.c:328-343.  Either pointer may be NULL. */
int oracle_steihaug_solve_rayleigh(OracleFact* f, int n, const int* hc, const int* hr, const double* hx,
                                   const double* gradient, double trust_radius, double stat_tol, int max_iter,
                                   double* newton_step, double* min_rayleigh_out, double* max_rayleigh_out);

int oracle_steihaug_solve(OracleFact* f, int n, const int* hc, const int* hr, const double* hx,
                          const double* gradient, double trust_radius, double stat_tol, int max_iter,
                          double* newton_step)
{
  return oracle_steihaug_solve_rayleigh(f, n, hc, hr, hx, gradient, trust_radius, stat_tol, max_iter, newton_step, NULL,
                                        NULL);
}

int oracle_steihaug_solve_rayleigh(OracleFact* f, int n, const int* hc, const int* hr, const double* hx,
                                   const double* gradient, double trust_radius, double stat_tol, int max_iter,
                                   double* newton_step, double* min_rayleigh_out, double* max_rayleigh_out)
{
  double min_rayleigh = 1., max_rayleigh = 1.;
  if (min_rayleigh_out)
    *min_rayleigh_out = min_rayleigh;
  if (max_rayleigh_out)
    *max_rayleigh_out = max_rayleigh;
  const double rel_tol    = stat_tol * 1e-2;
  const double rel_tol_sq = rel_tol * rel_tol;
  double* buf             = (double*)calloc((size_t)6 * (n > 0 ? n : 1), sizeof(double));
  if (!buf)
    return ORACLE_ERROR;
  double *z = buf, *r = buf + n, *g = buf + 2 * n, *d = buf + 3 * n, *Bd = buf + 4 * n, *cache = buf + 5 * n;
  double z_curr_nrm_sq = 0.;
  int iteration        = 0;
  for (int i = 0; i < n; ++i)
    newton_step[i] = 0.;
  memcpy(r, gradient, sizeof(double) * (size_t)n);
  project_dense(f, n, r, g); /* g0 = P[r0] */
  for (int i = 0; i < n; ++i)
    d[i] = -g[i];
  if (dotn(n, d, d) < rel_tol_sq)
  {
    free(buf);
    return 0;
  }
  double r_dot_g = dotn(n, r, g);
  for (iteration = 0;; ++iteration)
  {
    if (max_iter != -1 && iteration >= max_iter)
      break;
    if (fabs(r_dot_g) < rel_tol_sq)
    {
      memcpy(newton_step, z, sizeof(double) * (size_t)n);
      break;
    }
    if (oracle_hess_prod_lower(n, hc, hr, hx, d, Bd) != ORACLE_OK)
    {
      free(buf);
      return ORACLE_ERROR;
    }
    {
      /* steihaug_collect_rayleigh */
      const double dir_normsq = dotn(n, d, d);
      if (dir_normsq != 0.)
      {
        const double cur_rayleigh = dotn(n, d, Bd) / dir_normsq;
        min_rayleigh              = cur_rayleigh < min_rayleigh ? cur_rayleigh : min_rayleigh;
        max_rayleigh              = cur_rayleigh > max_rayleigh ? cur_rayleigh : max_rayleigh;
      }
    }
    const double dBd = dotn(n, d, Bd);
    if (dBd <= 0.)
    {
      const double z_dot_d  = dotn(n, z, d);
      const double d_nrm_sq = dotn(n, d, d);
      const double inner    = z_dot_d * z_dot_d - d_nrm_sq * (z_curr_nrm_sq - trust_radius * trust_radius);
      const double tau_min  = 1. / d_nrm_sq * (-z_dot_d - sqrt(inner));
      const double tau_max  = 1. / d_nrm_sq * (-z_dot_d + sqrt(inner));
      const double gd       = dotn(n, gradient, d);
      const double zBd      = dotn(n, z, Bd);
      const double tau_min_obj = tau_min * ((gd + zBd) + 0.5 * tau_min * dBd);
      const double tau_max_obj = tau_max * ((gd + zBd) + 0.5 * tau_max * dBd);
      const double tau         = (tau_min_obj < tau_max_obj) ? tau_min : tau_max;
      for (int i = 0; i < n; ++i)
        newton_step[i] = z[i] + tau * d[i];
      break;
    }
    const double alpha = r_dot_g / dBd;
    for (int i = 0; i < n; ++i)
      cache[i] = z[i] + alpha * d[i];
    const double z_next_nrm_sq = dotn(n, cache, cache);
    if (z_next_nrm_sq >= trust_radius * trust_radius)
    {
      /* sleqp_tr_compute_bdry_sol, tr_util.c:8-58 */
      const double prev_dot_d = dotn(n, z, d);
      const double d_norm_sq  = dotn(n, d, d);
      const double p_norm_sq  = dotn(n, z, z);
      const double inner      = prev_dot_d * prev_dot_d - d_norm_sq * (p_norm_sq - trust_radius * trust_radius);
      const double factor     = 1. / d_norm_sq * (-prev_dot_d + sqrt(inner));
      for (int i = 0; i < n; ++i)
        newton_step[i] = z[i] + factor * d[i];
      break;
    }
    memcpy(z, cache, sizeof(double) * (size_t)n);
    z_curr_nrm_sq = z_next_nrm_sq;
    for (int i = 0; i < n; ++i)
      r[i] += alpha * Bd[i];
    project_dense(f, n, r, g);
    double beta = 1. / r_dot_g;
    r_dot_g     = dotn(n, r, g);
    beta *= r_dot_g;
    for (int i = 0; i < n; ++i)
      d[i] = -g[i] + beta * d[i];
  }
  free(buf);
  if (min_rayleigh_out)
    *min_rayleigh_out = min_rayleigh;
  if (max_rayleigh_out)
    *max_rayleigh_out = max_rayleigh;
  return iteration;
}

/* ------------------------------------------------------------------------ */
/* simplicial sparse LDL^T (large-N CPU baseline)                           */
/* ------------------------------------------------------------------------ */

typedef struct
{
  int n;
  int* Lp;
  int* Li;
  double* Lx;
  double* D;
  int* P;    /* P[k] = original index of pivot k */
  int* Pinv;
  /* permuted upper-triangular CSC of the input */
  int* Up;
  int* Ui;
  double* Ux;
  int* Parent;
  double* work;
  long lnz;
  double flops;
} OracleLdl;

void oracle_ldl_free(OracleLdl* F)
{
  if (!F)
    return;
  free(F->Lp);
  free(F->Li);
  free(F->Lx);
  free(F->D);
  free(F->P);
  free(F->Pinv);
  free(F->Up);
  free(F->Ui);
  free(F->Ux);
  free(F->Parent);
  free(F->work);
  free(F);
}

/* Symbolic + numeric up-looking LDL^T of P K P^T, K given by its lower
 * triangle in CSC (what SLEQP_FACT_FLAGS_LOWER backends receive).  No
 * pivoting: the caller supplies a pivot order for which every leading block is
 * nonsingular (for K = [I A^T; A 0]: all x before any y).  symbolic_only != 0
 * stops after the analysis.  Returns NULL on a zero pivot / allocation failure. */
OracleLdl* oracle_ldl_factor(int n, const int* Kp, const int* Ki, const double* Kx, const int* perm,
                             int symbolic_only)
{
  OracleLdl* F = (OracleLdl*)calloc(1, sizeof(OracleLdl));
  if (!F)
    return NULL;
  F->n      = n;
  F->P      = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  F->Pinv   = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  F->Parent = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  F->Lp     = (int*)malloc(sizeof(int) * (size_t)(n + 2));
  F->D      = (double*)malloc(sizeof(double) * (size_t)(n + 1));
  F->Up     = (int*)calloc((size_t)n + 2, sizeof(int));
  const int nnz = n > 0 ? Kp[n] : 0;
  F->Ui     = (int*)malloc(sizeof(int) * (size_t)(nnz + 1));
  F->Ux     = (double*)malloc(sizeof(double) * (size_t)(nnz + 1));
  F->work   = (double*)calloc((size_t)n + 1, sizeof(double));
  int* Lnz     = (int*)calloc((size_t)n + 1, sizeof(int));
  int* Flag    = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int* Pattern = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  if (!F->P || !F->Pinv || !F->Parent || !F->Lp || !F->D || !F->Up || !F->Ui || !F->Ux || !F->work || !Lnz
      || !Flag || !Pattern)
    goto fail;
  for (int k = 0; k < n; ++k)
  {
    F->P[k]          = perm ? perm[k] : k;
    F->Pinv[F->P[k]] = k;
  }
  /* permuted upper triangle: entry (r, c) of lower K -> (min, max) of (Pinv[r], Pinv[c]) */
  for (int c = 0; c < n; ++c)
    for (int e = Kp[c]; e < Kp[c + 1]; ++e)
    {
      const int a = F->Pinv[Ki[e]], b = F->Pinv[c];
      ++F->Up[(a > b ? a : b) + 1];
    }
  for (int k = 0; k < n; ++k)
    F->Up[k + 1] += F->Up[k];
  {
    int* fill = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    if (!fill)
      goto fail;
    memcpy(fill, F->Up, sizeof(int) * (size_t)n);
    for (int c = 0; c < n; ++c)
      for (int e = Kp[c]; e < Kp[c + 1]; ++e)
      {
        const int a = F->Pinv[Ki[e]], b = F->Pinv[c];
        const int col = a > b ? a : b, row = a > b ? b : a;
        const int q   = fill[col]++;
        F->Ui[q]      = row;
        F->Ux[q]      = Kx ? Kx[e] : 0.;
      }
    free(fill);
  }
  /* ldl_symbolic: elimination tree and column counts */
  for (int k = 0; k < n; ++k)
  {
    F->Parent[k] = -1;
    Flag[k]      = k;
    Lnz[k]       = 0;
    for (int p = F->Up[k]; p < F->Up[k + 1]; ++p)
    {
      int i = F->Ui[p];
      if (i < k)
        for (; Flag[i] != k; i = F->Parent[i])
        {
          if (F->Parent[i] == -1)
            F->Parent[i] = k;
          ++Lnz[i];
          Flag[i] = k;
        }
    }
  }
  F->Lp[0] = 0;
  F->flops = 0.;
  {
    long tot = 0;
    for (int k = 0; k < n; ++k)
    {
      tot += Lnz[k];
      if (tot > 2000000000L)
        goto fail;
      F->Lp[k + 1] = (int)tot;
      F->flops += (double)(Lnz[k] + 1) * (Lnz[k] + 1);
    }
    F->lnz = tot + n;
  }
  if (symbolic_only)
  {
    free(Lnz);
    free(Flag);
    free(Pattern);
    return F;
  }
  F->Li = (int*)malloc(sizeof(int) * (size_t)(F->Lp[n] + 1));
  F->Lx = (double*)malloc(sizeof(double) * (size_t)(F->Lp[n] + 1));
  if (!F->Li || !F->Lx)
    goto fail;
  /* ldl_numeric */
  {
    double* Y = F->work;
    for (int k = 0; k < n; ++k)
    {
      Y[k]    = 0.;
      int top = n;
      Flag[k] = k;
      Lnz[k]  = 0;
      for (int p = F->Up[k]; p < F->Up[k + 1]; ++p)
      {
        int i = F->Ui[p];
        if (i <= k)
        {
          Y[i] += F->Ux[p];
          int len;
          for (len = 0; Flag[i] != k; i = F->Parent[i])
          {
            Pattern[len++] = i;
            Flag[i]        = k;
          }
          while (len > 0)
            Pattern[--top] = Pattern[--len];
        }
      }
      F->D[k] = Y[k];
      Y[k]    = 0.;
      for (; top < n; ++top)
      {
        const int i     = Pattern[top];
        const double yi = Y[i];
        Y[i]            = 0.;
        const int p2    = F->Lp[i] + Lnz[i];
        for (int p = F->Lp[i]; p < p2; ++p)
          Y[F->Li[p]] -= F->Lx[p] * yi;
        const double l_ki = yi / F->D[i];
        F->D[k] -= l_ki * yi;
        F->Li[p2] = k;
        F->Lx[p2] = l_ki;
        ++Lnz[i];
      }
      if (F->D[k] == 0.)
        goto fail;
    }
  }
  free(Lnz);
  free(Flag);
  free(Pattern);
  return F;
fail:
  free(Lnz);
  free(Flag);
  free(Pattern);
  oracle_ldl_free(F);
  return NULL;
}

/* Numeric-only refactorisation: new values of the SAME pattern, permutation, elimination tree and column pointers
 * of F reused (what a CPU backend that kept its symbolic phase would pay per SQP iteration; the reference's backends
 * do not, fact_ma57.c:529-625 - reported next to the symbolic-inclusive figure as `numeric_only`).  Returns 0, or -1
 * on a zero pivot. */
int oracle_ldl_refactor(OracleLdl* F, const int* Kp, const int* Ki, const double* Kx)
{
  const int n = F->n;
  if (!F->Li || !F->Lx)
    return -1;
  int* Lnz     = (int*)calloc((size_t)n + 1, sizeof(int));
  int* Flag    = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int* Pattern = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int* fill    = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  int rc       = 0;
  if (!Lnz || !Flag || !Pattern || !fill)
  {
    rc = -1;
    goto done;
  }
  memcpy(fill, F->Up, sizeof(int) * (size_t)n);
  for (int c = 0; c < n; ++c)
    for (int e = Kp[c]; e < Kp[c + 1]; ++e)
    {
      const int a = F->Pinv[Ki[e]], b = F->Pinv[c];
      const int col = a > b ? a : b;
      F->Ux[fill[col]++] = Kx[e];
    }
  {
    double* Y = F->work;
    for (int k = 0; k < n; ++k)
    {
      Y[k]    = 0.;
      int top = n;
      Flag[k] = k;
      Lnz[k]  = 0;
      for (int p = F->Up[k]; p < F->Up[k + 1]; ++p)
      {
        int i = F->Ui[p];
        if (i <= k)
        {
          Y[i] += F->Ux[p];
          int len;
          for (len = 0; Flag[i] != k; i = F->Parent[i])
          {
            Pattern[len++] = i;
            Flag[i]        = k;
          }
          while (len > 0)
            Pattern[--top] = Pattern[--len];
        }
      }
      F->D[k] = Y[k];
      Y[k]    = 0.;
      for (; top < n; ++top)
      {
        const int i     = Pattern[top];
        const double yi = Y[i];
        Y[i]            = 0.;
        const int p2    = F->Lp[i] + Lnz[i];
        for (int p = F->Lp[i]; p < p2; ++p)
          Y[F->Li[p]] -= F->Lx[p] * yi;
        const double l_ki = yi / F->D[i];
        F->D[k] -= l_ki * yi;
        F->Li[p2] = k;
        F->Lx[p2] = l_ki;
        ++Lnz[i];
      }
      if (F->D[k] == 0.)
      {
        rc = -1;
        goto done;
      }
    }
  }
done:
  free(Lnz);
  free(Flag);
  free(Pattern);
  free(fill);
  return rc;
}

long oracle_ldl_lnz(const OracleLdl* F) { return F->lnz; }
double oracle_ldl_flops(const OracleLdl* F) { return F->flops; }

/* x = K^-1 b: permute, L solve, D solve, L^T solve, unpermute */
void oracle_ldl_solve(const OracleLdl* F, const double* b, double* x)
{
  const int n = F->n;
  double* y   = F->work;
  for (int k = 0; k < n; ++k)
    y[k] = b[F->P[k]];
  for (int j = 0; j < n; ++j)
    for (int p = F->Lp[j]; p < F->Lp[j + 1]; ++p)
      y[F->Li[p]] -= F->Lx[p] * y[j];
  for (int j = 0; j < n; ++j)
    y[j] /= F->D[j];
  for (int j = n - 1; j >= 0; --j)
    for (int p = F->Lp[j]; p < F->Lp[j + 1]; ++p)
      y[j] -= F->Lx[p] * y[F->Li[p]];
  for (int k = 0; k < n; ++k)
    x[F->P[k]] = y[k];
  for (int k = 0; k < n; ++k)
    y[k] = 0.;
}
