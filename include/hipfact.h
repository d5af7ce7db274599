/* hipfact — C ABI of the MI355X-native KKT linear-algebra backend for SLEQP.
 *
 * This is the drop-in boundary.  Every entry point takes plain pointers and
 * sizes (no torch types, no C++ types) and returns 0 on success or a negative
 * HIPFACT_E* code; hipfact_last_error() gives the message.  The reference
 * interface each entry point replaces is cited as (file:line) relative to
 * chrhansk/sleqp v1.0.2 `src/main/`.
 *
 * The SLEQP-side binding (fact_hipfact.c: the five SleqpFactCallbacks plus
 * sleqp_fact_create_default) is in shim/ and described in INTEGRATION.md.
 */
#ifndef HIPFACT_H
#define HIPFACT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPFACT_VERSION "0.1.0"

enum
{
  HIPFACT_OK         = 0,
  HIPFACT_EINVAL     = -1, /* bad argument / malformed matrix              */
  HIPFACT_EDEVICE    = -2, /* no usable HIP device, HIP runtime error       */
  HIPFACT_ESINGULAR  = -3, /* zero / non-finite pivot (working set rank-deficient) */
  HIPFACT_ENOMEM     = -4,
  HIPFACT_ESTATE     = -5, /* call protocol violated (solve before set_matrix ...) */
  HIPFACT_EINTERNAL  = -6
};

typedef struct hipfact_handle hipfact_handle;

/* ---- lifetime ---------------------------------------------------------- */

/* Creates one backend instance bound to HIP device `device` (-1: read
 * SLEQP_HIP_DEVICE, then LOCAL_RANK, else 0).  All device state (stream,
 * buffers, symbolic cache) lives inside the handle; hipSetDevice is issued on
 * every entry because the calling thread is arbitrary (thread_test.c:77-110).
 * Fails with HIPFACT_EDEVICE when no GPU is present: there is no CPU fallback.
 * Replaces: the `fact_data` constructors, e.g. ma57_data_create
 * (fact/fact_ma57.c:733-807), lapack_data_create (fact/fact_lapack.c:36-50). */
int hipfact_create(hipfact_handle** out, int device);

/* Replaces SLEQP_FACT_FREE (fact/fact_types.h:23). Nulls *handle.  Handles are
 * reference counted like every SLEQP object (sleqp_fact_capture / _release,
 * fact/fact.c:120-146): the device state is destroyed when the last reference
 * is freed.  hipfact_spmat objects hold a reference to their handle. */
int hipfact_free(hipfact_handle** handle);

/* One more reference to the handle (released with hipfact_free): lets a second
 * SLEQP object (the TR solver of shim/tr_hipfact.c) share the factorisation of
 * the augmented Jacobian without process-global bookkeeping. */
int hipfact_retain(hipfact_handle* handle);

/* Thread-local-free error text of the last failing call on this handle (or of
 * hipfact_create when handle == NULL). */
const char* hipfact_last_error(const hipfact_handle* handle);

/* Warning of the last hipfact_set_matrix / hipfact_assemble_kkt that returned HIPFACT_OK, or NULL.  One exists today:
 * the working set is rank deficient (a zero or wrongly signed pivot of A A^T) and K was factored with static pivoting -
 * what MA57 reports as "Success - rank deficient", a positive status that the reference's MA57_CHECK_ERROR lets pass
 * (fact/fact_ma57.c:41-42, 118-133).  Every pivot of A A^T is shifted by "static_pivot_delta" (1e-8; rows are
 * equilibrated to unit norm) and the solves are refined against the caller's K until the backward error stalls: the
 * null-space projection and min-norm solves (consistent right-hand sides, unique in x) come out as for the
 * deduplicated working set; a right-hand side outside the range of K still ends in HIPFACT_ESINGULAR at the solve.
 * Info keys: "num_perturbed" (pivots the unshifted attempt reported), "static_pivot_shift" (of the active
 * factorisation, 0 = none), "static_pivot_runs".  Option "static_pivot" = 0 restores HIPFACT_ESINGULAR at once. */
const char* hipfact_last_warning(const hipfact_handle* handle);

/* ---- SleqpFact callbacks ------------------------------------------------ */

/* Replaces SLEQP_FACT_SET_MATRIX (fact/fact_types.h:9-10; dispatched from
 * sleqp_fact_set_matrix, fact/fact.c:59-75).  K is the lower-triangular CSC
 * matrix built by fill_aug_jac (aug_jac/standard_aug_jac.c:135-237) and read
 * through sleqp_mat_num_cols / sleqp_mat_cols / sleqp_mat_rows / sleqp_mat_data
 * (sparse/pub_mat.h:65-114): `colptr[N+1]`, `rowidx[nnz]` strictly ascending per
 * column, `vals[nnz]`.  Host pointers; the arrays are copied/uploaded before the
 * call returns (K is refilled in place by the caller afterwards,
 * standard_aug_jac.c:143).  Symbolic analysis is cached and reused while the
 * pattern is unchanged.  The numeric factorisation runs on the device. */
int hipfact_set_matrix(hipfact_handle* h, int N, const int* colptr, const int* rowidx, const double* vals);

/* Replaces SLEQP_FACT_SOLVE (fact/fact_types.h:12) for a sparse right-hand
 * side given as the public SleqpVec fields (sparse/pub_vec.h:16-25): dimension
 * `dim` (must equal N), `nnz` entries `indices[k]` ascending / `data[k]`.
 * Equivalent to sleqp_vec_to_raw (sparse/vec.c:105-119) followed by the solve
 * (fact_ma57.c:627-711, fact_lapack.c:125-154).  The solution stays on the
 * device until hipfact_solution. */
int hipfact_solve_sparse(hipfact_handle* h, int dim, int nnz, const int* indices, const double* data);

/* Same with a dense host right-hand side of length N. */
int hipfact_solve_dense(hipfact_handle* h, const double* rhs);

/* Replaces SLEQP_FACT_SOLUTION (fact/fact_types.h:14-18): copies entries
 * [begin, end) of the last solution into `out` (host, end-begin doubles).  The
 * caller sparsifies with sleqp_vec_set_from_raw(sol, out, end-begin, zero_eps)
 * exactly as every reference backend does (fact_ma57.c:713-730,
 * fact_lapack.c:156-171).  May be called several times per solve with
 * different ranges (standard_aug_jac.c:337-338 vs :382-386). */
int hipfact_solution(hipfact_handle* h, double* out, int begin, int end);

/* The same without the copy: *view points at entries [begin, end) of the last
 * solution in page-locked host memory owned by the handle (valid until the
 * next solve / set_matrix on this handle).  Every reference backend keeps the
 * dense solution in a buffer of its own and sparsifies straight out of it
 * (sleqp_vec_set_from_raw(sol, ma57_data->rhs_sol + begin, ...),
 * fact_ma57.c:713-730; fact_cholmod.c:211-228; fact_umfpack.c:245-262); this
 * entry point gives the shim that buffer.  The solve entry points above queue
 * the transfer of the whole solution behind the solve, so this call waits for
 * one event and touches no device API otherwise. */
int hipfact_solution_view(hipfact_handle* h, const double** view, int begin, int end);

/* Replaces SLEQP_FACT_CONDITION (fact/fact_types.h:20-21; the callback may be
 * NULL in the reference, fact.c:104-118).  Returns the pivot-ratio estimate
 * max|d| / min|d| of the block-diagonal factor, like CHOLMOD's rcond-based
 * value (fact_cholmod.c:197-209). */
int hipfact_condition(hipfact_handle* h, double* condition);

/* ---- device-resident variants (no PCIe in the hot loop) ----------------- */

/* Numeric refactorisation with new values already in HBM (same pattern as the
 * last hipfact_set_matrix).  `d_vals` is a device pointer to nnz doubles in the
 * layout of THAT matrix (the caller's K, unit rows of active bounds included:
 * the backend scatters them into its own structure).  HIPFACT_ESTATE when the
 * active plan was assembled from a Jacobian (hipfact_assemble_kkt is its
 * refactorisation). */
int hipfact_refactor_device(hipfact_handle* h, const double* d_vals);

/* Solve with the right-hand side resident in HBM (`d_rhs`, N doubles) and
 * write the solution to `d_sol` (N doubles, may alias d_rhs). */
int hipfact_solve_device(hipfact_handle* h, const double* d_rhs, double* d_sol);

/* Device pointer to the last solution (N doubles, owned by the handle). */
int hipfact_solution_device(hipfact_handle* h, const double** d_sol);

/* Blocks until the queued work has finished and reports what the asynchronous
 * entry points above could not: a singular / rank-deficient factorisation
 * (HIPFACT_ESINGULAR), a solve whose iterative refinement stalled far above its
 * tolerance (HIPFACT_ESINGULAR, "numerically singular"), a dependency wait that
 * timed out inside a single-launch kernel (HIPFACT_EINTERNAL; the handle is put
 * back into a clean state and needs a new factorisation).  A refinement that
 * needs more passes than the solve graph carries is continued here. */
int hipfact_check(hipfact_handle* h);

/* Blocks until all work queued on the handle's stream has finished. */
int hipfact_synchronize(hipfact_handle* h);

/* The handle's HIP stream (hipStream_t as void*), for event timing. */
int hipfact_stream(hipfact_handle* h, void** stream);

/* ---- KKT assembly on the device ------------------------------------------ */

/* Replaces reserve_aug_jac + fill_aug_jac (aug_jac/standard_aug_jac.c:106-237)
 * for SLEQP_FACT_FLAGS_LOWER backends: builds the lower-triangular CSC K from
 * the constraint Jacobian J (m_total x n, CSC: `j_colptr[n+1]`, `j_rowidx`,
 * `j_vals`, host pointers) and the working-set index maps
 * `var_index[n]` (sleqp_working_set_var_index, working_set.c:199-205; -1 = inactive)
 * and `cons_index[m_total]` (sleqp_working_set_cons_index, working_set.c:191-197;
 * already offset by the number of active variables, -1 = inactive).
 * `working_set_size` = |W|.  The result stays on the device and becomes the
 * matrix of the handle (as if hipfact_set_matrix had been called); its CSC
 * arrays are bit-identical to what fill_aug_jac produces.  When out pointers
 * are non-NULL the assembled arrays are also copied back (k_colptr: n+|W|+1
 * ints; k_rowidx/k_vals: *k_nnz entries, capacity n + nnz(J) + #active vars). */
int hipfact_assemble_kkt(hipfact_handle* h, int n, int m_total, const int* j_colptr, const int* j_rowidx,
                         const double* j_vals, const int* var_index, const int* cons_index, int working_set_size,
                         int* k_nnz, int* k_colptr, int* k_rowidx, double* k_vals);

/* ---- the sparse reduced matrix (PSD route) -------------------------------- */

/* Replaces the assembler of the reduced augmented Jacobian (compute_reduced_matrix,
 * aug_jac/reduced_aug_jac.c:323-377: O(|W|^2) merge-joins that push EVERY (row, col) pair, i.e. a dense
 * lower triangle, with an int overflow at |W| >= 46341, :271): S = A_W A_W^T as a SPARSE lower-triangular
 * CSC matrix in working-set row order, computed on the device from the product lists of the symbolic plan
 * (one fixed-order sum per structural entry of S).  This is the matrix the saddle mode factorises (after
 * equilibration): the device AugJac of shim/aug_jac_hipfact.c is at the same time the sparse, correct
 * reduced path - min-norm / LSQ / projection are solved through the Cholesky factor of S.
 * Valid after hipfact_set_matrix of a saddle matrix.  Two calls: with colptr == NULL only *nnz is
 * returned; then colptr (|W| + 1), rowidx and vals (*nnz) are filled. */
int hipfact_reduced_matrix(hipfact_handle* h, int* nnz, int* colptr, int* rowidx, double* vals);

/* ---- sparse matrix-vector products ------------------------------------- */

typedef struct hipfact_spmat hipfact_spmat;

/* Uploads a CSC matrix (the SleqpMat layout, sparse/mat.c:11-25) once and
 * keeps both orientations resident so that y = M x and y = M^T x are
 * gather-only CSR kernels. */
int hipfact_spmat_create(hipfact_handle* h, int num_rows, int num_cols, const int* colptr, const int* rowidx,
                         const double* vals, hipfact_spmat** out);
int hipfact_spmat_update_values(hipfact_spmat* M, const double* vals);
int hipfact_spmat_free(hipfact_spmat** M);

/* Replaces sleqp_mat_mult_vec (sparse/mat.c:282-310): y = M x, x dense host
 * vector of length num_cols, y dense host vector of length num_rows. */
int hipfact_spmat_mult_vec(hipfact_spmat* M, const double* x, double* y);
/* Replaces sleqp_mat_mult_vec_trans (sparse/mat.c:312-363): y = M^T x dense;
 * the caller applies the |y_j| > eps filter when packing the SleqpVec. */
int hipfact_spmat_mult_vec_trans(hipfact_spmat* M, const double* x, double* y);
/* Symmetric product from a lower-triangular CSC matrix, the explicit-Hessian
 * precedent prod_from_hess_matrix (bindings/mex/mex_hess.c:85-139). */
int hipfact_spmat_mult_vec_sym(hipfact_spmat* M, const double* x, double* y);
/* Device-resident forms: trans = 0 (M x), 1 (M^T x), 2 (symmetric-from-lower). */
int hipfact_spmat_mult_device(hipfact_spmat* M, int trans, const double* d_x, double* d_y);

/* ---- device-resident projected CG (the caller of the path) -------------- */

/* Replaces the loop of steihaug_solver_solve (tr/steihaug_solver.c:218-496, the
 * SleqpTRCallbacks.solve slot of tr/tr_types.h:9-29, selected with TR_SOLVER=CG or
 * AUTO + SLEQP_FUNC_HESS_PSD, newton.c:97-109) for problems whose Hessian of the
 * Lagrangian is available as an explicit matrix: every CG vector (z, r, g, d,
 * B d) stays in HBM, the null-space projection is the factorised KKT solve of
 * this handle (aug_jac_project_nullspace, standard_aug_jac.c:396-435), the
 * Hessian product is the symmetric SpMV of `hess` (lower-triangular CSC, the
 * prod_from_hess_matrix precedent, bindings/mex/mex_hess.c:85-139).  Per
 * iteration only three scalars cross PCIe instead of two n-vectors plus the
 * sparse<->dense marshal of the reference.
 *   gradient     host, n doubles           newton_step  host out, n doubles
 *   rel_tol      stat_eps * 1e-2 in the reference (steihaug_solver.c:21,241)
 *   max_iter     SLEQP_SETTINGS_INT_MAX_NEWTON_ITERATIONS (-1: none)
 *   tr_dual      out, multiplier of the trust-region constraint (steihaug_tr_dual,
 *                :187-216; -1 = SLEQP_NONE when the step is interior)
 *   iterations   out, number of CG iterations performed
 * Boundary hits use sleqp_tr_compute_bdry_sol (tr/tr_util.c:8-58).  Like the
 * reference, the step is zero when max_iter is exhausted before convergence. */
int hipfact_steihaug_solve(hipfact_handle* h, hipfact_spmat* hess, const double* gradient, double trust_radius,
                           double rel_tol, int max_iter, double* newton_step, double* tr_dual, int* iterations);

/* ---- matrix-free Hessian, Lanczos (the reference's default EQP solver) ---- */

/* product = H direction for host n-vectors; returns 0 on success.  The SLEQP
 * side wraps sleqp_problem_hess_prod (the matrix-free SLEQP_FUNC_HESS_PROD of
 * pub_func.h:168-172, dispatched at func.c:373-408) with the current
 * multipliers, shim/tr_hipfact.c. */
typedef int (*hipfact_hess_prod_fn)(void* user, const double* direction, double* product);

enum
{
  HIPFACT_TR_STEIHAUG = 0, /* projected Steihaug CG, tr/steihaug_solver.c:218-496 */
  HIPFACT_TR_GLTR     = 1  /* generalised Lanczos trust region, what trlib_krylov_min runs (tr/trlib_solver.c:322-352) */
};

/* Replaces the SleqpTRCallbacks.solve slot (tr/tr_types.h:9-29) for both of the
 * reference's Krylov solvers with every n-vector resident in HBM.  `hess` is an
 * explicit lower-triangular Hessian on this handle, or NULL: then `prod` is
 * called once per iteration with one n-vector crossing PCIe in each direction
 * (pinned staging).  GLTR continues on the trust-region boundary (Lanczos basis
 * kept in HBM, tridiagonal subproblem on the host) like trlib, where Steihaug
 * stops; convergence gamma_{k+1} |h_k| <= rel_tol * ||g||_P as trlib's
 * tol_rel_i / tol_rel_b (trlib_solver.c:272-275).  tr_dual: multiplier of the
 * trust-region constraint (GLTR: 0 when interior). */
int hipfact_tr_solve(hipfact_handle* h, int method, hipfact_spmat* hess, hipfact_hess_prod_fn prod, void* user,
                     const double* gradient, double trust_radius, double rel_tol, int max_iter, double* newton_step,
                     double* tr_dual, int* iterations);

/* The rest of the SleqpTRCallbacks contract (tr/tr_types.h:9-29): the `time_limit` argument of the solve slot and
 * the rayleigh slot (:18-20).
 *   time_limit    in, seconds from the call's entry, < 0 (SLEQP_NONE) = none.  The host loops look at the clock where
 *                 the reference does (steihaug_solver.c:297-310: top of every iteration; trlib_solver.c:631-636:
 *                 behind every completed iteration), the device-controlled loops whenever the host looks at their
 *                 control block (every 8 iterations).
 *   timed_out     out, 1 when the limit ended the iteration: the call still returns HIPFACT_OK and newton_step is the
 *                 iterate reached (feasible, inside the region); the SLEQP side returns SLEQP_ABORT_TIME
 *                 (steihaug_solver.c:490-492, trlib_solver.c:641-644), shim/tr_hipfact.c
 *   min_rayleigh  out, extremes of d.Bd / d.d over the CG directions, starting from 1 / 1 (steihaug_collect_rayleigh,
 *   max_rayleigh  steihaug_solver.c:150-182, 229-230); GLTR: of p.Hp / p.Mp over the equivalent CG directions while the
 *                 Lanczos tridiagonal is positive definite, of q.Hq / q.Mq = delta_k beyond (what trlib_rayleigh reads
 *                 from trlib's work array, trlib_solver.c:654-662); newton.c:328-343 reads them */
typedef struct hipfact_tr_extra
{
  double time_limit;
  int timed_out;
  double min_rayleigh;
  double max_rayleigh;
} hipfact_tr_extra;

/* hipfact_tr_solve with the extras above (extra == NULL: hipfact_tr_solve itself). */
int hipfact_tr_solve_ex(hipfact_handle* h, int method, hipfact_spmat* hess, hipfact_hess_prod_fn prod, void* user,
                        const double* gradient, double trust_radius, double rel_tol, int max_iter, double* newton_step,
                        double* tr_dual, int* iterations, hipfact_tr_extra* extra);

/* Host-only: the tridiagonal trust-region subproblem of GLTR (exposed for the tests).
 * min 1/2 h'Th + gamma0 e1'h, ||h|| <= radius; delta[0..k) diagonal, gamma[1..k) off-diagonal. */
int hipfact_tridiag_tr(int k, const double* delta, const double* gamma, double gamma0, double radius, double* h,
                       double* lambda);

/* ---- options / introspection ------------------------------------------- */

/* Options of hipfact_set_option.  Schedule options give bit-identical results for every setting unless their line
 * says otherwise (the parity tests sweep them); analysis options take effect at the next set_matrix / assemble, which
 * re-analyses.  The table is generated from the source: */
/* BEGIN OPTION TABLE (generated by scripts/gen_option_table.py from sleqp_amd/csrc/abi_options.inc)
 * 50 options; unknown names return HIPFACT_EINVAL.
 *   "refine_steps"
 *       correction passes carried by every solve graph (default 1; 0: plain solve, no residual); they
 *       return at once when the device-side control block reports convergence
 *   "refine_max"
 *       total correction passes of a solve, including those continued by hipfact_solution /
 *       hipfact_check (default 10)
 *   "refine_adaptive"
 *       0: every in-graph correction pass runs unconditionally
 *   "refine_tol"
 *       forward-error target (default 1e-10): the backward-error tolerance is refine_tol / condition
 *       estimate, clamped to [4.5e-16, 1e-12]
 *   "fail_omega"
 *       a solve whose refinement stalls above this backward error is reported as singular (default 1e-8)
 *   "static_pivot"
 *       0: a zero / wrongly signed pivot is HIPFACT_ESINGULAR at once (rounds 1 - 5)
 *   "static_pivot_delta"
 *       the shift of every pivot of A A^T when a rank-deficient working set is factored with static
 *       pivoting (default 1e-8; rows are equilibrated to unit norm)
 *   "equilibrate"
 *       takes effect at the next factorisation
 *   "use_graph"
 *       0: enqueue the launch sequences instead of replaying captured hipGraphs
 *   "top_max_fronts"
 *       0 disables the single-launch top-of-tree solve
 *   "factor_top_max"
 *       0: one launch per phase and level everywhere
 *   "wide_min_rows"
 *       fronts with at least this many update rows are solved by several workgroups in the per-level
 *       solve kernels (default 256; 0: off)
 *   "top_prefetch"
 *       0: the two-launch solve kernels fetch their panels behind the dependency wait (tests)
 *   "pull_max_children"
 *       0: extend-add always through the separate assembly kernel
 *   "debug_fake_timeout"
 *       test hook for the fallback to the per-level launches
 *   "factor_top_levels"
 *       at most this many levels in the single-launch top-of-tree factorisation (tests)
 *   "solve_slices"
 *       0: one item per front in the fused solve launch (fronts of up to 1024 rows only)
 *   "chain_pairs"
 *       dense chains: two fronts per trailing update (0: one Schur update per front)
 *   "solve_sorted"
 *       solve items of a level: biggest fronts first (0: plan order)
 *   "solve_whole_max"
 *       a front stays ONE solve item up to this many panel entries per thread
 *   "xupd_blocks"
 *       workgroups of the x update inside the tree launch
 *   "chain_fuse"
 *       0: single-front levels of a dense chain run pivot block and panel as two launches instead of one
 *       small dataflow launch
 *   "cg_residual_update"
 *       0: the projected CG keeps r as the reference's loop does
 *   "cg_device_loop"
 *       0: the host reads the dot products of every CG iteration (steihaug_impl)
 *   "lz_device_loop"
 *       0: GLTR with the host in every iteration (gltr_impl)
 *   "xupd_fused"
 *       0: x = b_x - A^T y as a launch of its own behind the tree (k_x_saddle)
 *   "refine_check_backoff"
 *       the interval between residual checks is multiplied by this after every check that passes
 *       (default 2; 1: fixed interval), up to 64 solves
 *   "refine_check_every"
 *       residual check on every k-th solve of a well-conditioned factorisation
 *   "decide_lazy"
 *       0: every solve graph ends with its own verdict launch
 *   "rhs_fused"
 *       0: k_rhs_saddle in front of the single-launch solve
 *   "spanel_fold"
 *       0: the solve panels in a launch of their own behind the factorisation
 *   "spanel_fold_room"
 *       solve-panel items dealt in beside a level's own pivot and panel items of the dataflow launch:
 *       workgroup slots per level (default 224)
 *   "factor_hint_peek"
 *       0: a refactorisation does not look at the last delivered refinement verdict (every first solve
 *       graph carries a correction pass)
 *   "top_block_breakeven"
 *       solves of one factorisation from which forming the dense top block of the solve tree pays
 *       (default 48)
 *   "solve_fused"
 *       0: the two-launch / per-level solve kernels on the factor panels
 *   "top_block_after"
 *       the top levels of the solve tree as one dense block from this solve of a factorisation on (0:
 *       never)
 *   "boundary_fast"
 *       0: host vectors through pageable borrows and three blocking points (round 3)
 *   "validate_rhs"
 *       walk the index array of every sparse right-hand side on the host (debug)
 *   "boundary_profile"
 *       where solve + solution spend their time (info keys bd_*); resets the sums
 *   "superset_vtable"
 *       0: no reuse across working sets (every changed pattern is analysed on its own rows; active
 *       bounds are still eliminated - exact_pattern = 1 for K as it is)
 *   "spmv_stream"
 *       0: every sparse product through the lanes-per-row kernel (default: matrices from spmv_stream_min
 *       entries on are streamed in row blocks)
 *   "spmv_stream_min"
 *       entries from which a sparse product is streamed (default 4 M: below, the matrix lives in the
 *       Infinity Cache)
 *   "exact_pattern"
 *       1: K is analysed exactly as given - no row dictionary, unit rows of active bounds stay in the
 *       structure (what hipfact_reduced_matrix needs)
 *   "assemble_superset"
 *       0: hipfact_assemble_kkt analyses every working set on its own rows (no superset plan)
 *   "plan_cache"
 *       inactive plan states kept (LRU); 0: one pattern at a time
 *   "profile"
 *       event-time every kernel class; value < 0 resets the counters
 *   "ordering"
 *       0 nested dissection + AMD leaves (default), 1 AMD on the whole graph, 2 natural
 *   "max_children"
 *       relaxed amalgamation keeps fronts at this many children (default 4 = what the pull extend-add
 *       takes in one block)
 *   "force_generic"
 *       1: no saddle-point structure detection, static 1 x 1 pivots on K as given (symmetric positive
 *       definite input: the PSD shim)
 *   "dense_mode"
 *       1: late elimination inside the tree (default), 2: low-rank correction, 0: off
 * END OPTION TABLE */
int hipfact_set_option(hipfact_handle* h, const char* name, double value);

/* Info (of the last finished solve: "last_omega" backward error in the
 * equilibrated space, "last_iters", "last_status" 0 converged / 1 stalled /
 * 2 non-finite / 3 pass limit, "kappa_est"); "N", "n", "m", "saddle", "nnzK", "nnzL", "nnzL_true", "flops",
 * "flops_dense", "nsuper", "nlevels", "nprod", "L_bytes", "U_bytes",
 * "analysis_s", "num_perturbed", "cache_hits", "max_r", "max_w",
 * "solve_bytes", "factor_bytes", ... */
int hipfact_get_info(const hipfact_handle* h, const char* name, double* value);

/* Debugging aid (tests, scripts): copies the first `bytes` of a named device buffer of the active plan
 * state to the host ("L", "SPf", "SPb", "sitems", "y", "xhat", "ysol", "uvec", "dscale"). */
int hipfact_debug_copy(hipfact_handle* h, const char* name, void* out, size_t bytes);

/* Host-only self-test of the worker pool behind the row dictionary's passes over K (no GPU needed): `callers` threads
 * run `rounds` threaded regions each, concurrently; returns the number of regions that did not cover their range exactly
 * once (0 = sound). */
int hipfact_debug_pool_selftest(int callers, int rounds);

/* ---- host-only symbolic plan (no GPU needed; used by the tests) ---------- */

typedef struct hipfact_plan hipfact_plan;
int hipfact_plan_create(int N, const int* colptr, const int* rowidx, const double* vals /* nullable */,
                        hipfact_plan** out);
void hipfact_plan_free(hipfact_plan** plan);
const char* hipfact_plan_error(const hipfact_plan* plan);
/* Exposes a named array of the plan (see sleqp_amd/csrc/plan.h); elem_size is
 * 4 (int32) or 8 (int64). */
int hipfact_plan_array(const hipfact_plan* plan, const char* name, const void** data, int64_t* len,
                       int* elem_size);
int hipfact_plan_scalar(const hipfact_plan* plan, const char* name, double* value);

#ifdef __cplusplus
}
#endif

#endif /* HIPFACT_H */
