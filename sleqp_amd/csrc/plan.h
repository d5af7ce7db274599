// Symbolic plan of the hipfact KKT backend: everything the device numeric
// phase needs, computed on the host once per sparsity pattern of K and cached.
//
// K is the lower-triangular CSC matrix handed to SLEQP_FACT_SET_MATRIX
// (reference fact/fact_types.h:9-10), built by fill_aug_jac
// (aug_jac/standard_aug_jac.c:135-237):  K = [ I  A^T ; A  0 ],  N = n + m.
//
// Saddle mode (K has exactly that shape, unit (1,1) block): the LDL^T of K
// under the constrained pivot order "every x before every y" is
//     L = [ I 0 ; A  L_s ],  D = diag(I, -D_s),  L_s D_s L_s^T = S = A A^T,
// so no pivot is ever zero (A has full row rank, pub_working_set.h:42-44) and the
// x columns are structurally trivial leaf supernodes.  They are eliminated by
// one product-list kernel (S values), the remaining m columns by a supernodal
// multifrontal LDL^T over the elimination tree of S; the x part of a solve is
// two SpMVs with A.
//
// Generic mode (anything else, e.g. an SPD / quasi-definite matrix): the same
// supernodal engine runs on M = K itself with static 1x1 pivots.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "hostmem.h"

namespace hipfact {

struct PlanParams {
  int ordering = 0;        // 0 = nested dissection + AMD leaves, 1 = AMD only, 2 = natural
  int nd_leaf = 0;         // 0 = automatic
  double nd_sep_frac = 0.2;
  int wmax = 128;          // widest supernode (diagonal block is LDS resident)
  // allowed explicit-zero fraction when merging a child into its parent: generous,
  // because every tree level costs a fixed ~0.15 ms of dependent kernel latency
  double relax_small = 0.6;   // merged width <= 32
  double relax_mid = 0.4;     // ... <= 64
  double relax_big = 0.3;     // ... wider
  int max_children = 4;       // merges must not create fronts with more children (0 = unlimited); = MAXCH of the device
  // Dense columns of the Jacobian (a variable that appears in a large share of the constraints): every such column
  // makes the rows it touches a clique of S = A A^T.  Columns with more than max(dense_min, dense_tau sqrt(m)) entries
  // (at most dense_max of them, the densest first) are left out of S; the device handles them by a low-rank
  // correction of every solve (hipfact.hip: dense columns).  0 = off.
  double dense_tau = 4.0;
  int dense_min = 64;
  int dense_max = 64;
  // How such columns are treated.
  //   1 (default)  LATE ELIMINATION inside the tree: x_d is not eliminated with the other leaf columns; it stays a
  //                vertex of the graph that is ordered behind every ordinary constraint row, i.e. the engine factors
  //                the symmetric quasi-definite matrix  M = [ A_s A_s^T  A_d ; A_d^T  -I ]  (y first: the pivots of
  //                the late variables are the negated capacitance matrix I + A_d^T S_s^-1 A_d, negative by inertia).
  //                What an ordering on K itself does with a dense column (AMD's dense-row rule behind MA57,
  //                fact_ma57.c:314-345, 761-763) - any number of columns, no extra solves per factorisation.
  //   2            the low-rank correction of round 3 (dense_cols.inc: at most 64 columns, k solves per factorisation)
  //   0            off: the cliques go into S
  int dense_mode = 1;
  // mode 1: a column is late from max(late_min, late_tau sqrt(m)) entries (its clique in S, c^2 / 2 entries, then
  // outweighs the at most m entries its late row of L can get), the densest late_max of them at most (0: automatic,
  // max(64, m / 4)); and whatever the counts, columns go late - densest first - until the product lists of S hold at
  // most prod_budget pairs (sum of c (c + 1) / 2 over the ordinary columns: the analysis never spends seconds building
  // lists before it notices that the pattern needs another strategy)
  double late_tau = 1.5;
  int late_min = 48;
  int late_max = 0;
  double prod_budget = 2.5e8;
  // Hub rows: constraint rows with more than max(hub_min, hub_tau sqrt(m)) neighbours in the graph of S (a budget-type
  // constraint that touches every variable is adjacent to every other row: no vertex separator exists around it) are
  // taken out of the graph the nested dissection sees and ordered last, like AMD's dense rows.  0 = off.
  double hub_tau = 10.0;
  int hub_min = 256;
  bool reuse_update_arena = true;  // a front may take over the update-matrix slot of a descendant two generations down
  bool adopt_leaves = true;   // childless fronts that are not adjacent to their parent are renumbered and merged into it
  bool force_generic = false;
};

struct Plan {
  // ---- problem
  int N = 0;  // dimension of K
  int n = 0;  // saddle: number of x columns; generic: 0
  int m = 0;  // order of M (saddle: constraint rows + late variables, generic: N)
  int my = 0;      // saddle: constraint rows (= N - n); vertices my .. m-1 of M are the late variables
  int n_late = 0;  // late variables (dense_mode 1): M = [A_s A_s^T  A_d; A_d^T  -I], exactly n_late negative pivots
  int n_late_rows = 0;  // constraint rows ordered behind the late variables (hub rows, rows without an ordinary entry)
  bool saddle = false;
  bool saddle_shape = false;  // the STRUCTURE is [I A^T; A 0] (saddle = structure and unit diagonal values)
  int n_shape = 0;            // ... with this many x columns
  int64_t nnzK = 0;
  std::vector<int> Kp, Ki;  // pattern of K the plan was built for (cache key)

  // ---- pivot order of M: perm[k] = original index of the k-th pivot
  std::vector<int> perm, iperm;

  // ---- M in pivot order, lower CSC (diagonal first in each column)
  std::vector<int64_t> Mp;  // m+1
  BigVec<int> Mi;           // row indices (pivot order)
  BigVec<int64_t> Mtarget;  // per entry: offset into the L arena

  // value sources.  saddle: M[e] = sum_t Kval[prod_a[t]] * Kval[prod_b[t]],
  // t in [prod_ptr[e], prod_ptr[e+1]).  generic: M[e] = Kval[src[e]] (-1: 0).
  BigVec<int64_t> prod_ptr;
  BigVec<int> prod_a, prod_b;
  std::vector<int> src;

  // ---- supernodes (fronts)
  int nsuper = 0;
  std::vector<int> sn_c0;       // nsuper+1 first column of each supernode
  std::vector<int> sn_r;        // rows in front (including own columns)
  std::vector<int64_t> sn_rowptr;  // nsuper+1 into sn_rows
  std::vector<int> sn_rows;     // sorted row structure, own columns first
  std::vector<int> sn_parent;   // -1 root
  std::vector<int> sn_level;    // 0 = leaf
  std::vector<int64_t> sn_Loff;  // panel offset in L arena (r x w, ld = r)
  std::vector<int64_t> sn_Uoff;  // update-matrix offset in U arena (u x u, ld = u)
  std::vector<int64_t> sn_uoff;  // update-vector offset (solve), length u
  std::vector<int> child_ptr, child_idx;  // children lists
  std::vector<int64_t> rel_ptr;  // nsuper+1 into rel (length u each)
  std::vector<int> rel;          // position of each below-row in the parent's front
  int nlevels = 0;
  std::vector<int> level_ptr, level_sn;  // supernodes grouped by level
  int64_t L_size = 0, U_size = 0, u_size = 0;

  // ---- saddle-mode SpMV structures
  std::vector<int> Ar_ptr;   // m+1: CSR of A with rows in pivot order
  std::vector<int> Ar_col;   // column (x index)
  std::vector<int> Ar_src;   // index into Kval
  std::vector<int> Kc_y;     // per K entry in columns < n: pivot position of its y row, -1 for the diagonal
  std::vector<int> dense_cols;  // x columns left out of S = A A^T (ascending); see PlanParams::dense_tau (dense_mode 2)
  std::vector<int> late_cols;   // dense_mode 1: late variable t (vertex my + t of M) is x column late_cols[t] (ascending)

  // ---- active bounds eliminated in front of the analysis (build_plan_bounds).  The plan above then is that of the
  // REDUCED matrix K' = [I A'^T; A' 0]: the unit rows of active bounds (working_set.c:139, standard_aug_jac.c:163-185)
  // are not rows of A', and the columns of the fixed variables hold nothing but their diagonal.  With beta = the
  // right-hand side of the unit rows:  x_B = beta,  K' [x; y'] = [b_x; b_y' - A'_B beta],  y_B = b_B - beta - A'_B^T y'.
  int N_ext = 0;                  // order of the caller's K (= N when nothing was eliminated)
  int n_bounds = 0;
  std::vector<int> bnd_row;       // per bound: its row among the caller's constraint rows (0 .. N_ext - n)
  std::vector<int> bnd_col;       // ... and the variable it fixes
  std::vector<int> row_ext;       // per row of A': its row among the caller's constraint rows
  std::vector<int> ent_ext;       // per entry of K': the entry of the caller's K it is
  std::vector<int> cut_ptr;       // per bound: the entries of A'_B in its column, [cut_ptr[t], cut_ptr[t + 1])
  std::vector<int> cut_row;       // ... row of A'
  std::vector<int> cut_ent;       // ... entry of the caller's K

  // ---- statistics
  int64_t nnzL = 0;       // entries of L incl. diagonal (M part, dense panels)
  int64_t nnzL_true = 0;  // same without relaxation zeros (column counts)
  double flops = 0;       // sum_j c_j^2 (true column counts)
  double flops_dense = 0; // flops executed on the dense fronts
  int64_t nprod = 0;
  int max_r = 0, max_w = 0, max_u = 0;
  double t_order = 0, t_symbolic = 0, t_total = 0;
  std::string error;
};

// Builds the plan for the pattern (N, Kp, Ki) of a lower-triangular CSC matrix.
// Kx may be null (pattern-only: the unit-diagonal test is then skipped).
// Returns false and sets plan.error on failure.
bool build_plan(int N, const int* Kp, const int* Ki, const double* Kx,
                const PlanParams& prm, Plan& plan);

// The same with the unit rows of active bounds eliminated first (see Plan::n_bounds): a row of the (2,1) block with a
// single entry, of value one when values are given, fixes its variable.  Left as vertices of S = A A^T such rows turn
// the eleven tree levels of SURVEY's config 4 into 28 at 10 % active bounds (every bound on x_j sits between all rows
// that hold x_j); eliminated, the tree is that of the other rows.  Anything that is not of the augmented shape, and
// matrices without such rows, go to build_plan unchanged.
bool build_plan_bounds(int N, const int* Kp, const int* Ki, const double* Kx,
                       const PlanParams& prm, Plan& plan);

}  // namespace hipfact
