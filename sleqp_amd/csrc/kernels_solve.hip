// gfx950 (MI355X / CDNA4) kernels of the hipfact KKT backend: translation unit of the SOLVES, the saddle-point front / back end,
// the vector and product kernels, the dense-column correction and the device-controlled CG.
//
// Numeric phase of the supernodal multifrontal LDL^T (replacing what the
// reference delegates to MA57 / CHOLMOD / UMFPACK / LAPACK behind
// SLEQP_FACT_SET_MATRIX and SLEQP_FACT_SOLVE, fact/fact_types.h:9-12), the
// level-scheduled triangular solves, the saddle-point SpMV front/back ends,
// the device restatement of fill_aug_jac (aug_jac/standard_aug_jac.c:135-237)
// and the CSR SpMV replacing sleqp_mat_mult_vec / sleqp_mat_mult_vec_trans
// (sparse/mat.c:282-363).
//
// Wave = 64 lanes; blocks are 256 (streaming kernels, Schur tiles), 512 (pivot / panel
// workgroups, the dataflow factorisation) or 1024 threads (solves).  All arithmetic is fp64.
// The numeric factorisation and the solves are bound by the critical path of the elimination
// tree, so the code is organised around dependent memory round trips and issue slots rather
// than bandwidth: self-contained work items, pull-mode extend-add, prefetch before every
// dependency wait, and single-launch dataflow kernels (k_factor_top, k_solve_tree; k_fwd_top /
// k_bwd_top as the fallback) in which workgroups synchronise through counters and posted data
// (DESIGN.md sections 2 and 4).
//
// One translation unit, split by role (included below in this order):
//   kernels_mvals.inc         product lists -> entries of S = A A^T in the front panels
//   kernels_front_pivot.inc   pivot block of a front: blocked LDL^T as free-running waves, posted tiles
//   kernels_front_update.inc  panel solve, Schur tiles, the per-level kernels
//   kernels_solve_level.inc   level-scheduled / one-launch-per-direction solves (fallback)
//   kernels_factor_top.inc    k_factor_top: the upper levels of the factorisation as one dataflow launch
//   kernels_solve_wide.inc    wide fronts of the fallback solves
//   kernels_solve_tree.inc    k_solve_tree (the whole solve in one launch), solve panels
//   kernels_saddle.inc        row scaling, right-hand side, x update, residual, refinement verdict
//   kernels_vector.inc        Krylov vector kernels, CSR SpMV, fill_aug_jac on the device
// (dense_cols.inc and krylov_device.inc carry their own kernels next to the host code that launches them.)
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "kernel_types.h"

namespace hipfact {
#include "kernels_common.inc"
#include "kernels_solve_level.inc"
#include "kernels_solve_wide.inc"
#include "kernels_solve_tree.inc"
#include "kernels_saddle.inc"
#include "kernels_vector.inc"
}  // namespace hipfact

// kernels that live next to the host code that launches them (their host halves are compiled with hipfact.hip)
#define DENSE_COLS_KERNELS
#include "dense_cols.inc"
#undef DENSE_COLS_KERNELS
#define KRYLOV_DEVICE_KERNELS
#include "krylov_device.inc"
#undef KRYLOV_DEVICE_KERNELS

// ---- the instances of the kernel templates the host runtime launches (it sees declarations only: kernels_decl.h)
namespace hipfact {
#define INST_CG(L)                                                                                                   \
  template __global__ void k_cg_spmv_dots<L>(int, const CgCtl*, const int*, const int*, const double*, const int*,   \
                                             const int*, const double*, const double*, const double*, const double*, \
                                             double*, double*);
INST_CG(1)
INST_CG(4)
INST_CG(16)
INST_CG(64)
#undef INST_CG
#define INST_CGH(L)                                                                                                    \
  template __global__ void k_cg_head<L>(int, const CgCtl*, CgCtl*, const double*, int, int, const int*, const int*,    \
                                        const double*, const int*, const int*, const double*, const double*, double*,  \
                                        double*, const double*, const double*, double*);
INST_CGH(1)
INST_CGH(4)
INST_CGH(16)
INST_CGH(64)
#undef INST_CGH
#define INST_LZ(L)                                                                                                  \
  template __global__ void k_lz_spmv_dot<L>(int, const LzCtl*, const int*, const int*, const double*, const int*,   \
                                            const int*, const double*, const double*, double*, double*);
INST_LZ(1)
INST_LZ(4)
INST_LZ(16)
INST_LZ(64)
#undef INST_LZ
template __global__ void k_x_saddle<true>(int, int, const int*, const double*, const int*, const int*, SaddleMaps,
                                          const double*, const double*, double*, const int*, int*);
template __global__ void k_x_saddle<false>(int, int, const int*, const double*, const int*, const int*, SaddleMaps,
                                           const double*, const double*, double*, const int*, int*);
}  // namespace hipfact
