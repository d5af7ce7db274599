// gfx950 (MI355X / CDNA4) kernels of the hipfact KKT backend: translation unit of the numeric FACTORISATION.
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
//   kernels_front_fused.inc   panel solve + Schur tile of a front as one role of the dataflow launch
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
#include "kernels_mvals.inc"
#include "kernels_front_pivot.inc"
#include "kernels_front_update.inc"
#include "kernels_front_fused.inc"
#include "kernels_solve_panels.inc"
#include "kernels_factor_top.inc"
}  // namespace hipfact

// ---- the instances of the kernel templates the host runtime launches (it sees declarations only: kernels_decl.h)
namespace hipfact {
#define INST_MVALS(IDX, PK)                                                                                          \
  template __global__ void k_mvals_prod<IDX, PK>(long long, const IDX*, const int*, const int*, const IDX*, const double*, \
                                                double*, int, const LongProd*,                                       \
                                                double*, unsigned int*);
INST_MVALS(unsigned int, true)
INST_MVALS(unsigned int, false)
INST_MVALS(long long, true)
INST_MVALS(long long, false)
#undef INST_MVALS
#define INST_FRONT(CH)                                                                                              \
  template __global__ void k_front_pivot<CH>(const FrontItem*, double*, double*, int*, const int*, const int*,     \
                                             const PullDesc*, int);                                                 \
  template __global__ void k_front_panel<CH>(const FrontItem*, double*, double*, const int*, const int*,           \
                                             const PullDesc*, int);                                                 \
  template __global__ void k_front_schur<CH>(const FrontItem*, double*, double*, const int*, const int*,           \
                                             const PullDesc*, int);
INST_FRONT(true)
INST_FRONT(false)
#undef INST_FRONT
}  // namespace hipfact

#ifdef HIPFACT_TRACE
// in-kernel timeline of the dataflow launch (scripts/timeline.py; never part of the product build)
extern "C" int hipfact_debug_trace(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_trace), sizeof(long long) * hipfact::TRACE_WGS * 8);
}
extern "C" int hipfact_debug_trace_owner(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_own), sizeof(long long) * hipfact::TRACE_WGS * 8);
}
extern "C" int hipfact_debug_trace_pivot(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hipfact::g_piv), sizeof(long long) * hipfact::TRACE_WGS * 24);
}
#endif
