// gfx950 (MI355X / CDNA4) kernels of the hipfact KKT backend.
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

namespace hipfact {

constexpr int FB = 256;  // threads per block

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// sum over the workgroup in a fixed order (shuffle tree inside the waves, then the waves' partials in wave order);
// the result is valid in thread 0.  lds: NT / 64 doubles.
template <int NT>
__device__ __forceinline__ double block_sum_fixed(double s, double* lds) {
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
  __syncthreads();
  double a = 0.0;
  if (threadIdx.x == 0)
    for (int q = 0; q < NT / 64; ++q) a += lds[q];
  return a;
}
// Thread 0 of a workgroup hands in the partial sum of its segment.  Returns true (in thread 0) for the workgroup that
// arrives last, with the sum of all partials in segment order; the counter is put back to zero for the next use.
__device__ __forceinline__ bool seg_arrive(double partial, int poff, int idx, int nseg, double* __restrict__ segpart,
                                           unsigned int* __restrict__ cnt, double& total) {
  if (nseg == 1) {
    total = partial;
    return true;
  }
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(segpart + poff + idx),
                     (unsigned long long)__double_as_longlong(partial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned int seen = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  if (seen + 1u != (unsigned int)nseg) return false;
  double a = 0.0;
  for (int q = 0; q < nseg; ++q)
    a += __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(segpart + poff + q),
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  total = a;
  return true;
}
#include "kernels_mvals.inc"
#include "kernels_front_pivot.inc"
#include "kernels_front_update.inc"
#include "kernels_solve_level.inc"
#include "kernels_factor_top.inc"
#include "kernels_solve_wide.inc"
#include "kernels_solve_tree.inc"
#include "kernels_saddle.inc"
#include "kernels_vector.inc"
}  // namespace hipfact
