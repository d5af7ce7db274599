// Constants and plain-old-data argument types shared by the device translation units (kernels_factor.hip, kernels_solve.hip) and the host
// runtime (hipfact.hip): what a launch site has to know about a kernel besides its declaration (kernels_decl.h,
// generated from the kernel sources by scripts/gen_kernel_decls.py).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace hipfact {

constexpr int FB = 256;  // threads per block of the streaming kernels
constexpr int MV_LONG = 4096;  // product lists longer than this are summed in segments (kernels_mvals.inc)
constexpr int CL = 8;    // lanes per column of K in the x update / residual (columns hold ~11 entries)
constexpr int SB = 1024;  // threads per block of the solve kernels (16 waves hide the panel-read latency)
constexpr int DOT_BLOCKS = 128;
constexpr int RL = 16;  // lanes per row of A^_p in the right-hand side product
constexpr int ST = 1024;  // threads per workgroup
constexpr int SPB = 512;
constexpr int ZT = 64;   // tile of k_top_syrk (top block of the solve as one product)
constexpr int ZC = 32;   // ... pivots per staged chunk
constexpr int ZS = 512;  // ... pivots per segment of a tile's sum (k_top_syrk_mfma / _reduce)
constexpr int CG_BLOCKS = 2048;  // most blocks (= partial sums) of the product kernel: one pass over the rows per block when they suffice
constexpr int CG_CHUNK = 8;  // iterations per graph launch (= per look of the host at the control block)

// ---- refinement control block (described with the residual kernels below)
struct RefineCtl {
  int done;     // 1: stop (converged, stagnated or non-finite)
  int iters;    // correction passes applied so far
  int status;   // 0 converged, 1 stagnated above the tolerance, 2 non-finite residual, 3 still running
  int pending;  // 1: the residual of a solve has left its partial maxima and nobody has judged them yet (deferred verdict)
  int seq;      // number of solves whose first residual has been judged (lets the host match a copy to a solve)
  int pad;
  double omega;       // ||r^||_inf / (||z^||_inf + ||b^||_inf), equilibrated space
  double omega_prev;
  double tol;         // effective tolerance of this solve
  double kappa;       // pivot-ratio condition estimate used for it
  double rnorm, bnorm;  // ||r^||_inf, ||b^||_inf of the last residual (a statically pivoted factor is judged on r / b too)
};
struct DecideIn {
  RefineCtl* ctl;
  RefineCtl* hctl;  // pinned copy for the host
  const double* partials;
  int nblk;
  double target;
  const unsigned long long* minmax;
};
// (working-set maps and equilibration of the saddle-point front end: described with the saddle kernels below)
struct SaddleMaps {
  const int* __restrict__ vmap;
  const int* __restrict__ cmap;
  const double* __restrict__ dscale;  // per pivot position
  int n;
  // segments of the long rows of A and of the long columns of K (device_types.h: LONG_ROW / LONG_COL / LongSeg); the
  // streaming kernels skip such rows and columns in their lane-group loops and give every segment to a workgroup
  int nrseg, ncseg;
  const LongSeg* __restrict__ rseg;
  const LongSeg* __restrict__ cseg;
  double* __restrict__ segpart;     // partial sums, one slot per segment
  unsigned int* __restrict__ segcnt;  // arrival counters, one per long row / column (zero between uses)
};
// what the forward items of the single-launch solve need to form their own rows of t = A^_p b~_x - D b_y[perm]
// (Ar_ptr null: t was left in y by a launch in front)
struct RhsIn {
  const int* __restrict__ Ar_ptr;
  const int* __restrict__ Ar_col;
  const double* __restrict__ Ar_val;
  const int* __restrict__ perm;
  SaddleMaps M;
  const double* __restrict__ b;
};
struct XupdIn {
  int n;
  const int* __restrict__ Kp;
  const double* __restrict__ Ksc;
  const int* __restrict__ Kc_y;
  const int* __restrict__ perm;
  SaddleMaps M;
  const double* __restrict__ b;
  double* __restrict__ z;
  int acc;
  int nblocks;
  double* __restrict__ dot_out;  // per workgroup: partial of b_x . z_x (the r.g of a CG iteration), or null
};
struct CgCtl {
  double rg, z_nrm_sq, rel_tol_sq, rad_sq, alpha, beta, tau;
  double ray_min, ray_max;  // smallest / largest d.Bd / d.d seen (steihaug_collect_rayleigh, steihaug_solver.c:150-182); start at 1
  int state;  // 0 running, 1 interior solution (|r.g| small), 2 boundary, 3 negative curvature, 4 iteration cap
  int stop, apply, it, max_iter, pad;
};
// Control block of the device-controlled phase of GLTR (krylov_device.inc): the LDL^T recurrence of the Lanczos
// tridiagonal and the CG quantities that follow from it, O(1) per iteration
struct LzCtl {
  double tol, rad_sq, gamma0;
  double gam_prev, gam_cur;  // gamma_{k-1}, gamma_k of the iteration in flight
  double del_cur;            // delta_k of the iteration in flight
  double piv, fsub;          // T = L D L^T: pivot d_{k-1} and the forward-substitution value f_{k-1}
  double rho, s_nrm_sq, sp, p_nrm_sq;  // ||r||^2, ||s||^2, <s, p>, ||p||^2 of the equivalent CG iterate (M-norms)
  int k;      // Lanczos iterations 0 .. k-1 are complete when `stop` is set
  int stop;
  int state;  // 0 running, 1 converged, 2 the interior solution left the trust region, 3 pivot <= 0 (T not positive
              // definite), 5 iteration cap
  int kmax;
};

}  // namespace hipfact
