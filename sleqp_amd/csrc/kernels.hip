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
// Wave = 64 lanes; every block is 256 threads (4 waves, one per SIMD).  All
// arithmetic is fp64.  The paths are HBM/L2-bound integer + fp64 streaming
// work: coalesced column-major panel accesses, LDS for the dense diagonal
// blocks and the Schur-update tiles.
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace hipfact {

constexpr int FB = 256;  // threads per block

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------
// M values -> panels
// ---------------------------------------------------------------------------

// Saddle mode: elimination of the n leaf columns x of K = [I A^T; A 0].
// M(i,k) = S(i,k) = sum_j A(i,j) A(k,j), summed in the fixed order of the
// product list (deterministic), written straight into its front panel.
__global__ __launch_bounds__(FB) void k_mvals_prod(long long nM, const long long* __restrict__ prod_ptr,
                                                   const int* __restrict__ prod_a, const int* __restrict__ prod_b,
                                                   const long long* __restrict__ target,
                                                   const double* __restrict__ Kval, double* __restrict__ L) {
  for (long long e = blockIdx.x * (long long)FB + threadIdx.x; e < nM; e += (long long)gridDim.x * FB) {
    double s = 0.0;
    const long long p1 = prod_ptr[e + 1];
    for (long long p = prod_ptr[e]; p < p1; ++p) s += Kval[prod_a[p]] * Kval[prod_b[p]];
    L[target[e]] = s;
  }
}

// Generic mode: M(e) = K(src[e]).
__global__ __launch_bounds__(FB) void k_mvals_src(long long nM, const int* __restrict__ src,
                                                  const long long* __restrict__ target,
                                                  const double* __restrict__ Kval, double* __restrict__ L) {
  for (long long e = blockIdx.x * (long long)FB + threadIdx.x; e < nM; e += (long long)gridDim.x * FB) {
    const int s = src[e];
    L[target[e]] = s >= 0 ? Kval[s] : 0.0;
  }
}

__global__ __launch_bounds__(FB) void k_gather(long long n, const int* __restrict__ src,
                                               const double* __restrict__ in, double* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    out[i] = in[src[i]];
}

// ---------------------------------------------------------------------------
// Front factorisation: one workgroup per front, all fronts of one
// elimination-tree level per launch.
//
//   A0  zero the update matrix U_s
//   A1  extend-add the children's update matrices (relative indices)
//   B   dense LDL^T of the w x w pivot block in LDS, then its inverse
//       (recursive doubling) so that the solves are pure GEMVs
//   C   L21 = P21 * inv(L11)^T * D^-1   (row-parallel)
//   D   U_s -= L21 D L21^T              (64 x 64 tiles through LDS)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FB) void k_factor_level(const SnDesc* __restrict__ sn,
                                                     const int* __restrict__ level_sn, double* __restrict__ L,
                                                     double* __restrict__ U, const int* __restrict__ rel,
                                                     const int* __restrict__ child_idx, int* __restrict__ info) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const int w = S.w, r = S.r, u = r - w;
  double* __restrict__ P = L + S.Loff;
  double* __restrict__ Us = U + S.Uoff;
  const int wp = (w + 15) & ~15;
  double* A = lds;             // wp x wp, ld = wp
  double* dd = A + wp * wp;    // wp pivots
  double* TA = dd + wp;        // 16 x 64
  double* TB = TA + 1024;      // 16 x 64

  // ---- A0
  for (long long i = tid; i < (long long)u * u; i += FB) Us[i] = 0.0;
  __syncthreads();
  // ---- A1
  for (int ci = S.child_begin; ci < S.child_end; ++ci) {
    const SnDesc Cd = sn[child_idx[ci]];
    const int uc = Cd.r - Cd.w;
    const double* __restrict__ Uc = U + Cd.Uoff;
    const int* __restrict__ rc = rel + Cd.reloff;
    for (int b = wave; b < uc; b += 4) {
      const int tb = rc[b];
      const double* col = Uc + (long long)b * uc;
      if (tb < w) {
        double* dst = P + (long long)tb * r;
        for (int a = b + lane; a < uc; a += 64) dst[rc[a]] += col[a];
      } else {
        double* dst = Us + (long long)(tb - w) * u - w;
        for (int a = b + lane; a < uc; a += 64) dst[rc[a]] += col[a];
      }
    }
    __syncthreads();
  }

  // ---- B: pivot block into LDS (lower triangle, identity padding)
  for (int k = wave; k < wp; k += 4)
    for (int i = lane; i < wp; i += 64) {
      double v = (i == k) ? 1.0 : 0.0;
      if (i < w && k < w) v = (i >= k) ? P[i + (long long)k * r] : 0.0;
      A[i + k * wp] = v;
    }
  __syncthreads();
  {
    const int tx = tid & 15, ty = tid >> 4;
    for (int k = 0; k < w; ++k) {
      double d = A[k + k * wp];
      if (d == 0.0 || !(fabs(d) <= 1.7e308)) {  // exactly singular or non-finite
        if (tid == 0) atomicAdd(&info[INFO_ZERO_PIVOT], 1);
        d = 1.0;
      }
      if (tid == 0) {
        dd[k] = d;
        if (d < 0.0) atomicAdd(&info[INFO_NEG_PIVOT], 1);
      }
      const double dinv = 1.0 / d;
      for (int j = k + 1 + ty; j < w; j += 16) {
        const double ajk = A[j + k * wp] * dinv;
        for (int i = j + tx; i < w; i += 16) A[i + j * wp] -= A[i + k * wp] * ajk;
      }
      __syncthreads();
    }
    if (tid >= w && tid < wp) dd[tid] = 1.0;
    __syncthreads();
    // unit lower L11 (strict part scaled by 1/d), unit diagonal
    for (int k = wave; k < w; k += 4) {
      const double dinv = 1.0 / dd[k];
      for (int i = k + lane; i < w; i += 64) A[i + k * wp] = (i == k) ? 1.0 : A[i + k * wp] * dinv;
    }
    __syncthreads();
  }
  // ---- B2: inverse of the unit lower triangular block, in place.
  {
    // base: 16 x 16 diagonal blocks, one thread per column, staged in registers
    const int nb = wp >> 4;
    double x[16];
    const int q = tid >> 4, c = tid & 15;
    if (tid < nb * 16) {
      const double* Bq = A + (q * 16) + (q * 16) * wp;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
        for (int t = 0; t < 16; ++t)
          if (t < i) s -= Bq[i + t * wp] * x[t];
        x[i] = s;
      }
    }
    __syncthreads();
    if (tid < nb * 16) {
      double* Bq = A + (q * 16) + (q * 16) * wp;
#pragma unroll
      for (int i = 0; i < 16; ++i) Bq[i + c * wp] = x[i];
    }
    __syncthreads();
    // doubling: inv([X11 0; B X22]) has off-diagonal block -X22 * B * X11
    for (int h = 16; h < wp; h <<= 1) {
      const int hs = 31 - __clz(h);  // log2(h)
      const int ntask = (wp + 2 * h - 1) / (2 * h);
      const int total = ntask << (2 * hs);
      double acc[16];
      // step 1: B := B * X11
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        acc[qq] = 0.0;
        const int e = tid + qq * FB;
        if (e < total) {
          const int task = e >> (2 * hs), rem = e & ((1 << (2 * hs)) - 1);
          const int i = rem & (h - 1), cc = rem >> hs;
          const int b = task * 2 * h;
          const int h2 = min(h, wp - b - h);
          if (i < h2) {
            const double* Brow = A + (b + h + i) + b * wp;
            const double* X = A + b + (b + cc) * wp;
            double s = 0.0;
            for (int t = cc; t < h; ++t) s += Brow[t * wp] * X[t];
            acc[qq] = s;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        const int e = tid + qq * FB;
        if (e < total) {
          const int task = e >> (2 * hs), rem = e & ((1 << (2 * hs)) - 1);
          const int i = rem & (h - 1), cc = rem >> hs;
          const int b = task * 2 * h;
          const int h2 = min(h, wp - b - h);
          if (i < h2) A[(b + h + i) + (b + cc) * wp] = acc[qq];
        }
      }
      __syncthreads();
      // step 2: B := -X22 * B
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        acc[qq] = 0.0;
        const int e = tid + qq * FB;
        if (e < total) {
          const int task = e >> (2 * hs), rem = e & ((1 << (2 * hs)) - 1);
          const int i = rem & (h - 1), cc = rem >> hs;
          const int b = task * 2 * h;
          const int h2 = min(h, wp - b - h);
          if (i < h2) {
            const double* X22row = A + (b + h + i) + (b + h) * wp;  // X22[i,t] = X22row[t*wp]
            const double* Bcol = A + (b + h) + (b + cc) * wp;       // B[t,cc] = Bcol[t]
            double s = 0.0;
            for (int t = 0; t <= i; ++t) s += X22row[t * wp] * Bcol[t];
            acc[qq] = -s;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) {
        const int e = tid + qq * FB;
        if (e < total) {
          const int task = e >> (2 * hs), rem = e & ((1 << (2 * hs)) - 1);
          const int i = rem & (h - 1), cc = rem >> hs;
          const int b = task * 2 * h;
          const int h2 = min(h, wp - b - h);
          if (i < h2) A[(b + h + i) + (b + cc) * wp] = acc[qq];
        }
      }
      __syncthreads();
    }
  }
  // store inv(L11) (strict lower) and the pivots (diagonal) back to the panel
  for (int k = wave; k < w; k += 4)
    for (int i = k + lane; i < w; i += 64) P[i + (long long)k * r] = (i == k) ? dd[k] : A[i + k * wp];

  // ---- C: L21 = P21 * X^T * D^-1, X = inv(L11); one thread per row, in place,
  // column chunks in descending order (a chunk only reads columns <= its own).
  for (int i = w + tid; i < r; i += FB) {
    double* Prow = P + i;
    for (int c0 = ((w - 1) >> 3) << 3; c0 >= 0; c0 -= 8) {
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const int cmax = min(c0 + 8, w) - 1;
      const double* Xc = A + c0;
      for (int t = 0; t <= cmax; ++t) {
        const double p = Prow[(long long)t * r];
#pragma unroll
        for (int qq = 0; qq < 8; ++qq) acc[qq] += p * Xc[qq + t * wp];
      }
#pragma unroll
      for (int qq = 0; qq < 8; ++qq)
        if (c0 + qq < w) Prow[(long long)(c0 + qq) * r] = acc[qq] / dd[c0 + qq];
    }
  }
  __syncthreads();

  // ---- D: U_s -= L21 D L21^T
  {
    const int tx = tid & 15, ty = tid >> 4;
    const double* __restrict__ P21 = P + w;
    for (int J0 = 0; J0 < u; J0 += 64)
      for (int I0 = J0; I0 < u; I0 += 64) {
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
        for (int k0 = 0; k0 < w; k0 += 16) {
          __syncthreads();
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = tid + qq * FB;
            const int i = e & 63, kk = e >> 6;
            const int k = k0 + kk;
            double va = 0.0, vb = 0.0;
            if (k < w) {
              if (I0 + i < u) va = P21[(I0 + i) + (long long)k * r];
              if (J0 + i < u) vb = P21[(J0 + i) + (long long)k * r] * dd[k];
            }
            TA[e] = va;
            TB[e] = vb;
          }
          __syncthreads();
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = TA[kk * 64 + tx + 16 * a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = TB[kk * 64 + ty + 16 * b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
              for (int b = 0; b < 4; ++b) acc[a][b] += av[a] * bv[b];
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int j = J0 + ty + 16 * b;
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            const int i = I0 + tx + 16 * a;
            if (i < u && j < u && i >= j) Us[i + (long long)j * u] -= acc[a][b];
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------
// Level-scheduled solves.  y holds the right-hand side in pivot order on entry
// and the solution on exit.  Forward: children -> parents, the coupling to the
// ancestors travels as update vectors (deterministic, no atomics).  Backward:
// parents -> children, gathers the ancestors' solution.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FB) void k_fwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
                                                  const double* __restrict__ L, const int* __restrict__ rel,
                                                  const int* __restrict__ child_idx, double* __restrict__ y,
                                                  double* __restrict__ uvec) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const int w = S.w, r = S.r, u = r - w;
  const double* __restrict__ P = L + S.Loff;
  double* f = lds;       // r
  double* xs = lds + r;  // w
  for (int t = tid; t < r; t += FB) f[t] = (t < w) ? y[S.c0 + t] : 0.0;
  __syncthreads();
  for (int ci = S.child_begin; ci < S.child_end; ++ci) {
    const SnDesc Cd = sn[child_idx[ci]];
    const int uc = Cd.r - Cd.w;
    const double* __restrict__ uv = uvec + Cd.uoff;
    const int* __restrict__ rc = rel + Cd.reloff;
    for (int a = tid; a < uc; a += FB) f[rc[a]] += uv[a];
    __syncthreads();
  }
  // x = inv(L11) f_top
  for (int k = tid; k < w; k += FB) {
    double s = f[k];
    const double* Xk = P + k;
    for (int t = 0; t < k; ++t) s += Xk[(long long)t * r] * f[t];
    xs[k] = s;
    y[S.c0 + k] = s;
  }
  __syncthreads();
  double* __restrict__ us = uvec + S.uoff;
  for (int a = tid; a < u; a += FB) {
    double s = f[w + a];
    const double* Lr = P + w + a;
    for (int k = 0; k < w; ++k) s -= Lr[(long long)k * r] * xs[k];
    us[a] = s;
  }
}

__global__ __launch_bounds__(FB) void k_bwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
                                                  const double* __restrict__ L, const int* __restrict__ rows,
                                                  double* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const int w = S.w, r = S.r, u = r - w;
  const double* __restrict__ P = L + S.Loff;
  const int* __restrict__ rw = rows + S.rowoff + w;
  double* g = lds;      // u
  double* v = lds + u;  // w
  for (int a = tid; a < u; a += FB) g[a] = y[rw[a]];
  __syncthreads();
  for (int k = wave; k < w; k += 4) {
    const double* col = P + w + (long long)k * r;
    double s = 0.0;
    for (int a = lane; a < u; a += 64) s += col[a] * g[a];
    s = wave_sum(s);
    if (lane == 0) v[k] = y[S.c0 + k] / P[k + (long long)k * r] - s;
  }
  __syncthreads();
  for (int k = wave; k < w; k += 4) {
    const double* col = P + (long long)k * r;
    double s = 0.0;
    for (int t = k + 1 + lane; t < w; t += 64) s += col[t] * v[t];
    s = wave_sum(s);
    if (lane == 0) y[S.c0 + k] = v[k] + s;
  }
}

// ---------------------------------------------------------------------------
// Saddle-point front / back end: K = [I A^T; A 0].
//   t_p  = A_p b_x - b_y[perm]          (right-hand side of S y = t)
//   z_x  = b_x - A^T y ;  z_y = y       (back substitution of the leaf columns)
//   res  = b - K z                      (iterative refinement)
// A_p is the CSR of A with rows in pivot order; the x update walks the columns
// of K itself (its CSC arrays are the CSR of A^T).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FB) void k_rhs_saddle(int m, int n, const int* __restrict__ Ar_ptr,
                                                   const int* __restrict__ Ar_col, const double* __restrict__ Ar_val,
                                                   const int* __restrict__ perm, const double* __restrict__ b,
                                                   double* __restrict__ t) {
  for (int k = blockIdx.x * FB + threadIdx.x; k < m; k += gridDim.x * FB) {
    double s = 0.0;
    const int p1 = Ar_ptr[k + 1];
    for (int p = Ar_ptr[k]; p < p1; ++p) s += Ar_val[p] * b[Ar_col[p]];
    t[k] = s - b[n + perm[k]];
  }
}

__global__ __launch_bounds__(FB) void k_x_saddle(int n, int m, const int* __restrict__ Kp,
                                                 const double* __restrict__ Kval, const int* __restrict__ Kc_y,
                                                 const int* __restrict__ perm, const double* __restrict__ yp,
                                                 const double* __restrict__ b, double* __restrict__ z) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < n + m; j += gridDim.x * FB) {
    if (j < n) {
      double s = b[j];
      const int e1 = Kp[j + 1];
      for (int e = Kp[j] + 1; e < e1; ++e) s -= Kval[e] * yp[Kc_y[e]];
      z[j] = s;
    } else {
      const int k = j - n;
      z[n + perm[k]] = yp[k];
    }
  }
}

// res = b - K z for the saddle matrix (columns < n of the lower CSC hold I and A).
__global__ __launch_bounds__(FB) void k_residual_saddle(int n, int m, const int* __restrict__ Kp,
                                                        const int* __restrict__ Ki, const double* __restrict__ Kval,
                                                        const int* __restrict__ Ar_ptr, const int* __restrict__ Ar_col,
                                                        const double* __restrict__ Ar_val,
                                                        const int* __restrict__ perm, const double* __restrict__ b,
                                                        const double* __restrict__ z, double* __restrict__ res) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < n + m; j += gridDim.x * FB) {
    if (j < n) {
      double s = b[j];
      const int e1 = Kp[j + 1];
      for (int e = Kp[j]; e < e1; ++e) s -= Kval[e] * z[Ki[e]];
      res[j] = s;
    } else {
      const int k = j - n;
      const int i = n + perm[k];
      double s = b[i];
      const int p1 = Ar_ptr[k + 1];
      for (int p = Ar_ptr[k]; p < p1; ++p) s -= Ar_val[p] * z[Ar_col[p]];
      res[i] = s;
    }
  }
}

// Generic mode residual: res = b - (L + L^T - diag) z with L lower CSC and its
// transpose (CSR of L) both resident.
__global__ __launch_bounds__(FB) void k_residual_sym(int N, const int* __restrict__ Kp, const int* __restrict__ Ki,
                                                     const double* __restrict__ Kval, const int* __restrict__ Tp,
                                                     const int* __restrict__ Ti, const int* __restrict__ Tsrc,
                                                     const double* __restrict__ b, const double* __restrict__ z,
                                                     double* __restrict__ res) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < N; j += gridDim.x * FB) {
    double s = b[j];
    for (int e = Kp[j]; e < Kp[j + 1]; ++e) s -= Kval[e] * z[Ki[e]];  // column j: rows >= j
    for (int p = Tp[j]; p < Tp[j + 1]; ++p)                            // row j: columns < j
      if (Ti[p] != j) s -= Kval[Tsrc[p]] * z[Ti[p]];
    res[j] = s;
  }
}

__global__ __launch_bounds__(FB) void k_axpy(long long n, double a, const double* __restrict__ x,
                                             double* __restrict__ y) {
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    y[i] += a * x[i];
}

__global__ __launch_bounds__(FB) void k_scatter(long long n, const int* __restrict__ idx,
                                                const double* __restrict__ in, double* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)FB + threadIdx.x; i < n; i += (long long)gridDim.x * FB)
    out[idx[i]] = in[i];
}

// sparse right-hand side -> dense (sleqp_vec_to_raw, sparse/vec.c:105-119); out pre-zeroed
__global__ __launch_bounds__(FB) void k_scatter_sparse(int nnz, const int* __restrict__ idx,
                                                       const double* __restrict__ val, double* __restrict__ out) {
  for (int i = blockIdx.x * FB + threadIdx.x; i < nnz; i += gridDim.x * FB) out[idx[i]] = val[i];
}

// pivot statistics: min/max |d| over all fronts (condition estimate)
__global__ __launch_bounds__(FB) void k_pivot_minmax(int ns, const SnDesc* __restrict__ sn,
                                                     const double* __restrict__ L, double* __restrict__ out) {
  __shared__ double smin[FB], smax[FB];
  double lo = 1.7e308, hi = 0.0;
  for (int s = blockIdx.x; s < ns; s += gridDim.x) {
    const SnDesc S = sn[s];
    for (int k = threadIdx.x; k < S.w; k += FB) {
      const double d = fabs(L[S.Loff + k + (long long)k * S.r]);
      lo = fmin(lo, d);
      hi = fmax(hi, d);
    }
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int o = FB / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = smin[0];
    out[2 * blockIdx.x + 1] = smax[0];
  }
}

// ---------------------------------------------------------------------------
// CSR SpMV (gather only, deterministic): y = M x with LANES lanes per row.
// mode 0: y_i = sum_p val[p] x[idx[p]]
// mode 2: symmetric from lower storage: rows of L (ptr/idx/val) plus columns
//         of L (ptr2/idx2/val2) without the diagonal.
// ---------------------------------------------------------------------------
template <int LANES>
__global__ __launch_bounds__(FB) void k_spmv_csr(int nrows, const int* __restrict__ ptr, const int* __restrict__ idx,
                                                 const double* __restrict__ val, const int* __restrict__ ptr2,
                                                 const int* __restrict__ idx2, const double* __restrict__ val2,
                                                 const double* __restrict__ x, double* __restrict__ y) {
  const int sub = threadIdx.x % LANES;
  const int rows_per_block = FB / LANES;
  for (int row = blockIdx.x * rows_per_block + threadIdx.x / LANES; row < nrows; row += gridDim.x * rows_per_block) {
    double s = 0.0;
    const int p1 = ptr[row + 1];
    for (int p = ptr[row] + sub; p < p1; p += LANES) s += val[p] * x[idx[p]];
    if (ptr2) {
      const int q1 = ptr2[row + 1];
      for (int q = ptr2[row] + sub; q < q1; q += LANES) {
        const int c = idx2[q];
        if (c != row) s += val2[q] * x[c];
      }
    }
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) s += __shfl_down(s, o, LANES);
    if (sub == 0) y[row] = s;
  }
}

template __global__ void k_spmv_csr<1>(int, const int*, const int*, const double*, const int*, const int*,
                                       const double*, const double*, double*);
template __global__ void k_spmv_csr<4>(int, const int*, const int*, const double*, const int*, const int*,
                                       const double*, const double*, double*);
template __global__ void k_spmv_csr<16>(int, const int*, const int*, const double*, const int*, const int*,
                                        const double*, const double*, double*);
template __global__ void k_spmv_csr<64>(int, const int*, const int*, const double*, const int*, const int*,
                                        const double*, const double*, double*);

// ---------------------------------------------------------------------------
// KKT assembly on the device: restatement of fill_aug_jac for LOWER backends
// (aug_jac/standard_aug_jac.c:135-237).  Column j < n of K receives, in this
// order: (j, j, 1); (n + var_index[j], j, 1) if the bound of x_j is active;
// (n + cons_index[i], j, J_ij) for every Jacobian entry of column j whose row
// is in the working set.  The |W| trailing columns stay empty.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(FB) void k_asm_count(int n, const int* __restrict__ jp, const int* __restrict__ ji,
                                                  const int* __restrict__ var_index,
                                                  const int* __restrict__ cons_index, int* __restrict__ cnt) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < n; j += gridDim.x * FB) {
    int c = 1 + (var_index[j] >= 0 ? 1 : 0);
    for (int e = jp[j]; e < jp[j + 1]; ++e) c += (cons_index[ji[e]] >= 0) ? 1 : 0;
    cnt[j] = c;
  }
}

// exclusive scan of cnt[0..n) into kp[0..n], kp[n..N] = total; single block
__global__ __launch_bounds__(1024) void k_asm_scan(int n, int N, const int* __restrict__ cnt, int* __restrict__ kp) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int j = base + tid;
    const int v = (j < n) ? cnt[j] : 0;
    part[tid] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int add = (tid >= o) ? part[tid - o] : 0;
      __syncthreads();
      part[tid] += add;
      __syncthreads();
    }
    if (j < n) kp[j] = carry + part[tid] - v;
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  for (int j = n + tid; j <= N; j += 1024) kp[j] = carry;
}

__global__ __launch_bounds__(FB) void k_asm_fill(int n, const int* __restrict__ jp, const int* __restrict__ ji,
                                                 const double* __restrict__ jx, const int* __restrict__ var_index,
                                                 const int* __restrict__ cons_index, const int* __restrict__ kp,
                                                 int* __restrict__ ki, double* __restrict__ kx) {
  for (int j = blockIdx.x * FB + threadIdx.x; j < n; j += gridDim.x * FB) {
    int e = kp[j];
    ki[e] = j;
    kx[e] = 1.0;
    ++e;
    const int vi = var_index[j];
    if (vi >= 0) {
      ki[e] = n + vi;
      kx[e] = 1.0;
      ++e;
    }
    for (int q = jp[j]; q < jp[j + 1]; ++q) {
      const int ci = cons_index[ji[q]];
      if (ci >= 0) {
        ki[e] = n + ci;
        kx[e] = jx[q];
        ++e;
      }
    }
  }
}

}  // namespace hipfact
