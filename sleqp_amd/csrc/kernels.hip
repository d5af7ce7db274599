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
// Front factorisation: one workgroup (4 waves) per front, all fronts of one
// elimination-tree level per launch.  Every dense phase runs on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64, verified lane layout: A[l&15][l>>4],
// B[l>>4][l&15], D[(l>>4)+4q][l&15]).
//
//   A0  zero the update matrix U_s
//   A1  extend-add the children's update matrices (relative indices)
//   B   blocked LDL^T (nb = 16) of the w x w pivot block in LDS:
//         S1  16x16 diagonal block: LDL^T and its inverse in registers of one
//             wave (cross-lane shuffles, no memory traffic)
//         S2  block column  L_Ik = A_Ik inv(L_kk)^T D^-1          (MFMA)
//         S3  trailing update A_IJ -= L_Ik D L_Jk^T                (MFMA)
//       then the inverse of the unit lower factor by recursive doubling with
//       MFMA products, so that the solves are pure GEMVs
//   C   L21 = P21 inv(L11)^T D^-1, operands streamed from the panel  (MFMA)
//   D   U_s -= L21 D L21^T, 32x32 blocks per wave, operands streamed (MFMA)
//
// LDS: A (wp x lda, lda = wp + 1: both the row-fragment and the transposed
// fragment reads stay <= 2-way bank conflicted), the pivots, a 16-column panel Y.
// ---------------------------------------------------------------------------
typedef double d4_t __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(FB) void k_factor_level(const SnDesc* __restrict__ sn,
                                                     const int* __restrict__ level_sn, double* __restrict__ L,
                                                     double* __restrict__ U, const int* __restrict__ rel,
                                                     const int* __restrict__ child_idx, int* __restrict__ info,
                                                     int phases) {
  // phases: bit mask A(1) B(2) C(4) D(8); anything but 15 is a timing-only build
  // of the same kernel (results are then wrong by construction).
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const int w = S.w, r = S.r, u = r - w;
  double* __restrict__ P = L + S.Loff;
  double* __restrict__ Us = U + S.Uoff;
  const int wp = (w + 15) & ~15;
  const int nbk = wp >> 4;
  const int lda = wp + 1;
  double* A = lds;            // wp x lda
  double* dd = A + wp * lda;  // wp pivots
  double* Yp = dd + wp;       // wp x 16 panel (L * D of the current block column), ld = lda

  // ---- A0
  if (phases & 1) {
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
  }
  if (!(phases & 2)) return;

  // ---- B: pivot block into LDS (lower triangle, identity padding)
  for (int k = wave; k < wp; k += 4)
    for (int i = lane; i < wp; i += 64) {
      double v = (i == k) ? 1.0 : 0.0;
      if (i < w && k < w) v = (i >= k) ? P[i + (long long)k * r] : 0.0;
      A[i + k * lda] = v;
    }
  __syncthreads();

  for (int kb = 0; kb < nbk; ++kb) {
    const int k0 = kb << 4;
    // S1: wave 0 factors the diagonal block and inverts its unit lower factor.
    // lane (i = li, q = lk) owns A[i][4q..4q+3] and X[i][4q..4q+3].
    if (wave == 0) {
      double a[4], x[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a[c] = A[(k0 + li) + (k0 + 4 * lk + c) * lda];
        x[c] = (li == 4 * lk + c) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int qk = k >> 2, kr = k & 3;
        const double ck_i = __shfl(a[kr], (qk << 4) | li, 64);  // A[i][k]
        double d = __shfl(a[kr], (qk << 4) | k, 64);            // A[k][k]
        if (d == 0.0 || !(fabs(d) <= 1.7e308)) {
          if (lane == 0) atomicAdd(&info[INFO_ZERO_PIVOT], 1);
          d = 1.0;
        }
        if (lane == 0) {
          dd[k0 + k] = d;
          if (d < 0.0) atomicAdd(&info[INFO_NEG_PIVOT], 1);
        }
        const double l_ik = (li > k) ? ck_i / d : 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const double xk = __shfl(x[c], (lk << 4) | k, 64);  // X[k][4q+c]
          x[c] -= l_ik * xk;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = 4 * lk + c;
          const double ck_j = __shfl(a[kr], (qk << 4) | j, 64);  // A[j][k]
          if (j > k) a[c] -= l_ik * ck_j;
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) A[(k0 + li) + (k0 + 4 * lk + c) * lda] = x[c];
    }
    __syncthreads();
    // S2: block column.  Y_Ik = A_Ik X_kk^T, L_Ik = Y_Ik D^-1.
    for (int I = kb + 1 + wave; I < nbk; I += 4) {
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double av = A[(16 * I + li) + (k0 + 4 * s + lk) * lda];
        const double bv = A[(k0 + li) + (k0 + 4 * s + lk) * lda];
        acc = MFMA_F64(av, bv, acc);
      }
      const double dinv = 1.0 / dd[k0 + li];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * I + lk + 4 * q;
        Yp[row + li * lda] = acc[q];
        A[row + (k0 + li) * lda] = acc[q] * dinv;
      }
    }
    __syncthreads();
    // S3: trailing update of the lower block triangle.
    {
      const int T = nbk - kb - 1;
      const int ntiles = T * (T + 1) / 2;
      for (int t = wave; t < ntiles; t += 4) {
        int J = 0, rem = t;
        while (rem >= T - J) {
          rem -= T - J;
          ++J;
        }
        const int I = kb + 1 + J + rem;
        J += kb + 1;
        d4_t acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = A[(16 * I + lk + 4 * q) + (16 * J + li) * lda];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const double av = -A[(16 * I + li) + (k0 + 4 * s + lk) * lda];
          const double bv = Yp[(16 * J + li) + (4 * s + lk) * lda];
          acc = MFMA_F64(av, bv, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) A[(16 * I + lk + 4 * q) + (16 * J + li) * lda] = acc[q];
      }
    }
    __syncthreads();
  }

  // ---- B2: inverse of the unit lower block factor by recursive doubling.
  // A holds inv(L_kk) in the diagonal blocks and L_IJ below.  For the block
  // [X11 0; B X22] the off-diagonal block of the inverse is -X22 B X11; one wave
  // computes one 16-column strip of it in registers (the accumulator tiles of the
  // first product are the B operands of the second), then all strips are stored.
  for (int h = 16; h < wp; h <<= 1) {
    const int ht = h >> 4;
    const int ntask = (wp + 2 * h - 1) / (2 * h);
    const int units = ntask * ht;
    const int task = wave / ht, tc = wave - task * ht;
    const int b = task * 2 * h;
    const int h2 = min(h, wp - b - h);
    const bool active = (wave < units) && (h2 > 0);
    const int h2t = active ? (h2 >> 4) : 0;
    d4_t Tt[4], R[4];
    if (active) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        if (ti < h2t) {
          for (int tt = tc; tt < ht; ++tt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const double av = A[(b + h + 16 * ti + li) + (b + 16 * tt + 4 * s + lk) * lda];
              const double bv = A[(b + 16 * tt + 4 * s + lk) + (b + 16 * tc + li) * lda];
              acc = MFMA_F64(av, bv, acc);
            }
        }
        Tt[ti] = acc;
      }
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) {
        d4_t acc = {0.0, 0.0, 0.0, 0.0};
        if (ti < h2t) {
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            if (tt <= ti) {
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                const double av = A[(b + h + 16 * ti + li) + (b + h + 16 * tt + 4 * s + lk) * lda];
                acc = MFMA_F64(av, Tt[tt][s], acc);
              }
            }
        }
        R[ti] = acc;
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
        if (ti < h2t) {
#pragma unroll
          for (int q = 0; q < 4; ++q) A[(b + h + 16 * ti + lk + 4 * q) + (b + 16 * tc + li) * lda] = -R[ti][q];
        }
    }
    __syncthreads();
  }
  // store inv(L11) (strict lower) and the pivots (diagonal) back to the panel
  for (int k = wave; k < w; k += 4)
    for (int i = k + lane; i < w; i += 64) P[i + (long long)k * r] = (i == k) ? dd[k] : A[i + k * lda];

  if (!(phases & 4)) return;
  // ---- C: L21^T tiles = X P21^T, scaled by D^-1.  Each wave owns 16 panel rows
  // at a time: the B operand streams from the panel (16 consecutive rows per
  // k-step), the A operand is X from LDS; results are stored row-contiguous.
  for (int R0 = w + 16 * wave; R0 < r; R0 += 64) {
    const bool rok = (R0 + li) < r;
    const double* __restrict__ Prow = P + R0 + li;
    d4_t acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) acc[ct] = (d4_t){0.0, 0.0, 0.0, 0.0};
    double pv[4], pn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int col = 4 * s + lk;
      pv[s] = (rok && col < w) ? Prow[(long long)col * r] : 0.0;
    }
    for (int tt = 0; tt < nbk; ++tt) {
      if (tt + 1 < nbk) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int col = 16 * (tt + 1) + 4 * s + lk;
          pn[s] = (rok && col < w) ? Prow[(long long)col * r] : 0.0;
        }
      }
#pragma unroll
      for (int ct = 0; ct < 8; ++ct)
        if (ct >= tt && ct < nbk) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double xv = A[(16 * ct + li) + (16 * tt + 4 * s + lk) * lda];
            acc[ct] = MFMA_F64(xv, pv[s], acc[ct]);
          }
        }
#pragma unroll
      for (int s = 0; s < 4; ++s) pv[s] = pn[s];
    }
    if (rok) {
#pragma unroll
      for (int ct = 0; ct < 8; ++ct)
        if (ct < nbk) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int col = 16 * ct + lk + 4 * q;
            if (col < w) P[(R0 + li) + (long long)col * r] = acc[ct][q] / dd[col];
          }
        }
    }
  }
  __syncthreads();

  // ---- D: U_s -= L21 D L21^T.  One 32 x 32 block of the lower triangle per
  // wave and iteration; computed transposed (rows of the accumulator = j) so that
  // the read-modify-write of U is contiguous in i.
  if (u > 0 && (phases & 8)) {
    const double* __restrict__ P21 = P + w;
    const int nb32 = (u + 31) >> 5;
    const long long nblk = (long long)nb32 * (nb32 + 1) / 2;
    int J = 0;
    long long base = 0;  // first block index of block column J
    for (long long blk = wave; blk < nblk; blk += 4) {
      while (blk - base >= nb32 - J) {
        base += nb32 - J;
        ++J;
      }
      const int I = J + (int)(blk - base);
      const int i0 = 32 * I, j0 = 32 * J;
      d4_t acc[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = (d4_t){0.0, 0.0, 0.0, 0.0};
      const bool jok0 = (j0 + li) < u, jok1 = (j0 + 16 + li) < u;
      const bool iok0 = (i0 + li) < u, iok1 = (i0 + 16 + li) < u;
      const double* __restrict__ Pj = P21 + j0 + li;
      const double* __restrict__ Pi = P21 + i0 + li;
      double aj0, aj1, bi0, bi1;
      {
        const bool kok = lk < w;
        const long long off = (long long)lk * r;
        const double dk = kok ? dd[lk] : 0.0;
        aj0 = (kok && jok0) ? Pj[off] * dk : 0.0;
        aj1 = (kok && jok1) ? Pj[off + 16] * dk : 0.0;
        bi0 = (kok && iok0) ? Pi[off] : 0.0;
        bi1 = (kok && iok1) ? Pi[off + 16] : 0.0;
      }
      for (int k0 = 0; k0 < w; k0 += 4) {
        double naj0 = 0.0, naj1 = 0.0, nbi0 = 0.0, nbi1 = 0.0;
        const int kn = k0 + 4 + lk;
        if (kn < w) {
          const long long off = (long long)kn * r;
          const double dk = dd[kn];
          if (jok0) naj0 = Pj[off] * dk;
          if (jok1) naj1 = Pj[off + 16] * dk;
          if (iok0) nbi0 = Pi[off];
          if (iok1) nbi1 = Pi[off + 16];
        }
        acc[0][0] = MFMA_F64(aj0, bi0, acc[0][0]);
        acc[0][1] = MFMA_F64(aj0, bi1, acc[0][1]);
        acc[1][0] = MFMA_F64(aj1, bi0, acc[1][0]);
        acc[1][1] = MFMA_F64(aj1, bi1, acc[1][1]);
        aj0 = naj0;
        aj1 = naj1;
        bi0 = nbi0;
        bi1 = nbi1;
      }
      // acc[x][y][q] = update of U(i = i0 + 16 y + li, j = j0 + 16 x + lk + 4 q)
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
          const int i = i0 + 16 * y + li;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = j0 + 16 * x + lk + 4 * q;
            if (i < u && j < u && i >= j) Us[i + (long long)j * u] -= acc[x][y][q];
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
constexpr int SB = 1024;  // threads per block of the solve kernels (16 waves hide the panel-read latency)

__global__ __launch_bounds__(SB) void k_fwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
                                                  const double* __restrict__ L, const int* __restrict__ rel,
                                                  const int* __restrict__ child_idx, double* __restrict__ y,
                                                  double* __restrict__ uvec) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const SnDesc S = sn[level_sn[blockIdx.x]];
  const int w = S.w, r = S.r, u = r - w;
  const double* __restrict__ P = L + S.Loff;
  double* f = lds;              // r
  double* xs = f + r;           // w
  double* ps = xs + w;          // 8 x w partial sums of the triangular product
  double* part = ps + 8 * w;    // <= 1024 partial sums of the rectangular product
  for (int t = tid; t < r; t += SB) f[t] = (t < w) ? y[S.c0 + t] : 0.0;
  __syncthreads();
  for (int ci = S.child_begin; ci < S.child_end; ++ci) {
    const SnDesc Cd = sn[child_idx[ci]];
    const int uc = Cd.r - Cd.w;
    const double* __restrict__ uv = uvec + Cd.uoff;
    const int* __restrict__ rc = rel + Cd.reloff;
    for (int a = tid; a < uc; a += SB) f[rc[a]] += uv[a];
    __syncthreads();
  }
  // x = inv(L11) f_top: row k, eighth p of the column range [0, k)
  {
    const int k = tid & 127, p = tid >> 7;
    if (k < w) {
      const int lo = (int)(((long long)k * p) >> 3), hi = (int)(((long long)k * (p + 1)) >> 3);
      const double* Xk = P + k;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int t = lo;
      for (; t + 3 < hi; t += 4) {
        s0 += Xk[(long long)t * r] * f[t];
        s1 += Xk[(long long)(t + 1) * r] * f[t + 1];
        s2 += Xk[(long long)(t + 2) * r] * f[t + 2];
        s3 += Xk[(long long)(t + 3) * r] * f[t + 3];
      }
      for (; t < hi; ++t) s0 += Xk[(long long)t * r] * f[t];
      ps[p * w + k] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  for (int k = tid; k < w; k += SB) {
    double s = f[k];
#pragma unroll
    for (int p = 0; p < 8; ++p) s += ps[p * w + k];
    xs[k] = s;
    y[S.c0 + k] = s;
  }
  __syncthreads();
  if (u > 0) {
    // u_s = f_below - L21 x: 64-row chunks x column slices, fixed-order reduction
    const int nchunk = (u + 63) >> 6;
    const int nslice = nchunk >= 16 ? 1 : 16 / nchunk;
    if (nslice == 1) {
      double* __restrict__ us = uvec + S.uoff;
      for (int ch = wave; ch < nchunk; ch += 16) {
        const int a = (ch << 6) + lane;
        if (a < u) {
          const double* Lr = P + w + a;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
          int k = 0;
          for (; k + 3 < w; k += 4) {
            s0 += Lr[(long long)k * r] * xs[k];
            s1 += Lr[(long long)(k + 1) * r] * xs[k + 1];
            s2 += Lr[(long long)(k + 2) * r] * xs[k + 2];
            s3 += Lr[(long long)(k + 3) * r] * xs[k + 3];
          }
          for (; k < w; ++k) s0 += Lr[(long long)k * r] * xs[k];
          us[a] = f[w + a] - ((s0 + s1) + (s2 + s3));
        }
      }
    } else {
      const int ch = wave % nchunk, sl = wave / nchunk;
      if (sl < nslice) {
        const int a = (ch << 6) + lane;
        const int lo = (int)(((long long)w * sl) / nslice), hi = (int)(((long long)w * (sl + 1)) / nslice);
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (a < u) {
          const double* Lr = P + w + a;
          int k = lo;
          for (; k + 3 < hi; k += 4) {
            s0 += Lr[(long long)k * r] * xs[k];
            s1 += Lr[(long long)(k + 1) * r] * xs[k + 1];
            s2 += Lr[(long long)(k + 2) * r] * xs[k + 2];
            s3 += Lr[(long long)(k + 3) * r] * xs[k + 3];
          }
          for (; k < hi; ++k) s0 += Lr[(long long)k * r] * xs[k];
        }
        part[sl * (nchunk << 6) + (ch << 6) + lane] = (s0 + s1) + (s2 + s3);
      }
      __syncthreads();
      double* __restrict__ us = uvec + S.uoff;
      for (int a = tid; a < u; a += SB) {
        double s = 0.0;
        for (int sl2 = 0; sl2 < nslice; ++sl2) s += part[sl2 * (nchunk << 6) + a];
        us[a] = f[w + a] - s;
      }
    }
  }
}

__global__ __launch_bounds__(SB) void k_bwd_level(const SnDesc* __restrict__ sn, const int* __restrict__ level_sn,
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
  for (int a = tid; a < u; a += SB) g[a] = y[rw[a]];
  __syncthreads();
  // v_k = z_k / d_k - L21(:,k)^T g : four columns per wave and pass
  for (int k0 = 4 * wave; k0 < w; k0 += 64) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c;
      if (k < w) {
        const double* col = P + w + (long long)k * r;
        for (int a = lane; a < u; a += 64) s[c] += col[a] * g[a];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += __shfl_down(s[c], o, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        if (k < w) v[k] = y[S.c0 + k] / P[k + (long long)k * r] - s[c];
      }
    }
  }
  __syncthreads();
  // x_k = v_k + inv(L11)(:,k)^T v below the diagonal
  for (int k0 = 4 * wave; k0 < w; k0 += 64) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c;
      if (k < w) {
        const double* col = P + (long long)k * r;
        for (int t = k + 1 + lane; t < w; t += 64) s[c] += col[t] * v[t];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) s[c] += __shfl_down(s[c], o, 64);
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = k0 + c;
        if (k < w) y[S.c0 + k] = v[k] + s[c];
      }
    }
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
